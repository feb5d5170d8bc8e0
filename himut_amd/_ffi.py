"""ctypes binding of libhimut_hip.so (include/himut_hip.h).

The HIP extension is the only compute path: if the library cannot be loaded
the import of this module's ``lib()`` raises -- there is no CPU fallback.
"""
import ctypes
import os
import sys

import numpy as np

from . import build

STATUS_NAMES = ["PASS", "LowBQ", "LowGQ", "IndelSite", "HetSite", "HetAltSite", "HomAltSite", "ComSnp",
                "PanelOfNormal", "LowDepth", "HighDepth", "Unphased"]

RECORD_DTYPE = np.dtype([("tpos", "<i4"), ("chunk", "<i4"), ("phase_set", "<i4"), ("gq", "<i4"), ("ref", "u1"),
                         ("alt", "u1"), ("gt0", "u1"), ("gt1", "u1"), ("status", "u1"), ("gt_state", "u1"),
                         ("flags", "u1"), ("pad", "u1"), ("counts", "<u4", (6,)), ("bqsum", "<u4", (4,))])
assert RECORD_DTYPE.itemsize == 64


class Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in (
        "min_qv", "min_mapq", "qlen_lower_limit", "qlen_upper_limit", "min_gq", "min_bq", "max_mismatch_count",
        "mismatch_window_size", "md_threshold", "min_ref_count", "min_alt_count", "min_hap_count", "phase",
        "reserved")] + [("min_sequence_identity", ctypes.c_double), ("min_trim", ctypes.c_double)]


class ReadBatchStruct(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_int64)] + [(k, ctypes.c_void_p) for k in (
        "tstart", "tend", "qstart", "qlen", "mapq", "flag", "qid", "qoff", "cs_off", "seq", "bq", "cs")] + [
        ("seq_bytes", ctypes.c_int64), ("bq_bytes", ctypes.c_int64), ("cs_bytes", ctypes.c_int64)]


class RunStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in ("ms_total", "ms_parse", "ms_bqsum", "ms_hap", "ms_emit", "ms_index",
                                               "ms_capture", "ms_eval", "ms_finalize")] + \
               [(k, ctypes.c_int64) for k in ("n_reads", "read_bases", "positions", "n_unique_positions", "n_candidates",
                                              "n_records", "column_slots", "reran")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class IngestResult(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int64) for k in ("n_reads", "bases_padded", "cs_bytes", "read_bases", "n_missing_cs",
                                              "n_unsorted", "n_malformed")]


EXPORTS = ["himut_abi_version", "himut_create", "himut_destroy", "himut_last_error", "himut_set_params",
           "himut_set_gt_lut", "himut_set_chunks", "himut_set_site_set", "himut_set_phase", "himut_push_reads",
           "himut_run", "himut_get_records", "himut_get_log", "himut_get_stats", "himut_records_device",
           "himut_copy_records_to_device", "himut_pile_counts", "himut_set_reference", "himut_run_normcounts",
           "himut_get_normcounts", "himut_ref_tricounts", "himut_run_edges", "himut_set_stage_timing", "himut_sbs96_counts", "himut_ingest_begin", "himut_ingest_buffer",
           "himut_ingest_wait", "himut_ingest_window", "himut_ingest_end", "himut_ingest_read_meta", "himut_download_reads",
           "himut_run_begin", "himut_run_end", "himut_debug_normcounts"]

_lib = None


class HimutError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libhimut_hip error {}: {}".format(code, message))
        self.code = code
        self.message = message


def _maybe_import_torch_first():
    """torch ships its own libamdhip64.so.7; whichever copy is loaded first
    serves the whole process.  When torch is going to be used in this process
    (bench.py, the RCCL gather) it must come up before our library so both bind
    to the same HIP runtime instance."""
    if "torch" in sys.modules or os.environ.get("HIMUT_NO_TORCH") == "1":
        return
    if os.environ.get("HIMUT_WITH_TORCH") == "1":
        import torch  # noqa: F401


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _maybe_import_torch_first()
    path = os.environ.get("HIMUT_HIP_LIB_OVERRIDE") or build.HIP_LIB   # override: diagnostic builds only
    if not os.path.exists(path):
        path = build.build_hip()
    L = ctypes.CDLL(path)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise ImportError("libhimut_hip.so lacks symbol " + name)
    L.himut_abi_version.restype = ctypes.c_int
    L.himut_create.restype = ctypes.c_int
    L.himut_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.himut_destroy.restype = None
    L.himut_destroy.argtypes = [ctypes.c_void_p]
    L.himut_last_error.restype = ctypes.c_char_p
    L.himut_last_error.argtypes = [ctypes.c_void_p]
    L.himut_set_params.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params)]
    L.himut_set_gt_lut.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_void_p]
    L.himut_set_chunks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.himut_set_site_set.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    L.himut_set_phase.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64]
    L.himut_push_reads.argtypes = [ctypes.c_void_p, ctypes.POINTER(ReadBatchStruct)]
    L.himut_run.argtypes = [ctypes.c_void_p]
    L.himut_get_records.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
    L.himut_get_log.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.himut_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(RunStats)]
    L.himut_records_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
    L.himut_copy_records_to_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.himut_pile_counts.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.himut_set_reference.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
    L.himut_run_normcounts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.himut_get_normcounts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.himut_ref_tricounts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.himut_set_stage_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.himut_sbs96_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p]
    L.himut_run_edges.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int64, ctypes.c_void_p]
    L.himut_ingest_begin.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    L.himut_ingest_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.himut_ingest_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.himut_ingest_window.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    L.himut_ingest_end.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(IngestResult)]
    L.himut_ingest_read_meta.argtypes = [ctypes.c_void_p] * 6
    L.himut_download_reads.argtypes = [ctypes.c_void_p, ctypes.POINTER(ReadBatchStruct), ctypes.c_void_p]
    for name in EXPORTS:
        if name not in ("himut_destroy", "himut_last_error", "himut_ingest_buffer"):
            getattr(L, name).restype = ctypes.c_int
    L.himut_ingest_buffer.restype = ctypes.c_void_p
    if L.himut_abi_version() != 2:
        raise ImportError("libhimut_hip.so ABI version mismatch")
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """One worker bound to one GPU (himut_ctx)."""

    def __init__(self, device=0):
        self._L = lib()
        h = ctypes.c_void_p()
        rc = self._L.himut_create(int(device), ctypes.byref(h))
        if rc:
            raise HimutError(rc, "himut_create failed (no usable HIP device {}?)".format(device))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.himut_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc:
            raise HimutError(rc, self._L.himut_last_error(self._h).decode("utf-8", "replace"))

    def set_params(self, **kw):
        p = Params()
        for k, _ in Params._fields_:
            if k in kw:
                setattr(p, k, kw[k])
        self._check(self._L.himut_set_params(self._h, ctypes.byref(p)))

    def set_gt_lut(self, hom, het, err, log_prior):
        hom = np.ascontiguousarray(hom, np.float64)
        het = np.ascontiguousarray(het, np.float64)
        err = np.ascontiguousarray(err, np.float64)
        pr = np.ascontiguousarray(log_prior, np.float64)
        self._check(self._L.himut_set_gt_lut(self._h, _ptr(hom), _ptr(het), _ptr(err), int(hom.shape[0]), _ptr(pr)))

    def set_chunks(self, chunks):
        s = np.array([c[0] for c in chunks], np.int32)
        e = np.array([c[1] for c in chunks], np.int32)
        self._check(self._L.himut_set_chunks(self._h, _ptr(s), _ptr(e), len(chunks)))

    def set_site_set(self, which, keys):
        k = np.ascontiguousarray(keys, np.uint64)
        self._check(self._L.himut_set_site_set(self._h, int(which), _ptr(k), int(k.shape[0])))

    def set_phase(self, off, hpos, href, halt, hbit):
        off = np.ascontiguousarray(off, np.int64)
        arrs = [np.ascontiguousarray(hpos, np.int32), np.ascontiguousarray(href, np.uint8),
                np.ascontiguousarray(halt, np.uint8), np.ascontiguousarray(hbit, np.uint8)]
        self._check(self._L.himut_set_phase(self._h, _ptr(off), *[_ptr(a) for a in arrs], int(off.shape[0]) - 1))

    def push_reads(self, b):
        keep = [np.ascontiguousarray(x, dt) for x, dt in (
            (b.tstart, np.int32), (b.tend, np.int32), (b.qstart, np.int32), (b.qlen, np.int32), (b.mapq, np.uint8),
            (b.flag, np.uint16), (b.qid, np.int32), (b.qoff, np.int64), (b.cs_off, np.int64), (b.seq, np.uint8),
            (b.bq, np.uint8), (b.cs, np.uint8))]
        st = ReadBatchStruct(b.n, *[_ptr(x) for x in keep], int(keep[9].shape[0]), int(keep[10].shape[0]),
                             int(keep[11].shape[0]))
        self._check(self._L.himut_push_reads(self._h, ctypes.byref(st)))

    def run(self):
        self._check(self._L.himut_run(self._h))

    def run_begin(self):
        """The run queued, not waited for (run_end does that): several contexts' runs back to back on one GPU."""
        self._check(self._L.himut_run_begin(self._h))

    def run_end(self):
        self._check(self._L.himut_run_end(self._h))

    def records(self):
        p = ctypes.c_void_p()
        n = ctypes.c_int64()
        self._check(self._L.himut_get_records(self._h, ctypes.byref(p), ctypes.byref(n)))
        if n.value == 0:
            return np.zeros(0, RECORD_DTYPE)
        buf = (ctypes.c_char * (n.value * 64)).from_address(p.value)
        return np.frombuffer(buf, dtype=RECORD_DTYPE).copy()

    def log(self):
        out = np.zeros(15, np.int64)
        self._check(self._L.himut_get_log(self._h, _ptr(out)))
        return [int(x) for x in out]

    def stats(self):
        s = RunStats()
        self._check(self._L.himut_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def stats_brief(self):
        """(ms_total, ms_capture, reran) of the last run without building the whole dictionary: what a timed loop reads."""
        s = getattr(self, "_st", None)
        if s is None:
            s = self._st = RunStats()
        self._L.himut_get_stats(self._h, ctypes.byref(s))
        return s.ms_total, s.ms_capture, s.reran

    def set_stage_timing(self, level):
        """0: total only; 1: + column capture (default); 2: every stage (costs a few microseconds per event)."""
        self._check(self._L.himut_set_stage_timing(self._h, int(level)))

    def records_device(self):
        p = ctypes.c_void_p()
        n = ctypes.c_int64()
        self._check(self._L.himut_records_device(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def copy_records_to_device(self, dev_ptr, capacity_records):
        self._check(self._L.himut_copy_records_to_device(self._h, ctypes.c_void_p(dev_ptr), int(capacity_records)))

    def set_reference(self, seq, cls, n_classes):
        raw = np.frombuffer(seq.encode("ascii") if isinstance(seq, str) else bytes(seq), np.uint8)
        cls = np.ascontiguousarray(cls, np.uint8)
        self._n_classes = int(n_classes)
        self._check(self._L.himut_set_reference(self._h, _ptr(raw), int(raw.shape[0]), _ptr(cls), int(n_classes)))

    def run_normcounts(self, alt_order, non_human_sample=False):
        tab = np.ascontiguousarray(alt_order, np.uint8).reshape(12)
        self._check(self._L.himut_run_normcounts(self._h, _ptr(tab), 1 if non_human_sample else 0))

    def debug_normcounts(self, sweep=0, dirty_cap=0, pool_slots=0):
        """Test hook (himut_debug_normcounts): which sweep, the capacity of a part of the left-over list, pool slots."""
        self._check(self._L.himut_debug_normcounts(self._h, int(sweep), ctypes.c_int64(int(dirty_cap)), int(pool_slots)))

    def normcounts(self):
        k3 = self._n_classes ** 3
        ccs = np.zeros(k3, np.int64)
        ref = np.zeros(k3, np.int64)
        log = np.zeros(14, np.int64)
        self._check(self._L.himut_get_normcounts(self._h, _ptr(ccs), _ptr(ref), _ptr(log)))
        return ccs, ref, [int(x) for x in log]

    def run_edges(self, hpos, href, min_bq, min_mapq, band):
        hpos = np.ascontiguousarray(hpos, np.int32)
        href = np.ascontiguousarray(href, np.uint8)
        counts = np.zeros(max(1, hpos.shape[0]) * int(band) * 4, np.uint32)
        self._check(self._L.himut_run_edges(self._h, _ptr(hpos), _ptr(href), int(hpos.shape[0]), int(min_bq), int(min_mapq),
                                            int(band), _ptr(counts)))
        return counts

    def ref_tricounts(self):
        out = np.zeros(64, np.int64)
        self._check(self._L.himut_ref_tricounts(self._h, _ptr(out)))
        return out

    # ---- device-side BAM ingest (bamio.ingest_contig drives it)
    def ingest_begin(self, inflated_bound, window_bytes):
        self._check(self._L.himut_ingest_begin(self._h, int(inflated_bound), int(window_bytes)))
        return [self._L.himut_ingest_buffer(self._h, k) for k in (0, 1)]

    @property
    def handle(self):
        """The himut_ctx* (for a host library that calls the C ABI itself: bamio's ingest pump)."""
        return self._h

    def fn_address(self, name):
        """Address of an exported function of the library, as a void*."""
        return ctypes.cast(getattr(self._L, name), ctypes.c_void_p)

    def raise_for(self, rc):
        self._check(rc)

    def ingest_wait(self, slot):
        self._check(self._L.himut_ingest_wait(self._h, int(slot)))

    def ingest_window(self, slot, start, nbytes, rec_off, qid, n_rec, padded_bases, tag_bytes):
        self._check(self._L.himut_ingest_window(self._h, int(slot), int(start), int(nbytes), _ptr(rec_off), _ptr(qid),
                                                int(n_rec), int(padded_bases), int(tag_bytes)))

    def ingest_end(self, unique_qnames):
        r = IngestResult()
        rc = self._L.himut_ingest_end(self._h, 1 if unique_qnames else 0, ctypes.byref(r))
        self._check(rc)
        return {k: getattr(r, k) for k, _ in IngestResult._fields_}

    def ingest_read_meta(self, n):
        """(tstart, tend, qlen, mapq, tp) of the resident reads: what bamlib.get_thresholds looks at."""
        a = [np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint8), np.zeros(n, np.uint8)]
        self._check(self._L.himut_ingest_read_meta(self._h, *[_ptr(x) for x in a]))
        return a

    def download_reads(self, res, name="", length=0):
        """The resident read batch as a host ReadBatch (tests)."""
        from .readbatch import ReadBatch
        n = int(res["n_reads"])
        a = dict(tstart=np.zeros(n, np.int32), tend=np.zeros(n, np.int32), qstart=np.zeros(n, np.int32),
                 qlen=np.zeros(n, np.int32), mapq=np.zeros(n, np.uint8), flag=np.zeros(n, np.uint16),
                 qid=np.zeros(n, np.int32), qoff=np.zeros(n, np.int64), cs_off=np.zeros(n + 1, np.int64),
                 seq=np.zeros(int(res["bases_padded"]) // 2, np.uint8), bq=np.zeros(int(res["bases_padded"]), np.uint8),
                 cs=np.zeros(int(res["cs_bytes"]), np.uint8))
        tp = np.zeros(n, np.uint8)
        order = ("tstart", "tend", "qstart", "qlen", "mapq", "flag", "qid", "qoff", "cs_off", "seq", "bq", "cs")
        st = ReadBatchStruct(n, *[_ptr(a[k]) for k in order], int(a["seq"].shape[0]), int(a["bq"].shape[0]),
                             int(a["cs"].shape[0]))
        self._check(self._L.himut_download_reads(self._h, ctypes.byref(st), _ptr(tp)))
        return ReadBatch(name=name, length=length, tp=tp, **a)

    def sbs96_counts(self, pos0, ref, alt):
        """99 bins (see himut_sbs96_counts) for the substitutions (0-based pos, ASCII ref / alt) of the contig whose
        string was given to set_reference."""
        pos0 = np.ascontiguousarray(pos0, np.int32)
        ref = np.ascontiguousarray(ref, np.uint8)
        alt = np.ascontiguousarray(alt, np.uint8)
        out = np.zeros(99, np.int64)
        self._check(self._L.himut_sbs96_counts(self._h, _ptr(pos0), _ptr(ref), _ptr(alt), int(pos0.shape[0]), _ptr(out)))
        return out

    def pile_counts(self, p0, p1):
        counts = np.zeros((p1 - p0, 6), np.uint32)
        bqsum = np.zeros((p1 - p0, 4), np.uint32)
        self._check(self._L.himut_pile_counts(self._h, int(p0), int(p1), _ptr(counts), _ptr(bqsum)))
        return counts, bqsum
