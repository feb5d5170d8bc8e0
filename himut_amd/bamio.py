"""BAM in / out for the hot path (ctypes front-end of csrc/bam_ingest.cpp).

``read_bam`` stands where the reference opens ``pysam.AlignmentFile`` and wraps every record
in ``bamlib.BAM`` (caller.py:267,299-300; bamlib.py:14-32, 89-129): it returns the header
facts the driver needs (contig sizes, sample name) and one ReadBatch per contig, reads in
file order.  ``write_bam`` turns read batches (e.g. synthetic ones) into a BAM file."""
import ctypes
import os

import numpy as np

from . import build
from .readbatch import ReadBatch

_lib = None


class _WriteContig(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("length", ctypes.c_int64), ("n", ctypes.c_int64)] + [
        (k, ctypes.c_void_p) for k in ("tstart", "qstart", "qlen", "mapq", "flag", "qid", "qoff", "cs_off", "seq",
                                       "bq", "cs", "tp")]


def _load():
    global _lib
    if _lib is None:
        path = build.build_host()
        L = ctypes.CDLL(path)
        L.bam_load.restype = ctypes.c_void_p
        L.bam_load.argtypes = [ctypes.c_char_p]
        L.bam_load_threads.restype = ctypes.c_void_p
        L.bam_load_threads.argtypes = [ctypes.c_char_p, ctypes.c_int]
        for f in ("bam_error", "bam_header_text"):
            getattr(L, f).restype = ctypes.c_char_p
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.bam_ref_name.restype = ctypes.c_char_p
        L.bam_ref_name.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.bam_n_ref.restype = ctypes.c_int64
        L.bam_n_ref.argtypes = [ctypes.c_void_p]
        for f in ("bam_ref_len", "bam_ref_nreads", "bam_ref_bases_padded", "bam_ref_cs_bytes"):
            getattr(L, f).restype = ctypes.c_int64
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.bam_count.restype = ctypes.c_int64
        L.bam_count.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.bam_ref_copy.restype = None
        L.bam_ref_copy.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 13
        L.vcf_format_records.restype = ctypes.c_int64
        L.vcf_format_records.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_int64]
        L.bam_ref_bytes.restype = ctypes.c_void_p
        L.bam_ref_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        L.bam_free.restype = None
        L.bam_free.argtypes = [ctypes.c_void_p]
        L.bam_stream_open.restype = ctypes.c_void_p
        L.bam_stream_open.argtypes = [ctypes.c_char_p, ctypes.c_int]
        for f in ("bam_stream_error", "bam_stream_header_text"):
            getattr(L, f).restype = ctypes.c_char_p
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.bam_stream_ref_name.restype = ctypes.c_char_p
        L.bam_stream_ref_name.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.bam_stream_n_ref.restype = ctypes.c_int64
        L.bam_stream_n_ref.argtypes = [ctypes.c_void_p]
        L.bam_stream_ref_len.restype = ctypes.c_int64
        L.bam_stream_ref_len.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        for f in ("bam_stream_indexed", "bam_stream_unique_names"):
            getattr(L, f).restype = ctypes.c_int
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.bam_stream_close.restype = None
        L.bam_stream_close.argtypes = [ctypes.c_void_p]
        L.bam_stream_select.restype = ctypes.c_int
        L.bam_stream_select.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]
        L.bam_stream_next.restype = ctypes.c_int64
        L.bam_stream_next.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                      ctypes.c_void_p]
        L.bam_stream_prefetch.restype = ctypes.c_int
        L.bam_stream_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.bam_stream_pump.restype = ctypes.c_int
        L.bam_stream_pump.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int64]
        L.bam_stream_wait.restype = ctypes.c_int
        L.bam_stream_wait.argtypes = [ctypes.c_void_p]
        L.bam_stream_head.restype = ctypes.c_int64
        L.bam_stream_head.argtypes = []
        for f in ("bam_stream_inflated_bytes", "bam_stream_scan_parts"):
            getattr(L, f).restype = ctypes.c_int64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.bam_write.restype = ctypes.c_int
        L.bam_write.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(_WriteContig), ctypes.c_int64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class BamFile:
    """All contigs of one BAM, loaded once."""

    def __init__(self, path, threads=0):
        """threads: BGZF inflate threads (0 = HIMUT_INGEST_THREADS or one per hardware thread, at most 16)."""
        L = _load()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        h = L.bam_load_threads(path.encode(), int(threads))
        self._h = h            # the big arrays of the batches are views into the library's memory
        self._L = L
        err = L.bam_error(h).decode()
        if err:
            raise ValueError("{}: {}".format(path, err))
        if L.bam_count(h, 0):
            # the reference does line.get_tag("cs") on every record (bamlib.py:32)
            raise KeyError("tag 'cs' not present in {} records of {}".format(L.bam_count(h, 0), path))
        if L.bam_count(h, 2):
            raise ValueError("{} is not coordinate sorted".format(path))
        self.header_text = L.bam_header_text(h).decode("utf-8", "replace")
        self.tname2tsize = {}
        self.batches = {}

        def view(i, which, n):
            if n == 0:
                return np.zeros(0, np.uint8)
            buf = (ctypes.c_uint8 * n).from_address(L.bam_ref_bytes(h, i, which))
            a = np.frombuffer(buf, dtype=np.uint8)
            a.flags.writeable = True
            return a

        for i in range(L.bam_n_ref(h)):
            name = L.bam_ref_name(h, i).decode()
            length = L.bam_ref_len(h, i)
            self.tname2tsize[name] = length
            n = L.bam_ref_nreads(h, i)
            tot = L.bam_ref_bases_padded(h, i)
            csb = L.bam_ref_cs_bytes(h, i)
            a = dict(tstart=np.zeros(n, np.int32), tend=np.zeros(n, np.int32), qstart=np.zeros(n, np.int32),
                     qlen=np.zeros(n, np.int32), mapq=np.zeros(n, np.uint8), flag=np.zeros(n, np.uint16),
                     qid=np.zeros(n, np.int32), qoff=np.zeros(n, np.int64), cs_off=np.zeros(n + 1, np.int64),
                     tp=np.zeros(n, np.uint8))
            L.bam_ref_copy(h, i, _p(a["tstart"]), _p(a["tend"]), _p(a["qstart"]), _p(a["qlen"]), _p(a["mapq"]),
                           _p(a["flag"]), _p(a["qid"]), _p(a["qoff"]), _p(a["cs_off"]), None, None, None, _p(a["tp"]))
            a["seq"] = view(i, 0, tot // 2)
            a["bq"] = view(i, 1, tot)
            a["cs"] = view(i, 2, csb)
            b = ReadBatch(name=name, length=length, **a)
            b._owner = self        # keeps the library's memory alive as long as the batch is
            self.batches[name] = b

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.bam_free(self._h)
                self._h = None
        except Exception:
            pass

    def sample(self):
        """SM of the first @RG line (bamlib.get_sample, bamlib.py:89-106)."""
        for line in self.header_text.strip().split("\n"):
            if line.startswith("@RG"):
                for f in line.split():
                    if f.startswith("SM"):
                        return f.split(":")[1]
        raise ValueError("SM field is missing; provide a BAM file with an @RG group")


class BamStream:
    """One BAM opened for the device-side ingest: the header here, the records of one contig at a time to the GPU
    (``ingest_contig``).  With an index beside the file (x.bam.bai) only the contig's own BGZF blocks are inflated."""

    def __init__(self, path, threads=0):
        L = _load()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self._L = L
        self._h = L.bam_stream_open(path.encode(), int(threads))
        err = L.bam_stream_error(self._h).decode()
        if err:
            raise ValueError("{}: {}".format(path, err))
        self.path = path
        self.header_text = L.bam_stream_header_text(self._h).decode("utf-8", "replace")
        self.names = [L.bam_stream_ref_name(self._h, i).decode() for i in range(L.bam_stream_n_ref(self._h))]
        self.tname2tsize = {n: L.bam_stream_ref_len(self._h, i) for i, n in enumerate(self.names)}
        self.indexed = bool(L.bam_stream_indexed(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.bam_stream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    sample = BamFile.sample

    def inflated_bytes(self):
        """Inflated bytes this stream has produced since it was opened (the header's blocks not counted)."""
        return int(self._L.bam_stream_inflated_bytes(self._h))

    def ingest_contig(self, ctx, chrom, window_bytes=None):
        """Streams the records of ``chrom`` into the context ``ctx`` (an _ffi.Context), parsed on the device.  Two
        pinned windows: while the GPU copies and parses one, the host's pool inflates the next.  Leaves the context as
        ``push_reads`` would and returns the ingest result (n_reads, bases_padded, cs_bytes, read_bases)."""
        L, h = self._L, self._h
        if window_bytes is None:
            window_bytes = int(os.environ.get("HIMUT_INGEST_WINDOW_KB", str(64 << 10))) << 10
        bound = ctypes.c_int64()
        if L.bam_stream_select(h, self.names.index(chrom), ctypes.byref(bound)):
            raise ValueError(L.bam_stream_error(h).decode())
        cap = window_bytes + L.bam_stream_head()       # head room for the record a window boundary cuts
        bufs = ctx.ingest_begin(bound.value, cap)
        rec_cap = window_bytes // 64 + 16
        # the loop -- wait for window k's inflate, start window k + 1's, hop over k's records, hand k to the GPU -- is one
        # call into the host library, which calls the device library's himut_ingest_wait / himut_ingest_window itself
        rc = L.bam_stream_pump(h, ctx.handle, ctx.fn_address("himut_ingest_wait"), ctx.fn_address("himut_ingest_window"),
                               bufs[0], bufs[1], cap, rec_cap)
        if rc == -2:
            raise ValueError("{}: {}".format(self.path, L.bam_stream_error(h).decode()))
        if rc:
            ctx.raise_for(rc)
        res = ctx.ingest_end(bool(L.bam_stream_unique_names(h)))
        if res["n_missing_cs"]:
            # the reference does line.get_tag("cs") on every record (bamlib.py:32)
            raise KeyError("tag 'cs' not present in {} records of {}".format(res["n_missing_cs"], self.path))
        if res["n_unsorted"]:
            raise ValueError("{} is not coordinate sorted".format(self.path))
        return res


_cache = {}


def read_bam(path, threads=0):
    key = (os.path.abspath(path), os.path.getmtime(path))
    if key not in _cache:
        _cache.clear()
        _cache[key] = BamFile(path, threads)
    return _cache[key]


def read_contig(path, chrom):
    return read_bam(path).batches[chrom]


def write_bam(path, batches, sample="syn"):
    """batches: list of ReadBatch in @SQ order.  CIGARs are derived from the cs tags."""
    L = _load()
    arr = (_WriteContig * len(batches))()
    keep = []
    for k, b in enumerate(batches):
        cols = [np.ascontiguousarray(x) for x in (b.tstart, b.qstart, b.qlen, b.mapq, b.flag, b.qid, b.qoff, b.cs_off,
                                                  b.seq, b.bq, b.cs, b.tp)]
        keep.append(cols)
        nm = b.name.encode()
        keep.append(nm)
        arr[k] = _WriteContig(nm, int(b.length), int(b.n), *[_p(x) for x in cols])
    rc = L.bam_write(path.encode(), sample.encode(), arr, len(batches))
    if rc:
        raise IOError("bam_write failed ({})".format(rc))


def format_records(recs, chrom, phased=False, single_molecule_file=False):
    """VCF body lines (bytes) of a records array, printed by the host library exactly as
    caller.records_to_tuples + vcflib._body_line would."""
    L = _load()
    recs = np.ascontiguousarray(recs)
    cap = int(recs.shape[0]) * 256 + 1024
    out = np.empty(cap, np.uint8)
    n = L.vcf_format_records(_p(recs), int(recs.shape[0]), chrom.encode(), 1 if phased else 0,
                             1 if single_molecule_file else 0, _p(out), cap)
    if n < 0:
        raise RuntimeError("vcf_format_records: buffer too small")
    return out[:n].tobytes()
