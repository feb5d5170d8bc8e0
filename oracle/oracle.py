"""ctypes front-end of oracle/himut_oracle.c (test infrastructure only).

The LUT builder below restates gtlib.py:47-69 with the reference's own Python
expressions so that the doubles come from the same libm calls."""
import ctypes
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhimut_oracle.so")

STATUS = ["PASS", "LowBQ", "LowGQ", "IndelSite", "HetSite", "HetAltSite", "HomAltSite", "ComSnp", "PanelOfNormal",
          "LowDepth", "HighDepth", "Unphased"]
BASE2IDX = {"A": 0, "T": 1, "G": 2, "C": 3}  # util.py:14-20
ERRORS = {1: "unparseable cs tag", 2: "KeyError: base outside ATGC", 3: "ValueError: math domain error (BQ 0)",
          4: "ValueError: invalid coordinates (chunk start > end)", 5: "record capacity", 6: "KeyError: tpos2qbase",
          7: "out of memory"}

RECORD_DTYPE = np.dtype([("tpos", "<i4"), ("chunk", "<i4"), ("phase_set", "<i4"), ("gq", "<i4"), ("ref", "u1"),
                         ("alt", "u1"), ("gt0", "u1"), ("gt1", "u1"), ("status", "u1"), ("gt_state", "u1"),
                         ("flags", "u1"), ("pad", "u1"), ("counts", "<u4", (6,)), ("bqsum", "<u4", (4,))])


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(ERRORS.get(code, "error {}".format(code)))
        self.code = code


class _Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("min_qv", "min_mapq", "qlen_lower", "qlen_upper", "min_gq", "min_bq",
                                               "max_mismatch_count", "mismatch_window", "md_threshold",
                                               "min_ref_count", "min_alt_count", "min_hap_count", "phase", "pad")] + \
               [("min_sequence_identity", ctypes.c_double), ("min_trim", ctypes.c_double)]


class _Reads(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64)] + [(k, ctypes.c_void_p) for k in (
        "tstart", "tend", "qstart", "qlen", "mapq", "flag", "qid", "qoff", "cs_off", "seq", "bq", "cs")]


class _Lut(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("lut_hom", "lut_het", "lut_err", "log_prior")]


class _Phase(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("off", "hpos", "href", "halt", "hbit")]


_lib = None


def build(force=False):
    src = os.path.join(HERE, "himut_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", "libhimut_oracle.so"], check=True, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_call.restype = ctypes.c_int
        _lib.orc_cs_ops.restype = ctypes.c_int
        _lib.orc_pile_counts.restype = ctypes.c_int
        _lib.orc_germ_gt.restype = ctypes.c_int
        assert _lib.orc_record_size() == RECORD_DTYPE.itemsize
    return _lib


def build_lut(germline_snv_prior):
    """gtlib.init + get_log10_* (gtlib.py:12-20, 47-69).  Index 0 of the three
    tables is NaN: the reference raises ValueError (log10(0)) for BQ 0."""
    hom = np.full(256, np.nan)
    het = np.full(256, np.nan)
    err = np.full(256, np.nan)
    for bq in range(1, 256):
        eps = 10 ** (-bq / 10)
        hom[bq] = math.log10(1 - eps)
        het[bq] = math.log10(0.5 - eps / 2.0)
        err[bq] = math.log10(10 ** (-(bq / 3) / 10))
    p = germline_snv_prior
    prior = {"het": p, "hetalt": p * p * 2, "homref": 1 - ((1.5 * p) + (p * p)), "homalt": p / 2}
    logp = np.array([math.log10(prior[k]) for k in ("homref", "het", "hetalt", "homalt")])
    return hom, het, err, logp


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _reads_struct(b):
    keep = [np.ascontiguousarray(x) for x in (b.tstart, b.tend, b.qstart, b.qlen, b.mapq, b.flag, b.qid, b.qoff,
                                               b.cs_off, b.seq, b.bq, b.cs)]
    r = _Reads(b.n, *[_ptr(x) for x in keep])
    r._keep = keep
    return r


def site_keys(sites):
    """(pos1, ref, alt) iterable -> sorted uint64 keys pos<<4 | ref<<2 | alt; non-ATGC entries dropped."""
    ks = [(int(p) << 4) | (BASE2IDX[r] << 2) | BASE2IDX[a] for (p, r, a) in sites if r in BASE2IDX and a in BASE2IDX]
    return np.array(sorted(set(ks)), dtype=np.uint64)


def pack_phase(chunks, ps2hbit, ps2hpos, ps2hetsnp):
    """Per chunk (keyed str(chunk_start), caller.py:292-295) flat arrays."""
    off = [0]
    hpos, href, halt, hbit = [], [], [], []
    for (s, e) in chunks:
        key = str(s)
        hp = ps2hpos.get(key, [])
        hs = ps2hetsnp.get(key, [])
        hb = ps2hbit.get(key, [])
        for p, (_, r, a), b in zip(hp, hs, hb):
            hpos.append(int(p))
            href.append(ord(r) if len(r) == 1 else 0)
            halt.append(ord(a) if len(a) == 1 else 0)
            hbit.append(ord(b))
        off.append(len(hpos))
    return (np.array(off, np.int64), np.array(hpos, np.int32), np.array(href, np.uint8), np.array(halt, np.uint8),
            np.array(hbit, np.uint8))


def call(batch, chunks, params, germline_snv_prior=1 / (10 ** 3), pon_keys=None, com_keys=None, phase=None,
         capacity=None):
    """Runs the restated worker.  params: dict with the reference's argument
    names.  Returns (records structured array, log list of 15 ints)."""
    L = lib()
    hom, het, err, logp = build_lut(germline_snv_prior)
    lut = _Lut(_ptr(hom), _ptr(het), _ptr(err), _ptr(logp))
    P = _Params(min_qv=params["min_qv"], min_mapq=params["min_mapq"], qlen_lower=params["qlen_lower_limit"],
                qlen_upper=params["qlen_upper_limit"], min_gq=params["min_gq"], min_bq=params["min_bq"],
                max_mismatch_count=params["max_mismatch_count"], mismatch_window=params["mismatch_window_size"],
                md_threshold=params["md_threshold"], min_ref_count=params["min_ref_count"],
                min_alt_count=params["min_alt_count"], min_hap_count=params["min_hap_count"],
                phase=1 if phase is not None else 0, pad=0,
                min_sequence_identity=params["min_sequence_identity"], min_trim=params["min_trim"])
    R = _reads_struct(batch)
    cs_ = np.array([c[0] for c in chunks], np.int32)
    ce_ = np.array([c[1] for c in chunks], np.int32)
    pon = np.zeros(0, np.uint64) if pon_keys is None else np.ascontiguousarray(pon_keys, np.uint64)
    com = np.zeros(0, np.uint64) if com_keys is None else np.ascontiguousarray(com_keys, np.uint64)
    ph = None
    if phase is not None:
        arrs = pack_phase(chunks, *phase)
        ph = _Phase(*[_ptr(a) for a in arrs])
    if capacity is None:
        capacity = max(1024, int(batch.cs.shape[0]) // 3 + 16)
    out = np.zeros(capacity, RECORD_DTYPE)
    nout = ctypes.c_int64(0)
    ncand = ctypes.c_int64(0)
    log = (ctypes.c_int64 * 15)()
    rc = L.orc_call(ctypes.byref(R), ctypes.byref(P), ctypes.byref(lut), ctypes.c_int64(len(chunks)), _ptr(cs_),
                    _ptr(ce_), _ptr(pon), ctypes.c_int64(pon.shape[0]), _ptr(com), ctypes.c_int64(com.shape[0]),
                    ctypes.byref(ph) if ph is not None else None, _ptr(out), ctypes.c_int64(capacity),
                    ctypes.byref(nout), log, ctypes.byref(ncand))
    if rc:
        raise OracleError(rc)
    return out[:nout.value].copy(), [int(x) for x in log]


def records_to_tuples(chrom, recs):
    """Reference 12-tuples (caller.py:351-620) from integer records; the
    divisions are the reference's (bamlib.py:204-208, caller.py:174-192)."""
    out = []
    for r in recs:
        c = [int(x) for x in r["counts"]]
        s = [int(x) for x in r["bqsum"]]
        depth = float(c[0] + c[1] + c[2] + c[3] + c[5])
        ref = chr(r["ref"])
        alt = chr(r["alt"])
        status = STATUS[int(r["status"])]
        ref_count = float(c[BASE2IDX[ref]])
        if status == "HetAltSite":
            a1, a2 = chr(r["gt0"]), chr(r["gt1"])
            p, q = BASE2IDX[a1], BASE2IDX[a2]
            pc, qc = float(c[p]), float(c[q])
            alt_bq = "{:0.1f},{:0.1f}".format(s[p] / pc, s[q] / qc)
            alt_count = "{:0.0f},{:0.0f}".format(pc, qc)
            alt_vaf = "{:.2f},{:.2f}".format(pc / depth, qc / depth)
            alt = "{},{}".format(a1, a2)
        else:
            ac = float(c[BASE2IDX[alt]])
            alt_bq = s[BASE2IDX[alt]] / ac if ac != 0 else 0.0
            alt_count = ac
            alt_vaf = ac / depth
        ps = str(int(r["phase_set"])) if int(r["phase_set"]) >= 0 else "."
        out.append((chrom, int(r["tpos"]), ref, alt, status, int(r["gq"]), alt_bq, depth, ref_count, alt_count,
                    alt_vaf, ps))
    return out


def cs_ops(batch, i):
    L = lib()
    R = _reads_struct(batch)
    cap = int(batch.cs_off[i + 1] - batch.cs_off[i]) + 2
    st = np.zeros(cap, np.uint8); rl = np.zeros(cap, np.int32); al = np.zeros(cap, np.int32)
    rf = np.zeros(cap, np.uint8); at = np.zeros(cap, np.uint8)
    n = L.orc_cs_ops(ctypes.byref(R), ctypes.c_int64(i), ctypes.c_int32(cap), _ptr(st), _ptr(rl), _ptr(al), _ptr(rf),
                     _ptr(at))
    if n < 0:
        raise OracleError(-n)
    return [(int(st[k]), int(rl[k]), int(al[k]), chr(rf[k]) if rf[k] else "", chr(at[k]) if at[k] else "")
            for k in range(n)]


def pile_counts(batch, p0, p1):
    L = lib()
    R = _reads_struct(batch)
    counts = np.zeros((p1 - p0, 6), np.uint32)
    bqsum = np.zeros((p1 - p0, 4), np.uint32)
    rc = L.orc_pile_counts(ctypes.byref(R), ctypes.c_int32(p0), ctypes.c_int32(p1), _ptr(counts), _ptr(bqsum))
    if rc:
        raise OracleError(rc)
    return counts, bqsum


def germ_gt(ref, alleles, bqs, germline_snv_prior=1 / (10 ** 3)):
    """alleles: himut indices (A0 T1 G2 C3) in fetch order.  Returns (gt, gq, state, pls)."""
    L = lib()
    hom, het, err, logp = build_lut(germline_snv_prior)
    lut = _Lut(_ptr(hom), _ptr(het), _ptr(err), _ptr(logp))
    a = np.ascontiguousarray(alleles, np.uint8)
    b = np.ascontiguousarray(bqs, np.uint8)
    g0 = ctypes.c_int(0); g1 = ctypes.c_int(0); st = ctypes.c_int(0)
    pl = (ctypes.c_double * 10)()
    gq = L.orc_germ_gt(ctypes.c_int(ord(ref)), ctypes.c_int32(a.shape[0]), _ptr(a), _ptr(b), ctypes.byref(lut),
                       ctypes.byref(g0), ctypes.byref(g1), ctypes.byref(st), pl)
    if gq < 0:
        raise OracleError(-gq)
    return chr(g0.value) + chr(g1.value), gq, ["homref", "het", "hetalt", "homalt"][st.value], list(pl)


TRI_LST = [f + m + l for f in "ACGT" for m in "CT" for l in "ACGT"]   # mutlib.py:17-50


def tri_classes(refseq):
    """Byte -> class id table for a reference string: one class per distinct byte (plus A C G T N).
    Returns (chars, cls[256])."""
    raw = refseq.encode("ascii") if isinstance(refseq, str) else bytes(refseq)
    present = np.flatnonzero(np.bincount(np.frombuffer(raw, np.uint8), minlength=256))
    chars = sorted(set(int(x) for x in present) | set(b"ACGTN"))
    cls = np.zeros(256, np.uint8)
    for i, ch in enumerate(chars):
        cls[ch] = i
    return chars, cls


def tri_dicts(chars, ccs, ref):
    """Histograms over class triples -> the two dicts of the reference (keys of tri_lst always present)."""
    K = len(chars)
    d_ccs = {t: 0 for t in TRI_LST}
    d_ref = {t: 0 for t in TRI_LST}
    for idx in np.flatnonzero((ccs != 0) | (ref != 0)):
        a, b, c = int(idx) // (K * K), (int(idx) // K) % K, int(idx) % K
        key = chr(chars[a]) + chr(chars[b]) + chr(chars[c])
        d_ccs[key] = d_ccs.get(key, 0) + int(ccs[idx])
        d_ref[key] = d_ref.get(key, 0) + int(ref[idx])
    return d_ccs, d_ref


def alt_order_table(order=None):
    """alt_order[ref allele][0..2] = alleles of list(base_set.difference(ref)) (normcounts.py:367); the
    order of a python set of strings depends on the interpreter's hash seed, so the caller supplies it."""
    tab = np.zeros((4, 3), np.uint8)
    for ref, ri in BASE2IDX.items():
        lst = order[ref] if order is not None else list(set("ATGC").difference(ref))
        tab[ri] = [BASE2IDX[a] for a in lst]
    return tab


def normcounts(batch, chunks, params, refseq, germline_snv_prior=1 / (10 ** 3), pon_keys=None, com_keys=None,
               alt_order=None, non_human_sample=False, phase=None):
    """Restated normcounts.get_callable_tricounts (phase = (hbit, hpos, hetsnp) dicts or None).
    Returns (ccs_tri2count, ref_tri2count, log[14])."""
    L = lib()
    L.orc_normcounts.restype = ctypes.c_int
    hom, het, err, logp = build_lut(germline_snv_prior)
    lut = _Lut(_ptr(hom), _ptr(het), _ptr(err), _ptr(logp))
    P = _Params(min_qv=params["min_qv"], min_mapq=params["min_mapq"], qlen_lower=params["qlen_lower_limit"],
                qlen_upper=params["qlen_upper_limit"], min_gq=params["min_gq"], min_bq=params["min_bq"],
                max_mismatch_count=params["max_mismatch_count"], mismatch_window=params["mismatch_window_size"],
                md_threshold=params["md_threshold"], min_ref_count=params["min_ref_count"],
                min_alt_count=params["min_alt_count"], min_hap_count=params["min_hap_count"],
                phase=1 if phase is not None else 0, pad=0,
                min_sequence_identity=params["min_sequence_identity"], min_trim=params["min_trim"])
    R = _reads_struct(batch)
    cs_ = np.array([c[0] for c in chunks], np.int32)
    ce_ = np.array([c[1] for c in chunks], np.int32)
    pon = np.zeros(0, np.uint64) if pon_keys is None else np.ascontiguousarray(pon_keys, np.uint64)
    com = np.zeros(0, np.uint64) if com_keys is None else np.ascontiguousarray(com_keys, np.uint64)
    ph = None
    if phase is not None:
        arrs = pack_phase(chunks, *phase)
        ph = _Phase(*[_ptr(a) for a in arrs])
    raw = np.frombuffer(refseq.encode("ascii") if isinstance(refseq, str) else bytes(refseq), np.uint8)
    chars, cls = tri_classes(refseq)
    K = len(chars)
    ccs = np.zeros(K * K * K, np.int64)
    ref = np.zeros(K * K * K, np.int64)
    tab = np.ascontiguousarray(alt_order_table(alt_order))
    log = (ctypes.c_int64 * 14)()
    rc = L.orc_normcounts(ctypes.byref(R), ctypes.byref(P), ctypes.byref(lut), ctypes.c_int64(len(chunks)), _ptr(cs_),
                          _ptr(ce_), _ptr(pon), ctypes.c_int64(pon.shape[0]), _ptr(com), ctypes.c_int64(com.shape[0]),
                          _ptr(raw), ctypes.c_int64(raw.shape[0]), _ptr(cls), ctypes.c_int(K), _ptr(tab),
                          ctypes.c_int(1 if non_human_sample else 0), ctypes.byref(ph) if ph is not None else None,
                          _ptr(ccs), _ptr(ref), log)
    if rc:
        raise OracleError(rc)
    d_ccs, d_ref = tri_dicts(chars, ccs, ref)
    return d_ccs, d_ref, [int(x) for x in log]


def edge_band(batch, hpos):
    """Largest number of hetSNPs one read spans, minus one: the band the edge table needs."""
    hp = np.asarray(hpos, np.int64)
    k = np.searchsorted(hp, batch.tend, side="right") - np.searchsorted(hp, batch.tstart, side="right")
    return max(1, int(k.max()) - 1 if k.shape[0] else 1)


def edges_from_table(counts, band, hidx=None):
    """Band table -> (edge_lst, edge2counts) as phaselib.get_edges returns them (keys that got a count, in natural order)."""
    nz = np.flatnonzero(counts.reshape(-1, 4).sum(1))
    edge2counts = {}
    for e in nz:
        i, d = int(e) // band, int(e) % band
        j = i + 1 + d
        key = (hidx[i], hidx[j]) if hidx is not None else (i, j)
        edge2counts[key] = [float(x) for x in counts.reshape(-1, 4)[e]]
    return sorted(edge2counts), edge2counts


def edges(batch, hetsnp_lst, min_bq, min_mapq):
    """Restated phaselib.get_edges.  hetsnp_lst: [(pos1, ref, alt)] sorted by position."""
    L = lib()
    L.orc_edges.restype = ctypes.c_int
    hpos = np.array([h[0] for h in hetsnp_lst], np.int32)
    href = np.array([ord(h[1]) for h in hetsnp_lst], np.uint8)
    band = edge_band(batch, hpos)
    counts = np.zeros(max(1, hpos.shape[0]) * band * 4, np.uint32)
    R = _reads_struct(batch)
    rc = L.orc_edges(ctypes.byref(R), ctypes.c_int(min_bq), ctypes.c_int(min_mapq), ctypes.c_int64(hpos.shape[0]), _ptr(hpos),
                     _ptr(href), ctypes.c_int64(band), _ptr(counts))
    if rc:
        raise OracleError(rc)
    return edges_from_table(counts, band)
