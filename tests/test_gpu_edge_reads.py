"""Hand-built alignments that drive the rarely taken paths of the kernels: long soft
clips, insertions and deletions longer than a capture window, reads with hundreds of
segments, reads shorter than one window, insertion/deletion neighbours, trailing
insertions, and candidate densities far above the synthetic workload's.  HIP (through the
C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

BASES = "ACGT"


def _make_read(rs, ref, tstart, target_len, kind):
    """One alignment starting at tstart; returns a record dict or None if it runs off the contig."""
    t = tstart
    seq = []
    cs = []
    L = len(ref)
    lead = trail = 0
    if kind in ("clip", "mixed") and rs.rand() < 0.8:
        lead = int(rs.randint(300, 4500))
    if kind in ("clip", "mixed") and rs.rand() < 0.5:
        trail = int(rs.randint(300, 4500))
    seq.append("".join(BASES[i] for i in rs.randint(0, 4, lead)))
    aligned = 0
    last = ""
    run_lo, run_hi = (15, 90) if kind == "noisy" else (200, 3500)
    sub_rate = 0.02 if kind == "dense" else (0.004 if kind != "noisy" else 0.01)
    first = True
    while aligned < target_len:
        n = int(rs.randint(run_lo, run_hi))
        if t + n >= L - 10:
            return None
        # a run of matches with substitutions sprinkled in
        i = 0
        while i < n:
            gap = int(rs.geometric(sub_rate)) if sub_rate > 0 else n
            m = min(gap - 1, n - i)
            if m > 0:
                cs.append(":{}".format(m))
                seq.append(ref[t:t + m])
                t += m
                i += m
                last = ":"
            if i < n:
                r = ref[t]
                a = BASES[(BASES.index(r) + int(rs.randint(1, 4))) % 4]
                cs.append("*{}{}".format(r.lower(), a.lower()))
                seq.append(a)
                t += 1
                i += 1
                last = "*"
        aligned += n
        first = False
        if aligned >= target_len:
            break
        # an indel (or two neighbouring ones) between runs
        x = rs.rand()
        big = kind in ("bigindel", "mixed") and rs.rand() < 0.35
        if x < 0.45:
            k = int(rs.randint(2300, 5200)) if big else int(rs.randint(1, 4))
            ins = "".join(BASES[j] for j in rs.randint(0, 4, k))
            cs.append("+" + ins.lower())
            seq.append(ins)
            last = "+"
            if rs.rand() < 0.15:                       # insertion directly followed by a deletion
                d = int(rs.randint(1, 4))
                if t + d >= L - 10:
                    return None
                cs.append("-" + ref[t:t + d].lower())
                t += d
                last = "-"
        else:
            d = int(rs.randint(2300, 6000)) if big else int(rs.randint(1, 4))
            if t + d >= L - 10:
                return None
            cs.append("-" + ref[t:t + d].lower())
            t += d
            last = "-"
            if rs.rand() < 0.15:                       # deletion directly followed by an insertion
                k = int(rs.randint(1, 4))
                ins = "".join(BASES[j] for j in rs.randint(0, 4, k))
                cs.append("+" + ins.lower())
                seq.append(ins)
                last = "+"
    if kind in ("mixed", "trailins") and last in (":", "*") and rs.rand() < 0.5:
        k = int(rs.randint(1, 4))                      # an insertion at the very end of the alignment
        ins = "".join(BASES[j] for j in rs.randint(0, 4, k))
        cs.append("+" + ins.lower())
        seq.append(ins)
    seq.append("".join(BASES[i] for i in rs.randint(0, 4, trail)))
    s = "".join(seq)
    bq = np.where(rs.rand(len(s)) < 0.8, 93, rs.randint(1, 93, len(s))).astype(np.uint8)
    return dict(tstart=tstart, tend=t, qstart=lead, seq=s, bq=bq, cs="".join(cs), mapq=60)


def _make_batch(seed, contig_len, n_reads, kinds):
    from himut_amd.readbatch import batch_from_records
    rs = np.random.RandomState(seed)
    ref = "".join(BASES[i] for i in rs.randint(0, 4, contig_len))
    recs = []
    while len(recs) < n_reads:
        kind = kinds[int(rs.randint(0, len(kinds)))]
        target = int(rs.randint(400, 1900)) if kind == "short" else int(rs.randint(3000, 16000))
        r = _make_read(rs, ref, int(rs.randint(0, contig_len - 500)), target, kind)
        if r is not None:
            recs.append(r)
    recs.sort(key=lambda r: r["tstart"])
    for i, r in enumerate(recs):
        r["qname"] = "m/{}/ccs".format(i)
    batch = batch_from_records("chrE", contig_len, recs)
    batch.refseq = ref          # the contig the reads were cut from (normcounts needs it)
    return batch


def _params(**kw):
    p = dict(util.CALL_DEFAULTS)
    p.update(min_qv=20, min_sequence_identity=0.5, min_bq=20, min_gq=0, max_mismatch_count=1000,
             qlen_lower_limit=100, qlen_upper_limit=100000, md_threshold=100000, min_ref_count=0, min_alt_count=1)
    p.update(kw)
    return p


def _check(batch, chunks, p):
    from oracle import oracle as O
    from himut_amd.caller import Worker
    from tests.test_gpu_parity import _run_hip
    w = Worker(0)
    try:
        orecs, olog = O.call(batch, chunks, p, p["germline_snv_prior"], None, None, None)
        hrecs, hlog = _run_hip(w, batch, chunks, p)
    finally:
        w.close()
    assert hlog == olog
    assert len(hrecs) == len(orecs)
    for name in ("tpos", "chunk", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"):
        assert np.array_equal(hrecs[name], orecs[name]), name
    return len(hrecs)


@pytest.mark.parametrize("seed,kinds", [
    (11, ["plain", "clip", "short"]),
    (12, ["bigindel", "plain"]),
    (13, ["noisy", "plain"]),
    (14, ["dense", "short", "trailins"]),
    (15, ["mixed", "noisy", "dense", "short", "bigindel", "clip", "trailins"]),
])
def test_edge_reads_against_oracle(seed, kinds):
    L = 120_000
    batch = _make_batch(seed, L, 260, kinds)
    chunks = [(1, 40_000), (40_000, 80_000), (80_000, L - 2)]
    n = _check(batch, chunks, _params())
    assert n > 500


def test_edge_reads_default_filters():
    """The same kind of reads under the default mismatch-window / trim rules."""
    L = 90_000
    batch = _make_batch(21, L, 200, ["mixed", "plain", "clip", "noisy"])
    chunks = [(1, 45_000), (45_000, L - 2)]
    p = _params(max_mismatch_count=0, mismatch_window_size=20, min_sequence_identity=0.9)
    _check(batch, chunks, p)


@pytest.mark.parametrize("seed,kinds", [
    (41, ["plain", "clip", "short", "trailins"]),
    (42, ["bigindel", "plain", "mixed"]),
    (43, ["noisy", "dense", "plain"]),
])
def test_edge_reads_normcounts_against_oracle(seed, kinds):
    """The same hand-built alignments through the dense normcounts sweep."""
    from oracle import oracle as O
    from himut_amd import normcounts
    from himut_amd.caller import Worker
    L = 100_000
    batch = _make_batch(seed, L, 240, kinds)
    refseq = batch.refseq.encode("ascii")
    chunks = [(1, 30_000), (30_000, 30_700), (45_000, L - 2)]
    p = _params(min_bq=30, min_gq=5, max_mismatch_count=3, mismatch_window_size=12, min_ref_count=2, md_threshold=60,
                min_sequence_identity=0.9)
    order = {"A": ["C", "T", "G"], "T": ["G", "C", "A"], "G": ["T", "A", "C"], "C": ["A", "G", "T"]}
    o_ccs, o_ref, o_log = O.normcounts(batch, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
    w = Worker(0)
    try:
        w.configure(p["min_qv"], p["min_mapq"], p["qlen_lower_limit"], p["qlen_upper_limit"], p["min_sequence_identity"],
                    p["min_gq"], p["min_bq"], p["min_trim"], p["max_mismatch_count"], p["mismatch_window_size"],
                    p["md_threshold"], p["min_ref_count"], p["min_alt_count"], p["min_hap_count"],
                    p["germline_snv_prior"], False)
        ccs, rf, log = normcounts.norm_contig(w, batch, chunks, refseq, alt_order=order)
    finally:
        w.close()
    assert log == o_log
    assert ccs == o_ccs and rf == o_ref
    assert log[13] > 10_000
