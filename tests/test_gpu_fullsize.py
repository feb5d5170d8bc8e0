"""BASELINE.json configs[3] and configs[4] at full size: a chr1-length contig (248 956 422 bp)
at 30x through the --phase path, and at 60x with 3x pile-ups (columns of >128 reads, depth above
md_threshold = 91).  Both exceed 2^32 read bases, so every 64-bit offset in the library is
exercised.  The oracle cannot finish these sizes in seconds, so each test checks
size-independent properties of the whole result and bit-exact parity with the oracle on a few
reference chunks (for config 5 the ones holding the deepest pile-ups)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

CHR1 = 248_956_422


@pytest.fixture(scope="module")
def worker():
    from himut_amd.caller import Worker
    w = Worker(0)
    yield w
    w.close()


def _sub_batch(b, idx):
    """The reads ``idx`` (ascending) of a batch, still pointing into the parent's seq / bq / cs bytes."""
    from himut_amd.readbatch import ReadBatch
    assert np.array_equal(b.qid[idx], idx)          # unique query names in the synthetic sample
    n = idx.shape[0]
    ln = b.cs_off[idx + 1] - b.cs_off[idx]
    cs_off = np.zeros(n + 1, np.int64)
    np.cumsum(ln, out=cs_off[1:])
    cs = np.concatenate([b.cs[b.cs_off[i]:b.cs_off[i + 1]] for i in idx]) if n else b.cs[:0]
    return ReadBatch(name=b.name, length=b.length, tstart=b.tstart[idx], tend=b.tend[idx], qstart=b.qstart[idx],
                     qlen=b.qlen[idx], mapq=b.mapq[idx], flag=b.flag[idx], qid=np.arange(n, dtype=np.int32),
                     qoff=b.qoff[idx], cs_off=cs_off, seq=b.seq, bq=b.bq, cs=cs, tp=b.tp[idx])


def _reads_for(b, chunks):
    """Indices of the reads that can overlap any of ``chunks`` (sorted by tstart; reads are <= 25 kb)."""
    parts = []
    for lo, hi in chunks:
        i0 = int(np.searchsorted(b.tstart, lo - 30_000, side="left"))
        i1 = int(np.searchsorted(b.tstart, hi + 1, side="left"))
        parts.append(np.arange(i0, i1, dtype=np.int64))
    return np.unique(np.concatenate(parts))


def _common_properties(recs, log):
    assert log[1] == log[2] + log[3] + log[4] + log[5] + log[6] + log[7]
    key = recs["tpos"].astype(np.int64) * 65536 + recs["ref"].astype(np.int64) * 256 + recs["alt"]
    assert np.all(np.diff(key) > 0)
    assert recs["status"].max() <= 11 and np.all(recs["flags"] == 0)
    assert np.all(recs["counts"][:, :4].sum(1) >= 1)


FIELDS = ("tpos", "phase_set", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum")


def _compare_window(worker, b, chunks, sel, p, full, phase_sets=None, pon=None, com=None):
    """HIP and oracle on the chunk sub-list ``sel`` (indices into ``chunks``); then the full
    run's records inside those chunks must be the same records (a chunk's result depends on
    earlier chunks only through som_seen at its first position, caller.py:325)."""
    from oracle import oracle as O
    from tests.test_gpu_parity import _run_hip
    sub_chunks = [chunks[k] for k in sel]
    sb = _sub_batch(b, _reads_for(b, sub_chunks))
    ps = None
    if phase_sets is not None:
        keep = {str(c[0]) for c in sub_chunks}
        ps = tuple({k: v for k, v in d.items() if k in keep} for d in phase_sets)
    orecs, olog = O.call(sb, sub_chunks, p, p["germline_snv_prior"], pon, com, ps)
    hrecs, hlog = _run_hip(worker, sb, sub_chunks, p, pon, com, ps)
    assert hlog == olog and len(hrecs) == len(orecs) and len(orecs) > 0
    for name in FIELDS + ("chunk",):
        assert np.array_equal(hrecs[name], orecs[name]), name
    starts = np.array([c[0] for c in sub_chunks])
    a = full[np.isin(full["chunk"], np.array(sel))]
    a = a[~np.isin(a["tpos"], starts)]
    h = hrecs[~np.isin(hrecs["tpos"], starts)]
    assert len(a) == len(h)
    for name in FIELDS:
        assert np.array_equal(a[name], h[name]), name
    return orecs


def test_config4_chr1_phase_full_size(worker):
    from himut_amd import bamlib, synth, vcflib
    import os
    import tempfile
    from tests.test_gpu_parity import _run_hip
    s = synth.generate(synth.SynthConfig(seed=4, contig_len=CHR1, name="chr1"))
    b = s.batch
    assert int(b.qlen.astype(np.int64).sum()) > 2 ** 32
    ql, qu, md = bamlib.get_thresholds({b.name: b}, [b.name], {b.name: b.length})
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    with tempfile.TemporaryDirectory() as d:
        pv = os.path.join(d, "p.vcf")
        synth.write_phased_vcf(pv, s, block=200)
        hb, hp, hs, c2c = vcflib.load_phased_hetsnps(pv, [b.name], {b.name: b.length})
    phase_sets = (dict(hb[b.name]), dict(hp[b.name]), dict(hs[b.name]))
    chunks = [(c[1], c[2]) for c in c2c[b.name]]
    assert len(chunks) > 700
    recs, log = _run_hip(worker, b, chunks, p, None, None, phase_sets)
    print("config4: reads", b.n, "bases", b.total_read_bases(), "chunks", len(chunks), "records", len(recs), "log", log)
    _common_properties(recs, log)
    # phased PASS records carry their chunk's phase set; nothing else does (caller.py:584-603)
    passed = recs["status"] == 0
    cs = np.array([c[0] for c in chunks], np.int32)
    assert passed.sum() > 1000 and np.array_equal(recs["phase_set"][passed], cs[recs["chunk"][passed]])
    assert np.all(recs["phase_set"][~passed] == -1)
    n = len(chunks)
    _compare_window(worker, b, chunks, [0, 1, n // 2, n - 2, n - 1], p, recs, phase_sets)


def test_config5_chr1_60x_pileups_full_size(worker):
    from himut_amd import bamlib, synth, util as hutil
    from tests.test_gpu_parity import _run_hip
    s = synth.generate(synth.SynthConfig(seed=5, contig_len=CHR1, depth=60.0, pile_frac=0.001, pile_mult=3.0, name="chr1"))
    b = s.batch
    assert int(b.qlen.astype(np.int64).sum()) > 3 * 2 ** 32
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    assert len(chunks) == 1245
    ql, qu, md = bamlib.get_thresholds({b.name: b}, [b.name], {b.name: b.length})
    assert md >= 91                               # bamlib.py:132-178 on 100 sampled windows
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    recs, log = _run_hip(worker, b, chunks, p)
    print("config5: reads", b.n, "bases", b.total_read_bases(), "records", len(recs), "md", md, "log", log)
    _common_properties(recs, log)
    depth = recs["counts"].sum(1) - recs["counts"][:, 4]
    assert depth.max() > 128                      # the wide-column path ran
    assert (recs["status"] == 10).sum() > 0       # HighDepth fired (caller.py:520-545)
    # the three chunks with the deepest candidate columns, plus the first and the last
    deep = np.unique(recs["chunk"][np.argsort(depth)[-200:]])[-3:]
    sel = sorted(set([0, len(chunks) - 1] + [int(k) for k in deep]))
    orecs = _compare_window(worker, b, chunks, sel, p, recs)
    od = orecs["counts"].sum(1) - orecs["counts"][:, 4]
    assert od.max() > 128
