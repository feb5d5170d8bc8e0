"""The call path (`himut call`'s worker, caller.py:208-642) with its thresholds away from the defaults, against the oracle:
the quality threshold on both sides of 93 and 128, the read filters switched off and at their extremes, trimming, the
mismatch window from 0 to 100 with up to 8 mismatches allowed, the genotype-quality / depth / count thresholds
(caller.py:310-317,349-550, bamlib.py:222-282), and min_hap_count 0 / 6 under --phase -- with base qualities of 94 to 255
in every other case (the reference's tables have 256 entries; a CCS read carries at most 93).  Bit-exact records and counters."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def worker():
    from himut_amd.caller import Worker
    w = Worker(0)
    yield w
    w.close()


_SWEEP = [
    dict(min_bq=0, min_qv=0),
    dict(min_bq=1, min_trim=0.0),
    dict(min_bq=60, max_mismatch_count=1),
    dict(min_bq=93, max_mismatch_count=2, mismatch_window_size=5),
    dict(min_bq=94, min_qv=93),
    dict(min_bq=127, min_trim=0.2),
    dict(min_bq=128, min_qv=0),
    dict(min_bq=200, min_qv=0, min_mapq=0),
    dict(min_bq=255, min_gq=0),
    dict(min_bq=30, mismatch_window_size=0),
    dict(min_bq=30, max_mismatch_count=8, mismatch_window_size=100, min_sequence_identity=0.0, min_mapq=0),
    dict(min_gq=0, min_ref_count=0, min_alt_count=0, md_threshold=25),
    dict(min_gq=99, min_ref_count=40, min_alt_count=3, md_threshold=10_000),
    dict(min_gq=60, min_alt_count=2, md_threshold=1),
]


def _high_qualities(b, seed, n):
    """A copy of the batch with n base qualities replaced by values of 94 .. 255."""
    from himut_amd.readbatch import ReadBatch
    rs = np.random.RandomState(seed)
    bq = b.bq.copy()
    idx = rs.randint(0, len(bq), n)
    bq[idx] = rs.randint(94, 256, n)
    # (the padding behind a read stays what it was: only bytes inside reads matter, and these are all inside or padding)
    return ReadBatch(name=b.name, length=b.length, tstart=b.tstart, tend=b.tend, qstart=b.qstart, qlen=b.qlen, mapq=b.mapq,
                     flag=b.flag, qid=b.qid, qoff=b.qoff, cs_off=b.cs_off, seq=b.seq, bq=bq, cs=b.cs, tp=b.tp)


def _compare(worker, b, chunks, p, pon=None, com=None, phase_sets=None):
    from oracle import oracle as O
    orecs, olog = O.call(b, chunks, p, p["germline_snv_prior"], pon, com, phase_sets)
    worker.configure(p["min_qv"], p["min_mapq"], p["qlen_lower_limit"], p["qlen_upper_limit"], p["min_sequence_identity"],
                     p["min_gq"], p["min_bq"], p["min_trim"], p["max_mismatch_count"], p["mismatch_window_size"],
                     p["md_threshold"], p["min_ref_count"], p["min_alt_count"], p["min_hap_count"], p["germline_snv_prior"],
                     phase_sets is not None)
    hrecs, hlog = worker.call_contig(b, chunks, pon, com, phase_sets)
    assert hlog == olog
    assert len(hrecs) == len(orecs)
    for name in ("tpos", "chunk", "phase_set", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"):
        assert np.array_equal(hrecs[name], orecs[name]), name
    return hrecs, hlog


@pytest.mark.parametrize("k", range(len(_SWEEP)))
def test_call_parameter_sweep_oracle_parity(worker, k):
    from oracle import oracle as O
    from himut_amd import synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=80 + k, contig_len=260_000, depth=32.0, sub_rate=1e-3, som_rate=2e-4,
                                         frac_noisy=0.05, frac_lowbq=0.05, frac_lowmapq=0.05, hetalt_frac=0.03, name="chrW"))
    b = s.batch
    if k & 1:
        b = _high_qualities(b, k, 20_000)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=60)
    p.update(_SWEEP[k])
    rs = np.random.RandomState(k)
    sites = [(int(x) + 1, chr(r), chr(a)) for x, r, a in zip(s.snp_pos, s.snp_ref, s.snp_alt)]
    extra = [(int(rs.randint(1, b.length)), "ACGT"[i], "ACGT"[j]) for i, j in rs.randint(0, 4, (2000, 2)) if i != j]
    pon = O.site_keys(extra[::2] + sites[::3])
    com = O.site_keys(extra[1::2] + sites[1::3])
    recs, log = _compare(worker, b, chunks, p, pon, com)
    assert log[1] > 0 or p["min_qv"] == 93          # (a mean quality of 93: no read proposes anything)


@pytest.mark.parametrize("min_hap,extra,high", [(0, {}, False), (6, {}, True), (3, dict(min_bq=128, min_qv=0), True),
                                                (1, dict(min_bq=40, max_mismatch_count=3, mismatch_window_size=60), False)])
def test_call_parameter_sweep_phase_oracle_parity(worker, tmp_path, min_hap, extra, high):
    from himut_amd import synth, vcflib
    s = synth.generate(synth.SynthConfig(seed=95 + min_hap, contig_len=300_000, read_len_mean=8000, read_len_sd=1500,
                                         read_len_min=3000, read_len_max=14000, snp_rate=2e-3, som_rate=2e-4, name="chrP"))
    b = s.batch
    if high:
        b = _high_qualities(b, 7 + min_hap, 20_000)
    pv = str(tmp_path / "p.vcf")
    synth.write_phased_vcf(pv, s, block=40)
    hb, hp, hs, c2c = vcflib.load_phased_hetsnps(pv, [b.name], {b.name: b.length})
    phase_sets = (dict(hb[b.name]), dict(hp[b.name]), dict(hs[b.name]))
    chunks = [(c[1], c[2]) for c in c2c[b.name]]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=4000, qlen_upper_limit=13000, md_threshold=60, min_hap_count=min_hap)
    p.update(extra)
    recs, log = _compare(worker, b, chunks, p, phase_sets=phase_sets)
    assert log[1] > 0


def test_call_reads_with_thousands_of_marked_positions(worker):
    """One error in 300 bases at 60x: a tenth of the reference positions carry some read's substitution, a read crosses two
    thousand of them -- the capture keeps 512 marked positions of its read at a time and fills the list again inside a
    window -- and reads span more than the 16 k positions one pass over the bitmap takes."""
    from himut_amd import synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=131, contig_len=150_000, depth=60.0, sub_rate=3e-3, som_rate=2e-4,
                                         read_len_mean=19000, read_len_sd=2500, read_len_min=9000, read_len_max=26000, name="chrM"))
    b = s.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=8000, qlen_upper_limit=30000, md_threshold=200, min_sequence_identity=0.9, max_mismatch_count=50)
    recs, log = _compare(worker, b, chunks, p)
    assert log[1] > 1000


def test_call_indel_heavy_reads(worker):
    """An insertion or a deletion every 250 bases: dozens of segments a read (the capture's segment window in LDS is walked
    and reloaded), long mismatch lists in the window filter -- against the oracle."""
    from himut_amd import synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=133, contig_len=150_000, depth=40.0, sub_rate=1e-3, ins_rate=2e-3, del_rate=2e-3,
                                         som_rate=2e-4, name="chrJ"))
    b = s.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=8000, qlen_upper_limit=30000, md_threshold=200, min_sequence_identity=0.9, max_mismatch_count=6,
             mismatch_window_size=30)
    recs, log = _compare(worker, b, chunks, p)
    assert log[1] > 100
