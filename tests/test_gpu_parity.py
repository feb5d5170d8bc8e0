"""Parity of the HIP path (through the C ABI) with (a) golden vectors captured
from the reference and (b) the CPU oracle on fresh seeded inputs.  Bit-exact:
integer records and counters must be identical."""
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


def _configure(worker, p, phase):
    worker.configure(p["min_qv"], p["min_mapq"], p["qlen_lower_limit"], p["qlen_upper_limit"],
                     p["min_sequence_identity"], p["min_gq"], p["min_bq"], p["min_trim"], p["max_mismatch_count"],
                     p["mismatch_window_size"], p["md_threshold"], p["min_ref_count"], p["min_alt_count"],
                     p["min_hap_count"], p["germline_snv_prior"], phase)


def _run_hip(worker, batch, chunks, p, pon=None, com=None, phase_sets=None):
    _configure(worker, p, phase_sets is not None)
    return worker.call_contig(batch, chunks, pon, com, phase_sets)


def _diff(got, want, limit=5):
    out = []
    for i, (g, w) in enumerate(zip(got, want)):
        if g != w:
            out.append("#{}: got {} want {}".format(i, g, w))
            if len(out) >= limit:
                break
    return "\n".join(out)


@pytest.mark.parametrize("case", util.WORKER_CASES + util.PHASE_CASES)
def test_golden_cases(worker, case):
    from himut_amd import caller
    batch, exp = util.load_case(case)
    if case == "worker_flags":
        pass  # shared query names; non-phase, so allowed
    p = util.params_of(exp)
    pon = caller.site_keys([tuple(t) for t in exp["pon_set"]]) if "pon_set" in exp else None
    com = caller.site_keys([tuple(t) for t in exp["common_set"]]) if "common_set" in exp else None
    recs, log = _run_hip(worker, batch, util.chunks_of(exp), p, pon, com, util.phase_of(exp))
    got = caller.records_to_tuples(exp["contig"], recs)
    want = util.expected_tuples(exp)
    assert got == want, _diff(got, want) + " | log {} want {}".format(log, exp["log"])
    assert log == exp["log"]


def _oracle_vs_hip(worker, cfg, chunks=None, overrides=None, md=52, qlim=None, with_sets=False, phase_block=0):
    from oracle import oracle as O
    from himut_amd import caller, synth, util as hutil, vcflib
    s = synth.generate(cfg)
    b = s.batch
    if chunks is None:
        chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(overrides or {})
    ql, qu = qlim if qlim else (int(cfg.read_len_mean * 0.6), int(cfg.read_len_mean * 1.5))
    p.update(qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    pon = com = None
    if with_sets:
        rs = np.random.RandomState(cfg.seed)
        sites = [(int(x) + 1, chr(r), chr(a)) for x, r, a in zip(s.snp_pos, s.snp_ref, s.snp_alt)]
        extra = [(int(rs.randint(1, b.length)), "ACGT"[i], "ACGT"[j]) for i, j in rs.randint(0, 4, (3000, 2)) if i != j]
        pon = caller.site_keys(extra[::2] + sites[::3])
        com = caller.site_keys(extra[1::2] + sites[1::3])
    phase_sets = None
    if phase_block:
        import os
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            pv = os.path.join(d, "p.vcf")
            synth.write_phased_vcf(pv, s, block=phase_block)
            hb, hp, hs, c2c = vcflib.load_phased_hetsnps(pv, [b.name], {b.name: b.length})
        phase_sets = (dict(hb[b.name]), dict(hp[b.name]), dict(hs[b.name]))
        chunks = [(c[1], c[2]) for c in c2c[b.name]]
    orecs, olog = O.call(b, chunks, p, p["germline_snv_prior"], pon, com, phase_sets)
    hrecs, hlog = _run_hip(worker, b, chunks, p, pon, com, phase_sets)
    assert hlog == olog
    assert len(hrecs) == len(orecs)
    for name in ("tpos", "chunk", "phase_set", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"):
        assert np.array_equal(hrecs[name], orecs[name]), name
    return len(hrecs), hlog


def test_oracle_parity_1mb_config1(worker):
    """BASELINE.json configs[0]: 1 Mb contig, 30x, no side VCFs (5 reference chunks)."""
    from himut_amd.synth import SynthConfig
    n, log = _oracle_vs_hip(worker, SynthConfig(seed=1, contig_len=1_000_000, name="chr1"))
    assert n > 1000 and log[1] >= n


def test_config1_full_size_matches_reference(worker):
    """BASELINE.json configs[0] at full size against the records the reference itself produced
    for the same seeded input (tests/golden/config1_reference.json)."""
    from himut_amd import caller
    b, exp = util.load_config1_reference()
    recs, log = _run_hip(worker, b, util.chunks_of(exp), util.params_of(exp))
    got, want = caller.records_to_tuples(exp["contig"], recs), util.expected_tuples(exp)
    assert got == want, _diff(got, want)
    assert log == exp["log"]


def test_oracle_parity_sets_and_boundaries(worker):
    from himut_amd.synth import SynthConfig
    cfg = SynthConfig(seed=2, contig_len=450_000, read_len_mean=6000, read_len_sd=1500, read_len_min=1500,
                      read_len_max=12000, som_rate=1e-4, hetalt_frac=0.05, het_frac=0.6, name="chr2")
    _oracle_vs_hip(worker, cfg, with_sets=True)


def test_oracle_parity_deep_pile(worker):
    """Columns deeper than one LDS row batch (> 64 rows): the spill loop."""
    from himut_amd.synth import SynthConfig
    cfg = SynthConfig(seed=3, contig_len=60_000, depth=60.0, read_len_mean=4000, read_len_sd=800, read_len_min=1500,
                      read_len_max=8000, pile_frac=0.2, pile_mult=4.0, name="chr3")
    _oracle_vs_hip(worker, cfg, md=400)


def test_oracle_parity_phase(worker):
    from himut_amd.synth import SynthConfig
    cfg = SynthConfig(seed=4, contig_len=300_000, read_len_mean=8000, read_len_sd=1500, read_len_min=3000,
                      read_len_max=14000, snp_rate=2e-3, som_rate=1e-4, name="chr4")
    _oracle_vs_hip(worker, cfg, phase_block=40)


def test_oracle_parity_overlapping_region_chunks(worker):
    """--region_list style chunk lists: overlapping and unordered chunks share
    som_seen in list order (caller.py:268,347)."""
    from himut_amd.synth import SynthConfig
    cfg = SynthConfig(seed=5, contig_len=120_000, read_len_mean=5000, read_len_sd=900, read_len_min=2000,
                      read_len_max=9000, som_rate=2e-4, name="chr5")
    _oracle_vs_hip(worker, cfg, chunks=[(50_000, 90_000), (1000, 60_000), (59_990, 60_010), (100_000, 120_000)])


def test_pile_counts_match_oracle(worker):
    from oracle import oracle as O
    from himut_amd.synth import SynthConfig, generate
    cfg = SynthConfig(seed=6, contig_len=40_000, read_len_mean=3000, read_len_sd=600, read_len_min=1000,
                      read_len_max=6000, ins_rate=2e-3, del_rate=2e-3, name="chr6")
    b = generate(cfg).batch
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=1000, qlen_upper_limit=6000, md_threshold=52)
    _configure(worker, p, False)
    worker.ctx.push_reads(b)
    for (p0, p1) in ((0, 40_000), (12_345, 13_000), (39_000, 40_000)):
        hc, hb = worker.ctx.pile_counts(p0, p1)
        oc, ob = O.pile_counts(b, p0, p1)
        assert np.array_equal(hc, oc)
        assert np.array_equal(hb, ob)


def test_error_codes_mirror_reference_failures(worker):
    """Inputs the reference crashes on raise here too (no silent output)."""
    from himut_amd import _ffi
    from himut_amd.readbatch import batch_from_records
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=10, qlen_upper_limit=10000, md_threshold=52)
    seq = "ACGT" * 50

    def run(recs, chunks=((0, 1000),)):
        b = batch_from_records("c", 1000, recs)
        _configure(worker, p, False)
        return worker.call_contig(b, list(chunks))

    ok = dict(tstart=10, tend=210, seq=seq, bq=[93] * 200, cs=":200")
    cand = dict(tstart=10, tend=210, seq=seq[:100] + "T" + seq[101:], bq=[93] * 200, cs=":100*at:99")
    run([ok, cand])
    # query N inside a match: KeyError in the reference, wherever it sits in a read that some chunk fetches
    # (caller.py:57,299) -- in a candidate column or far from one
    with pytest.raises(_ffi.HimutError) as e:
        run([dict(ok, seq=seq[:100] + "N" + seq[101:]), cand])
    assert e.value.code == 4
    with pytest.raises(_ffi.HimutError) as e:
        run([dict(ok, seq=seq[:30] + "N" + seq[31:]), cand])
    assert e.value.code == 4
    with pytest.raises(_ffi.HimutError) as e:   # garbage in cs
        run([dict(ok, cs=":100~ac:100")])
    assert e.value.code == 3
    with pytest.raises(_ffi.HimutError) as e:   # start > end chunk
        run([ok], chunks=((500, 100),))
    assert e.value.code == 6
    # BQ 0 in a candidate column: ValueError (log10(0)) in the reference
    sub = dict(tstart=10, tend=210, seq="T" + seq[1:], bq=[93] * 200, cs="*at:199")
    recs = [dict(ok, bq=[0] + [93] * 199) for _ in range(3)] + [sub]
    p2 = dict(p, min_trim=0.0)
    b = batch_from_records("c", 1000, recs)
    _configure(worker, p2, False)
    with pytest.raises(_ffi.HimutError) as e:
        worker.call_contig(b, [(0, 1000)])
    assert e.value.code == 5


def test_query_base_outside_atgc_raises_only_where_the_reference_does(worker):
    """caller.py:57: base2idx has only ATGC, so an ALIGNED query base outside them (N, an IUPAC code) ends the reference with
    a KeyError as soon as a chunk fetches the read -- and only then: a soft-clipped or inserted base is never looked up, and a
    read no chunk fetches is never piled.  The oracle restates that (ORC_ERR_BASE); the reads are flagged once per pushed
    batch (k_flag_bases) and a flagged read's aligned bases are looked at by its capture wave."""
    from oracle import oracle as O
    from himut_amd import _ffi
    from himut_amd.readbatch import batch_from_records
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=10, qlen_upper_limit=10000, md_threshold=52)
    seq = "ACGT" * 50
    base = [dict(tstart=10, tend=210, seq=seq, bq=[93] * 200, cs=":200") for _ in range(4)]
    base.append(dict(tstart=10, tend=210, seq=seq[:100] + "T" + seq[101:], bq=[93] * 200, cs=":100*at:99"))

    def both(extra, chunks=((0, 1000),), want_error=False):
        recs = sorted(base + extra, key=lambda r: r["tstart"])
        b = batch_from_records("c", 3000, recs)
        _configure(worker, p, False)
        if want_error:
            with pytest.raises(O.OracleError) as oe:
                O.call(b, list(chunks), p, p["germline_snv_prior"])
            assert "KeyError" in str(oe.value)
            with pytest.raises(_ffi.HimutError) as e:
                worker.call_contig(b, list(chunks))
            assert e.value.code == 4
        else:
            orecs, olog = O.call(b, list(chunks), p, p["germline_snv_prior"])
            hrecs, hlog = worker.call_contig(b, list(chunks))
            assert hlog == olog and len(hrecs) == len(orecs) and np.array_equal(hrecs["tpos"], orecs["tpos"])

    n_mid = dict(tstart=10, tend=210, seq=seq[:150] + "N" + seq[151:], bq=[93] * 200, cs=":200")
    both([n_mid], want_error=True)                                        # aligned, far from any candidate
    both([dict(n_mid, seq=seq[:150] + "R" + seq[151:])], want_error=True)  # an IUPAC code is not in base2idx either
    both([dict(n_mid, seq=seq[:199] + "N")], want_error=True)              # the read's last aligned base
    # soft-clipped N (the leading 20 bases are clipped), inserted N: never looked up
    both([dict(tstart=10, tend=190, qstart=20, seq="N" * 20 + seq[:180], bq=[93] * 200, cs=":180")])
    both([dict(tstart=10, tend=208, seq=seq[:60] + "NN" + seq[60:198], bq=[93] * 200, cs=":60+nn:138")])
    # a read with an aligned N that no chunk fetches: never piled
    far = dict(tstart=2000, tend=2200, seq=seq[:150] + "N" + seq[151:], bq=[93] * 200, cs=":200")
    both([far])
    both([far], chunks=((0, 1000), (1500, 2001)), want_error=True)         # ... until a chunk reaches it (start < end and end > start)


def test_full_size_chr20_properties_and_prefix_parity(worker):
    """BASELINE.json configs[1] at full size (64.4 Mb, 30x): size-independent properties on
    the whole result, and bit-exact parity with the oracle on the first 8 Mb of chunks."""
    from oracle import oracle as O
    from himut_amd import bamlib, synth, util as hutil
    from himut_amd.readbatch import ReadBatch
    cfg = synth.SynthConfig(seed=2, contig_len=64_444_167, name="chr20")
    s = synth.generate(cfg)
    b = s.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    assert len(chunks) == 323 and chunks[0] == (1, 200000) and chunks[-1][1] == b.length - 2
    ql, qu, md = bamlib.get_thresholds({b.name: b}, [b.name], {b.name: b.length})
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    rs = np.random.RandomState(9)
    from himut_amd import caller
    sites = [(int(x) + 1, chr(r), chr(a)) for x, r, a in zip(s.snp_pos[::2], s.snp_ref[::2], s.snp_alt[::2])]
    extra = [(int(rs.randint(1, b.length)), "ACGT"[i], "ACGT"[j]) for i, j in rs.randint(0, 4, (6000, 2)) if i != j]
    pon, com = caller.site_keys(extra), caller.site_keys(sites)
    recs, log = _run_hip(worker, b, chunks, p, pon, com)
    recs2, log2 = _run_hip(worker, b, chunks, p, pon, com)
    assert log == log2 and np.array_equal(recs.view(np.uint8), recs2.view(np.uint8))      # idempotent
    # counters partition the candidates (caller.py:328-605)
    assert log[1] == log[2] + log[3] + log[4] + log[5] + log[6] + log[7]
    assert log[6] == log[8] + log[9] + log[10] + log[11] + log[12] + log[13] + log[14]
    # sorted by (tpos, ref, alt) in ASCII order, one record per key, statuses in range
    key = recs["tpos"].astype(np.int64) * 65536 + recs["ref"].astype(np.int64) * 256 + recs["alt"]
    assert np.all(np.diff(key) > 0)
    assert recs["status"].max() <= 10 and np.all(recs["flags"] == 0)
    d = recs["counts"]
    assert np.all(d[:, :4].sum(1) >= 1)
    # prefix parity: oracle on the reads of the first 40 chunks
    nch = 40
    end = chunks[nch - 1][1]
    n = int(np.searchsorted(b.tstart, end, side="left"))
    tot = int(b.qoff[n - 1] + ((int(b.qlen[n - 1]) + 31) & ~31))
    sub = ReadBatch(name=b.name, length=b.length, tstart=b.tstart[:n], tend=b.tend[:n], qstart=b.qstart[:n],
                    qlen=b.qlen[:n], mapq=b.mapq[:n], flag=b.flag[:n], qid=b.qid[:n], qoff=b.qoff[:n],
                    cs_off=b.cs_off[:n + 1], seq=b.seq[:tot // 2], bq=b.bq[:tot], cs=b.cs[:int(b.cs_off[n])], tp=b.tp[:n])
    orecs, olog = O.call(sub, chunks[:nch], p, p["germline_snv_prior"], pon, com)
    mine = recs[recs["chunk"] < nch]
    assert len(mine) == len(orecs)
    for name in ("tpos", "chunk", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"):
        assert np.array_equal(mine[name], orecs[name]), name
