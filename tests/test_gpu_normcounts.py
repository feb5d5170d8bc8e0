"""normcounts sweep through the C ABI against the reference's golden
vectors and against the CPU oracle on fresh inputs.  Bit-exact: both trinucleotide dicts
and the 14 counters."""
import numpy as np
import pytest

from tests import util
from tests.test_oracle_golden import NORM_CASES, load_norm_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def worker():
    from himut_amd.caller import Worker
    w = Worker(0)
    yield w
    w.close()


def _configure(worker, p, phase=False):
    worker.configure(p["min_qv"], p["min_mapq"], p["qlen_lower_limit"], p["qlen_upper_limit"],
                     p["min_sequence_identity"], p["min_gq"], p["min_bq"], p["min_trim"], p["max_mismatch_count"],
                     p["mismatch_window_size"], p["md_threshold"], p["min_ref_count"], p["min_alt_count"],
                     p["min_hap_count"], p["germline_snv_prior"], phase)


@pytest.mark.parametrize("case", NORM_CASES)
def test_normcounts_golden(worker, case):
    from himut_amd import normcounts
    batch, exp, p, refseq, pon, com = load_norm_case(case)
    _configure(worker, p, util.phase_of(exp) is not None)
    ccs, rf, log = normcounts.norm_contig(worker, batch, util.chunks_of(exp), refseq, pon, com,
                                          exp["non_human_sample"], exp["alt_order"], phase_sets=util.phase_of(exp))
    assert log == exp["log"]
    assert ccs == {k: int(v) for k, v in exp["ccs_tri2count"].items()}
    assert rf == {k: int(v) for k, v in exp["ref_tri2count"].items()}


def test_normcounts_alt_order_variants(worker):
    """norm_order (see test_oracle_golden): every order python gave set("ATGC").difference(ref) in the reference runs,
    with the PoN / common counters that order produced."""
    from himut_amd import normcounts
    batch, exp, p, refseq, pon, com = load_norm_case("norm_order")
    _configure(worker, p, False)
    for v in exp["variants"]:
        ccs, rf, log = normcounts.norm_contig(worker, batch, util.chunks_of(exp), refseq, pon, com,
                                              exp["non_human_sample"], v["alt_order"])
        assert log == v["log"], v["hashseed"]
        assert {k: c for k, c in ccs.items() if c} == {k: int(c) for k, c in v["ccs_tri2count"].items() if c}
        assert {k: c for k, c in rf.items() if c} == {k: int(c) for k, c in v["ref_tri2count"].items() if c}


@pytest.mark.parametrize("seed,length,chunks", [
    (31, 300_000, None),                                     # reference chunking, two chunks
    (32, 120_000, [(500, 40_000), (40_000, 41_000), (90_000, 119_000)]),   # gaps and a tiny chunk
])
def test_normcounts_oracle_parity(worker, seed, length, chunks):
    from oracle import oracle as O
    from himut_amd import normcounts, synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=seed, contig_len=length, name="chrN"), want_ref=True)
    refseq = bytes(s.ref)
    if chunks is None:
        chunks = [(c[1], c[2]) for c in hutil.chunkloci((s.batch.name, 0, s.batch.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=52)
    rs = np.random.RandomState(seed)
    sites = [(int(x) + 1, chr(r), chr(a)) for x, r, a in zip(s.snp_pos, s.snp_ref, s.snp_alt)]
    extra = [(int(rs.randint(1, length)), "ACGT"[i], "ACGT"[j]) for i, j in rs.randint(0, 4, (2000, 2)) if i != j]
    pon = O.site_keys(extra[::2] + sites[::3])
    com = O.site_keys(extra[1::2] + sites[1::3])
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    o_ccs, o_ref, o_log = O.normcounts(s.batch, chunks, p, refseq, p["germline_snv_prior"], pon, com, alt_order=order)
    _configure(worker, p)
    ccs, rf, log = normcounts.norm_contig(worker, s.batch, chunks, refseq, pon, com, False, order)
    assert log == o_log
    assert ccs == o_ccs and rf == o_ref
    assert log[13] > 0 and log[11] + log[12] > 0


def _golden_with(worker, case, **dbg):
    from himut_amd import normcounts
    batch, exp, p, refseq, pon, com = load_norm_case(case)
    _configure(worker, p, util.phase_of(exp) is not None)
    worker.ctx.debug_normcounts(**dbg)
    try:
        ccs, rf, log = normcounts.norm_contig(worker, batch, util.chunks_of(exp), refseq, pon, com,
                                              exp["non_human_sample"], exp["alt_order"], phase_sets=util.phase_of(exp))
        reran = worker.ctx.stats()["reran"]
        redo = worker.ctx.stats()["column_slots"]
    finally:
        worker.ctx.debug_normcounts()
    assert log == exp["log"]
    assert ccs == {k: int(v) for k, v in exp["ccs_tri2count"].items()}
    assert rf == {k: int(v) for k, v in exp["ref_tri2count"].items()}
    return reran, redo


@pytest.mark.parametrize("case", ["norm_dense", "norm_phase", "norm_nsub"])
def test_normcounts_golden_tile_sweep(worker, case):
    """k_norm_tile (cells built in LDS by the workgroup) stays in the library: it does the tiles k_norm_quad leaves alone,
    and the whole contig when a list was too short.  The whole contig through it (himut_debug_normcounts), the same
    golden vectors."""
    _golden_with(worker, case, sweep=1)


@pytest.mark.parametrize("case", ["norm_dense", "norm_sets", "norm_phase"])
def test_normcounts_left_over_positions_do_not_fit(worker, case):
    """k_norm_quad hands the positions it does not classify itself (a column with another allele) to k_norm_dirty through
    a list with a part per workgroup, sized for one position in four; when a part is too short the sweep is repeated once
    with the room its counters ask for, and the context keeps that room.  A list with one entry per part: the same golden
    vectors, one repeat."""
    reran, _ = _golden_with(worker, case, dirty_cap=1)
    assert reran == 1


@pytest.mark.parametrize("case,slots", [("norm_dense", 1), ("norm_sets", 2), ("norm_phase", 1), ("norm_nsub", 1), ("norm_dense", 4)])
def test_normcounts_pool_runs_out(worker, case, slots):
    """A wave of k_norm_quad keeps the sums of the alleles that are not the reference's in a pool of 64 accumulators for its
    256 columns; a wave that needs more leaves its tile to k_norm_tile through a list.  With one to four slots nearly every
    tile of the goldens goes that way: the same vectors, no repeat of the contig."""
    reran, redo = _golden_with(worker, case, pool_slots=slots)
    assert reran == 0 and redo > 0


def test_normcounts_noisy_contig_goes_tile_by_tile_to_the_tile_kernel(worker):
    """Reads with one error in 80 bases at 35x: a third of the positions hold another allele, more than a wave of
    k_norm_quad has accumulators for (64 of its 256 columns), so nearly every tile is listed for k_norm_tile -- the list
    has room for every tile of the contig: no repeat of the contig, the oracle's counts."""
    from oracle import oracle as O
    from himut_amd import normcounts, synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=46, contig_len=150_000, depth=35.0, sub_rate=1.2e-2, name="chrZ"), want_ref=True)
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((s.batch.name, 0, s.batch.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=60, min_sequence_identity=0.9)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    o_ccs, o_ref, o_log = O.normcounts(s.batch, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
    _configure(worker, p)
    ccs, rf, log = normcounts.norm_contig(worker, s.batch, chunks, refseq, alt_order=order)
    st = worker.ctx.stats()
    assert log == o_log and ccs == o_ccs and rf == o_ref
    assert st["reran"] == 0 and st["column_slots"] > 150_000 // 256 // 2


def test_normcounts_indel_heavy_reads(worker):
    """An insertion or a deletion every 250 bases (and a substitution every 1000): a read's piece over a tile is cut in two or
    three, nearly every word of the bit array lies near a mismatch entry, reads have more segments than the plan's first
    look takes in -- against the oracle."""
    from oracle import oracle as O
    from himut_amd import normcounts, synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=47, contig_len=120_000, depth=40.0, sub_rate=1e-3, ins_rate=2e-3, del_rate=2e-3,
                                         name="chrI"), want_ref=True)
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((s.batch.name, 0, s.batch.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=80, min_sequence_identity=0.9, max_mismatch_count=3)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    o_ccs, o_ref, o_log = O.normcounts(s.batch, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
    _configure(worker, p)
    ccs, rf, log = normcounts.norm_contig(worker, s.batch, chunks, refseq, alt_order=order)
    assert log == o_log and ccs == o_ccs and rf == o_ref
    assert log[13] > 0


def test_normcounts_soft_clips_and_failing_reads(worker):
    """Long soft clips (the sweep's loads start behind them), a fifth of the reads failing the quality filter, a fifth the
    mapping-quality filter (piled, not counted), against the oracle."""
    from oracle import oracle as O
    from himut_amd import normcounts, synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=44, contig_len=80_000, depth=35.0, frac_softclip=0.6, softclip_max=5000,
                                         frac_lowbq=0.2, frac_lowmapq=0.2, name="chrF"), want_ref=True)
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((s.batch.name, 0, s.batch.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=60)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    o_ccs, o_ref, o_log = O.normcounts(s.batch, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
    _configure(worker, p)
    ccs, rf, log = normcounts.norm_contig(worker, s.batch, chunks, refseq, alt_order=order)
    assert log == o_log and ccs == o_ccs and rf == o_ref
    assert log[13] > 0


def test_normcounts_qualities_of_128_and_more(worker):
    """Qualities up to 255 in passing and in failing reads, against the oracle (its tables have 256 entries like the
    reference's; round 2's sweep kept the callable bit in bit 7 of the quality byte and sent such a contig down a slower
    path); then the same context takes an ordinary contig again."""
    from oracle import oracle as O
    from himut_amd import normcounts, synth, util as hutil
    from himut_amd.readbatch import ReadBatch
    s = synth.generate(synth.SynthConfig(seed=35, contig_len=150_000, name="chrQ"), want_ref=True)
    b = s.batch
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=52)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    rs = np.random.RandomState(35)
    bq = b.bq.copy()
    for r in rs.choice(len(b.qlen), 40, replace=False):          # a few bases each of forty reads
        o = int(b.qoff[r]) + rs.randint(0, int(b.qlen[r]), 25)
        bq[o] = rs.randint(128, 256, 25)
    hi = ReadBatch(name=b.name, length=b.length, tstart=b.tstart, tend=b.tend, qstart=b.qstart, qlen=b.qlen, mapq=b.mapq,
                   flag=b.flag, qid=b.qid, qoff=b.qoff, cs_off=b.cs_off, seq=b.seq, bq=bq, cs=b.cs, tp=b.tp)
    _configure(worker, p)
    for batch in (hi, b, hi):
        o_ccs, o_ref, o_log = O.normcounts(batch, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
        ccs, rf, log = normcounts.norm_contig(worker, batch, chunks, refseq, alt_order=order)
        assert log == o_log and ccs == o_ccs and rf == o_ref
        assert log[13] > 0


@pytest.mark.parametrize("seed,depth,extra", [
    (41, 150.0, {}),                                          # more than 64 rows over a quarter: several row batches
    (42, 40.0, dict(ins_rate=4e-3, del_rate=4e-3, sub_rate=5e-3, frac_softclip=0.5, softclip_max=300)),   # few spanning rows
    (43, 90.0, dict(pile_frac=0.05, pile_mult=8.0, read_len_mean=3000.0, read_len_sd=800.0, read_len_min=600,
                    read_len_max=6000)),                      # short reads, piles several hundred deep
])
def test_normcounts_deep_and_ragged_piles_oracle_parity(worker, seed, depth, extra):
    """k_norm_quad's row batches (64 reads of the window index at a time) and its general rows (an indel or a read end
    inside a wave's 256 positions) against the oracle where they are the rule, not the exception."""
    from oracle import oracle as O
    from himut_amd import normcounts, synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=seed, contig_len=60_000, depth=depth, name="chrD", **extra), want_ref=True)
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((s.batch.name, 0, s.batch.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=500, qlen_upper_limit=30000, md_threshold=1000, min_sequence_identity=0.9)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    o_ccs, o_ref, o_log = O.normcounts(s.batch, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
    _configure(worker, p)
    ccs, rf, log = normcounts.norm_contig(worker, s.batch, chunks, refseq, alt_order=order)
    assert log == o_log
    assert ccs == o_ccs and rf == o_ref
    assert log[1] > 0


_SWEEP = [
    dict(min_bq=0), dict(min_bq=1, min_trim=0.0), dict(min_bq=60, max_mismatch_count=1), dict(min_bq=93, max_mismatch_count=2, mismatch_window_size=5),
    dict(min_bq=30, max_mismatch_count=5, mismatch_window_size=50), dict(min_bq=127, min_trim=0.2), dict(min_bq=128), dict(min_bq=200, min_qv=0),
    dict(min_bq=20, mismatch_window_size=0), dict(min_gq=0, min_ref_count=0, md_threshold=25), dict(min_gq=99, min_ref_count=40),
    dict(min_qv=93, min_mapq=0), dict(min_sequence_identity=0.0, min_mapq=0, max_mismatch_count=8, mismatch_window_size=100),
]


@pytest.mark.parametrize("k", range(len(_SWEEP)))
def test_normcounts_parameter_sweep_oracle_parity(worker, k):
    """The filters' parameters away from their defaults (quality thresholds on both sides of 128, mismatch windows from 0
    to 100 with up to 8 mismatches allowed, trimming, genotype-quality and depth thresholds), qualities up to 255 in every
    other case, against the oracle."""
    from oracle import oracle as O
    from himut_amd import normcounts, synth, util as hutil
    from himut_amd.readbatch import ReadBatch
    s = synth.generate(synth.SynthConfig(seed=60 + k, contig_len=50_000, depth=32.0, sub_rate=1e-3, frac_noisy=0.05,
                                         name="chrW"), want_ref=True)
    b = s.batch
    if k & 1:
        rs = np.random.RandomState(k)
        bq = b.bq.copy()
        idx = rs.randint(0, len(bq), 3000)
        bq[idx] = rs.randint(94, 256, 3000)
        b = ReadBatch(name=b.name, length=b.length, tstart=b.tstart, tend=b.tend, qstart=b.qstart, qlen=b.qlen, mapq=b.mapq,
                      flag=b.flag, qid=b.qid, qoff=b.qoff, cs_off=b.cs_off, seq=b.seq, bq=bq, cs=b.cs, tp=b.tp)
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=60)
    p.update(_SWEEP[k])
    order = {"A": ["G", "C", "T"], "T": ["A", "G", "C"], "G": ["T", "C", "A"], "C": ["T", "A", "G"]}
    o_ccs, o_ref, o_log = O.normcounts(b, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
    _configure(worker, p)
    ccs, rf, log = normcounts.norm_contig(worker, b, chunks, refseq, alt_order=order)
    assert log == o_log
    assert ccs == o_ccs and rf == o_ref


def test_normcounts_phase_oracle_parity(worker, tmp_path):
    """--phase: chunks are the phase-set spans, reads need haplotype 0/1 there, positions need both haplotypes."""
    from oracle import oracle as O
    from himut_amd import normcounts, synth, vcflib
    s = synth.generate(synth.SynthConfig(seed=33, contig_len=400_000, snp_rate=2e-3, name="chrP"), want_ref=True)
    b = s.batch
    pv = str(tmp_path / "p.vcf")
    synth.write_phased_vcf(pv, s, block=40)
    hb, hp, hs, c2c = vcflib.load_phased_hetsnps(pv, [b.name], {b.name: b.length})
    phase_sets = (dict(hb[b.name]), dict(hp[b.name]), dict(hs[b.name]))
    chunks = [(c[1], c[2]) for c in c2c[b.name]]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=52, min_hap_count=6)
    refseq = bytes(s.ref)
    o_ccs, o_ref, o_log = O.normcounts(b, chunks, p, refseq, p["germline_snv_prior"], phase=phase_sets)
    _configure(worker, p, True)
    ccs, rf, log = normcounts.norm_contig(worker, b, chunks, refseq, phase_sets=phase_sets)
    assert log == o_log
    assert ccs == o_ccs and rf == o_ref
    assert log[2] > 0 and log[13] > 0


def test_normcounts_full_size_properties_and_prefix_parity(worker):
    """BASELINE configs[1] size (chr20-sized contig, 30x): the counters must add up (every counted base lands in
    exactly one class) and the first chunks must match the oracle run on the reads under them."""
    from oracle import oracle as O
    from himut_amd import bamlib, normcounts, synth, util as hutil
    from himut_amd.readbatch import ReadBatch
    L = 64_444_167
    s = synth.generate(synth.SynthConfig(seed=2, contig_len=L, name="chr20"), want_ref=True)
    b = s.batch
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    ql, qu, md = bamlib.get_thresholds({b.name: b}, [b.name], {b.name: b.length})
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    _configure(worker, p)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    ccs, rf, log = normcounts.norm_contig(worker, b, chunks, refseq, alt_order=order)
    assert log[1] == log[2] + log[3] + log[4] + log[5] + log[6]
    assert log[6] == sum(log[7:14])
    assert sum(ccs.values()) == log[13] and log[13] > 1_000_000_000
    assert sorted(ccs) == sorted(O.TRI_LST) and all(v > 0 for v in rf.values())
    # prefix parity: 12 chunks (2.4 Mb); the counters and dicts of a prefix are those of the oracle on the reads under it
    nch = 12
    end = chunks[nch - 1][1]
    n = int(np.searchsorted(b.tstart, end, side="left"))
    tot = int(b.qoff[n - 1] + ((int(b.qlen[n - 1]) + 31) & ~31))
    sub = ReadBatch(name=b.name, length=b.length, tstart=b.tstart[:n], tend=b.tend[:n], qstart=b.qstart[:n],
                    qlen=b.qlen[:n], mapq=b.mapq[:n], flag=b.flag[:n], qid=b.qid[:n], qoff=b.qoff[:n],
                    cs_off=b.cs_off[:n + 1], seq=b.seq[:tot // 2], bq=b.bq[:tot], cs=b.cs[:int(b.cs_off[n])], tp=b.tp[:n])
    o_ccs, o_ref, o_log = O.normcounts(sub, chunks[:nch], p, refseq, p["germline_snv_prior"], alt_order=order)
    h_ccs, h_ref, h_log = normcounts.norm_contig(worker, sub, chunks[:nch], refseq, alt_order=order)
    assert h_log == o_log and h_ccs == o_ccs and h_ref == o_ref


def test_ref_tricounts_match_reference_golden(worker):
    """reflib.get_chrom_tricount on the device against the reference's own counts (norm_host golden: N and
    lower-case stretches) and against the numpy host version on a large random string."""
    from himut_amd import normcounts as N
    exp = util.load_json("norm_host")
    seq = "".join(l.strip() for l in exp["fasta_text"].splitlines() if not l.startswith(">"))
    assert N.get_chrom_tricount_device(worker.ctx, seq) == exp["chrom_tricount"]
    rs = np.random.RandomState(7)
    big = bytes(rs.choice(np.frombuffer(b"ACGTACGTACGTNacgt", np.uint8), 5_000_000))
    assert N.get_chrom_tricount_device(worker.ctx, big) == N.get_chrom_tricount(big)


def test_sbs96_counts_on_device_match_reference(worker, tmp_path):
    """mutlib.load_sbs96_counts / get_sbs96 with the classification and counting on the device
    (himut_sbs96_counts) against the reference's own counts (tests/golden/norm_host.json), and against the host mirror
    on a string with N and lower-case stretches where classes are dropped or raise."""
    from himut_amd import normcounts as N
    exp = util.load_json("norm_host")
    fa = tmp_path / "ref.fa"
    fa.write_text(exp["fasta_text"])
    sbs = tmp_path / "calls.vcf"
    sbs.write_text(exp["sbs_vcf_text"])
    refseq = N.read_fasta(str(fa))
    chrom = exp["contig"]

    def ctx_for(c):
        chars, cls = N.tri_classes(refseq[c])
        worker.ctx.set_reference(refseq[c], cls, len(chars))
        return worker.ctx
    got = N.load_sbs96_counts_device(ctx_for, str(sbs), refseq, [chrom])
    assert got == exp["sbs96_counts"] and list(got) == N.SBS96_LST and sum(got.values()) > 0
    # direct: every (ref, alt) at every position of a string with N / soft-masked stretches, bins against the host mirror
    rs = np.random.RandomState(5)
    seq = "".join(rs.choice(list("ACGT"), 4000))
    seq = seq[:500] + "N" * 7 + seq[507:900] + seq[900:960].lower() + seq[960:]
    ref = {"c": seq}
    pos, rr, aa = [], [], []
    want = {k: 0 for k in N.SBS96_LST}
    n_drop = n_key = 0
    for p in range(0, len(seq) - 1):
        r = seq[p]
        if r not in "ACGT":
            continue
        for a in "ACGT":
            if a == r:
                continue
            k = N.get_sbs96("c", p, r, a, ref)
            pos.append(p); rr.append(ord(r)); aa.append(ord(a))
            if "N" in k:
                n_drop += 1
            elif k in want:
                want[k] += 1
            else:
                n_key += 1
    chars, cls = N.tri_classes(seq)
    worker.ctx.set_reference(seq, cls, len(chars))
    h = worker.ctx.sbs96_counts(pos, rr, aa)
    subs = N.SUB_LST
    dev = {"{}[{}]{}".format("ACGT"[u], subs[s6], "ACGT"[d]): int(h[s6 * 16 + u * 4 + d]) for s6 in range(6) for u in range(4) for d in range(4)}
    assert dev == want and int(h[96]) == n_drop and int(h[97]) == n_key and n_drop > 0 and n_key > 0 and int(h[98]) == 0
    assert int(worker.ctx.sbs96_counts([len(seq) - 1], [ord("C")], [ord("T")])[98]) == 1          # IndexError in the reference


def test_call_and_normcounts_interleaved_on_one_context(worker):
    """A call run leaves its scalars and position bitmap empty for the next call run; the normcounts sweep and the edge
    counts use the same buffers in between: every run gives what it gives on a fresh context, in any order."""
    from himut_amd import normcounts, synth, util as hutil
    s = synth.generate(synth.SynthConfig(seed=35, contig_len=260_000, name="chrI"), want_ref=True)
    s2 = synth.generate(synth.SynthConfig(seed=36, contig_len=900_000, name="chrJ"))
    refseq = bytes(s.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((s.batch.name, 0, s.batch.length))]
    chunks2 = [(c[1], c[2]) for c in hutil.chunkloci((s2.batch.name, 0, s2.batch.length))]
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=52)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}

    def call(b, ch):
        _configure(worker, p)
        recs, log = worker.call_contig(b, ch)
        return recs.tobytes(), list(log)

    def norm():
        _configure(worker, p)
        return normcounts.norm_contig(worker, s.batch, chunks, refseq, None, None, False, order)

    c1, n1, c2 = call(s.batch, chunks), norm(), call(s2.batch, chunks2)
    assert len(c1[0]) > 64 * 100 and n1[2][13] > 0
    for k in range(3):
        assert norm() == n1
        assert call(s2.batch, chunks2) == c2            # a longer contig: more bitmap than the run before left empty
        assert call(s.batch, chunks) == c1
        worker.ctx.run()                                  # the same reads again: nothing pushed, nothing cleared up front
        assert (worker.ctx.records().tobytes(), list(worker.ctx.log())) == c1
        assert norm() == n1
        assert call(s.batch, chunks) == c1


def test_normcounts_query_base_outside_atgc_in_a_fetched_read(worker):
    """normcounts.py:117 piles every base of every read a chunk fetches (:289): an aligned query base outside ATGC is a KeyError
    there even when it lies outside the chunk's own positions -- and none when it is soft-clipped or when no chunk fetches the
    read.  The oracle restates that; here k_read_live looks at the reads k_flag_bases has flagged."""
    from oracle import oracle as O
    from himut_amd import _ffi, normcounts
    from himut_amd.readbatch import batch_from_records
    rs = np.random.RandomState(3)
    refseq = bytes(rs.choice(np.frombuffer(b"ACGT", np.uint8), 3000))
    ref = refseq.decode()
    p = dict(util.CALL_DEFAULTS)
    p.update(qlen_lower_limit=10, qlen_upper_limit=10000, md_threshold=60, min_trim=0.0)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    reads = [dict(tstart=400 + 7 * i, tend=1000 + 7 * i, seq=ref[400 + 7 * i:1000 + 7 * i], bq=[93] * 600, cs=":600") for i in range(12)]

    def both(extra, chunks, want_error):
        recs = sorted(reads + extra, key=lambda r: r["tstart"])
        b = batch_from_records("c", 3000, recs)
        _configure(worker, p)
        if want_error:
            with pytest.raises(O.OracleError):
                O.normcounts(b, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
            with pytest.raises(_ffi.HimutError) as e:
                normcounts.norm_contig(worker, b, chunks, refseq, alt_order=order)
            assert e.value.code == 4
        else:
            o = O.normcounts(b, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
            h = normcounts.norm_contig(worker, b, chunks, refseq, alt_order=order)
            assert h[2] == o[2] and h[0] == o[0] and h[1] == o[1]

    s = ref[100:700]
    n_out = dict(tstart=100, tend=700, seq=s[:50] + "N" + s[51:], bq=[93] * 600, cs=":600")     # the N at position 150
    both([n_out], [(500, 1500)], True)           # fetched by the chunk (tend > 500), the N in front of the chunk
    both([n_out], [(800, 1500)], False)          # not fetched
    both([dict(tstart=120, tend=700, qstart=20, seq="N" * 20 + ref[120:700], bq=[93] * 600, cs=":580")], [(500, 1500)], False)
