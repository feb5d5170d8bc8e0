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
