"""`himut phase` (SURVEY §8f row 3 and its host side): hetSNP loading, graph / binomial test / haplotype blocks,
statistics and the phased-hetSNP VCF against fixtures captured from the reference
(tests/golden/make_golden.py phase_case).  CPU tests take the edge counts from the oracle; the GPU test takes
them from the device through the C ABI and runs the `himut phase` driver end to end, then `himut call --phase`
on its output."""
import os
import re

import numpy as np
import pytest

from tests import util

PHASE_CASES = ["phase_blocks", "phase_sparse"]


def _norm(text):
    return re.sub(r"##fileDate=\d+", "##fileDate=X", text)


def _write_inputs(tmp_path, exp):
    vcf = os.path.join(str(tmp_path), "{}.germline.vcf".format(exp["case"]))
    with open(vcf, "w") as o:
        o.write(exp["vcf_text"])
    return vcf


@pytest.mark.parametrize("case", PHASE_CASES)
def test_blocks_statistics_and_vcf_match_reference(case, tmp_path):
    from oracle import oracle as O
    from himut_amd import phaselib, vcflib
    batch, exp = util.load_case(case)
    exp["case"] = case
    vcf = _write_inputs(tmp_path, exp)
    chrom = exp["contig"]
    hetsnp_lst, hidx2hetsnp, hetsnp2hidx = vcflib.load_hetsnps(vcf, chrom, exp["length"])
    assert len(hetsnp_lst) == exp["statistics"][0] and hetsnp2hidx[hetsnp_lst[-1]] == len(hetsnp_lst) - 1
    assert hidx2hetsnp[0] == hetsnp_lst[0]
    edge_lst, e2c = O.edges(batch, hetsnp_lst, exp["min_bq"], exp["min_mapq"])
    e2c = {k: np.array(v) for k, v in e2c.items()}
    blocks = phaselib.build_haplotype_block(edge_lst, e2c, exp["min_p_value"], exp["min_phase_proportion"])
    assert [[[h, st] for h, st in b] for b in blocks] == exp["hblock_lst"]
    assert list(phaselib.get_hblock_statistics(blocks, hetsnp_lst)) == exp["statistics"]
    out = os.path.join(str(tmp_path), case + ".phased.vcf")
    vcflib.dump_phased_hetsnps("in.bam", vcf, chrom, None, exp["sizes"], exp["min_bq"], exp["min_mapq"], exp["min_p_value"],
                               exp["min_phase_proportion"], 1, [chrom], {chrom: blocks}, "1.0.4", out, "syn")
    got = open(out).read().replace(str(tmp_path) + "/", "")
    assert _norm(got) == _norm(exp["phased_vcf_text"])
    # what `himut call --phase` makes of it: the phase-set chunks the reference derived from its own output
    _, _, _, c2c = vcflib.load_phased_hetsnps(out, [chrom], exp["sizes"])
    assert [[int(x) for x in c[1:]] for c in c2c[chrom]] == exp["phase_chunks"]


def test_binomial_test_matches_scipy_reference_values():
    from himut_amd import phaselib
    # scipy.stats.binom_test(k, n, 0.5, "two-sided") values (exact method)
    assert phaselib.table2binom_test(np.array([5.0, 5.0, 0.0, 0.0])) == pytest.approx(2 * 0.5 ** 10, rel=1e-12)
    assert phaselib.table2binom_test(np.array([3.0, 2.0, 3.0, 2.0])) == pytest.approx(1.0, rel=1e-12)
    assert phaselib.table2binom_test(np.array([0.0, 1.0, 6.0, 7.0])) == pytest.approx(2 * (1 + 14) * 0.5 ** 14, rel=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case", PHASE_CASES)
def test_phase_driver_on_device_then_call_phase(case, tmp_path):
    from himut_amd import bamio, phaselib, vcflib
    from himut_amd.__main__ import main
    batch, exp = util.load_case(case)
    exp["case"] = case
    vcf = _write_inputs(tmp_path, exp)
    chrom = exp["contig"]
    bam = os.path.join(str(tmp_path), case + ".bam")
    bamio.write_bam(bam, [batch], sample="syn")
    out = os.path.join(str(tmp_path), case + ".phased.vcf")
    main(["phase", "-i", bam, "--vcf", vcf, "--region", chrom, "--min_p_value", str(exp["min_p_value"]),
          "--min_phase_proportion", str(exp["min_phase_proportion"]), "-o", out])
    got = open(out).read().replace(str(tmp_path) + "/", "").replace(case + ".bam", "in.bam")
    want = exp["phased_vcf_text"]
    # the driver's header lists the contigs of the BAM (one here; the fixture's size table has a second name)
    body = lambda t: [l for l in t.split("\n") if not l.startswith(("##contig", "##fileDate", "##source_version"))]
    assert body(got) == body(want)
    # blocks straight from the device edge counts
    blocks = {}
    phaselib.get_hblock(chrom, exp["length"], bam, vcf, exp["min_bq"], exp["min_mapq"], exp["min_p_value"],
                        exp["min_phase_proportion"], blocks, read_batch=batch)
    assert [[[h, st] for h, st in b] for b in blocks[chrom]] == exp["hblock_lst"]
    # and the phased file feeds `himut call --phase`
    called = os.path.join(str(tmp_path), case + ".call.vcf")
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        main(["call", "-i", bam, "--phased_vcf", out, "--phase", "--region", chrom, "--min_bq", "20", "-o", called])
    finally:
        os.chdir(cwd)
    lines = [l for l in open(called) if not l.startswith("#")]
    assert len(lines) > 10 and all(l.split("\t")[0] == chrom for l in lines)
