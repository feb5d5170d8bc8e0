"""End to end through the command line: synthetic BAM + side VCFs -> VCF files and
himut.log, compared with the CPU oracle run on the same inputs and printed by the same
(golden-pinned) writer."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_call_two_contigs_with_side_vcfs(tmp_path):
    from himut_amd import __main__ as cli
    from himut_amd import bamio, bamlib, caller, synth, util as hutil, vcflib
    from oracle import oracle as O
    s1 = synth.generate(synth.SynthConfig(seed=51, contig_len=260_000, read_len_mean=6000, read_len_sd=1200,
                                          read_len_min=2000, read_len_max=12000, som_rate=1e-4, name="chr2"))
    s2 = synth.generate(synth.SynthConfig(seed=52, contig_len=90_000, read_len_mean=6000, read_len_sd=1200,
                                          read_len_min=2000, read_len_max=12000, som_rate=1e-4, name="chr10"))
    bam = str(tmp_path / "in.bam")
    bamio.write_bam(bam, [s2.batch, s1.batch], sample="SMP")      # @SQ order differs from natural order
    com = str(tmp_path / "common.vcf")
    pon = str(tmp_path / "pon.vcf")
    synth.write_common_snps_vcf(com, s1, seed=1, other_contig="chr10")
    synth.write_pon_vcf(pon, s1, seed=1, rate=2e-3)
    out = str(tmp_path / "out.vcf")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        cli.main(["call", "-i", bam, "--common_snps", com, "--panel_of_normals", pon, "-o", out])
    finally:
        os.chdir(cwd)
    # expected: oracle per contig with the same host-side preparation
    batches = {"chr2": s1.batch, "chr10": s2.batch}
    sizes = {"chr10": 90_000, "chr2": 260_000}
    chrom_lst, c2c = hutil.load_loci(None, None, sizes)
    assert chrom_lst == ["chr2", "chr10"]
    ql, qu, md = bamlib.get_thresholds(batches, chrom_lst, sizes)
    p = dict(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99, min_gq=20,
             min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20, md_threshold=md,
             min_ref_count=3, min_alt_count=1, min_hap_count=3)
    want, wlog = {}, {}
    for c in chrom_lst:
        chunks = [(x[1], x[2]) for x in c2c[c]]
        recs, log = O.call(batches[c], chunks, p, 1 / (10 ** 3), caller.site_keys(vcflib.load_pon(c, pon)),
                           caller.site_keys(vcflib.load_common_snp(c, com)))
        want[c] = O.records_to_tuples(c, recs)
        wlog[c] = log
    exp = str(tmp_path / "exp.vcf")
    header = open(out).read().split("#CHROM")[0]
    hdr_full = header + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSMP"
    vcflib.dump_sbs(exp, hdr_full, chrom_lst, want)
    assert open(out).read() == open(exp).read()
    assert open(out.replace(".vcf", ".single_molecule_mutations.vcf")).read() == \
        open(exp.replace(".vcf", ".single_molecule_mutations.vcf")).read()
    vcflib.dump_call_log(chrom_lst, wlog, path=str(tmp_path / "exp.log"))
    assert open(tmp_path / "himut.log").read() == open(tmp_path / "exp.log").read()
    assert "##himut_command=himut call -i {}".format(bam) in header
    assert sum(len(v) for v in want.values()) > 500
