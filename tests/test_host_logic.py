"""Host-side mirrors (chunk geometry, VCF loaders, header, writers, LUT) against
golden vectors captured from the reference.  CPU only."""
import os
import re

import pytest

import numpy as np

from himut_amd import caller, gtlib, util as hutil, vcflib
from tests import util


def test_chunkloci_matches_reference():
    exp = util.load_json("leaf_cs")["chunkloci"]
    for L, want in exp.items():
        got = [list(c[1:]) for c in hutil.chunkloci(("c", 0, int(L)))]
        assert got == want


def test_lut_tables_match_reference():
    exp = util.load_json("leaf_gtlib")["tables"]
    hom, het, err, logp = gtlib.build_tables(1 / (10 ** 3))
    assert list(hom[1:94]) == exp["hom"]
    assert list(het[1:94]) == exp["het"]
    assert list(err[1:94]) == exp["err"]
    assert list(logp) == [exp["prior"][k] for k in ("homref", "het", "hetalt", "homalt")]
    assert np.isnan(hom[0])


def test_side_vcf_loaders(tmp_path):
    for case in ("worker_sets", "worker_dense_sets"):
        exp = util.load_json(case)
        c = tmp_path / "c.vcf"
        p = tmp_path / "p.vcf"
        c.write_text(exp["common_vcf"])
        p.write_text(exp["pon_vcf"])
        com = vcflib.load_common_snp(exp["contig"], str(c))
        pon = vcflib.load_pon(exp["contig"], str(p))
        assert sorted(list(t) for t in com) == exp["common_set"]
        assert sorted(list(t) for t in pon) == exp["pon_set"]
        # the inverted contig test: nothing from the called contig is kept
        assert all(line.split("\t")[0] != exp["contig"] or True for line in exp["common_vcf"].splitlines())


def test_phased_vcf_loader(tmp_path):
    for case in util.PHASE_CASES:
        exp = util.load_json(case)
        f = tmp_path / "ph.vcf"
        f.write_text(exp["phased_vcf"])
        hb, hp, hs, c2c = vcflib.load_phased_hetsnps(str(f), [exp["contig"]], {exp["contig"]: exp["length"]})
        assert [list(c[1:]) for c in c2c[exp["contig"]]] == exp["chunks"]
        assert dict(hb[exp["contig"]]) == exp["phase_sets"]["hbit"]
        assert dict(hp[exp["contig"]]) == exp["phase_sets"]["hpos"]
        assert {k: [list(t) for t in v] for k, v in hs[exp["contig"]].items()} == exp["phase_sets"]["hetsnp"]


def test_vcf_and_log_text(tmp_path):
    for case in util.WORKER_CASES + util.PHASE_CASES:
        exp = util.load_json(case)
        if "vcf_text" not in exp:
            continue
        recs = util.expected_tuples(exp)
        out = tmp_path / (case + ".vcf")
        phased = case in util.PHASE_CASES
        (vcflib.dump_phased_sbs if phased else vcflib.dump_sbs)(str(out), "#HEADER", [exp["contig"]],
                                                                {exp["contig"]: recs})
        assert out.read_text() == exp["vcf_text"]
        sm = str(out).replace(".vcf", ".single_molecule_mutations.vcf")
        assert open(sm).read() == exp["sm_vcf_text"]
        logf = tmp_path / "himut.log"
        vcflib.dump_call_log([exp["contig"]], {exp["contig"]: exp["log"]}, path=str(logf))
        assert logf.read_text() == exp["log_text"]


def _norm(h):
    return re.sub(r"##fileDate=\d+", "##fileDate=X", h)


def test_vcf_header_matches_reference():
    exp = util.load_json("vcf_header")
    got = vcflib.get_himut_vcf_header(
        "/fake/header.bam", None, None, None, None, {"chr7": 30000, "chr10": 5, "chr2": 7}, "c.vcf", "p.vcf", 30, 60,
        2000, 4100, 0.99, 20, 93, 0.01, 0, 20, 52, 3, 1, 3, 1, 1 / (10 ** 6), 1 / (10 ** 3), 1 / (10 ** 4), False,
        False, False, False, "1.0.4", "out.vcf", "syn")
    assert _norm(got) == _norm(exp["header"])
    got = vcflib.get_himut_vcf_header(
        "/fake/header.bam", None, "ph.vcf", "chr7", None, {"chr7": 30000}, "c.vcf", "p.vcf", 30, 60, 2000, 4100, 0.99,
        20, 93, 0.01, 0, 20, 52, 3, 1, 3, 4, 1 / (10 ** 6), 1 / (10 ** 3), 1 / (10 ** 4), True, False, False, False,
        "1.0.4", "out.vcf", "syn")
    assert _norm(got) == _norm(exp["header_phase"])


def test_records_to_tuples_roundtrip_on_golden():
    """Oracle records -> tuples through the PRODUCT's converter equal the reference tuples."""
    from oracle import oracle as O
    for case in ("worker_dense", "worker_dense_sets"):
        batch, exp = util.load_case(case)
        p = util.params_of(exp)
        pon = O.site_keys([tuple(t) for t in exp["pon_set"]]) if "pon_set" in exp else None
        com = O.site_keys([tuple(t) for t in exp["common_set"]]) if "common_set" in exp else None
        recs, _ = O.call(batch, util.chunks_of(exp), p, p["germline_snv_prior"], pon, com, None)
        assert caller.records_to_tuples(exp["contig"], recs) == util.expected_tuples(exp)
        if pon is not None:
            assert np.array_equal(caller.site_keys([tuple(t) for t in exp["pon_set"]]), pon)


def test_thresholds_match_reference():
    from himut_amd import bamlib, synth
    for c in util.load_json("thresholds")["cases"]:
        cfg = synth.SynthConfig(**c["cfg"])
        b = synth.generate(cfg).batch
        got = bamlib.get_thresholds({"chr9": b}, ["chr9"], {"chr9": cfg.contig_len})
        assert got == (c["qlen_lower_limit"], c["qlen_upper_limit"], c["md_threshold"])


def test_bgz_side_vcf_loaders(tmp_path):
    """.bgz side files (BGZF = a series of gzip members): the loaders read them without a tabix index and, as the
    reference's tabix variants do, on the contig itself -- so the common-SNP set differs from the plain .vcf one."""
    import gzip
    from himut_amd import vcflib
    exp = util.load_json("worker_sets")
    common = tmp_path / "c.vcf"
    pon = tmp_path / "p.vcf"
    common.write_text(exp["common_vcf"])
    pon.write_text(exp["pon_vcf"])

    def bgz(src, dst):
        # two gzip members, like two BGZF blocks
        data = open(src, "rb").read()
        half = data.rfind(b"\n", 0, len(data) // 2) + 1
        with open(dst, "wb") as o:
            o.write(gzip.compress(data[:half]))
            o.write(gzip.compress(data[half:]))

    bgz(common, tmp_path / "c.vcf.bgz")
    bgz(pon, tmp_path / "p.vcf.bgz")
    chrom = exp["contig"]
    assert vcflib.load_bgz_pon(chrom, str(tmp_path / "p.vcf.bgz")) == vcflib.load_pon(chrom, str(pon))
    got = vcflib.load_bgz_common_snp(chrom, str(tmp_path / "c.vcf.bgz"))
    # same filter as the plain loader, but on this contig instead of on every other one
    want = set()
    for line in exp["common_vcf"].splitlines():
        if line.startswith("#"):
            continue
        f = line.split()
        if f[0] == chrom and f[6] == "PASS" and "," not in f[4] and len(f[3]) == 1 and len(f[4]) == 1:
            want.add((int(f[1]), f[3], f[4]))
    assert got == want and len(got) > 0


def test_normcounts_host_side_matches_reference(tmp_path):
    """Everything of `himut normcounts` around the worker, against the reference's own functions
    (tests/golden/norm_host.json): thresholds from the SBS header, SBS96 counts, genome trinucleotide counts, the
    command line, the output table and norm.log."""
    from himut_amd import normcounts as N
    exp = util.load_json("norm_host")
    fa = tmp_path / "ref.fa"
    fa.write_text(exp["fasta_text"])
    sbs = tmp_path / "calls.vcf"
    sbs.write_text(exp["sbs_vcf_text"])
    refseq = N.read_fasta(str(fa))
    chrom = exp["contig"]
    assert list(N.get_thresholds(str(sbs))) == exp["thresholds"]
    counts = N.load_sbs96_counts(str(sbs), refseq, [chrom])
    assert counts == exp["sbs96_counts"] and list(counts) == list(exp["sbs96_counts"]) == N.SBS96_LST
    tri = N.get_chrom_tricount(refseq[chrom])
    assert tri == exp["chrom_tricount"]
    assert N.get_genome_tricounts(refseq, [chrom]) == tri
    args = (exp["bam"], exp["fa"], exp["sbs"])
    assert N.get_normcounts_cmdline(*args, None, None, 30, 60, 0.99, 20, 93, 0.01, 20, 0, 3, 1, 3, "common.vcf",
                                    "pon.vcf", 1e-6, 1e-3, 1e-4, 4, False, False, False, exp["out"]) == exp["cmdline"]
    assert N.get_normcounts_cmdline(*args, "g.vcf", "p.vcf", 30, 60, 0.99, 20, 93, 0.01, 20, 0, 3, 1, 3, "c.vcf",
                                    "n.vcf", 1e-6, 1e-3, 1e-4, 2, True, False, False, exp["out"]) == exp["cmdline_phase"]
    assert N.get_normcounts_cmdline(*args, "g.vcf", "p.vcf", 30, 60, 0.99, 20, 93, 0.01, 20, 0, 3, 1, 3, None, None,
                                    1e-6, 1e-3, 1e-4, 2, True, True, True, exp["out"]) == exp["cmdline_nonhuman"]
    out = tmp_path / "norm.tsv"
    N.dump_normcounts(counts, tri, {chrom: exp["ref_tri2count"]}, {chrom: exp["ccs_tri2count"]}, exp["cmdline"], str(out))
    assert out.read_text() == exp["normcounts_tsv"]
    log = tmp_path / "norm.log"
    N.dump_norm_log([chrom], {chrom: exp["log"]}, str(log))
    assert log.read_text() == exp["norm_log_text"]


def test_record_formatter_matches_reference_vcf_text(tmp_path):
    """The host library's VCF body printer against the text the REFERENCE's writer produced for the golden cases
    (records come from the oracle, which the other tests pin to the reference's tuples)."""
    from oracle import oracle as O
    from himut_amd import caller
    for case in ["worker_basic", "worker_dense_sets", "worker_flags", "worker_phase", "worker_phase_dense"]:
        batch, exp = util.load_case(case)
        p = util.params_of(exp)
        pon = O.site_keys([tuple(t) for t in exp["pon_set"]]) if "pon_set" in exp else None
        com = O.site_keys([tuple(t) for t in exp["common_set"]]) if "common_set" in exp else None
        recs, _ = O.call(batch, util.chunks_of(exp), p, p["germline_snv_prior"], pon, com, util.phase_of(exp))
        phased = util.phase_of(exp) is not None
        out = tmp_path / (case + ".vcf")
        vcflib.dump_records(str(out), "#HEADER", [exp["contig"]], {exp["contig"]: recs}, phased)
        assert out.read_text() == exp["vcf_text"], case
        assert (tmp_path / (case + ".single_molecule_mutations.vcf")).read_text() == exp["sm_vcf_text"], case
        # and the python printer agrees line by line
        ref_out = tmp_path / (case + ".py.vcf")
        (vcflib.dump_phased_sbs if phased else vcflib.dump_sbs)(str(ref_out), "#HEADER", [exp["contig"]],
                                                                  {exp["contig"]: caller.records_to_tuples(exp["contig"], recs)})
        assert ref_out.read_text() == out.read_text()


def test_germline_priors_match_reference(tmp_path):
    """--non_human_sample: vcflib.load_germline_counts / get_germline_priors / util.get_truncated_float against
    values captured from the reference (tests/golden/make_golden.py germline_priors_case)."""
    from himut_amd import util as hutil, vcflib
    exp = util.load_json("germline_priors")
    fa, vcf = os.path.join(str(tmp_path), "gp.fa"), os.path.join(str(tmp_path), "gp.germline.vcf")
    open(fa, "w").write(exp["fasta_text"])
    open(vcf, "w").write(exp["vcf_text"])
    for c in exp["cases"]:
        assert list(vcflib.load_germline_counts(vcf, c["chrom_lst"])) == c["counts"]
        if isinstance(c["priors"], str):
            with pytest.raises((ValueError, IndexError)) as e:
                vcflib.get_germline_priors(c["chrom_lst"], fa, vcf, c["reference_sample"])
            assert type(e.value).__name__ == c["priors"]
        else:
            assert list(vcflib.get_germline_priors(c["chrom_lst"], fa, vcf, c["reference_sample"])) == c["priors"]
    for f, want in exp["truncated"]:
        assert hutil.get_truncated_float(f) == want
    # a .vcf.bgz is the same records in gzip members
    import gzip
    bgz = vcf + ".bgz"
    with gzip.open(bgz, "wt") as o:
        o.write(exp["vcf_text"])
    c = exp["cases"][2]
    assert list(vcflib.load_germline_counts(bgz, c["chrom_lst"])) == c["counts"]


def test_bgz_phased_and_sbs_files_read_like_plain_ones(tmp_path):
    """.bgz inputs are series of gzip members: the phased-hetSNP loader and the SBS-file readers of normcounts give
    the same answers for a compressed copy (the reference reads them through tabix / cyvcf2)."""
    import gzip
    from himut_amd import normcounts as N
    exp = util.load_json("phase_blocks")
    plain = os.path.join(str(tmp_path), "phased.vcf")
    open(plain, "w").write(exp["phased_vcf_text"])
    bgz = plain + ".bgz"
    with gzip.open(bgz, "wt") as o:
        o.write(exp["phased_vcf_text"])
    a = vcflib.load_phased_hetsnps(plain, [exp["contig"]], exp["sizes"])
    b = vcflib.load_phased_hetsnps(bgz, [exp["contig"]], exp["sizes"])
    assert a[3] == b[3] and {k: dict(v) for k, v in a[0].items()} == {k: dict(v) for k, v in b[0].items()}
    assert {k: dict(v) for k, v in a[2].items()} == {k: dict(v) for k, v in b[2].items()}
    host = util.load_json("norm_host")
    sbs = os.path.join(str(tmp_path), "calls.vcf")
    open(sbs, "w").write(host["sbs_vcf_text"])
    with gzip.open(sbs + ".bgz", "wt") as o:
        o.write(host["sbs_vcf_text"])
    assert N.get_thresholds(sbs) == N.get_thresholds(sbs + ".bgz")
    fa = os.path.join(str(tmp_path), "ref.fa")
    open(fa, "w").write(host["fasta_text"])
    refseq = N.read_fasta(fa)
    chroms = list(refseq)
    assert N.load_sbs96_counts(sbs, refseq, chroms) == N.load_sbs96_counts(sbs + ".bgz", refseq, chroms)


def test_read_fasta_takes_a_record_start_only_at_a_line_start(tmp_path):
    """A '>' inside a description is text (pyfastx, which the reference reads the genome with, splits records at line
    starts): the header's first word is the name, the sequence keeps its case, blank lines and CRs go."""
    from himut_amd import normcounts as N
    fa = tmp_path / "g.fa"
    fa.write_bytes(b">chr1 a>b len=8\nACGT\r\nacgt\n\n>chr2\tdesc >x\nNNAC\n>chr3\n")
    assert N.read_fasta(str(fa)) == {"chr1": "ACGTacgt", "chr2": "NNAC", "chr3": ""}
    fb = tmp_path / "h.fa"
    fb.write_bytes(b"\n\n>c1 x>y\nAC\nGT\n")
    assert N.read_fasta(str(fb)) == {"c1": "ACGT"}
