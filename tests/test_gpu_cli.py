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


def test_call_then_normcounts_end_to_end(tmp_path):
    """`himut call` followed by `himut normcounts` on its output, two contigs: the table and norm.log equal what the
    golden-pinned host functions print from the ORACLE's per-contig results on the same inputs."""
    import numpy as np
    from himut_amd import __main__ as cli
    from himut_amd import bamio, normcounts as N, synth, util as hutil
    from oracle import oracle as O
    s1 = synth.generate(synth.SynthConfig(seed=61, contig_len=230_000, read_len_mean=6000, read_len_sd=1200,
                                          read_len_min=2000, read_len_max=12000, som_rate=2e-4, name="chr3"), want_ref=True)
    s2 = synth.generate(synth.SynthConfig(seed=62, contig_len=80_000, read_len_mean=6000, read_len_sd=1200,
                                          read_len_min=2000, read_len_max=12000, som_rate=2e-4, name="chr11"), want_ref=True)
    bam = str(tmp_path / "in.bam")
    bamio.write_bam(bam, [s1.batch, s2.batch], sample="SMP")
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as o:
        for s in (s1, s2):
            seq = bytes(s.ref).decode()
            o.write(">{}\n".format(s.batch.name))
            for i in range(0, len(seq), 70):
                o.write(seq[i:i + 70] + "\n")
    com = str(tmp_path / "common.vcf")
    pon = str(tmp_path / "pon.vcf")
    synth.write_common_snps_vcf(com, s1, seed=1, other_contig="chr11")
    synth.write_pon_vcf(pon, s1, seed=1, rate=2e-3)
    sbs = str(tmp_path / "calls.vcf")
    out = str(tmp_path / "norm.tsv")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        cli.main(["call", "-i", bam, "--common_snps", com, "--panel_of_normals", pon, "-o", sbs])
        cli.main(["normcounts", "-i", bam, "--ref", fa, "--sbs", sbs, "--common_snps", com, "--panel_of_normals", pon,
                  "-o", out])
    finally:
        os.chdir(cwd)
    # expected, from the oracle
    from himut_amd import vcflib
    refseq = N.read_fasta(fa)
    sizes = {"chr3": 230_000, "chr11": 80_000}
    chrom_lst, c2c = hutil.load_loci(None, None, sizes)
    ql, qu, md = N.get_thresholds(sbs)
    assert md == int(md)          # the header prints ceil(...) with one decimal (vcflib.py:183)
    p = dict(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99, min_gq=20,
             min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20, md_threshold=int(md),
             min_ref_count=3, min_alt_count=1, min_hap_count=3)
    batches = {"chr3": s1.batch, "chr11": s2.batch}
    ccs, ref, log = {}, {}, {}
    for c in chrom_lst:
        chunks = [(x[1], x[2]) for x in c2c[c]]
        ccs[c], ref[c], log[c] = O.normcounts(batches[c], chunks, p, refseq[c], 1 / (10 ** 3),
                                              O.site_keys(vcflib.load_pon(c, pon)),
                                              O.site_keys(vcflib.load_common_snp(c, com)),
                                              alt_order={b: list(N.BASE_SET.difference(b)) for b in "ATGC"})
    exp_out = str(tmp_path / "exp.tsv")
    cmd = open(out).readline().rstrip("\n")
    N.dump_normcounts(N.load_sbs96_counts(sbs, refseq, chrom_lst), N.get_genome_tricounts(refseq, chrom_lst), ref, ccs,
                      cmd, exp_out)
    assert open(out).read() == open(exp_out).read()
    N.dump_norm_log(chrom_lst, log, str(tmp_path / "exp.log"))
    assert open(tmp_path / "norm.log").read() == open(tmp_path / "exp.log").read()
    assert cmd.startswith("##himut_command=himut normcounts -i {} --ref {} --sbs {}".format(bam, fa, sbs))
    assert sum(log[c][13] for c in chrom_lst) > 1_000_000


def test_call_non_human_sample_uses_germline_priors(tmp_path):
    """--non_human_sample: the germline prior comes from the sample's own VCF and the FASTA
    (vcflib.get_germline_priors, caller.py:720-723), PoN and common SNPs are ignored (caller.py:248-262); records and
    counters equal the oracle's with that prior."""
    import numpy as np
    from himut_amd import __main__ as cli
    from himut_amd import bamio, bamlib, caller, synth, util as hutil, vcflib
    from oracle import oracle as O
    from tests import util
    s = synth.generate(synth.SynthConfig(seed=81, contig_len=210_000, read_len_mean=6000, read_len_sd=1200, read_len_min=2000,
                                         read_len_max=12000, snp_rate=4e-3, som_rate=2e-4, name="chr2"), want_ref=True)
    b = s.batch
    bam = str(tmp_path / "in.bam")
    bamio.write_bam(bam, [b], sample="SMP")
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as o:
        seq = bytes(s.ref).decode()
        o.write(">chr2\n")
        for i in range(0, len(seq), 70):
            o.write(seq[i:i + 70] + "\n")
    vcf = str(tmp_path / "germline.vcf")
    with open(vcf, "w") as o:
        o.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSMP\n")
        for p, r, a, g in zip(s.snp_pos, s.snp_ref, s.snp_alt, s.snp_gt):
            o.write("chr2\t{}\t.\t{}\t{}\t40\tPASS\t.\tGT\t{}\n".format(int(p) + 1, chr(r), chr(a), "0/1" if g in (1, 2) else "1/1"))
        o.write("chr2\t50\t.\tAC\tA\t40\tPASS\t.\tGT\t0/1\n")
    snv_prior, indel_prior = vcflib.get_germline_priors(["chr2"], fa, vcf, False)
    assert 0.002 <= snv_prior <= 0.006 and indel_prior == 5e-06
    out = str(tmp_path / "calls.vcf")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        cli.main(["call", "-i", bam, "--ref", fa, "--vcf", vcf, "--non_human_sample", "-o", out])
    finally:
        os.chdir(cwd)
    header = [l for l in open(out) if l.startswith("##himut_command")][0]
    assert "--germline_snv_prior {} --germline_indel_prior 5e-06".format(snv_prior) in header
    assert header.rstrip().endswith("--non_human_sample")
    sizes = {"chr2": b.length}
    chrom_lst, c2c = hutil.load_loci(None, None, sizes)
    ql, qu, md = bamlib.get_thresholds({"chr2": b}, chrom_lst, sizes)
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    orecs, olog = O.call(b, [(c[1], c[2]) for c in c2c["chr2"]], p, snv_prior, None, None, None)
    vcflib.dump_sbs(str(tmp_path / "exp.vcf"), "#H", chrom_lst, {"chr2": caller.records_to_tuples("chr2", orecs)})
    body = lambda path: [l for l in open(path) if not l.startswith("#")]
    assert body(out) == body(str(tmp_path / "exp.vcf")) and len(body(out)) > 100
    log_rows = [l.split() for l in open(tmp_path / "himut.log")]
    assert [int(r[1]) for r in log_rows[1:]] == [int(x) for x in olog]


def test_call_one_process_per_gpu_matches_single_process(tmp_path):
    """`himut call` under torch.distributed.run (two ranks; gloo, both on cuda:0 of the one-GPU box): every rank
    scans its LPT share of the contigs, rank 0 gathers the record buffers and writes the files -- byte-identical
    to the single-process run."""
    import subprocess
    import sys
    from himut_amd import __main__ as cli
    from himut_amd import bamio, synth
    samples = [synth.generate(synth.SynthConfig(seed=90 + k, contig_len=L, read_len_mean=6000, read_len_sd=1200,
                                                read_len_min=2000, read_len_max=12000, som_rate=2e-4, name=name),
                              want_ref=True)
               for k, (name, L) in enumerate([("chr1", 260_000), ("chr2", 150_000), ("chr10", 90_000), ("chrX", 40_000)])]
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as o:
        for s_ in samples:
            seq = bytes(s_.ref).decode()
            o.write(">{}\n".format(s_.batch.name))
            for i in range(0, len(seq), 70):
                o.write(seq[i:i + 70] + "\n")
    bam = str(tmp_path / "in.bam")
    bamio.write_bam(bam, [s.batch for s in samples], sample="SMP")
    com, pon = str(tmp_path / "common.vcf"), str(tmp_path / "pon.vcf")
    synth.write_common_snps_vcf(com, samples[0], seed=1, other_contig="chr2")
    synth.write_pon_vcf(pon, samples[1], seed=1, rate=2e-3)
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    args = ["call", "-i", bam, "--common_snps", com, "--panel_of_normals", pon, "-o"]
    cwd = os.getcwd()
    os.chdir(one)
    try:
        cli.main(args + [str(one / "calls.vcf")])
    finally:
        os.chdir(cwd)
    env = dict(os.environ, HIMUT_DIST_BACKEND="gloo", PYTHONPATH=cwd + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29655", "-m", "himut_amd"] + args +
                       [str(two / "calls.vcf")], cwd=str(two), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    strip = lambda p: [l for l in open(p) if not l.startswith(("##fileDate", "##himut_command"))]
    assert strip(one / "calls.vcf") == strip(two / "calls.vcf") and len(strip(two / "calls.vcf")) > 500
    # the same driver path over RCCL, as a group of one rank (broadcast, all-reduce and gather on device tensors)
    solo = tmp_path / "solo"
    solo.mkdir()
    env1 = dict(os.environ, HIMUT_DIST_SINGLE="1", PYTHONPATH=env["PYTHONPATH"])
    r1 = subprocess.run([sys.executable, "-m", "himut_amd"] + args + [str(solo / "calls.vcf")], cwd=str(solo), env=env1,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert strip(one / "calls.vcf") == strip(solo / "calls.vcf")
    # `himut normcounts` the same two ways
    nargs = ["normcounts", "-i", bam, "--ref", fa, "--sbs", str(one / "calls.vcf"), "--common_snps", com,
             "--panel_of_normals", pon, "-o"]
    os.chdir(one)
    try:
        cli.main(nargs + [str(one / "norm.tsv")])
    finally:
        os.chdir(cwd)
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29656", "-m", "himut_amd"] + nargs +
                        [str(two / "norm.tsv")], cwd=str(two), env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    table = lambda p: [l for l in open(p) if not l.startswith("##himut_command")]
    assert table(one / "norm.tsv") == table(two / "norm.tsv") and len(table(two / "norm.tsv")) > 90
    assert open(one / "norm.log").read() == open(two / "norm.log").read()
    assert open(one / "himut.log").read() == open(two / "himut.log").read()
    sm = "calls.single_molecule_mutations.vcf"
    assert strip(one / sm) == strip(two / sm)
