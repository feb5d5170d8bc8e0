"""Command lines of ``himut call`` and ``himut normcounts`` (reference: src/himut/parse_args.py:37-227,
502-692): same flag names, types and defaults, plus ``--devices`` for the GPUs to use."""
import argparse
import sys


def build_parser(program_version):
    parser = argparse.ArgumentParser(
        prog="himut",
        description="himut identifies high-confidence single molecule somatic single-base substitutions from "
                    "PacBio CCS reads (MI355X build of the `call` path)")
    parser.add_argument("-v", "--version", action="version", version="%(prog)s {}".format(program_version))
    sub = parser.add_subparsers(dest="sub", metavar="")
    p = sub.add_parser("call", help="detects somatic mutations from circular consensus sequence (CCS) reads",
                       formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("-i", "--bam", type=str, required=True,
                   help="minimap2 (parameters: -ax map-hifi --cs=short) aligned BAM file")
    p.add_argument("--ref", type=str, required=False, help="reference genome FASTA file")
    p.add_argument("--vcf", type=str, required=False, help="VCF file with germline mutations")
    p.add_argument("--phased_vcf", type=str, required=False, help="phased germline VCF file")
    p.add_argument("--common_snps", type=str, required=False, help="common SNPs VCF file")
    p.add_argument("--panel_of_normals", type=str, required=False, help="panel of normal VCF file")
    p.add_argument("--region", type=str, required=False, help="target chromosome")
    p.add_argument("--region_list", type=str, required=False, help="list of target chromosomes, one per line")
    p.add_argument("--min_qv", type=int, default=30, help="minimum read accuracy score")
    p.add_argument("--min_mapq", type=int, default=60, help="minimum mapping quality score")
    p.add_argument("--min_sequence_identity", type=float, default=0.99, help="minimum sequence identity")
    p.add_argument("--min_gq", type=int, default=20, help="minimum germline genotype quality score")
    p.add_argument("--min_bq", type=int, default=93, help="minimum base quality score")
    p.add_argument("--min_ref_count", type=int, default=3, help="minimum reference allele depth")
    p.add_argument("--min_alt_count", type=int, default=1, help="minimum alternative allele depth")
    p.add_argument("--min_hap_count", type=int, default=3, help="minimum h0 and h1 haplotype count")
    p.add_argument("--min_trim", type=float, default=0.01, help="proportion of the read ends to ignore")
    p.add_argument("--max_mismatch_count", type=int, default=0, help="maximum mismatches within the window")
    p.add_argument("--mismatch_window_size", type=int, default=20, help="mismatch window size")
    p.add_argument("--somatic_snv_prior", type=float, default=1 / (10 ** 6), help="somatic SNV prior")
    p.add_argument("--germline_snv_prior", type=float, default=1 / (10 ** 3), help="germline SNV prior")
    p.add_argument("--germline_indel_prior", type=float, default=1 / (10 ** 4), help="germline indel prior")
    p.add_argument("-t", "--threads", type=int, default=1, help="kept for the header; GPUs do the work")
    p.add_argument("--phase", required=False, action="store_true", help="phase somatic mutations")
    p.add_argument("--non_human_sample", required=False, action="store_true", help="human (default) or non-human sample")
    p.add_argument("--reference_sample", required=False, action="store_true", help="reads from the reference sample")
    p.add_argument("--create_panel_of_normal", required=False, action="store_true",
                   help="call substitutions with relaxed parameters for panel of normal preparation")
    p.add_argument("-o", "--output", type=str, required=True, help="VCF file to write the substitutions")
    p.add_argument("--devices", type=str, default="0", help="comma separated GPU ids (contigs are spread over them)")
    # himut normcounts (reference: parse_args.py:502-692)
    n = sub.add_parser("normcounts", help="normalises SBS96 mutation counts based on genome and read trinucleotide "
                                          "context counts", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    n.add_argument("-i", "--bam", type=str, required=True,
                   help="minimap2 (parameters: -ax map-hifi --cs=short) aligned BAM file")
    n.add_argument("--ref", type=str, required=True, help="reference FASTA file")
    n.add_argument("--sbs", type=str, required=True, help="himut VCF file to read somatic single base substitutions")
    n.add_argument("--vcf", type=str, required=False, help="VCF file with germline mutations")
    n.add_argument("--phased_vcf", type=str, required=False, help="phased germline VCF file")
    n.add_argument("--common_snps", type=str, required=False, help="common SNPs VCF file")
    n.add_argument("--panel_of_normals", type=str, required=False, help="panel of normal VCF file")
    n.add_argument("--region", type=str, required=False, help="target chromosome")
    n.add_argument("--region_list", type=str, required=False, help="list of target chromosomes, one per line")
    n.add_argument("--min_qv", type=int, default=30, help="minimum read accuracy score")
    n.add_argument("--min_mapq", type=int, default=60, help="minimum mapping quality score")
    n.add_argument("--min_sequence_identity", type=float, default=0.99, help="minimum sequence identity")
    n.add_argument("--min_gq", type=int, default=20, help="minimum germline genotype quality score")
    n.add_argument("--min_bq", type=int, default=93, help="minimum base quality score")
    n.add_argument("--min_ref_count", type=int, default=3, help="minimum reference allele depth")
    n.add_argument("--min_alt_count", type=int, default=1, help="minimum alternative allele depth")
    n.add_argument("--min_hap_count", type=int, default=3, help="minimum h0 and h1 haplotype count")
    n.add_argument("--min_trim", type=float, default=0.01, help="proportion of the read ends to ignore")
    n.add_argument("--mismatch_window", type=int, default=20, help="mismatch window size")
    n.add_argument("--max_mismatch_count", type=int, default=0, help="maximum mismatches within the window")
    n.add_argument("--somatic_snv_prior", type=float, default=1 / (10 ** 6), help="somatic SNV prior")
    n.add_argument("--germline_snv_prior", type=float, default=1 / (10 ** 3), help="germline SNV prior")
    n.add_argument("--germline_indel_prior", type=float, default=1 / (10 ** 4), help="germline indel prior")
    n.add_argument("-t", "--threads", type=int, default=1, help="kept for the command line record; GPUs do the work")
    n.add_argument("--phase", required=False, action="store_true", help="use phased reads only")
    n.add_argument("--non_human_sample", required=False, action="store_true", help="human (default) or non-human sample")
    n.add_argument("--reference_sample", required=False, action="store_true", help="reads from the reference sample")
    n.add_argument("-o", "--output", type=str, required=True, help="file to write the normalised SBS96 counts")
    n.add_argument("--devices", type=str, default="0", help="comma separated GPU ids (contigs are spread over them)")
    # himut phase (reference: parse_args.py:343-415)
    h = sub.add_parser("phase", help="returns phased hetsnps", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    h.add_argument("-i", "--bam", type=str, required=True,
                   help="minimap2 (parameters: -ax map-hifi --cs=short) aligned BAM file")
    h.add_argument("--vcf", type=str, required=True, help="deepvariant VCF file with germline mutations")
    h.add_argument("--region", type=str, required=False, help="target chromosome")
    h.add_argument("--region_list", type=str, required=False, help="list of target chromosomes separated by new line")
    h.add_argument("--min_bq", type=int, default=20, help="minimum base quality score threshold")
    h.add_argument("--min_mapq", type=int, default=20, help="minimum mapping quality score")
    h.add_argument("--min_p_value", type=float, default=0.0001, help="maximum binomial p-value of a phase consistent edge")
    h.add_argument("--min_phase_proportion", type=float, default=0.2, help="minimum proportion of phase consistent edges")
    h.add_argument("-t", "--threads", type=int, default=1, help="BGZF inflate threads; the GPU counts the edges")
    h.add_argument("-o", "--output", type=str, required=True, help="VCF file to write phased hetsnps")
    h.add_argument("--devices", type=str, default="0", help="comma separated GPU ids (the first one is used)")
    return parser


def parse_args(program_version, arguments=None):
    parser = build_parser(program_version)
    if arguments is None:
        arguments = sys.argv[1:]
    if len(arguments) == 0:          # parse_args.py:693-695: help and exit 0
        parser.print_help()
        parser.exit()
    return parser, parser.parse_args(arguments)
