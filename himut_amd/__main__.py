#!/usr/bin/env python3
"""``python -m himut_amd call ...`` / ``normcounts ...`` / ``phase ...`` -- the `himut call`, `himut normcounts`
and `himut phase` entry points (reference: src/himut/__main__.py:15-63,115-146)."""
__version__ = "1.0.4+mi355x"

from himut_amd.parse_args import parse_args


def main(arguments=None):
    parser, options = parse_args(__version__, arguments)
    if options.sub == "call":
        from himut_amd import caller
        caller.call_somatic_substitutions(
            options.bam, options.ref, options.vcf, options.phased_vcf, options.common_snps, options.panel_of_normals,
            options.region, options.region_list, options.min_qv, options.min_mapq, options.min_sequence_identity,
            options.min_gq, options.min_bq, options.min_trim, options.max_mismatch_count, options.mismatch_window_size,
            options.min_ref_count, options.min_alt_count, options.min_hap_count, options.somatic_snv_prior,
            options.germline_snv_prior, options.germline_indel_prior, options.threads, options.phase,
            options.non_human_sample, options.reference_sample, options.create_panel_of_normal, __version__,
            options.output, devices=[int(d) for d in options.devices.split(",") if d != ""])
    elif options.sub == "normcounts":
        from himut_amd import normcounts
        normcounts.get_normcounts(
            options.bam, options.ref, options.sbs, options.vcf, options.phased_vcf, options.common_snps,
            options.panel_of_normals, options.region, options.region_list, options.min_qv, options.min_mapq,
            options.min_sequence_identity, options.min_gq, options.min_bq, options.min_trim, options.mismatch_window,
            options.max_mismatch_count, options.min_ref_count, options.min_alt_count, options.min_hap_count,
            options.somatic_snv_prior, options.germline_snv_prior, options.germline_indel_prior, options.threads,
            options.phase, options.non_human_sample, options.reference_sample, options.output,
            devices=[int(d) for d in options.devices.split(",") if d != ""])
    elif options.sub == "phase":
        from himut_amd import phaselib
        phaselib.get_chrom_hblock(
            options.bam, options.vcf, options.region, options.region_list, options.min_bq, options.min_mapq,
            options.min_p_value, options.min_phase_proportion, options.threads, __version__, options.output,
            devices=[int(d) for d in options.devices.split(",") if d != ""])
    else:
        parser.print_help()


if __name__ == "__main__":
    main()
