"""The command line mirrors the reference's: no arguments prints help and exits 0
(reference tests/test_main.py), `call` accepts the same flags with the same defaults."""
import pytest

from himut_amd.parse_args import parse_args


def test_no_arguments_exits_zero(capsys):
    with pytest.raises(SystemExit) as e:
        parse_args("x", [])
    assert e.value.code == 0
    assert "usage" in capsys.readouterr().out.lower()


def test_call_defaults_match_reference():
    _, o = parse_args("x", ["call", "-i", "a.bam", "-o", "o.vcf"])
    assert (o.min_qv, o.min_mapq, o.min_sequence_identity, o.min_gq, o.min_bq) == (30, 60, 0.99, 20, 93)
    assert (o.min_ref_count, o.min_alt_count, o.min_hap_count, o.min_trim) == (3, 1, 3, 0.01)
    assert (o.max_mismatch_count, o.mismatch_window_size, o.threads) == (0, 20, 1)
    assert (o.somatic_snv_prior, o.germline_snv_prior, o.germline_indel_prior) == (1e-6, 1e-3, 1e-4)
    assert not (o.phase or o.non_human_sample or o.reference_sample or o.create_panel_of_normal)
