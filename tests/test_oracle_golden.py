"""The CPU oracle (oracle/himut_oracle.c) against golden vectors captured from
the reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import util
from himut_amd.readbatch import batch_from_records


@pytest.mark.parametrize("case", util.WORKER_CASES + util.PHASE_CASES)
def test_worker_matches_reference(case):
    batch, exp = util.load_case(case)
    p = util.params_of(exp)
    pon = O.site_keys([tuple(t) for t in exp["pon_set"]]) if "pon_set" in exp else None
    com = O.site_keys([tuple(t) for t in exp["common_set"]]) if "common_set" in exp else None
    recs, log = O.call(batch, util.chunks_of(exp), p, p["germline_snv_prior"], pon, com, util.phase_of(exp))
    assert log == exp["log"]
    got = O.records_to_tuples(exp["contig"], recs)
    want = util.expected_tuples(exp)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


def test_germ_gt_vectors():
    exp = util.load_json("leaf_gtlib")
    idx = O.BASE2IDX
    for v in exp["vectors"]:
        gt, gq, state, pls = O.germ_gt(v["ref"], [idx[a] for a in v["alleles"]], v["bqs"])
        assert pls == v["pls"]
        assert (gt, gq, state) == (v["gt"], v["gq"], v["state"])
        # the reference hands get_germ_gq the two-letter genotype, so it equals gq
        assert v["germ_gq"] == v["gq"]


def test_lut_tables():
    exp = util.load_json("leaf_gtlib")["tables"]
    hom, het, err, logp = O.build_lut(1 / (10 ** 3))
    assert list(hom[1:94]) == exp["hom"]
    assert list(het[1:94]) == exp["het"]
    assert list(err[1:94]) == exp["err"]
    assert [logp[0], logp[1], logp[2], logp[3]] == [exp["prior"][k] for k in ("homref", "het", "hetalt", "homalt")]


def test_cs_tuples():
    exp = util.load_json("leaf_cs")
    for c in exp["cs"]:
        tend = c["tstart"] + sum(t[3] for t in c["tuples"])
        b = batch_from_records("c", 10 ** 6, [dict(tstart=c["tstart"], tend=tend, qstart=c["qstart"], seq=c["seq"],
                                                   bq=c["bq"], cs=c["cs"])])
        ops = O.cs_ops(b, 0)
        assert [(o[0], o[1], o[2]) for o in ops] == [(t[0], t[3], t[4]) for t in c["tuples"]]
        for o, t in zip(ops, c["tuples"]):
            if t[0] == 2:
                assert (o[3], o[4]) == (t[1], t[2])


NORM_CASES = ["norm_basic", "norm_sets", "norm_dense", "norm_softmask", "norm_phase", "norm_nsub", "norm_insins"]


def load_norm_case(case):
    """(batch, expected, params, refseq bytes, pon keys, common keys) of a normcounts fixture."""
    import numpy as np
    from oracle import oracle as O
    exp = util.load_json(case)
    with np.load(os.path.join(util.GOLDEN, case + ".npz")) as z:
        from himut_amd.readbatch import ReadBatch
        batch = ReadBatch.from_npz_dict(z)
        refseq = bytes(z["refseq"])
    p = util.params_of(exp)
    pon = O.site_keys([tuple(t) for t in exp["pon_set"]]) if "pon_set" in exp else None
    com = O.site_keys([tuple(t) for t in exp["common_set"]]) if "common_set" in exp else None
    return batch, exp, p, refseq, pon, com


@pytest.mark.parametrize("case", NORM_CASES)
def test_normcounts_oracle_matches_reference(case):
    """normcounts.get_callable_tricounts (non-phased): both trinucleotide dicts and the 14 counters."""
    from oracle import oracle as O
    batch, exp, p, refseq, pon, com = load_norm_case(case)
    ccs, rf, log = O.normcounts(batch, util.chunks_of(exp), p, refseq, p["germline_snv_prior"], pon, com,
                                alt_order=exp["alt_order"], non_human_sample=exp["non_human_sample"],
                                phase=util.phase_of(exp))
    assert log == exp["log"]
    assert {k: v for k, v in ccs.items() if v or k in O.TRI_LST} == \
        {k: v for k, v in exp["ccs_tri2count"].items() if v or k in O.TRI_LST}
    assert {k: v for k, v in rf.items() if v or k in O.TRI_LST} == \
        {k: v for k, v in exp["ref_tri2count"].items() if v or k in O.TRI_LST}


def test_normcounts_alt_order_decides_pon_vs_common():
    """norm_order: the panel of normals and the common SNPs each hold one alternative allele of the same positions; the
    reference was run under several PYTHONHASHSEED values and each distinct order of set("ATGC").difference(ref)
    (normcounts.py:367) gave its own PoN / common counters.  The oracle takes the order as an input and must
    reproduce every variant."""
    from oracle import oracle as O
    batch, exp, p, refseq, pon, com = load_norm_case("norm_order")
    assert len(exp["variants"]) >= 2 and len({(v["log"][11], v["log"][12]) for v in exp["variants"]}) >= 2
    for v in exp["variants"]:
        ccs, rf, log = O.normcounts(batch, util.chunks_of(exp), p, refseq, p["germline_snv_prior"], pon, com,
                                    alt_order=v["alt_order"], non_human_sample=exp["non_human_sample"])
        assert log == v["log"], v["hashseed"]
        assert {k: c for k, c in ccs.items() if c} == {k: c for k, c in v["ccs_tri2count"].items() if c}
        assert {k: c for k, c in rf.items() if c} == {k: c for k, c in v["ref_tri2count"].items() if c}


EDGE_CASES = ["edges_basic", "edges_lowq"]


@pytest.mark.parametrize("case", EDGE_CASES)
def test_edges_oracle_matches_reference(case):
    """phaselib.get_edges: the same edges in the same order with the same four counts."""
    batch, exp = util.load_case(case)
    hets = [tuple(h) for h in exp["hetsnps"]]
    edge_lst, e2c = O.edges(batch, hets, exp["min_bq"], exp["min_mapq"])
    assert [list(e) for e in edge_lst] == exp["edge_lst"]
    assert {"{},{}".format(*k): v for k, v in e2c.items()} == exp["edge2counts"]


def test_config1_full_size_matches_reference():
    """BASELINE.json configs[0] at full size: the oracle against the reference's own records."""
    from himut_amd import caller
    b, exp = util.load_config1_reference()
    recs, log = O.call(b, util.chunks_of(exp), util.params_of(exp), util.CALL_DEFAULTS["germline_snv_prior"], None, None, None)
    assert caller.records_to_tuples(exp["contig"], recs) == util.expected_tuples(exp)
    assert log == exp["log"]
