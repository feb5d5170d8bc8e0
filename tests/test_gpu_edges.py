"""phaselib.get_edges through the C ABI: the reference's golden vectors and the oracle
on a larger contig.  Bit-exact: the same edges, in the same order, with the same four counts."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["edges_basic", "edges_lowq"])
def test_edges_golden(case):
    from himut_amd import phaselib
    batch, exp = util.load_case(case)
    hets = [tuple(h) for h in exp["hetsnps"]]
    hidx = {h: i for i, h in enumerate(hets)}
    edge_lst, e2c = phaselib.get_edges(exp["contig"], None, exp["min_bq"], exp["min_mapq"], [h[0] for h in hets], hets, hidx,
                                       read_batch=batch)
    assert [list(e) for e in edge_lst] == exp["edge_lst"]
    assert {"{},{}".format(*k): [float(x) for x in v] for k, v in e2c.items()} == exp["edge2counts"]


def test_edges_oracle_parity_dense_snps():
    """2 Mb, hetSNPs dense enough that reads span more than 64 of them (several blocks of lanes per read)."""
    from oracle import oracle as O
    from himut_amd import phaselib, synth
    s = synth.generate(synth.SynthConfig(seed=71, contig_len=2_000_000, snp_rate=8e-3, name="chrH"))
    hets = sorted(set((int(p) + 1, chr(r), chr(a)) for p, r, a, g in zip(s.snp_pos, s.snp_ref, s.snp_alt, s.snp_gt)
                      if g in (1, 2)))
    hidx = {h: i for i, h in enumerate(hets)}
    o_lst, o_e2c = O.edges(s.batch, hets, 40, 20)
    edge_lst, e2c = phaselib.get_edges("chrH", None, 40, 20, [h[0] for h in hets], hets, hidx, read_batch=s.batch)
    assert O.edge_band(s.batch, [h[0] for h in hets]) > 64
    assert edge_lst == o_lst
    assert {k: [float(x) for x in v] for k, v in e2c.items()} == o_e2c
    assert len(edge_lst) > 100_000
