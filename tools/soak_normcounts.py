#!/usr/bin/env python3
"""Soak run of the normcounts sweep: three contigs of different sizes and settings (plain, high mismatch rate with
qualities of 128 and more in it, another filter setting) alternate through ONE context N times; every pass must return
the counts of the first pass of its contig, and the first pass those of the oracle.  Prints one line."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=300)
    a = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime)
    import numpy as np
    from himut_amd import caller, normcounts, synth, util as hutil
    from himut_amd.readbatch import ReadBatch
    from oracle import oracle as O
    from tests import util
    w = caller.Worker(0)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    cases = []
    for seed, L, extra, over in ((71, 400_000, {}, {}), (72, 90_000, dict(sub_rate=2e-3, ins_rate=1e-3, del_rate=1e-3), dict(min_bq=60, max_mismatch_count=1)),
                                 (73, 230_000, dict(depth=70.0), dict(min_gq=30, mismatch_window_size=40))):
        s = synth.generate(synth.SynthConfig(seed=seed, contig_len=L, name="chr{}".format(seed), **extra), want_ref=True)
        b = s.batch
        if seed == 72:
            rs = np.random.RandomState(seed)
            bq = b.bq.copy()
            idx = rs.randint(0, len(bq), 2000)
            bq[idx] = rs.randint(128, 256, 2000)
            b = ReadBatch(name=b.name, length=b.length, tstart=b.tstart, tend=b.tend, qstart=b.qstart, qlen=b.qlen, mapq=b.mapq,
                          flag=b.flag, qid=b.qid, qoff=b.qoff, cs_off=b.cs_off, seq=b.seq, bq=bq, cs=b.cs, tp=b.tp)
        chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
        p = dict(util.CALL_DEFAULTS)
        p.update(qlen_lower_limit=9000, qlen_upper_limit=22500, md_threshold=120)
        p.update(over)
        refseq = bytes(s.ref)
        want = O.normcounts(b, chunks, p, refseq, p["germline_snv_prior"], alt_order=order)
        cases.append((b, chunks, p, refseq, want))
    t0 = time.time()
    for i in range(a.passes):
        k = (i * 7 + i // 5) % 3
        b, chunks, p, refseq, want = cases[k]
        w.configure(p["min_qv"], p["min_mapq"], p["qlen_lower_limit"], p["qlen_upper_limit"], p["min_sequence_identity"], p["min_gq"],
                    p["min_bq"], p["min_trim"], p["max_mismatch_count"], p["mismatch_window_size"], p["md_threshold"],
                    p["min_ref_count"], p["min_alt_count"], p["min_hap_count"], p["germline_snv_prior"], False)
        ccs, rf, log = normcounts.norm_contig(w, b, chunks, refseq, alt_order=order)
        assert (ccs, rf, log) == want, "pass {} of contig {} differs: {} vs {}".format(i, k, log, want[2])
    print("soak ok: {} passes, {:.1f} s, callable bases {}".format(a.passes, time.time() - t0, [c[4][2][13] for c in cases]))
    w.close()


if __name__ == "__main__":
    main()
