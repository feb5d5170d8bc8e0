#!/usr/bin/env python3
"""Soak run of the call scan: the same contig N times through one context (capacities kept, no host round trip in
the middle), alternating with a second, smaller contig every 50th pass; every pass must return the same records
and counters as the first one of its contig.  Prints one line."""
import argparse
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=2000)
    ap.add_argument("--contig-len", type=int, default=16_000_000)
    a = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime)
    import numpy as np
    from bench import make_side_sets
    from himut_amd import bamlib, caller, synth, util as hutil
    w = caller.Worker(0)
    cases = []
    for seed, L in ((11, a.contig_len), (12, a.contig_len // 5)):
        s = synth.generate(synth.SynthConfig(seed=seed, contig_len=L, name="chr{}".format(seed)))
        b = s.batch
        chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
        ql, qu, md = bamlib.get_thresholds({b.name: b}, [b.name], {b.name: b.length})
        pon, com = make_side_sets(s, seed)
        cases.append((b, chunks, ql, qu, md, pon, com))
    want = {}
    t0 = time.time()
    k_cur = None
    for i in range(a.passes):
        k = 1 if i % 50 == 49 else 0
        b, chunks, ql, qu, md, pon, com = cases[k]
        if k != k_cur:
            w.configure(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99,
                        min_gq=20, min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20,
                        md_threshold=md, min_ref_count=3, min_alt_count=1, min_hap_count=3,
                        germline_snv_prior=1 / (10 ** 3), phase=False)
            recs, log = w.call_contig(b, chunks, pon, com, None)       # pushes the reads again
            k_cur = k
        else:
            w.ctx.run()
            recs, log = w.ctx.records(), w.ctx.log()
        sig = (zlib.crc32(np.ascontiguousarray(recs).view(np.uint8)), tuple(log), len(recs))
        if k not in want:
            want[k] = sig
        assert sig == want[k], "pass {} of contig {} differs: {} vs {}".format(i, k, sig, want[k])
    print("soak ok: {} passes, {:.1f} s, signatures {}".format(a.passes, time.time() - t0, {k: v[2] for k, v in want.items()}))
    w.close()


if __name__ == "__main__":
    main()
