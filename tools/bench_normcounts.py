#!/usr/bin/env python3
"""Measurement of the normcounts sweep on one GPU by itself: bench.py's `normcounts` leg (a chr20-sized
30x synthetic contig, inputs resident in HBM, K timed passes of himut_run_normcounts).  Prints one JSON line.

    python tools/bench_normcounts.py --steps 5 [--contig-len N] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)       # (the leg always makes two untimed passes first)
    ap.add_argument("--contig-len", type=int, default=64_444_167)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--error-rate", type=float, default=None, help="substitution, insertion and deletion rate each (generator defaults: 2e-4, 1e-4, 1e-4)")
    ap.add_argument("--bq93-prob", type=float, default=None, help="fraction of the bases with quality 93 (the generator's default if not given)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=float, default=16.0)
    a = ap.parse_args()
    from himut_amd import bamlib, caller, synth, util as hutil
    import bench as B

    extra = {} if a.bq93_prob is None else {"bq93_prob": a.bq93_prob}
    if a.error_rate is not None:
        extra.update(sub_rate=a.error_rate, ins_rate=a.error_rate, del_rate=a.error_rate)
    sample = synth.generate(synth.SynthConfig(seed=2, contig_len=a.contig_len, depth=a.depth, name="chr20", **extra), want_ref=True)
    batch = sample.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((batch.name, 0, batch.length))]
    ql, qu, md = bamlib.get_thresholds({batch.name: batch}, [batch.name], {batch.name: batch.length})
    pon, com = B.make_side_sets(sample, 100)
    params = dict(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99,
                  min_gq=20, min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20, md_threshold=md,
                  min_ref_count=3, min_alt_count=1, min_hap_count=3)
    w = caller.Worker(0)
    w.configure(germline_snv_prior=1 / (10 ** 3), phase=False, **params)
    ctx = w.ctx
    ctx.set_chunks(chunks)
    ctx.set_site_set(0, pon)
    ctx.set_site_set(1, com)
    ctx.push_reads(batch)
    out = B.leg_normcounts(ctx, sample, chunks, params, pon, com, a.steps, not a.no_cpu_baseline, a.cpu_sample_mb)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
