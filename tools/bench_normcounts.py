#!/usr/bin/env python3
"""Measurement of the normcounts sweep on one GPU: a chr20-sized 30x synthetic contig,
inputs resident in HBM, K timed passes of himut_run_normcounts.  Prints one JSON line.

    python tools/bench_normcounts.py --steps 5 --warmup 1 [--contig-len N] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contig-len", type=int, default=64_444_167)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=float, default=2.0)
    a = ap.parse_args()
    import numpy as np
    from himut_amd import bamlib, caller, normcounts, synth, util as hutil
    sys.path.insert(0, ROOT)
    import bench as B

    cfg = synth.SynthConfig(seed=2, contig_len=a.contig_len, depth=a.depth, name="chr20")
    sample = synth.generate(cfg, want_ref=True)
    batch = sample.batch
    refseq = bytes(sample.ref)
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((batch.name, 0, batch.length))]
    ql, qu, md = bamlib.get_thresholds({batch.name: batch}, [batch.name], {batch.name: batch.length})
    pon, com = B.make_side_sets(sample, 100)
    params = dict(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99,
                  min_gq=20, min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20, md_threshold=md,
                  min_ref_count=3, min_alt_count=1, min_hap_count=3)
    w = caller.Worker(0)
    w.configure(germline_snv_prior=1 / (10 ** 3), phase=False, **params)
    ctx = w.ctx
    chars, cls = normcounts.tri_classes(refseq)
    ctx.set_chunks(chunks)
    ctx.set_site_set(0, pon)
    ctx.set_site_set(1, com)
    ctx.set_reference(refseq, cls, len(chars))
    ctx.push_reads(batch)
    tab = normcounts.alt_order_table({"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"],
                                      "C": ["G", "T", "A"]})
    for _ in range(a.warmup):
        ctx.run_normcounts(tab)
    ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.run_normcounts(tab)
        ms.append(ctx.stats()["ms_total"])
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    ccs, ref, log = ctx.normcounts()
    positions = sum(e - s for s, e in chunks)
    out = {"metric": "Mbp swept/sec at 30x CCS (himut normcounts callable-tricount sweep)",
           "value": positions / 1e6 / (elapsed / a.steps), "unit": "Mbp/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "data": "synthetic",
           "dtype": "u8/int32 + f64 genotype", "device_ms": float(np.mean(ms)),
           "config": {"workload": "chr20-sized contig ({} bp) {:.0f}x synthetic CCS, common-SNP + PoN filtering, "
                                  "reference chunking ({} chunks), non-phased".format(a.contig_len, a.depth, len(chunks)),
                      "reads": st["n_reads"], "read_bases": st["read_bases"], "column_slots": st["column_slots"]},
           "callable_bases": log[13], "num_bases": log[1],
           # SURVEY 8d convention for the dense sweep: 1.5 B in + 2 B cell written + 2 B cell read per read base
           "roofline": {"bound": "hbm", "achieved": st["read_bases"] * 5.5 / (np.mean(ms) * 1e-3) / 1e9, "peak": 8000.0,
                        "unit": "GB/s", "frac": st["read_bases"] * 5.5 / (np.mean(ms) * 1e-3) / 1e9 / 8000.0,
                        "note": "whole pass against the SURVEY's 5.5 B per read base (qualities twice, packed bases once, one "
                                "2-byte cell written and read); the column sweep (k_norm_col) builds no cells at all and "
                                "waits on dependent loads more than it moves bytes (DESIGN.md section 8)"}}
    if not a.no_cpu_baseline:
        from oracle import oracle as O
        nch = max(1, int(a.cpu_sample_mb * 1e6 / 200000))
        sub_chunks = chunks[:nch]
        end = sub_chunks[-1][1]
        n = int(np.searchsorted(batch.tstart, end, side="left"))
        from himut_amd.readbatch import ReadBatch
        tot = int(batch.qoff[n - 1] + ((int(batch.qlen[n - 1]) + 31) & ~31))
        sub = ReadBatch(name=batch.name, length=batch.length, tstart=batch.tstart[:n], tend=batch.tend[:n],
                        qstart=batch.qstart[:n], qlen=batch.qlen[:n], mapq=batch.mapq[:n], flag=batch.flag[:n],
                        qid=batch.qid[:n], qoff=batch.qoff[:n], cs_off=batch.cs_off[:n + 1], seq=batch.seq[:tot // 2],
                        bq=batch.bq[:tot], cs=batch.cs[:int(batch.cs_off[n])], tp=batch.tp[:n])
        t1 = time.perf_counter()
        O.normcounts(sub, sub_chunks, params, refseq, 1 / (10 ** 3), pon, com)
        dt = time.perf_counter() - t1
        span = sum(e - s for s, e in sub_chunks)
        out["cpu_baseline"] = {"value": span / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port",
                               "sample": "first {} reference chunks ({:.1f} Mb, {} reads), oracle/himut_oracle.c "
                                         "orc_normcounts single thread, {:.1f} s".format(nch, span / 1e6, n, dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
