#!/usr/bin/env python3
"""phaselib.get_edges on the chr20-sized 30x contig: device time of himut_run_edges (inputs resident) and the
CPU restatement on a sample.  Prints one JSON line."""
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
    ap.add_argument("--contig-len", type=int, default=64_444_167)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import numpy as np
    from himut_amd import caller, phaselib, synth
    s = synth.generate(synth.SynthConfig(seed=2, contig_len=a.contig_len, name="chr20"))
    b = s.batch
    hets = sorted(set((int(p) + 1, chr(r), chr(al)) for p, r, al, g in zip(s.snp_pos, s.snp_ref, s.snp_alt, s.snp_gt)
                      if g in (1, 2)))
    hpos = np.array([h[0] for h in hets], np.int32)
    href = np.array([ord(h[1]) for h in hets], np.uint8)
    w = caller.Worker(0)
    w.configure(0, 0, 0, 1 << 30, 0.0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0, 1 / (10 ** 3), False)
    ctx = w.ctx
    ctx.push_reads(b)
    band = phaselib.edge_band(b, hpos)
    ctx.run_edges(hpos, href, 20, 20, band)
    ms = []
    for _ in range(a.steps):
        counts = ctx.run_edges(hpos, href, 20, 20, band)
        ms.append(ctx.stats()["ms_total"])
    n_edges = int(np.count_nonzero(counts.reshape(-1, 4).sum(1)))
    out = {"metric": "Mbp phased-edge-counted/sec at 30x CCS (himut phase get_edges)", "value": a.contig_len / 1e6 / (np.mean(ms) * 1e-3),
           "unit": "Mbp/s", "device_ms": float(np.mean(ms)), "hetsnps": len(hets), "band": band, "edges": n_edges,
           "pair_counts": int(counts.sum()), "reads": int(b.n)}
    if not a.no_cpu_baseline:
        from oracle import oracle as O
        from himut_amd.readbatch import ReadBatch
        n = int(np.searchsorted(b.tstart, 8_000_000, side="left"))
        tot = int(b.qoff[n - 1] + ((int(b.qlen[n - 1]) + 31) & ~31))
        sub = ReadBatch(name=b.name, length=b.length, tstart=b.tstart[:n], tend=b.tend[:n], qstart=b.qstart[:n],
                        qlen=b.qlen[:n], mapq=b.mapq[:n], flag=b.flag[:n], qid=b.qid[:n], qoff=b.qoff[:n],
                        cs_off=b.cs_off[:n + 1], seq=b.seq[:tot // 2], bq=b.bq[:tot], cs=b.cs[:int(b.cs_off[n])], tp=b.tp[:n])
        sh = [h for h in hets if h[0] < 8_100_000]
        t0 = time.perf_counter()
        O.edges(sub, sh, 20, 20)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 8.0 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port",
                               "sample": "reads of the first 8 Mb ({}), oracle orc_edges single thread, {:.1f} s".format(n, dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
