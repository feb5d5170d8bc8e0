#!/usr/bin/env python3
"""Benchmark of the hot path: `himut call`'s per-chromosome CCS pileup scan on
MI355X.  One "step" = one full pass of the scan (all kernels of himut_run) over
one synthetic 30x contig that is already resident in HBM.  With N ranks each
rank owns its own contig (the reference's starmap axis, caller.py:766-810) and
every step ends with the RCCL gather of the record buffers to rank 0.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHR20_LEN = 64_444_167
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# algorithmic HBM bytes per unit of work, per kernel (DESIGN.md "Kernels and their rooflines")
STAGE_KERNEL = {"ms_parse": "k_parse_cs", "ms_emit": "k_propose",
                "ms_capture": "k_stream_capture", "ms_eval": "k_eval_columns"}


def algorithmic_bytes(stage, st, cs_bytes):
    rb, pos, cand, slots = st["read_bases"], st["positions"], st["n_candidates"], st["column_slots"]
    if stage == "ms_capture":    # every quality byte + every packed base once, one 2-byte slot per pile cell kept
        return rb * 1.5 + slots * 2.0
    if stage == "ms_parse":      # every quality byte once; cs text in, ~16 B per cs operation out (segments + mismatch list)
        return rb * 1.0 + cs_bytes * 1.0 + cs_bytes / 4.0 * 16.0
    if stage == "ms_emit":       # mismatch list in, mask word + candidate out per candidate
        return cs_bytes / 4.0 * 8.0 + cand * 16.0
    if stage == "ms_eval":       # column slots in, one 64-byte record out
        return slots * 2.0 + cand * 64.0
    return 0.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--contig-len", type=int, default=CHR20_LEN)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=float, default=24.0)
    return ap.parse_args()


def make_side_sets(sample, seed):
    """PoN / common-SNP key arrays of the size BASELINE configs[1] describes
    (common = half the germline SNPs + 1e-4/bp decoys, PoN = 1e-4/bp random)."""
    import numpy as np
    from himut_amd import caller
    rs = np.random.RandomState(seed)
    L = sample.batch.length
    keep = rs.rand(sample.snp_pos.shape[0]) < 0.5
    common = [(int(p) + 1, chr(r), chr(a)) for p, r, a in zip(sample.snp_pos[keep], sample.snp_ref[keep], sample.snp_alt[keep])]
    n_decoy = int(1e-4 * L)
    pos = rs.randint(1, L + 1, size=2 * n_decoy)
    ra = rs.randint(0, 4, size=(2 * n_decoy, 2))
    decoys = [(int(p), "ACGT"[i], "ACGT"[j]) for p, (i, j) in zip(pos, ra) if i != j]
    h = len(decoys) // 2
    return caller.site_keys(decoys[:h]), caller.site_keys(common + decoys[h:])


def cpu_baseline(batch, chunks, params, pon, com, sample_mb):
    """The CPU oracle (a C restatement of the reference algorithm, "port") timed
    on one host core over a bounded prefix of the same workload."""
    import numpy as np
    from oracle import oracle as O
    from himut_amd.readbatch import ReadBatch
    limit = int(sample_mb * 1e6)
    sub_chunks = [c for c in chunks if c[1] <= limit]
    if not sub_chunks:
        sub_chunks = chunks[:1]
    end = sub_chunks[-1][1]
    n = int(np.searchsorted(batch.tstart, end, side="left"))
    tot = int(batch.qoff[n - 1] + ((int(batch.qlen[n - 1]) + 31) & ~31)) if n else 0
    sub = ReadBatch(name=batch.name, length=batch.length, tstart=batch.tstart[:n], tend=batch.tend[:n],
                    qstart=batch.qstart[:n], qlen=batch.qlen[:n], mapq=batch.mapq[:n], flag=batch.flag[:n],
                    qid=batch.qid[:n], qoff=batch.qoff[:n], cs_off=batch.cs_off[:n + 1], seq=batch.seq[:tot // 2],
                    bq=batch.bq[:tot], cs=batch.cs[:int(batch.cs_off[n])], tp=batch.tp[:n])
    t0 = time.perf_counter()
    recs, log = O.call(sub, sub_chunks, params, 1 / (10 ** 3), pon, com)
    dt = time.perf_counter() - t0
    span = sum(e - s + 1 for s, e in sub_chunks)
    return {"value": span / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port",
            "sample": "first {} reference chunks ({:.1f} Mb, {} reads) of the same contig, oracle/himut_oracle.c "
                      "single thread, {:.1f} s; candidate sites/s {:.0f}".format(len(sub_chunks), span / 1e6, n, dt,
                                                                              log[1] / dt)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch  # before libhimut_hip.so: both must bind to one HIP runtime
    import torch.distributed as dist
    import numpy as np
    # rehearsal knobs (not used by the driver): HIMUT_BENCH_BACKEND=gloo runs the N > 1 code path
    # on a box with one GPU, every rank on cuda:0
    backend = os.environ.get("HIMUT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from himut_amd import bamlib, caller, synth, util as hutil
    from himut_amd import dist as hdist

    # ---- synthetic workload (BASELINE.json configs[1] shape; one contig per rank)
    t_gen = time.perf_counter()
    names = ["chr{}".format(20 + k) for k in range(world)]   # one chr20-sized contig per rank
    cfg = synth.SynthConfig(seed=2 + rank, contig_len=a.contig_len, depth=a.depth, name=names[rank])
    sample = synth.generate(cfg)
    batch = sample.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((batch.name, 0, batch.length))]
    ql, qu, md = bamlib.get_thresholds({batch.name: batch}, [batch.name], {batch.name: batch.length})
    pon, com = make_side_sets(sample, 100 + rank)
    params = dict(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99,
                  min_gq=20, min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20, md_threshold=md,
                  min_ref_count=3, min_alt_count=1, min_hap_count=3)
    t_gen = time.perf_counter() - t_gen

    w = caller.Worker(local_rank)
    w.configure(germline_snv_prior=1 / (10 ** 3), phase=False, **params)
    ctx = w.ctx
    ctx.set_chunks(chunks)
    ctx.set_site_set(0, pon)
    ctx.set_site_set(1, com)
    t_h2d = time.perf_counter()
    ctx.push_reads(batch)                     # inputs resident in HBM before the timed region
    t_h2d = time.perf_counter() - t_h2d

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the final exchange of every step (record buffers of the step's contigs -> rank 0, RCCL over xGMI)
    # is pipelined: it runs while the next step scans.  All K exchanges complete inside the timed region.
    ex = None
    use_ex = world > 1 or os.environ.get("HIMUT_BENCH_FORCE_EXCHANGE") == "1"   # rehearsal of the N > 1 path on one rank
    if use_ex:
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1)
        ctx.run()                                           # untimed: sizes the exchange buffers
        cap = hdist.RecordExchange.plan(ctx.records_device()[1])
        ex = hdist.RecordExchange(rank, world, cap, depth=2)

    def step():
        ctx.run()
        if use_ex:
            _, n = ctx.records_device()
            if backend == "nccl":
                ex.submit(n, ctx.log(), ctx=ctx)
            else:
                ex.submit(n, ctx.log(), records=ctx.records())

    for _ in range(a.warmup):
        step()
    if ex is not None:
        ex.drain()
        ex.meta = []
    stage_ms = {}
    barrier()
    t0 = time.perf_counter()
    # timed region: the library records only the run's start / end and the events around the column capture
    # (the dominant kernel, whose launch time the roofline needs); an event costs a barrier packet on the queue
    for _ in range(a.steps):
        step()
        st = ctx.stats()
        for k in ("ms_total", "ms_capture"):
            stage_ms.setdefault(k, []).append(st[k])
    gathered = ex.drain() if ex is not None else None
    barrier()
    elapsed = time.perf_counter() - t0
    # untimed: three more passes with every stage event on, for the per-stage breakdown
    ctx.set_stage_timing(2)
    detail = {}
    for _ in range(3):
        ctx.run()
        st = ctx.stats()
        for k in ("ms_total", "ms_parse", "ms_hap", "ms_emit", "ms_index", "ms_capture", "ms_eval", "ms_finalize"):
            detail.setdefault(k, []).append(st[k])
    ctx.set_stage_timing(1)
    if gathered is not None:                                # untimed: host copy of the last step's gathered records
        gathered = (gathered[0], ex.last_records(gathered[0]))
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if use_ex:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = ctx.stats()
    log = ctx.log()
    totals = torch.tensor([st["positions"], log[1], st["read_bases"], st["n_records"]], dtype=torch.float64,
                          device=red_dev)
    if use_ex:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    positions, cand_sites, read_bases, n_records = [float(x) for x in totals.tolist()]

    if rank == 0:
        if gathered is not None:   # every step's exchange delivered every rank's records
            counts, last = gathered
            assert len(counts) == world and all(len(c) == a.steps for c in counts)
            assert sum(int(c[-1][0]) for c in counts) == int(n_records)
            assert sum(len(v) for v in last.values()) == int(n_records)
        ms_per_step = elapsed / a.steps * 1e3
        mbp_s = positions / 1e6 / (elapsed / a.steps)
        avg = {k: float(np.mean(v)) for k, v in detail.items()}
        timed = {k: float(np.mean(v)) for k, v in stage_ms.items()}
        dom = max(STAGE_KERNEL, key=lambda k: avg[k])           # the dominant kernel of the step
        dom_ms = timed[dom] if dom in timed else avg[dom]       # the column capture: measured inside the timed region
        cs_bytes = int(batch.cs.shape[0])
        alg_bytes = algorithmic_bytes(dom, st, cs_bytes)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        per_kernel = {STAGE_KERNEL[k]: {"ms": avg[k], "alg_GBps": algorithmic_bytes(k, st, cs_bytes) / (avg[k] * 1e-3) / 1e9}
                      for k in STAGE_KERNEL}
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                t = json.load(open(tp))
                if t.get("contig_len") == a.contig_len and t.get("depth") == a.depth:
                    traffic = t.get(STAGE_KERNEL[dom])
            except Exception:
                traffic = None
        out = {
            "metric": "Mbp scanned/sec at 30x CCS (himut call pileup scan)", "value": mbp_s, "unit": "Mbp/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 + f64 genotype tail",
            "data": "synthetic",
            "config": {"workload": "chr20-sized contig ({} bp) {:.0f}x synthetic CCS per GPU, common-SNP + PoN "
                                   "filtering, reference chunking ({} chunks)".format(a.contig_len, a.depth,
                                                                                        len(chunks)),
                       "reads_per_gpu": st["n_reads"], "read_bases_per_gpu": st["read_bases"],
                       "parallelism": "contig-per-gpu x{} + RCCL gather".format(world)},
            "candidate_sites_per_sec": cand_sites / (elapsed / a.steps),
            "candidate_sites_per_step": cand_sites, "records_per_step": n_records,
            "device_ms_per_step": timed["ms_total"],
            "stage_ms": avg, "stage_ms_note": "three untimed passes after the timed region with every stage event "
                                              "recorded (himut_set_stage_timing 2); the timed steps record run "
                                              "start / end and the events around k_stream_capture only",
            "kernels": per_kernel,
            "setup_s": {"generate": t_gen, "h2d": t_h2d},
            # the whole step against the SURVEY's own byte count (3.5 B per read base + 2 B per position +
            # (1.5 D + 48) B per candidate site), device time of the timed steps
            "roofline_step": (lambda by: {"bound": "hbm", "achieved": by / (timed["ms_total"] * 1e-3) / 1e9,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": by / (timed["ms_total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "algorithmic_bytes_per_step": by, "convention": "SURVEY 8d"})(
                (read_bases * 3.5 + positions * 2.0 + cand_sites * (1.5 * a.depth + 48.0)) / world),
            "roofline": {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms},
        }
        if not a.no_cpu_baseline and world == 1:           # the CPU leg runs on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(batch, chunks, params, pon, com, a.cpu_sample_mb)
        print(json.dumps(out), flush=True)
    w.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
