#!/usr/bin/env python3
"""BASELINE.json configs[2]: a whole synthetic GRCh38 (24 contigs with the primary-assembly
lengths, 3.09 Gb) at 30x through `himut call`'s scan, contigs packed onto the ranks by
longest-processing-time (himut_amd/dist.py) and the record buffers gathered to rank 0 at the
end.  Strong scaling: the genome is fixed, N ranks share it.

  python tools/bench_genome.py                       (one GPU scans all 24 contigs in turn)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P tools/bench_genome.py       (N ranks, RCCL gather)

Per contig the reads are generated on the host, uploaded (timed apart), scanned once untimed
(allocations) and once timed with inputs resident in HBM.  Prints ONE JSON line on rank 0.
--scale S divides every contig length by S (rehearsals)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GRCH38 = {"chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555, "chr5": 181538259,
          "chr6": 170805979, "chr7": 159345973, "chr8": 145138636, "chr9": 138394717, "chr10": 133797422,
          "chr11": 135086622, "chr12": 133275309, "chr13": 114364328, "chr14": 107043718, "chr15": 101991189,
          "chr16": 90338345, "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
          "chr21": 46709983, "chr22": 50818468, "chrX": 156040895, "chrY": 57227415}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--depth", type=float, default=30.0)
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch  # before libhimut_hip.so: both must bind to one HIP runtime
    import torch.distributed as dist
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
    from bench import make_side_sets
    from himut_amd import bamlib, caller, synth, util as hutil
    from himut_amd import dist as hdist

    sizes = {c: max(int(L / a.scale), 250_000) for c, L in GRCH38.items()}
    names = hutil.natsorted(list(sizes))
    mine = hdist.lpt_assign(sizes, world)[rank]
    w = caller.Worker(local_rank)
    local = {}
    t_wall = time.perf_counter()
    tot = dict(gen=0.0, h2d=0.0, run_ms=0.0, d2h=0.0, span=0, bases=0, cand=0, recs=0, reads=0)
    for c in mine:
        t0 = time.perf_counter()
        s = synth.generate(synth.SynthConfig(seed=300 + names.index(c), contig_len=sizes[c], depth=a.depth, name=c))
        b = s.batch
        chunks = [(k[1], k[2]) for k in hutil.chunkloci((c, 0, b.length))]
        ql, qu, md = bamlib.get_thresholds({c: b}, [c], {c: b.length})
        pon, com = make_side_sets(s, 500 + names.index(c))
        tot["gen"] += time.perf_counter() - t0
        w.configure(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99,
                    min_gq=20, min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20,
                    md_threshold=md, min_ref_count=3, min_alt_count=1, min_hap_count=3,
                    germline_snv_prior=1 / (10 ** 3), phase=False)
        ctx = w.ctx
        t0 = time.perf_counter()
        ctx.set_chunks(chunks)
        ctx.set_site_set(0, pon)
        ctx.set_site_set(1, com)
        ctx.push_reads(b)
        tot["h2d"] += time.perf_counter() - t0
        ctx.run()                      # first pass over this contig sizes the work buffers
        ctx.run()
        st = ctx.stats()
        tot["run_ms"] += st["ms_total"]
        t0 = time.perf_counter()
        recs, log = ctx.records(), ctx.log()
        tot["d2h"] += time.perf_counter() - t0
        local[c] = (recs.copy(), log)
        tot["span"] += st["positions"]
        tot["bases"] += st["read_bases"]
        tot["cand"] += log[1]
        tot["recs"] += len(recs)
        tot["reads"] += b.n
        del s, b
    t_scan_wall = time.perf_counter() - t_wall
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
        t0 = time.perf_counter()
        res = hdist.gather_contig_results(local, names, rank, world)
        t_gather = time.perf_counter() - t0
        keys = ("gen", "h2d", "run_ms", "d2h", "span", "bases", "cand", "recs", "reads")
        dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
        v = torch.tensor([float(tot[k]) for k in keys], dtype=torch.float64, device=dev)
        vmax = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        sums = dict(zip(keys, v.tolist()))
        maxs = dict(zip(keys, vmax.tolist()))
    else:
        res, t_gather = local, 0.0
        sums = maxs = tot
    if rank == 0:
        assert list(res) == names or world == 1
        n_rec = sum(len(r[0]) for r in res.values())
        assert n_rec == int(sums["recs"])
        device_s = maxs["run_ms"] / 1e3 + t_gather
        print(json.dumps({
            "metric": "Mbp scanned/sec at 30x CCS, whole synthetic GRCh38 (24 contigs)", "unit": "Mbp/s",
            "value": sums["span"] / 1e6 / device_s, "n_gpus": world, "scaling": "strong",
            "genome_bp": int(sums["span"]), "reads": int(sums["reads"]), "read_bases": int(sums["bases"]),
            "candidate_sites": int(sums["cand"]), "records": n_rec,
            "candidate_sites_per_sec": sums["cand"] / device_s,
            "scan_device_s_max_rank": maxs["run_ms"] / 1e3, "gather_s": t_gather,
            "host_s_max_rank": {"generate": maxs["gen"], "h2d_pageable": maxs["h2d"], "records_d2h": maxs["d2h"]},
            "wall_s_rank0_scan_loop": t_scan_wall, "contigs_rank0": mine,
            "note": "value = genome span / (slowest rank's summed himut_run device time + the final gather); "
                    "inputs resident in HBM when each run starts"}))
    w.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
