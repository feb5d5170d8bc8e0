#!/usr/bin/env python3
"""BASELINE.json configs[2] as a tool of its own: the whole synthetic GRCh38 (24 contigs, 3.09 Gb) at 30x through
`himut call`'s scan, strong scaling over the ranks (himut_amd/genome.py; bench.py --gpus N > 1 runs the same code and
prints the same dict as "genome_strong").

  python tools/bench_genome.py [--scale S] [--steps K]          one GPU, no exchange
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tools/bench_genome.py                                  N ranks, records to rank 0 over RCCL

HIMUT_BENCH_BACKEND=gloo rehearses N ranks on a box with one GPU (every rank on cuda:0).
Prints ONE JSON line on rank 0.  --scale S divides every contig length by S (rehearsals)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--group-of-one", action="store_true", help="one rank, but through a process group (the RCCL calls)")
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
    if world > 1 or a.group_of_one:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from himut_amd import genome
    out = genome.run_genome(rank, world, local_rank, scale=a.scale, depth=a.depth, steps=a.steps, backend=backend)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
