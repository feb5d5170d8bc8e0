"""Multi-GPU driver pieces: contigs are independent units (the reference's
``Pool.starmap`` axis, caller.py:766-810), so they are packed onto ranks and
scanned with no data-path collective; one exchange at the end brings the
fixed-width record buffers and the 15 counters of every contig to rank 0
(RCCL over xGMI when the backend is nccl; gloo on CPU for tests)."""
import numpy as np

from ._ffi import RECORD_DTYPE
from .util import natsorted

REC = RECORD_DTYPE.itemsize


def lpt_assign(contig_len, world):
    """Longest-processing-time packing of contigs onto ``world`` ranks.
    Returns a list (per rank) of contig names; deterministic."""
    order = sorted(contig_len, key=lambda c: (-contig_len[c], c))
    load = [0] * world
    out = [[] for _ in range(world)]
    for c in order:
        k = min(range(world), key=lambda i: (load[i], i))
        out[k].append(c)
        load[k] += contig_len[c]
    return out


def join_group(devices=(0,)):
    """(rank, world, device) when the process was started by torch.distributed.run with more than one rank -- the
    process group is created on first use (RCCL; ``HIMUT_DIST_BACKEND=gloo`` for rehearsals on a box with fewer
    GPUs than ranks, every rank then uses devices[0]) -- else None."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("HIMUT_DIST_SINGLE") != "1":     # =1: a group of one rank (rehearses the RCCL calls)
        return None
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("HIMUT_DIST_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else (list(devices) or [0])[0]
    torch.cuda.set_device(device)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def leave_group():
    """Barrier + teardown of the process group join_group created."""
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def broadcast_ints(values, src=0):
    """The integers of rank ``src`` on every rank."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return [int(x) for x in t.tolist()]


def gather_contig_results(local, contig_names, rank, world, device_buffers=None, materialize=True):
    """local: {contig: (records structured array, log list of 15)} for the
    contigs this rank scanned.  ``device_buffers``: optional {contig:
    (ctx, n_records)} to copy records device-to-device instead of from the host
    array (nccl path).  Returns on rank 0 {contig: (records, log)} for every
    contig, in natural contig order; None elsewhere.  With ``materialize=False`` rank 0
    gets the raw (table, per-rank buffers) still on the gather device (no host copy)."""
    import torch
    import torch.distributed as dist
    names = natsorted(list(contig_names))
    index = {c: i for i, c in enumerate(names)}
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    # 1. per contig: owner's record count and counters (all-reduce of a table that is zero elsewhere)
    table = torch.zeros((len(names), 17), dtype=torch.int64, device=dev)
    for c, (recs, log) in local.items():
        n = device_buffers[c][1] if device_buffers and c in device_buffers else len(recs)
        table[index[c], 0] = n
        table[index[c], 1] = rank
        table[index[c], 2:17] = torch.tensor(list(log), dtype=torch.int64)
    dist.all_reduce(table, op=dist.ReduceOp.SUM)
    tab = table.cpu().numpy()
    per_rank = [0] * world
    for i in range(len(names)):
        per_rank[int(tab[i, 1])] += int(tab[i, 0])
    cap = max(max(per_rank), 1)
    # 2. one fixed-size send per rank: its contigs' records back to back, in natural order
    send = torch.zeros(cap * REC, dtype=torch.uint8, device=dev)
    if dev.type == "cuda":
        # the library copies into `send` on its own stream: the zero fill (torch's stream) must be done first
        torch.cuda.current_stream().synchronize()
    off = 0
    for c in names:
        if c not in local:
            continue
        n = int(tab[index[c], 0])
        if n == 0:
            continue
        if device_buffers and c in device_buffers:
            ctx = device_buffers[c][0]
            ctx.copy_records_to_device(send.data_ptr() + off * REC, n)
        else:
            raw = np.ascontiguousarray(local[c][0]).view(np.uint8).reshape(-1)
            send[off * REC:(off + n) * REC] = torch.from_numpy(raw.copy()).to(dev)
        off += n
    recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    if not materialize:
        return tab, recv
    out = {}
    cursor = [0] * world
    bufs = [t.cpu().numpy() for t in recv]
    for i, c in enumerate(names):
        n, owner = int(tab[i, 0]), int(tab[i, 1])
        a = cursor[owner]
        recs = bufs[owner][a * REC:(a + n) * REC].view(RECORD_DTYPE).copy()
        cursor[owner] += n
        out[c] = (recs, [int(x) for x in tab[i, 2:17]])
    return out


class RecordExchange:
    """The same final exchange, pipelined: the gather of one contig's records to rank 0 runs while the next
    contig is scanned.  Every rank sends a fixed-size buffer per submit (``cap_records`` records, agreed once
    with ``plan``), `depth` of them in flight; the record counts and the 15 counters travel in one small
    all-gather when the exchange is drained."""

    def __init__(self, rank, world, cap_records, depth=2):
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.cap, self.depth = rank, world, int(cap_records), depth
        backend = dist.get_backend()
        self.dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.send = [torch.zeros(self.cap * REC, dtype=torch.uint8, device=self.dev) for _ in range(depth)]
        self.recv = [[torch.empty_like(self.send[0]) for _ in range(world)] for _ in range(depth)] if rank == 0 else None
        if self.dev.type == "cuda":
            torch.cuda.current_stream().synchronize()     # the zero fills are done before the library writes
        self.handles = [None] * depth
        self.meta = []          # per submit: [n, log[15]]
        self.k = 0
        # one untimed gather brings the point-to-point channels up (RCCL sets them up on first use)
        dist.gather(self.send[0], self.recv[0] if rank == 0 else None, dst=0)
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    @staticmethod
    def plan(n_local):
        """Collective: the capacity every rank must use = the largest record count of any rank (+25 %)."""
        import torch
        import torch.distributed as dist
        backend = dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item()) * 5 // 4 + 64

    def submit(self, n, log, ctx=None, records=None):
        """Records of one finished contig: from the context's device buffer (ctx) or from a host array."""
        import torch
        import torch.distributed as dist
        if n > self.cap:
            raise ValueError("RecordExchange: {} records exceed the planned capacity {}".format(n, self.cap))
        slot = self.k % self.depth
        if self.handles[slot] is not None:
            self.handles[slot].wait()
        if n:
            if ctx is not None:
                ctx.copy_records_to_device(self.send[slot].data_ptr(), self.cap)
            else:
                raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
                self.send[slot][:n * REC] = torch.from_numpy(raw.copy()).to(self.dev)
        self.handles[slot] = dist.gather(self.send[slot], self.recv[slot] if self.rank == 0 else None, dst=0,
                                         async_op=True)
        self.meta.append([int(n)] + [int(x) for x in log])
        self.k += 1

    def drain(self, materialize_last=False):
        """Waits for every gather in flight and exchanges the per-submit counts.  Rank 0 gets
        (counts[world][submits][16], last) where ``last`` is {rank: records array} of the last submit when
        ``materialize_last``; other ranks get None."""
        import torch
        import torch.distributed as dist
        for h in self.handles:
            if h is not None:
                h.wait()
        self.handles = [None] * self.depth
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        m = torch.tensor(self.meta if self.meta else [[0] * 16], dtype=torch.int64, device=self.dev)
        allm = [torch.empty_like(m) for _ in range(self.world)]
        dist.all_gather(allm, m)
        if self.rank != 0:
            return None
        counts = [t.cpu().numpy() for t in allm]
        last = self.last_records(counts) if materialize_last else None
        return counts, last

    def last_records(self, counts):
        """Rank 0: {rank: records array} of the last submit (host copy)."""
        if self.rank != 0 or self.k == 0:
            return None
        slot = (self.k - 1) % self.depth
        last = {}
        for r in range(self.world):
            n = int(counts[r][-1][0])
            last[r] = self.recv[slot][r][:n * REC].cpu().numpy().view(RECORD_DTYPE).copy()
        return last
