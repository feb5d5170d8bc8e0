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
