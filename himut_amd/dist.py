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


class RankError(RuntimeError):
    """Some rank of the group failed; raised on EVERY rank so that none is left waiting in a collective."""


def share_or_raise(payload, error=None):
    """The ranks' payloads (a list, rank order) -- or RankError on every rank when any rank passes an ``error``
    (the exception it caught while doing its share).  The first collective after a stretch of per-rank work: a rank
    that failed still joins it, so its peers do not hang until the backend's timeout; the group is torn down before
    raising."""
    import torch.distributed as dist
    parts = [None] * dist.get_world_size()
    msg = None if error is None else "{}: {}".format(type(error).__name__, error)
    dist.all_gather_object(parts, (msg, None if error is not None else payload))
    failed = [(r, m) for r, (m, _) in enumerate(parts) if m is not None]
    if failed:
        leave_group()
        text = "; ".join("rank {}: {}".format(r, m) for r, m in failed)
        if error is not None:
            raise RankError(text) from error
        raise RankError(text)
    return [p for _, p in parts]


def broadcast_ints(values, src=0):
    """The integers of rank ``src`` on every rank."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return [int(x) for x in t.tolist()]


def _gather_device():
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _p2p(ops):
    """Starts a batch of point-to-point transfers (each peer talks to rank 0 over its own xGMI link: no ring, no
    padding to a common size); returns the requests."""
    import torch.distributed as dist
    return dist.batch_isend_irecv(ops) if ops else []


def _wait_all(reqs, dev):
    for r in reqs:
        r.wait()
    if dev.type == "cuda":
        # on RCCL wait() orders torch's current stream only; the library copies into / out of these buffers on a
        # stream of its own, so the host itself has to see the transfers finished
        import torch
        torch.cuda.current_stream().synchronize()


def gather_contig_results(local, contig_names, rank, world, device_buffers=None, materialize=True):
    """local: {contig: (records structured array, log list of 15)} for the
    contigs this rank scanned.  ``device_buffers``: optional {contig:
    (ctx, n_records)} to copy records device-to-device instead of from the host
    array (nccl path).  Returns on rank 0 {contig: (records, log)} for every
    contig, in natural contig order; None elsewhere.  With ``materialize=False`` rank 0
    gets the raw (table, per-rank buffers) still on the gather device (no host copy).

    One small all-reduce tells every rank the record count and the counters of every contig; then each rank sends
    its records -- its contigs back to back in natural order, exactly as many bytes as it has -- straight to rank 0
    (``isend`` / ``irecv``; RCCL has no gatherv, and a padded gather would move world x the largest share)."""
    import torch
    import torch.distributed as dist
    names = natsorted(list(contig_names))
    index = {c: i for i, c in enumerate(names)}
    dev = _gather_device()
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
    # 2. this rank's records back to back, in natural contig order
    send = torch.zeros(max(per_rank[rank], 1) * REC, dtype=torch.uint8, device=dev)
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
    # 3. exact-size transfers to rank 0
    if rank != 0:
        if per_rank[rank]:
            _wait_all(_p2p([dist.P2POp(dist.isend, send[:per_rank[rank] * REC], 0)]), dev)
        return None
    recv = [send] + [torch.empty(max(per_rank[r], 1) * REC, dtype=torch.uint8, device=dev) for r in range(1, world)]
    _wait_all(_p2p([dist.P2POp(dist.irecv, recv[r][:per_rank[r] * REC], r) for r in range(1, world) if per_rank[r]]), dev)
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


class RecordExchangeOverflow(RuntimeError):
    """A submit held more records than its planned message carries.  Raised by ``drain`` on EVERY rank alike, with the
    capacities the exchange must be rebuilt with (``caps``) before the pass is repeated."""

    def __init__(self, caps, where):
        RuntimeError.__init__(self, "RecordExchange: more records than planned at (rank, submit) {}".format(where))
        self.caps = caps


class RecordExchange:
    """The same final exchange, pipelined: the records of one finished contig travel to rank 0 while the next
    contig is scanned.  A run is a sequence of ROUNDS (round k = the k-th contig of every rank; a rank with fewer
    contigs sends nothing in the later rounds).  A point-to-point message has one size that sender and receiver must
    agree on beforehand, so the sizes are agreed once (``plan``: per round and rank the record count of a rehearsal
    pass -- the scan is deterministic, so that IS the count of every later pass -- plus one record in 64 of slack):
    every transfer is a message of its own size from the owner to rank 0, each peer on its own xGMI link, nothing
    padded to the largest share and next to nothing padded at all.  A submit with more records than planned sends the
    planned part and is reported by ``drain`` on every rank (RecordExchangeOverflow: rebuild with its ``caps`` and
    repeat the pass).  ``depth`` passes over the rounds may be in flight (buffers per round and pass); the record
    counts and the 15 counters travel in one small all-gather when the exchange is drained."""

    def __init__(self, rank, world, caps, depth=2, keep=False):
        import torch
        import torch.distributed as dist
        if isinstance(caps, (int, np.integer)):
            caps = [[int(caps)] * world]
        self.rank, self.world, self.depth = rank, world, depth
        self.caps = [[int(x) for x in row] for row in caps]
        self.rounds = len(self.caps)
        self.dev = _gather_device()
        nslot = self.rounds * depth
        mk = lambda n: torch.zeros(max(int(n), 1) * REC, dtype=torch.uint8, device=self.dev)
        # slot s serves round s % rounds.  Rank 0 keeps one landing buffer per slot and rank (its own records are
        # copied straight there); the others keep one send buffer per slot
        if rank == 0:
            self.recv = [[mk(self.caps[s % self.rounds][r]) for r in range(world)] for s in range(nslot)]
            self.send = None
        else:
            self.recv = None
            self.send = [mk(self.caps[s % self.rounds][rank]) for s in range(nslot)]
        if self.dev.type == "cuda":
            torch.cuda.current_stream().synchronize()     # the zero fills are done before the library writes
        self.reqs = [[] for _ in range(nslot)]
        self.meta = []          # per submit: [n, log[15]]
        self.k = 0
        self.keep = keep        # rank 0: host copies of every completed submit in self.kept[k][rank] (tests)
        self.kept = {}
        self._pending = {}      # slot -> submit index whose transfers are in flight
        # one untimed transfer per peer brings the point-to-point channels up (RCCL sets them up on first use)
        tiny = torch.zeros(REC, dtype=torch.uint8, device=self.dev)
        if rank == 0:
            warm = [torch.empty_like(tiny) for _ in range(world)]
            _wait_all(_p2p([dist.P2POp(dist.irecv, warm[r], r) for r in range(1, world)]), self.dev)
        else:
            _wait_all(_p2p([dist.P2POp(dist.isend, tiny, 0)]), self.dev)

    @staticmethod
    def plan(n_local):
        """Collective.  ``n_local``: this rank's record count per round (a list; an int = one round).  Returns the
        capacities [round][rank] every rank must construct the exchange with (the count + one in 64 + 16)."""
        import torch
        import torch.distributed as dist
        dev = _gather_device()
        mine = [int(n_local)] if isinstance(n_local, (int, np.integer)) else [int(x) for x in n_local]
        world = dist.get_world_size()
        r = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        rounds = max(int(r.item()), 1)
        t = torch.zeros(rounds, dtype=torch.int64, device=dev)
        if mine:
            t[:len(mine)] = torch.tensor(mine, dtype=torch.int64)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        cnt = np.stack([x.cpu().numpy() for x in allt], axis=1)         # [round][rank]
        return [[int(n) + int(n) // 64 + 16 if n > 0 else 0 for n in row] for row in cnt]

    def _finish(self, slot):
        if not self.reqs[slot] and slot not in self._pending:
            return
        _wait_all(self.reqs[slot], self.dev)
        self.reqs[slot] = []
        k = self._pending.pop(slot, None)
        if self.keep and self.rank == 0 and k is not None:
            # counts of the peers are not known before drain(): keep the whole landing buffers
            self.kept[k] = [t.cpu().numpy().copy() for t in self.recv[slot]]

    def submit(self, n, log, ctx=None, records=None):
        """Records of one finished contig (this rank's contig of the current round; n = 0 with nothing to send):
        from the context's device buffer (ctx) or from a host array."""
        import torch
        import torch.distributed as dist
        rnd = self.k % self.rounds
        slot = self.k % (self.rounds * self.depth)
        cap = self.caps[rnd][self.rank]
        self._finish(slot)          # the transfers that last used this slot's buffers are over
        dst = self.recv[slot][0] if self.rank == 0 else self.send[slot]
        if n > cap:
            # more than the message carries: nothing of it is sent (the message keeps its planned size: the other side has
            # posted it), the count in the table tells every rank at drain()
            pass
        elif n:
            if ctx is not None:
                ctx.copy_records_to_device(dst.data_ptr(), cap)
            else:
                raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
                dst[:n * REC] = torch.from_numpy(raw.copy()).to(self.dev)
                if self.dev.type == "cuda":
                    torch.cuda.current_stream().synchronize()
        if self.rank == 0:
            ops = [dist.P2POp(dist.irecv, self.recv[slot][r][:self.caps[rnd][r] * REC], r)
                   for r in range(1, self.world) if self.caps[rnd][r] > 0]
        else:
            ops = [dist.P2POp(dist.isend, dst[:cap * REC], 0)] if cap > 0 else []
        self.reqs[slot] = _p2p(ops)
        self._pending[slot] = self.k
        self.meta.append([int(n)] + [int(x) for x in log])
        self.k += 1

    def drain(self, materialize_last=False):
        """Waits for every transfer in flight and exchanges the per-submit counts.  Rank 0 gets
        (counts[world][submits][16], last) where ``last`` is {rank: records array} of the last submit when
        ``materialize_last``; other ranks get None."""
        import torch
        import torch.distributed as dist
        for slot in range(len(self.reqs)):
            self._finish(slot)
        # every rank has made the same number of submits (a collective sequence), so the tables have one shape
        m = torch.tensor(self.meta if self.meta else [[0] * 16], dtype=torch.int64, device=self.dev)
        allm = [torch.empty_like(m) for _ in range(self.world)]
        dist.all_gather(allm, m)
        counts = [t.cpu().numpy() for t in allm]
        # a submit that did not fit its planned message: every rank sees the same table and raises the same error
        need = [row[:] for row in self.caps]
        over = []
        for r in range(self.world):
            for k in range(len(self.meta)):
                n, rnd = int(counts[r][k][0]), k % self.rounds
                if n > self.caps[rnd][r]:
                    over.append((r, k))
                    need[rnd][r] = max(need[rnd][r], n + n // 64 + 16)
        if over:
            self.meta = []
            raise RecordExchangeOverflow(need, over[:8])
        if self.rank != 0:
            return None
        last = self.last_records(counts) if materialize_last else None
        return counts, last

    def records_of(self, counts, k):
        """Rank 0, with keep=True: {rank: records array} of submit k."""
        out = {}
        for r in range(self.world):
            n = int(counts[r][k][0])
            out[r] = self.kept[k][r][:n * REC].view(RECORD_DTYPE).copy()
        return out

    def last_records(self, counts):
        """Rank 0: {rank: records array} of the last submit (host copy)."""
        if self.rank != 0 or self.k == 0:
            return None
        slot = (self.k - 1) % (self.rounds * self.depth)
        last = {}
        for r in range(self.world):
            n = int(counts[r][-1][0])
            last[r] = self.recv[slot][r][:n * REC].cpu().numpy().view(RECORD_DTYPE).copy()
        return last
