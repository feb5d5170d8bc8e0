"""The N > 1 path on CPU: contig packing and the final gather of per-contig record
buffers to rank 0 (gloo, world_size 2; the GPU run uses the same code over RCCL)."""
import os
import subprocess
import sys

import numpy as np

from himut_amd import dist as hdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from himut_amd import dist as hdist
from himut_amd._ffi import RECORD_DTYPE
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lens = {"chr1": 900, "chr2": 500, "chr10": 450, "chrX": 300, "chrM": 0}
assign = hdist.lpt_assign(lens, world)
def fake(c):
    n = lens[c] // 100
    r = np.zeros(n, RECORD_DTYPE)
    r["tpos"] = np.arange(n) + len(c) * 1000
    r["gq"] = len(c)
    r["counts"][:, 0] = np.arange(n)
    return r, [len(c) + k for k in range(15)]
local = {c: fake(c) for c in assign[rank]}
res = hdist.gather_contig_results(local, list(lens), rank, world)
if rank == 0:
    assert list(res) == ["chr1", "chr2", "chr10", "chrM", "chrX"], list(res)
    for c in lens:
        recs, log = res[c]
        want, wlog = fake(c)
        assert np.array_equal(recs, want), c
        assert log == wlog, c
    print("GATHER_OK", json.dumps(assign))
# the pipelined exchange: rounds of one "contig" per rank (rank 1 has one fewer), sizes planned per round and rank,
# three passes over the rounds with two in flight; the contents of EVERY submit are checked, not only the counts
names = ["chr1", "chr10", "chrX"] if rank == 0 else ["chr2", "chr10"]
other = ["chr2", "chr10"] if rank == 0 else ["chr1", "chr10", "chrX"]
mine = [fake(c) for c in names]
caps = hdist.RecordExchange.plan([len(r) for r, _ in mine])
assert len(caps) == 3 and all(len(row) == world for row in caps) and caps[2][1] == 0, caps
ex = hdist.RecordExchange(rank, world, caps, depth=2, keep=True)
PASSES = 3
for p in range(PASSES):
    for k in range(3):
        if k < len(mine):
            recs, log = mine[k]
            recs = recs.copy(); recs["gq"] += p               # every pass sends different bytes
            ex.submit(len(recs), log, records=recs)
        else:
            ex.submit(0, [0] * 15, records=None)
out = ex.drain(materialize_last=True)
if rank == 0:
    counts, last = out
    assert [int(x[0]) for x in counts[0]] == [9, 4, 3] * PASSES and [int(x[0]) for x in counts[1]] == [5, 4, 0] * PASSES
    assert [int(v) for v in counts[1][0][1:]] == fake("chr2")[1]
    for p in range(PASSES):
        for k in range(3):
            got = ex.records_of(counts, p * 3 + k)
            want0 = fake(names[k])[0]; want0["gq"] += p
            assert np.array_equal(got[0], want0), (p, k)
            if k < 2:
                want1 = fake(other[k])[0]; want1["gq"] += p
                assert np.array_equal(got[1], want1), (p, k)
            else:
                assert len(got[1]) == 0
    assert np.array_equal(last[0]["tpos"], fake("chrX")[0]["tpos"]) and len(last[1]) == 0
    # the one-round form the weak-scaling bench uses (an int capacity for every rank)
print("EXCHANGE_OK" if rank == 0 else "")
ex1 = hdist.RecordExchange(rank, world, hdist.RecordExchange.plan(9)[0][0], depth=2, keep=True)
for p in range(4):
    recs, log = fake("chr1" if rank == 0 else "chr2")
    recs["gq"] += p
    ex1.submit(len(recs), log, records=recs)
out = ex1.drain()
if rank == 0:
    counts, _ = out
    for p in range(4):
        got = ex1.records_of(counts, p)
        w0 = fake("chr1")[0]; w0["gq"] += p
        w1 = fake("chr2")[0]; w1["gq"] += p
        assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1), p
    print("EXCHANGE1_OK")
# the plan is the rehearsal's count plus one record in 64 (+ 16): what travels is the count, not a quarter more
caps9 = hdist.RecordExchange.plan(9 if rank == 0 else 5)
assert caps9 == [[9 + 16, 5 + 16]], caps9
big = hdist.RecordExchange.plan(64000 if rank == 0 else 0)
assert big == [[64000 + 1000 + 16, 0]], big
# a pass with more records than planned (rank 1's second submit): no rank hangs, drain() raises the same error on both with
# the capacities to rebuild with; the rebuilt exchange carries the pass
ex2 = hdist.RecordExchange(rank, world, [[3, 3]], depth=2, keep=True)
def pass2(ex):
    for p in range(3):
        n = 5 if (rank == 1 and p == 1) else 2
        recs = np.zeros(n, RECORD_DTYPE); recs["tpos"] = 100 * rank + np.arange(n) + p
        ex.submit(n, [p] * 15, records=recs)
try:
    pass2(ex2)
    ex2.drain()
    raise SystemExit("no overflow reported")
except hdist.RecordExchangeOverflow as e:
    assert e.caps == [[3, 5 + 16]], e.caps
    ex3 = hdist.RecordExchange(rank, world, e.caps, depth=2, keep=True)
pass2(ex3)
out = ex3.drain()
if rank == 0:
    counts, _ = out
    assert [int(x[0]) for x in counts[1]] == [2, 5, 2]
    got = ex3.records_of(counts, 1)
    assert list(got[1]["tpos"]) == [101, 102, 103, 104, 105] and list(got[0]["tpos"]) == [1, 2]
    print("OVERFLOW_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_lpt_assign_balances_and_is_deterministic():
    lens = {"chr{}".format(i): l for i, l in enumerate([248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114,
                                                        107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57], 1)}
    a = hdist.lpt_assign(lens, 8)
    assert sorted(c for r in a for c in r) == sorted(lens)
    loads = [sum(lens[c] for c in r) for r in a]
    assert max(loads) <= 1.12 * (sum(lens.values()) / 8)
    assert a == hdist.lpt_assign(lens, 8)
    assert hdist.lpt_assign({"a": 5}, 4) == [["a"], [], [], []]


def test_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2", HIMUT_NO_TORCH="0")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "GATHER_OK" in outs[0] and "EXCHANGE_OK" in outs[0] and "EXCHANGE1_OK" in outs[0] and "OVERFLOW_OK" in outs[0], outs[0]


ERR_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from himut_amd import dist as hdist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
# a stretch of per-rank work in which every rank succeeds ...
parts = hdist.share_or_raise({"rank": rank})
assert [p["rank"] for p in parts] == list(range(world))
# ... and one in which rank 1 fails: both ranks must leave with RankError naming rank 1, none may hang
err = None
try:
    if rank == 1:
        raise KeyError("tag 'cs' not present")
except Exception as e:
    err = e
try:
    hdist.share_or_raise("payload", err)
except hdist.RankError as e:
    assert "rank 1: KeyError" in str(e), str(e)
    assert not dist.is_initialized()
    print("RANKERROR_OK")
    sys.exit(3)
print("NOT RAISED")
'''


def test_failed_rank_takes_every_rank_down_gloo(tmp_path):
    """a rank that raises in its share must not leave the peers blocked in the next collective."""
    script = tmp_path / "e.py"
    script.write_text(ERR_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29619", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert [p.returncode for p in procs] == [3, 3], "\n".join(outs)
    assert all("RANKERROR_OK" in o for o in outs), "\n".join(outs)
