"""BASELINE.json configs[2] on one GPU: the 24 GRCh38-length contigs at 30x through the strong-scaling driver
(himut_amd/genome.py -- the code bench.py --gpus N runs), as a process group of ONE rank so that the RCCL
calls, the planned point-to-point exchange and the gathered buffers are the ones a multi-rank run uses.
Checks the partition / ordering properties of every contig's gathered records and bit-exact parity with the oracle
on chr21 and chrY (the oracle finishes those in seconds; the larger contigs are covered by the properties)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch                      # before libhimut_hip.so: both bind one HIP runtime
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from concurrent.futures import ThreadPoolExecutor
from himut_amd import genome, dist as hdist
from himut_amd.util import natsorted
out = genome.run_genome(0, 1, 0, scale=float(sys.argv[2]), steps=2, keep_records=True)
dist.barrier(); dist.destroy_process_group()
sizes, names = genome.genome_sizes(float(sys.argv[2]))
from himut_amd import util as hutil
recs, logs = out["records_by_contig"], out["logs_by_contig"]
assert sorted(recs) == sorted(names) and len(names) == 24
span = sum(e - s + 1 for c in names for (_c, s, e) in hutil.chunkloci((c, 0, sizes[c])))                 # util.py:119-132
assert out["genome_bp"] == span, (out["genome_bp"], span)
assert out["reran"] == 0 and out["scaling"] == "strong"
tot_r = tot_c = 0
for c in names:
    r, lg = recs[c], logs[c]
    assert lg[1] == sum(lg[2:8]), (c, lg)                                 # every candidate lands in exactly one class
    key = r["tpos"].astype(np.int64) * 65536 + r["ref"].astype(np.int64) * 256 + r["alt"]
    assert np.all(np.diff(key) > 0), c                                    # natsorted, de-duplicated (caller.py:622)
    assert r["tpos"].min() >= 1 and r["tpos"].max() <= sizes[c] and r["status"].max() <= 10 and np.all(r["flags"] == 0)
    assert np.all(r["counts"][:, :4].sum(1) >= 1)
    assert len(r) > sizes[c] * 4e-3                                        # ~5.3 records per kb at this error model
    tot_r += len(r); tot_c += lg[1]
assert tot_r == out["records"] and tot_c == out["candidate_sites"]
# the packing the multi-GPU run would use: every contig once, nobody above chr1 + 12 %
plan = hdist.lpt_assign(sizes, 8)
assert sorted(x for p in plan for x in p) == sorted(names)
assert max(sum(sizes[x] for x in p) for p in plan) <= 1.12 * sum(sizes.values()) / 8
# bit-exact parity with the oracle on two whole contigs
from oracle import oracle as O
ql, qu, md = out["thresholds"]
P = dict(genome.CALL_PARAMS, qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
def check(c):
    b, chunks, pon, com = genome.contig_inputs(c, sizes[c], 30.0, names)
    orecs, olog = O.call(b, chunks, P, genome.GERMLINE_SNV_PRIOR, pon, com)
    assert olog == logs[c], (c, olog, logs[c])
    assert len(orecs) == len(recs[c]), c
    for f in ("tpos", "chunk", "phase_set", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"):
        assert np.array_equal(orecs[f], recs[c][f]), (c, f)
    return len(orecs)
with ThreadPoolExecutor(2) as pool:
    n = list(pool.map(check, ["chr21", "chrY"]))
print("GENOME_OK", json.dumps({k: out[k] for k in ("Mbp_per_s", "s_per_genome", "slowest_rank_device_s", "records", "candidate_sites")}), n)
'''


def test_genome_24_contigs_one_rank_group(tmp_path):
    script = tmp_path / "g.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    scale = os.environ.get("HIMUT_TEST_GENOME_SCALE", "1")
    p = subprocess.run([sys.executable, str(script), ROOT, scale], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert p.returncode == 0 and "GENOME_OK" in p.stdout, p.stdout[-4000:]
