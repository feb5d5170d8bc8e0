"""Randomised parity of both paths with the oracle through ONE context (tools/fuzz_parity.py): contigs of changing length,
depth, error rates and read lengths, thresholds, chunkings and --phase, one after the other.  The rounds 60 .. 115 of seed 2
are the sequence on which a grown buffer that came back at its old address was taken for the old one (its new part was
never cleared: garbage candidates at the end of a contig) -- found by this fuzz in round 3; then a fresh stretch."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args", [["--seed", "2", "--start", "60", "--rounds", "116"], ["--seed", "7", "--rounds", "60"]])
def test_fuzz_parity_through_one_context(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--minutes", "4"] + args,
                       capture_output=True, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-6:])
    assert r.returncode == 0 and "fuzz ok" in r.stdout, tail + r.stderr[-2000:]
