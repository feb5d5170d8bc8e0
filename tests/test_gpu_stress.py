"""Context reuse: the golden cases run repeatedly in random order through ONE context.
Catches state that leaks between contigs (stale device buffers, out-of-range probes)."""
import random

import pytest

from tests import util

pytestmark = pytest.mark.gpu


def test_random_order_reuse():
    from himut_amd import caller
    from himut_amd.caller import Worker
    from tests.test_gpu_parity import _run_hip
    w = Worker(0)
    try:
        data = {}
        for c in util.WORKER_CASES + util.PHASE_CASES:
            batch, exp = util.load_case(c)
            p = util.params_of(exp)
            pon = caller.site_keys([tuple(t) for t in exp["pon_set"]]) if "pon_set" in exp else None
            com = caller.site_keys([tuple(t) for t in exp["common_set"]]) if "common_set" in exp else None
            data[c] = (batch, exp, p, pon, com)
        rnd = random.Random(1)
        prev = None
        for _ in range(4):
            order = list(data)
            rnd.shuffle(order)
            for c in order:
                batch, exp, p, pon, com = data[c]
                recs, log = _run_hip(w, batch, util.chunks_of(exp), p, pon, com, util.phase_of(exp))
                got = caller.records_to_tuples(exp["contig"], recs)
                assert got == util.expected_tuples(exp), "{} after {}".format(c, prev)
                assert log == exp["log"], "{} after {}".format(c, prev)
                prev = c
    finally:
        w.close()
