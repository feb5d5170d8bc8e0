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


def test_kept_capacities_small_large_small():
    """A context keeps the candidate / column capacities of its last sized run and launches the next
    one without waiting for the counts (design 6).  A larger contig after a smaller one
    overflows them and is repeated with exact sizes; a smaller one afterwards runs inside capacities
    far larger than it needs.  Every run must equal the oracle, whichever way it was sized."""
    import numpy as np
    from oracle import oracle as O
    from himut_amd import synth, util as hutil
    from himut_amd.caller import Worker
    from tests.test_gpu_parity import _run_hip
    sizes = [60_000, 900_000, 60_000, 250_000, 900_000]
    w = Worker(0)
    try:
        for k, L in enumerate(sizes):
            s = synth.generate(synth.SynthConfig(seed=70 + (k % 2), contig_len=L, read_len_mean=6000, read_len_sd=1200,
                                                 read_len_min=2000, read_len_max=12000, som_rate=1e-4, name="chr7"))
            b = s.batch
            chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
            p = dict(util.CALL_DEFAULTS, qlen_lower_limit=3000, qlen_upper_limit=11000, md_threshold=52)
            orecs, olog = O.call(b, chunks, p, p["germline_snv_prior"], None, None, None)
            for _ in range(2):        # the second pass of a size always runs on kept capacities
                hrecs, hlog = _run_hip(w, b, chunks, p)
                assert hlog == olog, (k, L)
                assert len(hrecs) == len(orecs)
                for name in ("tpos", "chunk", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"):
                    assert np.array_equal(hrecs[name], orecs[name]), (k, L, name)
                st = w.ctx.stats()
                assert st["n_records"] == len(orecs) and st["n_candidates"] >= len(orecs)
    finally:
        w.close()


def test_stage_timing_levels():
    """himut_set_stage_timing: which stage times a run reports (0 total, 1 + capture, 2 all)."""
    from himut_amd import synth, util as hutil
    from himut_amd.caller import Worker
    from tests.test_gpu_parity import _run_hip
    s = synth.generate(synth.SynthConfig(seed=77, contig_len=300_000, read_len_mean=6000, read_len_sd=1200,
                                         read_len_min=2000, read_len_max=12000, name="chr7"))
    b = s.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=3000, qlen_upper_limit=11000, md_threshold=52)
    w = Worker(0)
    try:
        ref = None
        for level in (1, 0, 2, 1):
            w.ctx.set_stage_timing(level)
            recs, log = _run_hip(w, b, chunks, p)
            ref = ref or (recs.tobytes(), log)
            assert (recs.tobytes(), log) == ref
            st = w.ctx.stats()
            assert st["ms_total"] > 0
            assert (st["ms_capture"] > 0) == (level >= 1)
            assert (st["ms_parse"] > 0) == (level >= 2) and (st["ms_eval"] > 0) == (level >= 2)
        with pytest.raises(Exception):
            w.ctx.set_stage_timing(3)
    finally:
        w.close()
