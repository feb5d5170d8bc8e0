"""Shared helpers for the test-suite: golden fixture loading and the mapping
from a fixture's stored arguments to worker parameters."""
import json
import os

import numpy as np

from himut_amd.readbatch import ReadBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CALL_DEFAULTS = dict(min_qv=30, min_mapq=60, min_sequence_identity=0.99, min_gq=20, min_bq=93, min_trim=0.01,
                     max_mismatch_count=0, mismatch_window_size=20, min_ref_count=3, min_alt_count=1,
                     min_hap_count=3, somatic_snv_prior=1 / (10 ** 6), germline_snv_prior=1 / (10 ** 3),
                     germline_indel_prior=1 / (10 ** 4))

WORKER_CASES = ["worker_basic", "worker_sets", "worker_dense", "worker_dense_sets", "worker_longcs",
                "worker_boundary", "worker_flags", "worker_pon_params", "worker_insins"]
PHASE_CASES = ["worker_phase", "worker_phase_dense", "worker_phase_dup"]


def load_json(case):
    with open(os.path.join(GOLDEN, case + ".json")) as f:
        return json.load(f)


def load_case(case):
    exp = load_json(case)
    with np.load(os.path.join(GOLDEN, case + ".npz")) as z:
        batch = ReadBatch.from_npz_dict(z)
    return batch, exp


def params_of(exp):
    p = dict(CALL_DEFAULTS)
    p.update(exp.get("overrides", {}))
    p["qlen_lower_limit"] = exp["qlen_lower_limit"]
    p["qlen_upper_limit"] = exp["qlen_upper_limit"]
    p["md_threshold"] = exp["md_threshold"]
    return p


def chunks_of(exp):
    return [tuple(c) for c in exp["chunks"]]


def phase_of(exp):
    if "phase_sets" not in exp:
        return None
    ps = exp["phase_sets"]
    hetsnp = {k: [tuple(t) for t in v] for k, v in ps["hetsnp"].items()}
    return ps["hbit"], ps["hpos"], hetsnp


def expected_tuples(exp):
    return [tuple([exp["contig"]] + r) for r in exp["records"]]


def load_config1_reference():
    """BASELINE.json configs[0] as the reference itself scanned it (tests/golden/config1_reference.py):
    regenerates the 1 Mb / 30x batch from its seed, verifies it byte for byte against the stored
    checksums, and returns (batch, fixture)."""
    import zlib
    from himut_amd import synth
    exp = load_json("config1_reference")
    b = synth.generate(synth.SynthConfig(**exp["synth"])).batch
    for k, v in exp["checksums"].items():
        assert zlib.crc32(np.ascontiguousarray(getattr(b, k)).view(np.uint8)) == v, "generated batch differs: " + k
    return b, exp
