"""Times the reference's own worker (caller.py:208) on BASELINE.json configs[0] — one 1 Mb
synthetic contig, 30x, no side VCFs, one process — in the build container, and checks the CPU
oracle against it on the same input (records and counters identical).  BAM decode is excluded:
reads are served from memory through the pysam stand-in of ref_harness.py.  Prints one JSON
line (the number is quoted in DESIGN.md §6) and writes tests/golden/config1_reference.json:
the reference's records and counters for that input plus checksums of the generated reads, so
the tests can regenerate the batch from its seed (the generator is deterministic) and compare
the oracle and the HIP path with the reference itself at the full config size.  Container-only,
like make_golden.py."""
import json
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("NPY_DISABLE_CPU_FEATURES", "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX2")

from himut_amd import caller, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import util  # noqa: E402
from tests.golden import ref_harness as H  # noqa: E402


def batch_checksums(b):
    return {k: zlib.crc32(np.ascontiguousarray(getattr(b, k)).view(np.uint8)) for k in
            ("tstart", "tend", "qstart", "qlen", "mapq", "flag", "qid", "qoff", "cs_off", "seq", "bq", "cs", "tp")}


def main():
    ref = H.load_reference()
    s = synth.generate(synth.SynthConfig(seed=1, contig_len=1_000_000, name="chr1"))
    b = s.batch
    bam = "/fake/config1.bam"
    H.register_bam(bam, {b.name: b})
    sizes = {b.name: b.length}
    ql, qu, md = ref.bamlib.get_thresholds(bam, [b.name], sizes)
    _, c2c = ref.util.load_loci(None, None, sizes)
    chunks = [(s_, e_) for (_, s_, e_) in c2c[b.name]]
    t0 = time.perf_counter()
    recs, log = H.run_reference_worker(bam, b.name, chunks, ql, qu, md)
    t_ref = time.perf_counter() - t0
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    t0 = time.perf_counter()
    orecs, olog = O.call(b, chunks, p, p["germline_snv_prior"], None, None, None)
    t_orc = time.perf_counter() - t0
    got = caller.records_to_tuples(b.name, orecs)
    same = [tuple(r) for r in recs] == got and list(log) == list(olog)
    span = sum(e - s_ + 1 for s_, e in chunks)
    print(json.dumps({"config": "1 Mb contig, 30x, {} reads, {} chunks, no side VCFs".format(b.n, len(chunks)),
                      "reference_s": round(t_ref, 2), "reference_Mbp_per_s": round(span / t_ref / 1e6, 4),
                      "oracle_s": round(t_orc, 3), "oracle_Mbp_per_s": round(span / t_orc / 1e6, 3),
                      "records": len(recs), "candidate_sites": int(log[1]), "oracle_identical": bool(same),
                      "cores": 1, "python": sys.version.split()[0], "numpy": np.__version__}))
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "config1_reference.json"), "w") as o:
        json.dump({"synth": {"seed": 1, "contig_len": 1_000_000, "name": "chr1"}, "contig": b.name,
                   "length": b.length, "checksums": batch_checksums(b), "chunks": chunks, "qlen_lower_limit": ql,
                   "qlen_upper_limit": qu, "md_threshold": md, "log": [int(x) for x in log],
                   "records": [[(float(x) if isinstance(x, (float, np.floating)) else
                                 int(x) if isinstance(x, (int, np.integer)) else x) for x in r[1:]] for r in recs],
                   "reference_seconds": round(t_ref, 2)}, o, indent=0, sort_keys=True)
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
