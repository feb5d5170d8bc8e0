#!/usr/bin/env python3
"""Host ingest rate: writes a synthetic 30x BAM for a contig of --contig-len, then times bamio.BamFile() with 1, 4, 8,
16 inflate threads.  CPU only."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig-len", type=int, default=16_000_000)
    a = ap.parse_args()
    from himut_amd import bamio, synth
    s = synth.generate(synth.SynthConfig(seed=3, contig_len=a.contig_len, name="chr1"))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.bam")
        bamio.write_bam(path, [s.batch])
        size = os.path.getsize(path)
        payload = s.batch.seq.nbytes + s.batch.bq.nbytes + s.batch.cs.nbytes
        out = {"bam_MB": size / 1e6, "payload_MB": payload / 1e6, "reads": int(s.batch.n), "runs": []}
        for th in (1, 4, 8, 16):
            best = 1e9
            for _ in range(2):
                t = time.perf_counter()
                bamio.BamFile(path, th)
                best = min(best, time.perf_counter() - t)
            out["runs"].append({"threads": th, "seconds": best, "bam_MB_per_s": size / 1e6 / best,
                                "read_Mbases_per_s": s.batch.total_read_bases() / 1e6 / best})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
