#!/usr/bin/env python3
"""The BGZF inflate on the GPU against the host pool: every block of a synthetic 30x BAM of --contig-len in one launch.
Prints one JSON line: kernel ms, GB/s of inflated bytes, the host pool's time for the same blocks."""
import argparse
import json
import os
import struct
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig-len", type=int, default=16_000_000)
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    import numpy as np
    from himut_amd import _ffi, bamio, synth
    from tests.test_gpu_inflate import BLOCK
    s = synth.generate(synth.SynthConfig(seed=3, contig_len=a.contig_len, name="chr20"))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.bam")
        bamio.write_bam(path, [s.batch], sample="S")
        del s
        raw = np.fromfile(path, np.uint8)
        t0 = time.perf_counter()
        host = bamio.BamFile(path)
        t_host = time.perf_counter() - t0
    offs, p, uo, n = [], 0, 0, raw.shape[0]
    rb = raw.tobytes()
    while p < n:
        xlen = struct.unpack_from("<H", rb, p + 10)[0]
        bsize = struct.unpack_from("<H", rb, p + 16)[0] + 1
        isize = struct.unpack_from("<I", rb, p + bsize - 4)[0]
        offs.append((uo, p + 12 + xlen, bsize - 12 - xlen - 8, isize, 0))
        uo += isize
        p += bsize
    blocks = np.array(offs, BLOCK)
    ctx = _ffi.Context(0)
    best = None
    for _ in range(a.repeat):
        out, status, ms = ctx.inflate_blocks(rb, blocks, uo)
        assert status == 0, status
        best = ms if best is None else min(best, ms)
    assert out[:4].tobytes() == b"BAM\x01"
    print(json.dumps({"blocks": int(blocks.shape[0]), "compressed_MB": n / 1e6, "inflated_MB": uo / 1e6, "kernel_ms": best,
                      "inflated_GB_per_s": uo / 1e6 / best, "host_load_s (inflate + parse, pool)": t_host}))
    ctx.close()


if __name__ == "__main__":
    main()
