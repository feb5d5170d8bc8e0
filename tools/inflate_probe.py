#!/usr/bin/env python3
"""Kernel time of the GPU inflate on 4096 blocks of 64 KB of several kinds of data (what costs: literals, matches, headers)."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from himut_amd import _ffi
from tests.test_gpu_inflate import BLOCK, _pack

def raw(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()

rs = np.random.RandomState(1)
N = 65280
qual = rs.randint(40, 80, size=N).astype(np.uint8).tobytes()
text = (b":1234*at:77-acg+t" * 5000)[:N]
kinds = {
    "literals only, 1 header (huffman-only, mem 9)": raw(qual, 6, zlib.Z_HUFFMAN_ONLY, 9),
    "literals only, 4+ headers (huffman-only, mem 8)": raw(qual, 6, zlib.Z_HUFFMAN_ONLY, 8),
    "literals only, many headers (mem 1)": raw(qual, 6, zlib.Z_HUFFMAN_ONLY, 1),
    "stored": raw(qual, 0),
    "matches, long (text level 6)": raw(text, 6),
    "matches, rle zeros": raw(bytes(N), 6),
    "level 1 quality bytes": raw(qual, 1),
}
ctx = _ffi.Context(0)
for name, comp in kinds.items():
    streams = [(comp, N)] * 4096
    c, blocks, total = _pack(streams)
    best = 1e9
    for _ in range(2):
        out, status, ms = ctx.inflate_blocks(c, blocks, total)
        assert status == 0
        best = min(best, ms)
    print("%-52s comp %6d B  %8.2f ms  %6.2f GB/s" % (name, len(comp), best, total / 1e6 / best))
