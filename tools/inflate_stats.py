#!/usr/bin/env python3
"""What the DEFLATE streams of a BAM's BGZF blocks are made of (symbols decoded by table / by the long-code path, matches
and their bytes and distances, dynamic headers): counted by the host build of the decoder the GPU runs
(csrc/himut_inflate.h).  usage: tools/inflate_stats.py [file.bam]   (default: a synthetic 30x BAM of 2 Mb)"""
import ctypes
import os
import struct
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from himut_amd import bamio, synth

L = bamio._load()
L.inflate_port.restype = ctypes.c_int
L.inflate_port.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    s = synth.generate(synth.SynthConfig(seed=3, contig_len=2_000_000, name="chr20"))
    path = os.path.join(tempfile.mkdtemp(), "x.bam")
    bamio.write_bam(path, [s.batch], sample="S")
raw = open(path, "rb").read()
st = (ctypes.c_longlong * 8)()
L.inflate_port_stats(st, 1)
p = n = tot = 0
while p < len(raw):
    xlen = struct.unpack_from("<H", raw, p + 10)[0]
    bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
    comp = raw[p + 12 + xlen:p + bsize - 8]
    isize = struct.unpack_from("<I", raw, p + bsize - 4)[0]
    src = np.frombuffer(comp + b"\0" * 192, np.uint8).copy()
    out = np.zeros(isize + 8, np.uint8)
    assert L.inflate_port(src.ctypes.data, len(comp), out.ctypes.data, isize) == 0
    p += bsize
    n += 1
    tot += isize
L.inflate_port_stats(st, 1)
names = ["symbols by table", "symbols by long-code path", "matches", "match bytes", "dynamic headers", "code-length symbols",
         "matches with distance <= 64", "matches with distance <= 1024"]
print("{} blocks, {} inflated bytes, {} compressed".format(n, tot, len(raw)))
for k, nm in enumerate(names):
    print("%-32s %12d   per block %9.1f" % (nm, st[k], st[k] / n))
