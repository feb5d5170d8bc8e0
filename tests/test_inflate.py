"""The DEFLATE decoder the GPU runs one lane per BGZF block (csrc/himut_inflate.h), compiled for the host: against
zlib on raw streams of every kind -- stored, fixed and dynamic blocks, every level and strategy, long matches, the
longest distances, several blocks per stream -- and on damaged streams (an error, never a crash or wrong bytes)."""
import ctypes
import zlib

import numpy as np
import pytest

from himut_amd import bamio


def _port():
    L = bamio._load()
    L.inflate_port.restype = ctypes.c_int
    L.inflate_port.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    return L


def _raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def _inflate(L, comp, n):
    src = np.frombuffer(comp + b"\0" * 192, np.uint8).copy()
    out = np.full(n + 8, 0xAA, np.uint8)
    rc = L.inflate_port(src.ctypes.data, len(comp), out.ctypes.data, n)
    assert out[n:].tobytes() == b"\xaa" * 8          # nothing behind the announced size
    return rc, out[:n].tobytes()


def _samples():
    rs = np.random.RandomState(5)
    qual = rs.randint(33, 127, size=60000).astype(np.uint8).tobytes()                     # high entropy: literals
    seq = rs.choice(np.frombuffer(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88", np.uint8), size=30000).tobytes()
    text = (b":1234*at:77-acg+t" * 4000)[:65280]                                            # repetitive: long matches
    runs = b"".join(bytes([i & 255]) * (1 + (i * 37) % 300) for i in range(400))[:65280]   # distance 1, lengths to 258
    far = rs.bytes(20000) + rs.bytes(12768) + b"needle" + rs.bytes(100)                     # matches 32 KB back
    far = far + far[:40000]
    bam_like = (qual[:20000] + seq[:10000] + text[:3000]) * 2
    return {"empty": b"", "one": b"x", "qual": qual, "seq": seq, "text": text, "runs": runs, "far": far[:65280],
            "bam_like": bam_like[:65280], "zeros": bytes(65280)}


@pytest.mark.parametrize("name", list(_samples()))
def test_port_equals_zlib(name):
    L = _port()
    data = _samples()[name]
    seen = set()
    for level in (0, 1, 2, 4, 6, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
            for mem in (1, 8):
                comp = _raw(data, level, strategy, mem)
                if comp in seen:
                    continue
                seen.add(comp)
                rc, out = _inflate(L, comp, len(data))
                assert rc == 0 and out == data, (name, level, strategy, mem, rc)
    assert len(seen) >= (1 if len(data) < 2 else 4)


def test_several_blocks_in_one_stream():
    """zlib starts a new DEFLATE block whenever its symbol buffer is full, and a full flush ends one: stored, fixed and
    dynamic blocks in one stream, the last one marked final."""
    L = _port()
    s = _samples()
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 1)
    parts = [s["qual"][:9000], s["text"][:9000], s["runs"][:9000], s["seq"][:9000]]
    comp = b""
    for k, p in enumerate(parts):
        comp += c.compress(p) + c.flush(zlib.Z_FULL_FLUSH if k % 2 else zlib.Z_SYNC_FLUSH)     # flushes emit empty stored blocks
    comp += c.flush()
    data = b"".join(parts)
    rc, out = _inflate(L, comp, len(data))
    assert rc == 0 and out == data


def test_damaged_streams_are_errors():
    L = _port()
    data = _samples()["bam_like"]
    comp = _raw(data, 6)
    assert _inflate(L, comp, len(data))[0] == 0
    assert _inflate(L, comp, len(data) - 1)[0] == 6           # more output than announced
    assert _inflate(L, comp, len(data) + 1)[0] != 0           # less
    assert _inflate(L, comp[:len(comp) // 2], len(data))[0] != 0          # truncated
    rs = np.random.RandomState(1)
    bad = 0
    for _ in range(200):
        b = bytearray(comp)
        for _ in range(3):
            b[rs.randint(0, len(b))] ^= 1 << rs.randint(0, 8)
        rc, out = _inflate(L, bytes(b), len(data))
        if rc != 0 or out != data:
            bad += 1
        assert rc != 0 or len(out) == len(data)
    assert bad > 150                                           # (a flipped bit in a literal's code can still give a valid stream)
    assert _inflate(L, b"\x07", 0)[0] == 1                     # block type 3
    assert _inflate(L, b"\x01\x05\x00\x00\x00hello", 5)[0] == 2           # stored: LEN / NLEN disagree
    assert _inflate(L, b"\x01\x05\x00\xfa\xffhello", 5) == (0, b"hello")


def test_bgzf_blocks_of_a_bam(tmp_path):
    """Every BGZF block of a BAM our writer made (zlib level 1), through the port, equals zlib's output."""
    import struct
    from himut_amd import synth
    L = _port()
    s = synth.generate(synth.SynthConfig(seed=9, contig_len=300_000, name="chr7"))
    path = str(tmp_path / "b.bam")
    bamio.write_bam(path, [s.batch], sample="x")
    raw = open(path, "rb").read()
    p = n = 0
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        comp = raw[p + 12 + xlen:p + bsize - 8]
        isize = struct.unpack_from("<I", raw, p + bsize - 4)[0]
        want = zlib.decompress(comp, -15)
        assert len(want) == isize
        rc, out = _inflate(L, comp, isize)
        assert rc == 0 and out == want
        p += bsize
        n += 1
    assert n > 50
