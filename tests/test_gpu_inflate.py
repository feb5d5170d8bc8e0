"""The BGZF inflate on the device through the C ABI -- both decoders: a wave per block (csrc/himut_inflate_wave.h, the
default) and a lane per block (csrc/himut_inflate.h, HIMUT_INFLATE=lane): streams of every kind in one launch, and every
block of a BAM file, against zlib."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLOCK = np.dtype([("uoff", "<u8"), ("coff", "<u4"), ("clen", "<u4"), ("isize", "<u4"), ("pad", "<u4")])


@pytest.fixture(scope="module")
def ctx():
    from himut_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def _pack(streams):
    """streams: list of (compressed, inflated length) -> (comp bytes, block table, total inflated length)"""
    blocks = np.zeros(len(streams), BLOCK)
    comp, co, uo = [], 0, 0
    for k, (c, n) in enumerate(streams):
        blocks[k] = (uo, co, len(c), n, 0)
        comp.append(c)
        co += len(c)
        uo += n
    return b"".join(comp), blocks, uo


@pytest.fixture(params=["wave", "lane"])
def decoder(request, monkeypatch):
    monkeypatch.setenv("HIMUT_INFLATE", request.param)
    return request.param


def test_streams_of_every_kind_in_one_launch(ctx, decoder):
    from tests.test_inflate import _raw, _samples
    want, streams = [], []
    for name, data in _samples().items():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                streams.append((_raw(data, level, strategy), len(data)))
                want.append(data)
    comp, blocks, total = _pack(streams)
    out, status, ms = ctx.inflate_blocks(comp, blocks, total)
    assert status == 0
    assert out.tobytes() == b"".join(want)
    # a damaged block raises its bit and leaves the others alone
    bad = bytearray(comp)
    k = len(streams) // 2
    bad[blocks[k]["coff"]] = 0x07                       # block type 3
    out2, status2, _ = ctx.inflate_blocks(bytes(bad), blocks, total)
    assert status2 == 1 << 1
    a, b = int(blocks[k]["uoff"]), int(blocks[k]["uoff"]) + int(blocks[k]["isize"])
    assert out2[:a].tobytes() == out[:a].tobytes() and out2[b:].tobytes() == out[b:].tobytes()


def test_every_block_of_a_bam(ctx, tmp_path, decoder):
    from himut_amd import bamio, synth
    s = synth.generate(synth.SynthConfig(seed=19, contig_len=3_000_000, name="chr7"))
    path = str(tmp_path / "b.bam")
    bamio.write_bam(path, [s.batch], sample="x")
    raw = open(path, "rb").read()
    streams, want, p = [], [], 0
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        c = raw[p + 12 + xlen:p + bsize - 8]
        streams.append((c, struct.unpack_from("<I", raw, p + bsize - 4)[0]))
        want.append(zlib.decompress(c, -15))
        p += bsize
    comp, blocks, total = _pack(streams)
    out, status, ms = ctx.inflate_blocks(comp, blocks, total)
    assert status == 0 and len(streams) > 1000
    assert out.tobytes() == b"".join(want)
