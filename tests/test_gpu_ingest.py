"""The device-side BAM ingest: BGZF blocks inflated by the host pool into pinned windows, records
parsed on the GPU (himut_ingest_*, csrc/himut_ingest.h).  The device-parsed read batch must equal the host-parsed one
byte for byte; `himut call` on it must give the records of the push_reads path."""
import os

import numpy as np
import pytest

from tests import util
from tests.test_bamio import check_hand_packed, hand_packed_bam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def worker():
    from himut_amd.caller import Worker
    w = Worker(0)
    yield w
    w.close()


def _same(a, b):
    assert a.n == b.n
    for k in ("tstart", "tend", "qstart", "qlen", "mapq", "flag", "qid", "qoff", "cs_off", "seq", "bq", "cs", "tp"):
        x, y = getattr(a, k), getattr(b, k)
        assert x.shape == y.shape and np.array_equal(x, y), k


def test_hand_packed_record_parsed_on_the_device(worker, tmp_path):
    from himut_amd import bamio
    path = str(tmp_path / "hand.bam")
    seq, qual = hand_packed_bam(path)
    st = bamio.BamStream(path)
    assert not st.indexed and st.tname2tsize == {"ctg": 1000}
    res = st.ingest_contig(worker.ctx, "ctg")
    b = worker.ctx.download_reads(res, "ctg", 1000)
    check_hand_packed(b, st.sample(), seq, qual)
    _same(b, bamio.BamFile(path).batches["ctg"])


@pytest.mark.parametrize("index", [True, False])
@pytest.mark.parametrize("window_kb", [96, 1024])
def test_two_contig_bam_device_parse_equals_host_parse(worker, tmp_path, monkeypatch, index, window_kb):
    """A secondary alignment, a supplementary pair sharing a name, records that straddle BGZF blocks and windows; with
    the .bai (only the contig's blocks are inflated) and without (the blocks in front are hopped over)."""
    from himut_amd import bamio, synth
    s1 = synth.generate(synth.SynthConfig(seed=41, contig_len=60_000, depth=14.0, read_len_mean=2500, read_len_sd=700,
                                          read_len_min=801, read_len_max=5000, name="chr2"))
    s2 = synth.generate(synth.SynthConfig(seed=42, contig_len=35_000, depth=12.0, read_len_mean=2500, read_len_sd=500,
                                          read_len_min=800, read_len_max=5000, name="chr10", cs_long=True))
    s2.batch.flag[3] |= 0x100
    s2.batch.qid[7] = 5
    path = str(tmp_path / "x.bam")
    bamio.write_bam(path, [s1.batch, s2.batch], sample="sampleA")
    if not index:
        monkeypatch.setenv("HIMUT_INGEST_NO_INDEX", "1")
    host = bamio.BamFile(path, threads=2)
    st = bamio.BamStream(path, threads=3)
    assert st.indexed == index and st.sample() == "sampleA" and st.tname2tsize == host.tname2tsize
    for chrom in ("chr10", "chr2", "chr10"):           # any order, a contig twice
        before = st.inflated_bytes()
        res = st.ingest_contig(worker.ctx, chrom, window_bytes=window_kb << 10)
        if index:
            # a process that takes only this contig (its rank's share under torch.distributed.run) inflates this
            # contig's blocks and no others: its record bytes + at most a block either side
            own = sum(36 + 64 + int(q) + (int(q) + 1) // 2 for q in host.batches[chrom].qlen) + len(host.batches[chrom].cs)
            assert res["n_reads"] * 40 < st.inflated_bytes() - before <= own + 2 * 65536 + 512 * res["n_reads"]
        got = worker.ctx.download_reads(res, chrom, st.tname2tsize[chrom])
        _same(got, host.batches[chrom])
        assert res["read_bases"] == host.batches[chrom].total_read_bases()
    st.close()


def test_call_from_device_ingest_equals_call_from_pushed_reads(worker, tmp_path):
    from himut_amd import bamio, bamlib, synth, util as hutil
    from tests.test_gpu_parity import _run_hip
    s = synth.generate(synth.SynthConfig(seed=77, contig_len=450_000, name="chr7"))
    path = str(tmp_path / "y.bam")
    bamio.write_bam(path, [s.batch])
    chunks = [(c[1], c[2]) for c in hutil.chunkloci(("chr7", 0, 450_000))]
    ql, qu, md = bamlib.get_thresholds({"chr7": s.batch}, ["chr7"], {"chr7": 450_000})
    p = dict(util.CALL_DEFAULTS, qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md)
    want, wlog = _run_hip(worker, s.batch, chunks, p)
    st = bamio.BamStream(path)
    ctx = worker.ctx
    ctx.set_chunks(chunks)
    res = st.ingest_contig(ctx, "chr7", window_bytes=4 << 20)
    assert res["n_reads"] == s.batch.n
    # the thresholds from the device-parsed per-read fields equal the ones from the host batch
    ts, te, qlen, mapq, tp = ctx.ingest_read_meta(res["n_reads"])
    assert np.array_equal(ts, s.batch.tstart) and np.array_equal(qlen, s.batch.qlen) and np.array_equal(tp, s.batch.tp)
    ctx.run()
    got, glog = ctx.records(), ctx.log()
    assert glog == wlog and np.array_equal(got, want)
