"""Neutral read-batch layout shared by the BAM ingest, the synthetic generator,
the HIP library (``himut_push_reads``) and the test oracle.

One batch = the non-secondary alignments of ONE contig in BAM file order
(coordinate sorted, i.e. the order ``pysam.AlignmentFile.fetch`` would yield;
reference call sites ``caller.py:299`` / ``bamlib.py:14-32``).  All arrays are
structure-of-arrays; sequence and qualities keep the BAM-native encoding
(4-bit packed bases, high nibble first; raw Phred bytes) so ingest is a plain
concatenation.

Fields
------
tstart, tend : int32[n]   0-based reference start / exclusive end (CIGAR derived)
qstart       : int32[n]   leading soft-clip length (query_alignment_start)
qlen         : int32[n]   len(query_sequence), soft clips included
mapq         : uint8[n]
flag         : uint16[n]  SAM flag (secondary reads are dropped before batching)
qid          : int32[n]   index of the first read in the batch with the same
                          query name (== own index for unique names)
qoff         : int64[n]   base offset of the read in ``seq``/``bq``; multiple of 32
cs_off       : int64[n+1] byte offsets into ``cs``
seq          : uint8[total/2]  packed nibbles (=ACMGRSVTWYHKDBN), 2 bases / byte
bq           : uint8[total]    Phred qualities
cs           : uint8[...]      concatenated cs tag strings (no "cs:Z:" prefix)
tp           : uint8[n]   minimap2 ``tp`` tag character ('P', 'S', ...; 0 if absent)
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

NIB2CHAR = "=ACMGRSVTWYHKDBN"
CHAR2NIB = {c: i for i, c in enumerate(NIB2CHAR)}


@dataclass
class ReadBatch:
    name: str
    length: int
    tstart: np.ndarray
    tend: np.ndarray
    qstart: np.ndarray
    qlen: np.ndarray
    mapq: np.ndarray
    flag: np.ndarray
    qid: np.ndarray
    qoff: np.ndarray
    cs_off: np.ndarray
    seq: np.ndarray
    bq: np.ndarray
    cs: np.ndarray
    tp: np.ndarray
    qnames: Optional[list] = field(default=None, repr=False)

    @property
    def n(self) -> int:
        return int(self.tstart.shape[0])

    def total_read_bases(self) -> int:
        return int(self.qlen.astype(np.int64).sum())

    def query_sequence(self, i: int) -> str:
        o = int(self.qoff[i])
        n = int(self.qlen[i])
        b = self.seq[o >> 1:(o + n + 1) >> 1]
        nib = np.empty(b.shape[0] * 2, dtype=np.uint8)
        nib[0::2] = b >> 4
        nib[1::2] = b & 15
        return "".join(NIB2CHAR[x] for x in nib[:n])

    def query_qualities(self, i: int) -> np.ndarray:
        o = int(self.qoff[i])
        return self.bq[o:o + int(self.qlen[i])]

    def cs_tag(self, i: int) -> str:
        return bytes(self.cs[int(self.cs_off[i]):int(self.cs_off[i + 1])]).decode("ascii")

    def query_name(self, i: int) -> str:
        if self.qnames is not None:
            return self.qnames[i]
        return "ccs/{}".format(int(self.qid[i]))

    def validate(self) -> None:
        n = self.n
        assert self.cs_off.shape[0] == n + 1
        for a in (self.tend, self.qstart, self.qlen, self.mapq, self.flag, self.qid, self.qoff, self.tp):
            assert a.shape[0] == n
        if n:
            assert np.all(np.diff(self.tstart) >= 0), "reads must be coordinate sorted"
            assert np.all(self.qoff % 32 == 0)
            assert int(self.qoff[-1]) + int(self.qlen[-1]) <= self.bq.shape[0]
            assert self.seq.shape[0] * 2 >= self.bq.shape[0]

    def to_npz_dict(self) -> dict:
        d = {k: getattr(self, k) for k in ("tstart", "tend", "qstart", "qlen", "mapq", "flag", "qid", "qoff",
                                           "cs_off", "seq", "bq", "cs", "tp")}
        d["name"] = np.frombuffer(self.name.encode(), dtype=np.uint8)
        d["length"] = np.array([self.length], dtype=np.int64)
        return d

    @staticmethod
    def from_npz_dict(d, prefix: str = "") -> "ReadBatch":
        g = lambda k: np.ascontiguousarray(d[prefix + k])
        return ReadBatch(name=bytes(g("name")).decode(), length=int(g("length")[0]), tstart=g("tstart"),
                         tend=g("tend"), qstart=g("qstart"), qlen=g("qlen"), mapq=g("mapq"), flag=g("flag"),
                         qid=g("qid"), qoff=g("qoff"), cs_off=g("cs_off"), seq=g("seq"), bq=g("bq"), cs=g("cs"),
                         tp=g("tp"))


def batch_from_records(name, length, records) -> ReadBatch:
    """Build a batch from python records (tests / hand-made edge cases).

    ``records`` is a list of dicts with keys tstart, tend, qstart, seq (str),
    bq (sequence of int), cs (str), and optional mapq, flag, qname, tp.
    """
    n = len(records)
    tstart = np.array([r["tstart"] for r in records], dtype=np.int32)
    tend = np.array([r["tend"] for r in records], dtype=np.int32)
    qstart = np.array([r.get("qstart", 0) for r in records], dtype=np.int32)
    qlen = np.array([len(r["seq"]) for r in records], dtype=np.int32)
    mapq = np.array([r.get("mapq", 60) for r in records], dtype=np.uint8)
    flag = np.array([r.get("flag", 0) for r in records], dtype=np.uint16)
    tp = np.array([ord(r.get("tp", "P")) if r.get("tp", "P") else 0 for r in records], dtype=np.uint8)
    qnames = [r.get("qname", "ccs/{}".format(i)) for i, r in enumerate(records)]
    first = {}
    qid = np.zeros(n, dtype=np.int32)
    for i, q in enumerate(qnames):
        qid[i] = first.setdefault(q, i)
    qoff = np.zeros(n, dtype=np.int64)
    cs_off = np.zeros(n + 1, dtype=np.int64)
    o = 0
    c = 0
    for i, r in enumerate(records):
        qoff[i] = o
        cs_off[i] = c
        o += (len(r["seq"]) + 31) & ~31
        c += len(r["cs"])
    cs_off[n] = c
    seq = np.zeros(o // 2, dtype=np.uint8)
    bq = np.zeros(o, dtype=np.uint8)
    cs = np.zeros(c, dtype=np.uint8)
    for i, r in enumerate(records):
        s = r["seq"].upper()
        nib = np.array([CHAR2NIB[ch] for ch in s], dtype=np.uint8)
        if nib.shape[0] & 1:
            nib = np.append(nib, 0).astype(np.uint8)
        packed = (nib[0::2] << 4) | nib[1::2]
        seq[qoff[i] // 2: qoff[i] // 2 + packed.shape[0]] = packed
        bq[qoff[i]: qoff[i] + len(s)] = np.asarray(r["bq"], dtype=np.uint8)
        cs[cs_off[i]: cs_off[i + 1]] = np.frombuffer(r["cs"].encode("latin-1"), dtype=np.uint8)
    return ReadBatch(name=name, length=length, tstart=tstart, tend=tend, qstart=qstart, qlen=qlen, mapq=mapq,
                     flag=flag, qid=qid, qoff=qoff, cs_off=cs_off, seq=seq, bq=bq, cs=cs, tp=tp, qnames=qnames)
