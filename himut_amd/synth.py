"""ctypes front-end of the synthetic CCS generator (csrc/synth.cpp) plus the
side-VCF writers that go with it (SURVEY.md §8(d) recipe).  Test and bench
infrastructure only; the product path never imports this module."""
import ctypes
import os
from dataclasses import dataclass

import numpy as np

from . import build
from .readbatch import ReadBatch


class _SynthParams(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("contig_len", ctypes.c_int32),
        ("read_len_min", ctypes.c_int32),
        ("read_len_max", ctypes.c_int32),
        ("softclip_max", ctypes.c_int32),
        ("cs_long", ctypes.c_int32),
        ("threads", ctypes.c_int32),
        ("depth", ctypes.c_double),
        ("read_len_mean", ctypes.c_double),
        ("read_len_sd", ctypes.c_double),
        ("snp_rate", ctypes.c_double),
        ("het_frac", ctypes.c_double),
        ("hetalt_frac", ctypes.c_double),
        ("sub_rate", ctypes.c_double),
        ("ins_rate", ctypes.c_double),
        ("del_rate", ctypes.c_double),
        ("som_rate", ctypes.c_double),
        ("frac_noisy", ctypes.c_double),
        ("noisy_mult", ctypes.c_double),
        ("frac_lowmapq", ctypes.c_double),
        ("frac_lowbq", ctypes.c_double),
        ("bq93_prob", ctypes.c_double),
        ("frac_softclip", ctypes.c_double),
        ("pile_frac", ctypes.c_double),
        ("pile_mult", ctypes.c_double),
    ]


@dataclass
class SynthConfig:
    seed: int = 1
    contig_len: int = 100_000
    depth: float = 30.0
    read_len_mean: float = 15000.0
    read_len_sd: float = 2500.0
    read_len_min: int = 5000
    read_len_max: int = 25000
    snp_rate: float = 1e-3
    het_frac: float = 2.0 / 3.0
    hetalt_frac: float = 0.0
    sub_rate: float = 2e-4
    ins_rate: float = 1e-4
    del_rate: float = 1e-4
    som_rate: float = 1e-5
    frac_noisy: float = 0.02
    noisy_mult: float = 30.0
    frac_lowmapq: float = 0.02
    frac_lowbq: float = 0.02
    bq93_prob: float = 0.85
    frac_softclip: float = 0.0
    softclip_max: int = 0
    cs_long: bool = False
    pile_frac: float = 0.0
    pile_mult: float = 1.0
    threads: int = 0
    name: str = "chrS"


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = build.build_synth()
        lib = ctypes.CDLL(path)
        lib.synth_create.restype = ctypes.c_void_p
        lib.synth_create.argtypes = [ctypes.POINTER(_SynthParams)]
        for f in ("synth_n_reads", "synth_total_bases_padded", "synth_cs_total", "synth_n_snps"):
            getattr(lib, f).restype = ctypes.c_int64
            getattr(lib, f).argtypes = [ctypes.c_void_p]
        lib.synth_fill.restype = None
        lib.synth_fill.argtypes = [ctypes.c_void_p] * 14
        lib.synth_get_snps.restype = None
        lib.synth_get_snps.argtypes = [ctypes.c_void_p] * 5
        lib.synth_get_ref.restype = None
        lib.synth_get_ref.argtypes = [ctypes.c_void_p] * 2
        lib.synth_destroy.restype = None
        lib.synth_destroy.argtypes = [ctypes.c_void_p]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@dataclass
class SynthSample:
    batch: ReadBatch
    snp_pos: np.ndarray   # 0-based
    snp_ref: np.ndarray   # ASCII
    snp_alt: np.ndarray   # ASCII
    snp_gt: np.ndarray    # 1 hap0 only, 2 hap1 only, 3 hom-alt, 4 tri-allelic (two alts)
    ref: np.ndarray = None


def generate(cfg: SynthConfig, want_ref: bool = False) -> SynthSample:
    lib = _load()
    threads = cfg.threads or min(16, os.cpu_count() or 1)
    p = _SynthParams(seed=cfg.seed, contig_len=cfg.contig_len, read_len_min=cfg.read_len_min,
                     read_len_max=cfg.read_len_max, softclip_max=cfg.softclip_max, cs_long=int(cfg.cs_long),
                     threads=threads, depth=cfg.depth, read_len_mean=cfg.read_len_mean,
                     read_len_sd=cfg.read_len_sd, snp_rate=cfg.snp_rate, het_frac=cfg.het_frac, hetalt_frac=cfg.hetalt_frac,
                     sub_rate=cfg.sub_rate, ins_rate=cfg.ins_rate, del_rate=cfg.del_rate, som_rate=cfg.som_rate,
                     frac_noisy=cfg.frac_noisy, noisy_mult=cfg.noisy_mult, frac_lowmapq=cfg.frac_lowmapq,
                     frac_lowbq=cfg.frac_lowbq, bq93_prob=cfg.bq93_prob, frac_softclip=cfg.frac_softclip,
                     pile_frac=cfg.pile_frac, pile_mult=cfg.pile_mult)
    h = lib.synth_create(ctypes.byref(p))
    try:
        n = lib.synth_n_reads(h)
        tot = lib.synth_total_bases_padded(h)
        cst = lib.synth_cs_total(h)
        ns = lib.synth_n_snps(h)
        tstart = np.zeros(n, np.int32); tend = np.zeros(n, np.int32); qstart = np.zeros(n, np.int32)
        qlen = np.zeros(n, np.int32); mapq = np.zeros(n, np.uint8); flag = np.zeros(n, np.uint16)
        qid = np.zeros(n, np.int32); qoff = np.zeros(n, np.int64); cs_off = np.zeros(n + 1, np.int64)
        seq = np.zeros(tot // 2, np.uint8); bq = np.zeros(tot, np.uint8); cs = np.zeros(max(cst, 1), np.uint8)[:cst]
        tp = np.zeros(n, np.uint8)
        lib.synth_fill(h, _p(tstart), _p(tend), _p(qstart), _p(qlen), _p(mapq), _p(flag), _p(qid), _p(qoff),
                       _p(cs_off), _p(seq), _p(bq), _p(cs), _p(tp))
        spos = np.zeros(ns, np.int32); sref = np.zeros(ns, np.uint8); salt = np.zeros(ns, np.uint8)
        sgt = np.zeros(ns, np.uint8)
        lib.synth_get_snps(h, _p(spos), _p(sref), _p(salt), _p(sgt))
        ref = None
        if want_ref:
            ref = np.zeros(cfg.contig_len, np.uint8)
            lib.synth_get_ref(h, _p(ref))
    finally:
        lib.synth_destroy(h)
    batch = ReadBatch(name=cfg.name, length=cfg.contig_len, tstart=tstart, tend=tend, qstart=qstart, qlen=qlen,
                      mapq=mapq, flag=flag, qid=qid, qoff=qoff, cs_off=cs_off, seq=seq, bq=bq, cs=cs, tp=tp)
    return SynthSample(batch=batch, snp_pos=spos, snp_ref=sref, snp_alt=salt, snp_gt=sgt, ref=ref)


# --------------------------------------------------------------------------
# side VCFs (plain-text .vcf; what vcflib.load_pon / load_common_snp /
# load_phased_hetsnps of the reference read)

_VCF_HEAD = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsyn\n"


def _rs(seed, tag):
    return np.random.RandomState((seed * 1000003 + tag) % (2 ** 31 - 1))


def write_common_snps_vcf(path, sample: SynthSample, seed=0, keep_frac=0.5, decoy_rate=1e-4,
                          other_contig="chrOther", extra_sites=()):
    """Common-SNP file.  The reference's plain-.vcf loader keeps records from
    every contig EXCEPT the one being called (vcflib.py:434), keyed by position
    only, so the records that are meant to hit are written under a second
    contig name and a few same-contig records are added that must NOT hit."""
    rs = _rs(seed, 11)
    L = sample.batch.length
    name = sample.batch.name
    keep = (rs.rand(sample.snp_pos.shape[0]) < keep_frac) & (sample.snp_gt != 4)
    rows = [(other_contig, int(p), r, a, "PASS") for (p, r, a) in extra_sites]
    for p, r, a in zip(sample.snp_pos[keep], sample.snp_ref[keep], sample.snp_alt[keep]):
        rows.append((other_contig, int(p) + 1, chr(r), chr(a), "PASS"))
    nd = rs.poisson(decoy_rate * L)
    for _ in range(nd):
        p = int(rs.randint(1, L + 1))
        r, a = rs.choice(4, 2, replace=False)
        rows.append((other_contig, p, "ACGT"[r], "ACGT"[a], "PASS" if rs.rand() < 0.9 else "q10"))
    # same-contig records: ignored by the reference for this contig
    for p, r, a in list(zip(sample.snp_pos, sample.snp_ref, sample.snp_alt))[:50]:
        rows.append((name, int(p) + 1, chr(r), chr(a), "PASS"))
    rows.sort(key=lambda t: (t[0], t[1]))
    with open(path, "w") as o:
        o.write(_VCF_HEAD)
        for c, p, r, a, f in rows:
            o.write("{}\t{}\t.\t{}\t{}\t.\t{}\t.\tGT\t0/1\n".format(c, p, r, a, f))


def write_pon_vcf(path, sample: SynthSample, seed=0, rate=1e-4, extra_sites=()):
    """Panel-of-normal file: random (pos, ref, alt) plus ``extra_sites``
    (1-based pos, ref, alt) so that some true candidates are hit."""
    rs = _rs(seed, 13)
    L = sample.batch.length
    name = sample.batch.name
    rows = set()
    for _ in range(rs.poisson(rate * L)):
        p = int(rs.randint(1, L + 1))
        r, a = rs.choice(4, 2, replace=False)
        rows.add((p, "ACGT"[r], "ACGT"[a]))
    for p, r, a in extra_sites:
        rows.add((int(p), r, a))
    with open(path, "w") as o:
        o.write(_VCF_HEAD)
        for p, r, a in sorted(rows):
            o.write("{}\t{}\t.\t{}\t{}\t.\tPASS\t.\tGT\t0/1\n".format(name, p, r, a))


def write_phased_vcf(path, sample: SynthSample, block=200):
    """Phased hetSNP file: runs of ``block`` hetSNPs per phase set, PS = first
    position of the run (the reference keys phase sets by str(chunk_start),
    caller.py:292)."""
    name = sample.batch.name
    het = (sample.snp_gt == 1) | (sample.snp_gt == 2)
    pos = sample.snp_pos[het]
    ref = sample.snp_ref[het]
    alt = sample.snp_alt[het]
    gt = sample.snp_gt[het]
    with open(path, "w") as o:
        o.write(_VCF_HEAD)
        for k in range(pos.shape[0]):
            ps = int(pos[(k // block) * block]) + 1
            g = "1|0" if gt[k] == 1 else "0|1"
            o.write("{}\t{}\t.\t{}\t{}\t.\tPASS\t.\tGT:PS\t{}:{}\n".format(
                name, int(pos[k]) + 1, chr(ref[k]), chr(alt[k]), g, ps))
