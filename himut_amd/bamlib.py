"""Sampling thresholds of `himut call` (reference: src/himut/bamlib.py:132-178).

Host-side by design: it runs once over the whole BAM before the
per-contig workers and yields three scalars."""
import math
import random

import numpy as np


def get_md_threshold(coverage):
    """bamlib.py:132-134"""
    return math.ceil(coverage + (4 * math.sqrt(coverage)))


SAMPLE_COUNT = 100
SAMPLE_RANGE = 100000


def sample_starts(chrom_lst, chrom2len):
    """The window starts of bamlib.py:150-153: one random stream (seed 10) consumed in contig order, so every process
    that lists the same contigs draws the same windows."""
    random.seed(10)
    return {chrom: random.sample(range(chrom2len[chrom]), SAMPLE_COUNT) for chrom in chrom_lst}


def sample_qlens(tstart, tend, qlen, mapq, tp, starts):
    """Query lengths of the primary ("tp:A:P", mapq > 0) reads over each sampled window of one contig, in the
    reference's order (window, then fetch order; bamlib.py:154-166).  int64 array."""
    primary = (mapq > 0) & (tp == ord("P"))                    # bamlib.py:161-163
    parts = []
    for start in starts:
        end = start + SAMPLE_RANGE
        hi = int(np.searchsorted(tstart, end, side="left"))           # reads with tstart < end
        sel = np.nonzero((tend[:hi] > start) & primary[:hi])[0]
        if sel.shape[0]:
            parts.append(qlen[sel].astype(np.int64))
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


def thresholds_from_samples(samples, chrom_lst):
    """qlen_lower_limit, qlen_upper_limit, md_threshold from the per-contig samples (bamlib.py:167-178).  The lengths
    are concatenated in contig order because np.mean / np.std sum pairwise in array order."""
    if len(chrom_lst) == 0:
        raise ValueError("target is missing")
    genome_sample_sum = SAMPLE_COUNT * SAMPLE_RANGE * len(chrom_lst)
    parts = [samples[c] for c in chrom_lst if samples[c].shape[0]]
    qlen_lst = np.concatenate(parts) if parts else np.zeros(0, np.int64)
    genome_read_sum = int(qlen_lst.sum())
    qlen_std = np.std(qlen_lst)
    qlen_mean = math.ceil(np.mean(qlen_lst))
    lower = math.ceil(qlen_mean - 2 * qlen_std)
    qlen_lower_limit = 0 if lower < 0 else lower
    qlen_upper_limit = math.ceil(qlen_mean + 2 * qlen_std)
    coverage = genome_read_sum / float(genome_sample_sum)
    return qlen_lower_limit, qlen_upper_limit, get_md_threshold(coverage)


def get_thresholds(batches, chrom_lst, chrom2len):
    """qlen_lower_limit, qlen_upper_limit, md_threshold from 100 random 100-kb
    windows per contig (bamlib.py:137-178).  ``batches``: contig -> ReadBatch."""
    if len(chrom_lst) == 0:
        raise ValueError("target is missing")
    starts = sample_starts(chrom_lst, chrom2len)
    samples = {}
    for chrom in chrom_lst:
        b = batches[chrom]
        samples[chrom] = sample_qlens(b.tstart, b.tend, b.qlen, b.mapq, b.tp, starts[chrom])
    return thresholds_from_samples(samples, chrom_lst)
