"""Sampling thresholds of `himut call` (reference: src/himut/bamlib.py:132-178).

Host-side by design: it runs once over the whole BAM before the
per-contig workers and yields three scalars."""
import math
import random

import numpy as np


def get_md_threshold(coverage):
    """bamlib.py:132-134"""
    return math.ceil(coverage + (4 * math.sqrt(coverage)))


def get_thresholds(batches, chrom_lst, chrom2len):
    """qlen_lower_limit, qlen_upper_limit, md_threshold from 100 random 100-kb
    windows per contig (bamlib.py:137-178).  ``batches``: contig -> ReadBatch.
    The query lengths are collected in the reference's order (contig, window,
    fetch order) because np.mean / np.std sum pairwise in array order."""
    if len(chrom_lst) == 0:
        raise ValueError("target is missing")
    random.seed(10)
    sample_count = 100
    sample_range = 100000
    genome_sample_sum = sample_count * sample_range * len(chrom_lst)
    parts = []
    for chrom in chrom_lst:
        b = batches[chrom]
        chrom_len = chrom2len[chrom]
        starts = random.sample(range(chrom_len), sample_count)
        primary = (b.mapq > 0) & (b.tp == ord("P"))                    # bamlib.py:161-163
        ts, te = b.tstart, b.tend
        for start in starts:
            end = start + 100000
            hi = int(np.searchsorted(ts, end, side="left"))           # reads with tstart < end
            sel = np.nonzero((te[:hi] > start) & primary[:hi])[0]
            if sel.shape[0]:
                parts.append(b.qlen[sel].astype(np.int64))
    qlen_lst = np.concatenate(parts) if parts else np.zeros(0, np.int64)
    genome_read_sum = int(qlen_lst.sum())
    qlen_std = np.std(qlen_lst)
    qlen_mean = math.ceil(np.mean(qlen_lst))
    lower = math.ceil(qlen_mean - 2 * qlen_std)
    qlen_lower_limit = 0 if lower < 0 else lower
    qlen_upper_limit = math.ceil(qlen_mean + 2 * qlen_std)
    coverage = genome_read_sum / float(genome_sample_sum)
    return qlen_lower_limit, qlen_upper_limit, get_md_threshold(coverage)
