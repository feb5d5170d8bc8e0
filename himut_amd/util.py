"""Chunk geometry and small helpers (reference: src/himut/util.py).

The chunk list decides results at chunk boundaries, so it is
reproduced exactly, including the reference's quirks."""
import re
from collections import defaultdict

BASE_LST = list("ATGC")                                   # util.py:14
BASE2IDX = {b: i for i, b in enumerate(BASE_LST)}         # util.py:17

CHUNK = 200000


def natural_key(s):
    """Digit-aware ordering of contig names (stand-in for natsort.natsorted,
    util.py:96; chr2 < chr10)."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


def natsorted(names):
    return sorted(names, key=natural_key)


def chunkloci(loci):
    """util.chunkloci (util.py:119-132): regions longer than 200 kb become
    (chrom, 1, 200000), (chrom, 200000, 400000), ..., and the last chunk ends
    at end - 2."""
    chrom, start, end = loci
    if end - start > CHUNK:
        out = [(chrom, 1, CHUNK)]
        starts = list(range(CHUNK, end, CHUNK))
        for i, s in enumerate(starts[:-1]):
            out.append((chrom, s, starts[i + 1]))
        if (chrom, starts[-1], end) not in out:
            out.append((chrom, starts[-1], end - 2))
        return out
    return [(chrom, start, end)]


def load_loci(region, region_list, tname2tsize):
    """util.load_loci (util.py:66-102)."""
    chrom2loci = defaultdict(list)
    if region_list is not None:
        for line in open(region_list).readlines():
            arr = line.strip().split()
            if len(arr) == 1:
                chrom2loci[arr[0]].append((arr[0], 0, tname2tsize[arr[0]]))
            else:
                chrom2loci[arr[0]].append((arr[0], int(arr[1]), int(arr[2])))
    elif region is not None:
        if region not in tname2tsize:
            raise KeyError("{} does not exist in the BAM file".format(region))
        chrom2loci[region].append((region, 0, tname2tsize[region]))
    else:
        for tname, tsize in tname2tsize.items():
            chrom2loci[tname].append((tname, 0, tsize))
    chrom_lst = natsorted(list(chrom2loci.keys()))
    chrom2chunks = {}
    for chrom, loci_lst in chrom2loci.items():
        chrom2chunks[chrom] = [c for loci in loci_lst for c in chunkloci(loci)]
    return chrom_lst, chrom2chunks


def load_pon_params():
    """Thresholds that --create_panel_of_normals switches to (util.py:44-63):
    (min_bq, min_gq, min_qv, min_mapq, min_trim, min_hap_count, min_sequence_identity, phase)."""
    return 20, 10, 20, 30, 0, 0, 0.8, False


def get_truncated_float(f):
    """f rounded to its first significant decimal place, for 0 <= f < 0.05 (util.py:539-544): the rounding one
    place past the last one that still gives 0.0.  Anything that never rounds to 0.0 raises ValueError (max() of
    an empty list), as in the reference."""
    rounded = [round(f, i) for i in range(1, 10)]
    return rounded[max(j for j, k in enumerate(rounded) if k == 0.0) + 1]
