"""`himut phase`: the edge counting (``himut.phaselib.get_edges``, src/himut/phaselib.py:16-67) runs on the
device behind libhimut_hip.so; graph building, the binomial test, the haplotype blocks and the driver
(phaselib.py:70-323) stay where the reference has them, on the host."""
import time
from collections import deque

import numpy as np

from .caller import _worker_for


def edge_band(batch, hpos):
    """Largest number of hetSNPs one read spans, minus one (at least 1)."""
    hp = np.asarray(hpos, np.int64)
    if batch.n == 0 or hp.shape[0] == 0:
        return 1
    k = np.searchsorted(hp, batch.tend, side="right") - np.searchsorted(hp, batch.tstart, side="right")
    return max(1, int(k.max()) - 1)


def get_edges(chrom, bam_file, min_bq, min_mapq, hpos_lst, hetsnp_lst, hetsnp2hidx, device=0, read_batch=None):
    """Drop-in for himut.phaselib.get_edges: (edge_lst, edge2counts) with the reference's keys (pairs of hidx in
    natural order) and four float counts per edge (cis1, cis2, trans1, trans2)."""
    if read_batch is None:
        from . import bamio
        read_batch = bamio.read_contig(bam_file, chrom)
    w = _worker_for(device)
    ctx = w.ctx
    if w._lut_prior is None:              # the cs decode only needs some parameter block
        w.configure(0, 0, 0, 1 << 30, 0.0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0, 1 / (10 ** 3), False)
    hpos = np.asarray(hpos_lst, np.int32)
    href = np.array([ord(h[1]) for h in hetsnp_lst], np.uint8)
    band = edge_band(read_batch, hpos)
    ctx.push_reads(read_batch)
    counts = ctx.run_edges(hpos, href, min_bq, min_mapq, band).reshape(-1, 4)
    hidx = [hetsnp2hidx[h] for h in hetsnp_lst]
    edge2counts = {}
    for e in np.flatnonzero(counts.sum(1)):
        i, d = int(e) // band, int(e) % band
        edge2counts[(hidx[i], hidx[i + 1 + d])] = counts[e].astype(np.float64)
    return sorted(edge2counts), edge2counts


# --------------------------------------------------------------------------
# graph, binomial test, haplotype blocks (phaselib.py:70-195)

def table2binom_test(table):
    """Two-sided binomial p-value of the cis count among all four counts, p = 0.5 (phaselib.py:70-74;
    scipy.stats.binom_test there, binomtest here: same exact method)."""
    from scipy.stats import binomtest
    return float(binomtest(int(table[0] + table[1]), int(np.sum(table)), 0.5, alternative="two-sided").pvalue)


def build_graph(edge_lst, edge2counts):
    """Adjacency lists in edge order; edges without counts are dropped (phaselib.py:77-88)."""
    graph = {}
    for i, j in edge_lst:
        if np.sum(edge2counts[i, j]) == 0:
            continue
        graph.setdefault(i, []).append(j)
        graph.setdefault(j, []).append(i)
    return graph


def get_phased_graph(edge_lst, edge2counts, min_p_value, min_phase_proportion):
    """Nodes whose share of significantly cis- or trans-skewed edges exceeds ``min_phase_proportion``, each with
    those edges (phaselib.py:91-119).  Nodes are visited in ascending order, so an edge's p-value is computed at
    its smaller end and looked up at the larger one."""
    graph = build_graph(edge_lst, edge2counts)
    p_of = {}
    phased = {}
    for i in sorted(graph):
        keep = []
        for j in graph[i]:
            if i < j:
                p_of[(i, j)] = table2binom_test(edge2counts[(i, j)])
            if p_of[(min(i, j), max(i, j))] < min_p_value:
                keep.append(j)
        if len(keep) / len(graph[i]) > min_phase_proportion:
            phased[i] = keep
    return phased


def _same_haplotype(table):
    return table[0] + table[1] > table[2] + table[3]


def build_haplotype_block(edge_lst, edge2counts, min_p_value, min_phase_proportion):
    """Breadth-first labelling of the phased graph: the first node of a component is haplotype "0", a neighbour
    keeps the label across a mostly-cis edge and flips it across a mostly-trans one; blocks of two or more
    hetSNPs are returned as sorted (hidx, label) lists (phaselib.py:122-195).  Traversal order decides the labels
    when evidence conflicts, so it follows the reference: components in dictionary order of the phased graph,
    a first-in first-out queue of (parent, child) edges, a node labelled when it is first taken off the queue
    (the start node's own neighbours are labelled up front), nodes outside the phased graph never labelled from
    the queue."""
    graph = get_phased_graph(edge_lst, edge2counts, min_p_value, min_phase_proportion)
    seen = set()
    blocks = []
    for start in graph:
        if start in seen:
            continue
        seen.add(start)
        label = {start: "0"}
        todo = deque()
        for nb in graph[start]:
            label[nb] = "0" if _same_haplotype(edge2counts[(min(start, nb), max(start, nb))]) else "1"
            todo.append((start, nb))
        while todo:
            a, b = todo.popleft()
            if b in seen or b not in graph:
                continue
            seen.add(b)
            same = _same_haplotype(edge2counts[(min(a, b), max(a, b))])
            label[b] = label[a] if same else ("1" if label[a] == "0" else "0")
            for c in graph[b]:
                if c not in seen:
                    todo.append((b, c))
        if len(label) >= 2:
            blocks.append(sorted(label.items()))
    return blocks


def get_hblock_statistics(hblock_lst, hetsnp_lst):
    """(hetSNPs, blocks, smallest, largest block in hetSNPs, shortest, longest block in bp) (phaselib.py:198-232)."""
    sizes = [len(b) for b in hblock_lst]
    spans = [hetsnp_lst[b[-1][0]][0] - hetsnp_lst[b[0][0]][0] for b in hblock_lst]
    if not sizes:
        return len(hetsnp_lst), 0, 0, 0, 0, 0
    return len(hetsnp_lst), len(hblock_lst), min(sizes), max(sizes), min(spans), max(spans)


# --------------------------------------------------------------------------
# drivers (phaselib.py:235-323)

def get_hblock(chrom, chrom_len, bam_file, vcf_file, min_bq, min_mapq, min_p_value, min_phase_proportion,
               chrom2hblock_lst, device=0, read_batch=None):
    """Drop-in for himut.phaselib.get_hblock: assigns chrom2hblock_lst[chrom]."""
    from . import vcflib
    hetsnp_lst, _, hetsnp2hidx = vcflib.load_hetsnps(vcf_file, chrom, chrom_len)
    hpos_lst = [h[0] for h in hetsnp_lst]
    edge_lst, edge2counts = get_edges(chrom, bam_file, min_bq, min_mapq, hpos_lst, hetsnp_lst, hetsnp2hidx,
                                      device=device, read_batch=read_batch)
    chrom2hblock_lst[chrom] = build_haplotype_block(edge_lst, edge2counts, min_p_value, min_phase_proportion)


def get_chrom_hblock(bam_file, vcf_file, region, region_list, min_bq, min_mapq, min_p_value, min_phase_proportion,
                     threads, version, out_file, devices=(0,)):
    """`himut phase` (phaselib.py:253-323): phases the hetSNPs of every target contig and writes the phased VCF.
    ``threads`` feeds the BAM ingest; contigs go through the device one after the other."""
    import os
    from . import bamio, util, vcflib
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # `himut phase` is one process (its device work is the edge counts of one contig at a time); started under
        # torch.distributed.run every rank would phase every contig and write the same file
        raise RuntimeError("himut phase runs as a single process: start it without torch.distributed.run "
                           "(WORLD_SIZE={})".format(os.environ["WORLD_SIZE"]))
    t0 = time.time() / 60
    print("phasing hetsnps with {} threads".format(threads))
    bam = bamio.BamFile(bam_file, threads=threads)
    tname2tsize = bam.tname2tsize
    chrom_lst, _ = util.load_loci(region, region_list, tname2tsize)
    chrom2hblock_lst = {}
    for chrom in chrom_lst:
        get_hblock(chrom, tname2tsize[chrom], bam_file, vcf_file, min_bq, min_mapq, min_p_value, min_phase_proportion,
                   chrom2hblock_lst, device=devices[0], read_batch=bam.batches[chrom])
    print("finished phasing hetsnps")
    print("returning phased hetsnps")
    vcflib.dump_phased_hetsnps(bam_file, vcf_file, region, region_list, tname2tsize, min_bq, min_mapq, min_p_value,
                               min_phase_proportion, threads, chrom_lst, chrom2hblock_lst, version, out_file, bam.sample())
    print("finished returning phased hetsnps")
    print("haplotype phasing took {} minutes".format(time.time() / 60 - t0))
    return chrom2hblock_lst
