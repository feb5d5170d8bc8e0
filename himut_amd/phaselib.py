"""Host side of the edge counting of `himut phase`: mirror of ``himut.phaselib.get_edges``
(src/himut/phaselib.py:16-67) in front of libhimut_hip.so.  Graph building, the binomial test and
the haplotype blocks (phaselib.py:70-195) stay where the reference has them, on the host."""
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
