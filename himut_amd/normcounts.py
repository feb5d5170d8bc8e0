"""Host side of the normcounts sweep: mirror of ``himut.normcounts.get_callable_tricounts``
(src/himut/normcounts.py:206-421) in front of libhimut_hip.so.

The device returns two histograms over class triples and 14 counters; this module builds
the byte -> class table from the contig's reference string, supplies the order python gives
``set("ATGC").difference(ref)`` (the worker's PoN/common precedence and its tie rule depend on
it, normcounts.py:367-386), and turns the histograms back into the reference's dicts."""
import numpy as np

from .caller import BASE2IDX, _worker_for, site_keys

TRI_LST = [f + m + l for f in "ACGT" for m in "CT" for l in "ACGT"]     # mutlib.py:17-50
BASE_SET = set("ATGC")                                                   # util.py:15


def tri_classes(refseq):
    """One class per distinct byte of the reference string (A C G T N always present)."""
    raw = refseq.encode("ascii") if isinstance(refseq, str) else bytes(refseq)
    present = np.flatnonzero(np.bincount(np.frombuffer(raw, np.uint8), minlength=256))
    chars = sorted(set(int(x) for x in present) | set(b"ACGTN"))
    if len(chars) > 32:
        raise ValueError("reference string holds more than 32 distinct characters")
    cls = np.zeros(256, np.uint8)
    for i, ch in enumerate(chars):
        cls[ch] = i
    return chars, cls


def alt_order_table(order=None):
    """[ref allele][0..2] = alleles of list(base_set.difference(ref)) (normcounts.py:367) as THIS
    interpreter orders the set (PYTHONHASHSEED decides, exactly as for the reference)."""
    tab = np.zeros((4, 3), np.uint8)
    for ref, ri in BASE2IDX.items():
        lst = order[ref] if order is not None else list(BASE_SET.difference(ref))
        tab[ri] = [BASE2IDX[a] for a in lst]
    return tab


def tri_dicts(chars, ccs, ref):
    """Histograms over class triples -> ccs_tri2count, ref_tri2count (tri_lst keys always present,
    normcounts.py:247-249)."""
    K = len(chars)
    d_ccs = {t: 0 for t in TRI_LST}
    d_ref = {t: 0 for t in TRI_LST}
    for idx in np.flatnonzero((ccs != 0) | (ref != 0)):
        a, b, c = int(idx) // (K * K), (int(idx) // K) % K, int(idx) % K
        key = chr(chars[a]) + chr(chars[b]) + chr(chars[c])
        d_ccs[key] = d_ccs.get(key, 0) + int(ccs[idx])
        d_ref[key] = d_ref.get(key, 0) + int(ref[idx])
    return d_ccs, d_ref


def norm_contig(worker, batch, chunks, refseq, pon_keys=None, common_keys=None, non_human_sample=False, alt_order=None,
                phase_sets=None):
    """Runs the sweep on one contig through a configured caller.Worker; returns (ccs dict, ref dict, log[14]).
    phase_sets = (hbit, hpos, hetsnp) dicts of the contig when the worker was configured with phase=True."""
    from .caller import pack_phase_sets
    ctx = worker.ctx
    chars, cls = tri_classes(refseq)
    ctx.set_chunks(chunks)
    ctx.set_site_set(0, pon_keys if pon_keys is not None else np.zeros(0, np.uint64))
    ctx.set_site_set(1, common_keys if common_keys is not None else np.zeros(0, np.uint64))
    ctx.set_reference(refseq, cls, len(chars))
    if phase_sets is not None:
        ctx.set_phase(*pack_phase_sets(chunks, *phase_sets))
    ctx.push_reads(batch)
    ctx.run_normcounts(alt_order_table(alt_order), non_human_sample)
    ccs, ref, log = ctx.normcounts()
    d_ccs, d_ref = tri_dicts(chars, ccs, ref)
    return d_ccs, d_ref, log


def get_callable_tricounts(
    chrom, seq, bam_file, common_snps, panel_of_normals, chunkloci_lst, phase_set2hbit_lst, phase_set2hpos_lst,
    phase_set2hetsnp_lst, min_qv, min_mapq, min_trim, qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq,
    min_bq, mismatch_window, max_mismatch_count, min_ref_count, min_alt_count, min_hap_count, md_threshold,
    somatic_snv_prior, germline_snv_prior, germline_indel_prior, phase, non_human_sample,
    chrom2ccs_callable_tri2count, chrom2ref_callable_tri2count, chrom2norm_log, device=0, read_batch=None,
):
    """Drop-in for himut.normcounts.get_callable_tricounts (normcounts.py:206): same arguments, same three
    assignments."""
    from . import vcflib
    pon_keys = com_keys = None
    if common_snps is not None and common_snps.endswith(".vcf"):              # normcounts.py:251-253
        com_keys = site_keys(vcflib.load_common_snp(chrom, common_snps))
    elif common_snps is not None and common_snps.endswith(".bgz"):            # normcounts.py:262-271
        com_keys = site_keys(vcflib.load_bgz_common_snp(chrom, common_snps))
    if panel_of_normals is not None and panel_of_normals.endswith(".vcf"):    # normcounts.py:255-257
        pon_keys = site_keys(vcflib.load_pon(chrom, panel_of_normals))
    elif panel_of_normals is not None and panel_of_normals.endswith(".bgz"):  # normcounts.py:273-282
        pon_keys = site_keys(vcflib.load_bgz_pon(chrom, panel_of_normals))
    if read_batch is None:
        from . import bamio
        read_batch = bamio.read_contig(bam_file, chrom)
    w = _worker_for(device)
    w.configure(min_qv, min_mapq, qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq, min_trim,
                max_mismatch_count, mismatch_window, md_threshold, min_ref_count, min_alt_count, min_hap_count,
                germline_snv_prior, phase)
    chunks = [(int(s), int(e)) for (_c, s, e) in chunkloci_lst]
    phase_sets = (phase_set2hbit_lst, phase_set2hpos_lst, phase_set2hetsnp_lst) if phase else None
    ccs, ref, log = norm_contig(w, read_batch, chunks, seq, pon_keys, com_keys, non_human_sample, phase_sets=phase_sets)
    chrom2ccs_callable_tri2count[chrom] = ccs
    chrom2ref_callable_tri2count[chrom] = ref
    chrom2norm_log[chrom] = log
