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
    if batch is not None:               # None: the contig's reads are in HBM already (bamio.BamStream.ingest_contig)
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
    resident_worker=None,
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
    if resident_worker is None and read_batch is None:
        from . import bamio
        read_batch = bamio.read_contig(bam_file, chrom)
    w = resident_worker if resident_worker is not None else _worker_for(device)
    w.configure(min_qv, min_mapq, qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq, min_trim,
                max_mismatch_count, mismatch_window, md_threshold, min_ref_count, min_alt_count, min_hap_count,
                germline_snv_prior, phase)
    chunks = [(int(s), int(e)) for (_c, s, e) in chunkloci_lst]
    phase_sets = (phase_set2hbit_lst, phase_set2hpos_lst, phase_set2hetsnp_lst) if phase else None
    ccs, ref, log = norm_contig(w, read_batch, chunks, seq, pon_keys, com_keys, non_human_sample, phase_sets=phase_sets)
    chrom2ccs_callable_tri2count[chrom] = ccs
    chrom2ref_callable_tri2count[chrom] = ref
    chrom2norm_log[chrom] = log


# ---------------------------------------------------------------------------------------------
# The rest of `himut normcounts` (host side): inputs around the worker and the output table.
# Mirrors, in order of use: vcflib.get_thresholds (vcflib.py:666-700), mutlib.load_sbs96_counts /
# get_sbs96 (mutlib.py:1998-2102), reflib.get_chrom_tricount / get_genome_tricounts (reflib.py:11-61),
# mutlib.get_normcounts_cmdline (:2396-2479), dump_normcounts (:2482-2539), dump_norm_log (:2605-2640).

SUB_LST = ["C>A", "C>G", "C>T", "T>A", "T>C", "T>G"]                                    # mutlib.py:14
SBS96_LST = ["{}[{}]{}".format(f, sub, l) for f in "ACGT" for sub in SUB_LST for l in "ACGT"]   # mutlib.py:52-149
PURINE = set("AG")
PUR2PYR = {"A": "T", "T": "A", "G": "C", "C": "G", "N": "N"}                            # mutlib.py:15
NORM_LOG_ROWS = ["num_ccs", "num_bases", "num_unphased_bases", "num_het_bases", "num_hetalt_bases",
                 "num_homalt_bases", "num_homref_bases", "num_uncallable_bases", "num_md_filtered_bases",
                 "num_ab_filtered_bases", "num_low_gq_bases", "num_pon_filtered_bases", "num_pop_filtered_bases",
                 "num_callable_bases"]


def read_fasta(path):
    """name -> sequence exactly as the file spells it (pyfastx keeps the case; so does the worker).  The file is taken
    whole and the line ends are deleted record by record (bytes.translate): a 3 Gb genome in seconds, where a loop over
    its fifty million lines takes a minute."""
    seqs = {}
    with open(path, "rb") as fh:
        data = fh.read()
    # a record starts with '>' at the START of a line only (a '>' inside a description is text)
    for rec in data[1:].split(b"\n>") if data.startswith(b">") else _fasta_records(data):
        header, _, body = rec.partition(b"\n")
        fields = header.split()
        if not fields:
            continue
        seqs[fields[0].decode()] = body.translate(None, b"\n\r\t ").decode("latin-1")
    return seqs


def _fasta_records(data):
    """Records of a file that does not begin with '>' (leading blank lines): everything behind a '>' at a line start."""
    i = data.find(b"\n>")
    return data[i + 2:].split(b"\n>") if i >= 0 else []


def _open_sbs(sbs_file):
    """Lines of the SBS file `himut call` wrote: plain .vcf, or .vcf.bgz read as the gzip members it is made of
    (the reference goes through cyvcf2, vcflib.py:666-701, mutlib.py:2058-2102)."""
    if sbs_file.endswith(".vcf"):
        return open(sbs_file)
    if sbs_file.endswith(".vcf.bgz"):
        import gzip
        return gzip.open(sbs_file, "rt")
    raise ValueError("SBS file must end in .vcf or .vcf.bgz")


def get_thresholds(sbs_file):
    """qlen_lower_limit, qlen_upper_limit, md_threshold from the header `himut call` wrote."""
    opts, md = {}, None
    for line in _open_sbs(sbs_file):
        if line.startswith("##FILTER=<ID=HighDepth"):
            md = line.strip().split()[-1].replace('">', "")
        elif line.startswith("##himut_command"):
            arr = line.strip().replace("##himut_command=himut call", "").split()
            key = None
            for i, tok in enumerate(arr):       # alternating option / value, flags have no value
                if i % 2 == 0:
                    key = tok
                elif not tok.startswith("--"):
                    opts[key] = tok
        elif line.startswith("#CHROM"):
            break
    return int(opts["--qlen_lower_limit"]), int(opts["--qlen_upper_limit"]), float(md)


def get_sbs96(chrom, pos, ref, alt, refseq):
    """pos 0-based; purine references are reported on the other strand."""
    seq = refseq[chrom]
    if ref in PURINE:
        return "{}[{}>{}]{}".format(PUR2PYR.get(seq[pos + 1], "N"), PUR2PYR.get(ref, "N"), PUR2PYR.get(alt, "N"),
                                    PUR2PYR.get(seq[pos - 1], "N"))
    return "{}[{}>{}]{}".format(seq[pos - 1], ref, alt, seq[pos + 1])


def load_sbs96_counts(vcf_file, refseq, chrom_lst):
    """SBS96 counts of the PASS bi-allelic SNVs of ``chrom_lst``; classes that contain an N are dropped."""
    from .vcflib import VcfRecord
    per_chrom = {}
    contigs = []
    for line in _open_sbs(vcf_file):
        if line.startswith("##"):
            if line.startswith("##contig"):
                contigs.append(line.strip().replace("##contig=<ID=", "").split(",")[0])
            continue
        if line.startswith("#CHROM"):
            per_chrom = {t: {} for t in contigs}
            continue
        v = VcfRecord(line)
        if v.is_snp and v.is_pass:
            k = get_sbs96(v.chrom, v.pos - 1, v.ref, v.alt, refseq)
            d = per_chrom[v.chrom]                  # KeyError for a contig the header does not list, as the reference
            d[k] = d.get(k, 0) + 1
    counts = {k: 0 for k in SBS96_LST}
    for chrom in chrom_lst:
        for k, c in per_chrom[chrom].items():
            if k.count("N") == 0:
                counts[k] += c                       # KeyError for a class outside the 96, as the reference
    return counts


def load_sbs96_counts_device(ctx_for, vcf_file, refseq, chrom_lst):
    """load_sbs96_counts with the classification and the counting on the device (himut_sbs96_counts, k_sbs96): the VCF
    text is parsed here, the PASS bi-allelic SNVs of every contig go to the device as three arrays and the contig's
    string -- resident for the sweep anyway -- supplies the neighbours.  ``ctx_for(chrom)`` returns a context whose
    reference is that contig's string.  Raises what the reference raises: KeyError for a contig the header does not
    list or a class outside the 96, IndexError for a position at the end of the string."""
    from .vcflib import VcfRecord
    per_chrom = None
    contigs = []
    for line in _open_sbs(vcf_file):
        if line.startswith("##"):
            if line.startswith("##contig"):
                contigs.append(line.strip().replace("##contig=<ID=", "").split(",")[0])
            continue
        if line.startswith("#CHROM"):
            per_chrom = {t: ([], [], []) for t in contigs}
            continue
        v = VcfRecord(line)
        if v.is_snp and v.is_pass:
            p, r, a = per_chrom[v.chrom]            # KeyError for a contig the header does not list, as the reference
            p.append(v.pos - 1); r.append(ord(v.ref)); a.append(ord(v.alt))
    counts = {k: 0 for k in SBS96_LST}
    subs = ("C>A", "C>G", "C>T", "T>A", "T>C", "T>G")
    for chrom in chrom_lst:
        p, r, a = per_chrom[chrom]
        if not p:
            continue
        h = ctx_for(chrom).sbs96_counts(p, r, a)
        if h[98]:
            raise IndexError("string index out of range")
        if h[97]:
            raise KeyError("SBS96 class outside the 96 (a neighbour or alternative allele that is not A/C/G/T/N)")
        for s6 in range(6):
            for u in range(4):
                for d in range(4):
                    counts["{}[{}]{}".format("ACGT"[u], subs[s6], "ACGT"[d])] += int(h[s6 * 16 + u * 4 + d])
    return counts


def get_chrom_tricount(seq):
    """Trinucleotide counts of one contig (reflib.py:11-33): triplets whose FIRST base is N are skipped, purine
    centres are reverse-complemented.  Only the 32 pyrimidine-centred ACGT keys are kept (all the callers read)."""
    raw = np.frombuffer(seq.encode("ascii") if isinstance(seq, str) else bytes(seq), np.uint8)
    code = np.full(256, 4, np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    c = code[raw]
    if c.shape[0] < 3:
        return {t: 0 for t in TRI_LST}
    a, b, d = c[:-2], c[1:-1], c[2:]
    ok = (a < 4) & (b < 4) & (d < 4)            # a key with any other letter is not in tri_lst
    a, b, d = a[ok], b[ok], d[ok]
    pur = (b == 0) | (b == 2)                    # A or G in the middle: read the other strand
    f = np.where(pur, 3 - d, a)
    m = np.where(pur, 3 - b, b)
    l = np.where(pur, 3 - a, d)
    hist = np.bincount(f * 16 + m * 4 + l, minlength=64)
    return {t: int(hist["ACGT".index(t[0]) * 16 + "ACGT".index(t[1]) * 4 + "ACGT".index(t[2])]) for t in TRI_LST}


def get_chrom_tricount_device(ctx, seq):
    """The same counts from the device (the contig string goes to HBM for the sweep anyway)."""
    chars, cls = tri_classes(seq)
    ctx.set_reference(seq, cls, len(chars))
    h = ctx.ref_tricounts()
    return {t: int(h["ACGT".index(t[0]) * 16 + "ACGT".index(t[1]) * 4 + "ACGT".index(t[2])]) for t in TRI_LST}


def get_genome_tricounts(refseq, chrom_lst):
    tot = {t: 0 for t in TRI_LST}
    for chrom in chrom_lst:
        for t, c in get_chrom_tricount(refseq[chrom]).items():
            tot[t] += c
    return tot


def get_normcounts_cmdline(bam_file, ref_file, sbs_file, vcf_file, phased_vcf_file, min_qv, min_mapq,
                           min_sequence_identity, min_gq, min_bq, min_trim, mismatch_window, max_mismatch_count,
                           min_ref_count, min_alt_count, min_hap_count, common_snps, panel_of_normals,
                           somatic_snv_prior, germline_snv_prior, germline_indel_prior, threads, phase,
                           non_human_sample, reference_sample, out_file):
    param = ("--min_qv {} --min_mapq {} --min_sequence_identity {} --min_gq {} --min_bq {} --min_ref_count {} "
             "--min_alt_count {} --min_hap_count {} --min_trim {} --mismatch_window {} --max_mismatch_count {} "
             "--somatic_snv_prior {} --germline_snv_prior {} --germline_indel_prior {} --threads {} -o {}").format(
        min_qv, min_mapq, min_sequence_identity, min_gq, min_bq, min_ref_count, min_alt_count, min_hap_count, min_trim,
        mismatch_window, max_mismatch_count, somatic_snv_prior, germline_snv_prior, germline_indel_prior, threads,
        out_file)
    head = "##himut_command=himut normcounts -i {} --ref {} --sbs {}".format(bam_file, ref_file, sbs_file)
    if non_human_sample:
        tail = " --vcf {}{} {}{} --non_human_sample{}".format(
            vcf_file, " --phased_vcf {}".format(phased_vcf_file) if phase else "", param, " --phase" if phase else "",
            " --reference_sample" if reference_sample else "")
        return head + tail
    if reference_sample:
        return None          # the reference has no branch for --reference_sample without --non_human_sample
    if phase:
        return head + " --phased_vcf {} {} --common_snps {} --panel_of_normals {} --phase".format(
            phased_vcf_file, param, common_snps, panel_of_normals)
    return head + " {} --common_snps {} --panel_of_normals {}".format(param, common_snps, panel_of_normals)


def _trifreq(d):
    tot = sum(d.values())
    return {k: (0 if tot == 0 else v / float(tot)) for k, v in d.items()}


def _ratio(num, den):
    fn, fd = _trifreq(num), _trifreq(den)
    return {t: (0 if fd[t] == 0 else fn[t] / float(fd[t])) for t in TRI_LST}


def dump_normcounts(sbs96_counts, ref_tri2count, chrom2ref_callable_tri2count, chrom2ccs_callable_tri2count, cmdline,
                    out_file):
    ref_call = {t: sum(d[t] for d in chrom2ref_callable_tri2count.values()) for t in TRI_LST}
    ccs_call = {t: sum(d[t] for d in chrom2ccs_callable_tri2count.values()) for t in TRI_LST}
    r_ref = _ratio(ref_call, ref_tri2count)
    r_call = _ratio(ref_call, ccs_call)
    with open(out_file, "w") as o:
        o.write("{}\n".format(cmdline))
        o.write("\t".join(["sub", "tri", "sbs96", "counts", "normcounts", "ref_tri_ratio", "ref_ccs_tri_ratio",
                           "ref_tri_count", "ref_callable_tri_count", "ccs_callable_tri_count"]) + "\n")
        for k in SBS96_LST:
            sub, tri = k[2:5], k[0] + k[2] + k[6]
            count = sbs96_counts[k]
            norm = count * r_ref[tri] * r_call[tri]
            o.write("{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\n".format(sub, tri, k, count, norm, r_ref[tri], r_call[tri],
                                                                      ref_tri2count[tri], ref_call[tri], ccs_call[tri]))


def dump_norm_log(chrom_lst, chrom2norm_log, path="norm.log"):
    dt = np.zeros((len(NORM_LOG_ROWS), len(chrom_lst)))
    for i, chrom in enumerate(chrom_lst):
        for j, count in enumerate(chrom2norm_log[chrom]):
            dt[j][i] = count
    with open(path, "w") as o:
        o.write("{:30}{}\n".format("", "\t".join(list(chrom_lst) + ["total"])))
        for k, row in enumerate(NORM_LOG_ROWS):
            cells = [str(int(r)) for r in dt[k].tolist()] + [str(int(np.sum(dt[k])))]
            o.write("{:30}{}\n".format(row, "\t".join(cells)))


def get_normcounts(bam_file, ref_file, sbs_file, vcf_file, phased_vcf_file, common_snps, panel_of_normals, region,
                   region_list, min_qv, min_mapq, min_sequence_identity, min_gq, min_bq, min_trim, mismatch_window,
                   max_mismatch_count, min_ref_count, min_alt_count, min_hap_count, somatic_snv_prior,
                   germline_snv_prior, germline_indel_prior, threads, phase, non_human_sample, reference_sample,
                   out_file, devices=(0,), log_path="norm.log"):
    """Driver of `himut normcounts` (normcounts.py:424-592): same arguments, the same table and norm.log; the PDF
    plot is left out.  Contigs go to the GPUs of ``devices`` round-robin.  A contig's reads come in through the
    device-side ingest (bamio.BamStream), one contig at a time: a process inflates only the BGZF blocks of the contigs it
    sweeps itself."""
    from . import bamio, dist, util, vcflib
    from .caller import Worker
    group = dist.join_group(devices)       # (rank, world, device) under torch.distributed.run, else None
    bam = bamio.BamStream(bam_file, threads if threads and threads > 1 else 0)
    tname2tsize = bam.tname2tsize
    chrom_lst, chrom2chunkloci_lst = util.load_loci(region, region_list, tname2tsize)
    ps2hbit, ps2hpos, ps2hetsnp = {}, {}, {}
    if phase:
        ps2hbit, ps2hpos, ps2hetsnp, chrom2chunkloci_lst = vcflib.load_phased_hetsnps(phased_vcf_file, chrom_lst,
                                                                                      tname2tsize)
    if non_human_sample:                                                                    # normcounts.py:487-490
        germline_snv_prior, germline_indel_prior = vcflib.get_germline_priors(chrom_lst, ref_file, vcf_file, reference_sample)
    qlen_lower_limit, qlen_upper_limit, md_threshold = get_thresholds(sbs_file)
    refseq = read_fasta(ref_file)
    ccs, ref, log = {}, {}, {}

    def sweep(chrom, dev):
        w = Worker(dev)
        try:
            bam.ingest_contig(w.ctx, chrom)
            sweep_resident(chrom, dev, w)
        finally:
            w.close()                      # the contig's reads leave HBM

    def sweep_resident(chrom, dev, w):
        get_callable_tricounts(
            chrom, refseq[chrom], bam_file, common_snps, panel_of_normals, chrom2chunkloci_lst[chrom],
            ps2hbit.get(chrom, {}), ps2hpos.get(chrom, {}), ps2hetsnp.get(chrom, {}), min_qv, min_mapq, min_trim,
            qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq, mismatch_window,
            max_mismatch_count, min_ref_count, min_alt_count, min_hap_count, md_threshold, somatic_snv_prior,
            germline_snv_prior, germline_indel_prior, phase, non_human_sample, ccs, ref, log,
            device=dev, resident_worker=w)

    if group is not None:
        # one process per GPU: each rank sweeps its LPT share of the contigs; the per-contig dictionaries (a few
        # hundred integers each) are collected on every rank and rank 0 writes the table
        rank, world, dev = group
        err = None
        try:
            for chrom in dist.lpt_assign({c: tname2tsize[c] for c in chrom_lst}, world)[rank]:
                sweep(chrom, dev)
        except Exception as e:              # noqa: BLE001 -- every rank leaves with the same error (dist.share_or_raise)
            err = e
        parts = dist.share_or_raise((ccs, ref, log), err)
        dist.leave_group()
        if rank != 0:
            return None, None, None
        ccs, ref, log = {}, {}, {}
        for c_, r_, l_ in parts:
            ccs.update(c_); ref.update(r_); log.update(l_)
    else:
        for k, chrom in enumerate(chrom_lst):
            sweep(chrom, devices[k % len(devices)])
    cmdline = get_normcounts_cmdline(bam_file, ref_file, sbs_file, vcf_file, phased_vcf_file, min_qv, min_mapq,
                                     min_sequence_identity, min_gq, min_bq, min_trim, mismatch_window,
                                     max_mismatch_count, min_ref_count, min_alt_count, min_hap_count, common_snps,
                                     panel_of_normals, somatic_snv_prior, germline_snv_prior, germline_indel_prior,
                                     threads, phase, non_human_sample, reference_sample, out_file)
    # SBS96 classes of the called substitutions, counted on the device against each contig's string
    from . import caller

    def ctx_for(chrom):
        ctx = caller._worker_for(devices[0] if group is None else group[2]).ctx
        chars, cls = tri_classes(refseq[chrom])
        ctx.set_reference(refseq[chrom], cls, len(chars))
        return ctx
    sbs2count = load_sbs96_counts_device(ctx_for, sbs_file, refseq, chrom_lst)
    ref_tri2count = get_genome_tricounts(refseq, chrom_lst)
    dump_normcounts(sbs2count, ref_tri2count, ref, ccs, cmdline, out_file)
    dump_norm_log(chrom_lst, log, log_path)
    return ccs, ref, log
