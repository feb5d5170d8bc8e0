"""Host mirror of the reference's per-chromosome worker and its driver
(reference: src/himut/caller.py).

``get_somatic_substitutions`` keeps the reference's 30-argument signature
(caller.py:208-241) so it can stand in for the starmap target; the body hands
the contig to the HIP library through the C ABI (include/himut_hip.h) and turns
the integer records back into the reference's 12-tuples with the reference's
own divisions (bamlib.py:181-219, caller.py:174-192).  There is no CPU
implementation of the scan in this package: without the HIP library the call
raises.
"""
import numpy as np

from . import _ffi, gtlib
from .util import BASE2IDX

STATUS_NAMES = _ffi.STATUS_NAMES


def site_keys(sites):
    """Sorted uint64 keys (pos1 << 4 | ref << 2 | alt) of a set of
    (pos, ref, alt); entries outside ATGC can never equal a candidate."""
    ks = {(int(p) << 4) | (BASE2IDX[r] << 2) | BASE2IDX[a] for (p, r, a) in sites if r in BASE2IDX and a in BASE2IDX}
    return np.array(sorted(ks), dtype=np.uint64)


def pack_phase_sets(chunks, ps2hbit, ps2hpos, ps2hetsnp):
    """Flat per-chunk hetSNP arrays.  The reference keys the phase set of a
    chunk by str(chunk_start) (caller.py:292-295); a missing key gives empty
    lists (they are defaultdicts, vcflib.py:627-629)."""
    off = [0]
    hpos, href, halt, hbit = [], [], [], []
    for (s, _e) in chunks:
        key = str(s)
        hp = ps2hpos.get(key, [])
        hs = ps2hetsnp.get(key, [])
        hb = ps2hbit.get(key, [])
        for p, (_, r, a), b in zip(hp, hs, hb):
            hpos.append(int(p))
            href.append(ord(r) if len(r) == 1 else 0)
            halt.append(ord(a) if len(a) == 1 else 0)
            hbit.append(ord(b))
        off.append(len(hpos))
    return (np.array(off, np.int64), np.array(hpos, np.int32), np.array(href, np.uint8), np.array(halt, np.uint8),
            np.array(hbit, np.uint8))


def records_to_tuples(chrom, recs):
    """Integer device records -> the tuples of caller.py:351-620."""
    out = []
    for r in recs:
        c = [int(x) for x in r["counts"]]
        s = [int(x) for x in r["bqsum"]]
        read_depth = float(c[0] + c[1] + c[2] + c[3] + c[5])          # bamlib.py:213-219 (np.float64 there)
        ref = chr(r["ref"])
        alt = chr(r["alt"])
        status = STATUS_NAMES[int(r["status"])]
        ref_count = float(c[BASE2IDX[ref]])
        if status == "HetAltSite":                                    # caller.py:368-391, 174-192
            a1, a2 = chr(r["gt0"]), chr(r["gt1"])
            pidx, qidx = BASE2IDX[a1], BASE2IDX[a2]
            p_count, q_count = float(c[pidx]), float(c[qidx])
            pbq = s[pidx] / float(p_count)
            qbq = s[qidx] / float(q_count)
            alt_bq = "{:0.1f},{:0.1f}".format(pbq, qbq)
            alt_count = "{:0.0f},{:0.0f}".format(p_count, q_count)
            alt_vaf = "{:.2f},{:.2f}".format(p_count / float(read_depth), q_count / float(read_depth))
            alt = "{},{}".format(a1, a2)
        else:                                                         # bamlib.py:197-210
            alt_count = float(c[BASE2IDX[alt]])
            alt_vaf = alt_count / float(read_depth)
            alt_bq = s[BASE2IDX[alt]] / float(alt_count) if alt_count != 0 else 0.0
        ps = str(int(r["phase_set"])) if int(r["phase_set"]) >= 0 else "."
        out.append((chrom, int(r["tpos"]), ref, alt, status, int(r["gq"]), alt_bq, read_depth, ref_count, alt_count,
                    alt_vaf, ps))
    return out


class Worker:
    """A GPU-bound worker: one context, reusable across contigs."""

    def __init__(self, device=0):
        self.ctx = _ffi.Context(device)
        self._lut_prior = None

    def close(self):
        self.ctx.close()

    def configure(self, min_qv, min_mapq, qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq,
                  min_trim, max_mismatch_count, mismatch_window_size, md_threshold, min_ref_count, min_alt_count,
                  min_hap_count, germline_snv_prior, phase):
        self.ctx.set_params(min_qv=int(min_qv), min_mapq=int(min_mapq), qlen_lower_limit=int(qlen_lower_limit),
                            qlen_upper_limit=int(qlen_upper_limit), min_gq=int(min_gq), min_bq=int(min_bq),
                            max_mismatch_count=int(max_mismatch_count),
                            mismatch_window_size=int(mismatch_window_size), md_threshold=int(md_threshold),
                            min_ref_count=int(min_ref_count), min_alt_count=int(min_alt_count),
                            min_hap_count=int(min_hap_count), phase=1 if phase else 0,
                            min_sequence_identity=float(min_sequence_identity), min_trim=float(min_trim))
        if self._lut_prior != germline_snv_prior:
            self.ctx.set_gt_lut(*gtlib.build_tables(germline_snv_prior))
            self._lut_prior = germline_snv_prior

    def call_contig(self, batch, chunks, pon_keys=None, common_keys=None, phase_sets=None):
        """Runs the scan on one contig; returns (records array, 15 counters)."""
        ctx = self.ctx
        ctx.set_chunks(chunks)
        ctx.set_site_set(0, pon_keys if pon_keys is not None else np.zeros(0, np.uint64))
        ctx.set_site_set(1, common_keys if common_keys is not None else np.zeros(0, np.uint64))
        if phase_sets is not None:
            ctx.set_phase(*pack_phase_sets(chunks, *phase_sets))
        ctx.push_reads(batch)
        ctx.run()
        return ctx.records(), ctx.log()

    def call_resident(self, chunks, pon_keys=None, common_keys=None, phase_sets=None):
        """The same on the reads the context already holds (bamio.BamStream.ingest_contig put them in HBM)."""
        ctx = self.ctx
        ctx.set_chunks(chunks)
        ctx.set_site_set(0, pon_keys if pon_keys is not None else np.zeros(0, np.uint64))
        ctx.set_site_set(1, common_keys if common_keys is not None else np.zeros(0, np.uint64))
        if phase_sets is not None:
            ctx.set_phase(*pack_phase_sets(chunks, *phase_sets))
        ctx.run()
        return ctx.records(), ctx.log()


_default_worker = {}


def _worker_for(device):
    w = _default_worker.get(device)
    if w is None:
        w = Worker(device)
        _default_worker[device] = w
    return w


def get_somatic_substitutions(
    chrom, bam_file, common_snps, panel_of_normals, chunkloci_lst, phase_set2hbit_lst, phase_set2hpos_lst,
    phase_set2hetsnp_lst, min_qv, min_mapq, qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq,
    min_trim, max_mismatch_count, mismatch_window_size, md_threshold, min_ref_count, min_alt_count, min_hap_count,
    somatic_snv_prior, germline_snv_prior, germline_indel_prior, phase, non_human_sample, create_panel_of_normals,
    chrom2tsbs_lst, chrom2tsbs_log, device=0, read_batch=None, chrom2records=None, resident_worker=None,
):
    """Drop-in for himut.caller.get_somatic_substitutions (caller.py:208).

    ``bam_file`` is read with the package's own BAM reader unless a prebuilt
    ``read_batch`` is given.  ``somatic_snv_prior`` and ``germline_indel_prior``
    are accepted and unused, as in the reference.  Assigns
    chrom2tsbs_lst[chrom] / chrom2tsbs_log[chrom] exactly like caller.py:622-641 (with ``chrom2records`` the
    integer records are kept instead of the tuples: the driver's fast printer works from those)."""
    from . import vcflib
    pon_keys = com_keys = None
    human = not non_human_sample and not create_panel_of_normals
    if common_snps is not None and human and common_snps.endswith(".vcf"):          # caller.py:248-254
        com_keys = site_keys(vcflib.load_common_snp(chrom, common_snps))
    elif common_snps is not None and human and common_snps.endswith(".bgz"):        # caller.py:269-278
        com_keys = site_keys(vcflib.load_bgz_common_snp(chrom, common_snps))
    if panel_of_normals is not None and human and panel_of_normals.endswith(".vcf"):  # caller.py:256-262
        pon_keys = site_keys(vcflib.load_pon(chrom, panel_of_normals))
    elif panel_of_normals is not None and human and panel_of_normals.endswith(".bgz"):  # caller.py:280-289
        pon_keys = site_keys(vcflib.load_bgz_pon(chrom, panel_of_normals))
    if resident_worker is None and read_batch is None:
        from . import bamio
        read_batch = bamio.read_contig(bam_file, chrom)
    w = resident_worker if resident_worker is not None else _worker_for(device)
    w.configure(min_qv, min_mapq, qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq, min_trim,
                max_mismatch_count, mismatch_window_size, md_threshold, min_ref_count, min_alt_count, min_hap_count,
                germline_snv_prior, phase)
    chunks = [(int(s), int(e)) for (_c, s, e) in chunkloci_lst]
    phase_sets = (phase_set2hbit_lst, phase_set2hpos_lst, phase_set2hetsnp_lst) if phase else None
    if resident_worker is not None:     # the contig's reads are in HBM already (device-side ingest)
        recs, log = w.call_resident(chunks, pon_keys, com_keys, phase_sets)
    else:
        recs, log = w.call_contig(read_batch, chunks, pon_keys, com_keys, phase_sets)
    if chrom2records is not None:       # the driver prints from the integer records (vcflib.dump_records)
        chrom2records[chrom] = recs
    else:
        chrom2tsbs_lst[chrom] = records_to_tuples(chrom, recs)
    chrom2tsbs_log[chrom] = log


def call_somatic_substitutions(
    bam_file, ref_file, vcf_file, phased_vcf_file, common_snps, panel_of_normals, region, region_list, min_qv,
    min_mapq, min_sequence_identity, min_gq, min_bq, min_trim, max_mismatch_count, mismatch_window_size,
    min_ref_count, min_alt_count, min_hap_count, somatic_snv_prior, germline_snv_prior, germline_indel_prior, threads,
    phase, non_human_sample, reference_sample, create_panel_of_normals, version, out_file, devices=(0,),
    log_path="himut.log",
):
    """Driver of `himut call` (reference: caller.py:645-838), host side.

    Same argument list and the same outputs (out_file, *.single_molecule_mutations.vcf,
    ./himut.log).  Under ``torch.distributed.run`` (one process per GPU) every rank scans its
    LPT share of the contigs on its own GPU and rank 0 gathers the record buffers and writes
    the files; in a single process the contigs go through the GPUs in ``devices`` one after the
    other.  Unlike the reference it returns instead of calling
    sys.exit(0), and input problems raise instead of printing and exiting."""
    import time
    from . import bamio, bamlib, dist, util, vcflib
    t0 = time.time()
    if not out_file.endswith(".vcf"):
        raise ValueError("VCF file must have .vcf suffix")
    group = dist.join_group(devices)       # (rank, world, device) under torch.distributed.run, else None
    bam = bamio.BamStream(bam_file, threads if threads and threads > 1 else 0)
    tname2tsize = bam.tname2tsize
    chrom_lst, chrom2chunkloci_lst = util.load_loci(region, region_list, tname2tsize)       # caller.py:681-682
    ps2hbit, ps2hpos, ps2hetsnp = {}, {}, {}
    if phase:                                                                               # caller.py:683-689
        ps2hbit, ps2hpos, ps2hetsnp, chrom2chunkloci_lst = vcflib.load_phased_hetsnps(phased_vcf_file, chrom_lst,
                                                                                      tname2tsize)
    # The contigs are the unit of work (the reference's starmap axis, caller.py:766-810): this process takes its share
    # -- its rank's under torch.distributed.run, else everything, spread over ``devices`` -- and brings ONLY those
    # contigs' BGZF blocks in (the index beside the BAM says where they are): inflated by the host pool into pinned
    # windows, parsed and placed by the GPU, one resident context per contig.
    sizes = {c: tname2tsize[c] for c in chrom_lst}
    devices = list(devices) or [0]
    if group is not None:
        rank, world, dev = group
        share = [(c, dev) for c in dist.lpt_assign(sizes, world)[rank]]
    else:
        share = [(c, d) for d, contigs in zip(devices, dist.lpt_assign(sizes, len(devices))) for c in contigs]
    starts = bamlib.sample_starts(chrom_lst, tname2tsize)
    resident, samples = {}, {}

    def ingest_share():
        for chrom, dev in share:
            w = Worker(dev)
            resident[chrom] = w
            res = bam.ingest_contig(w.ctx, chrom)
            ts, te, ql_, mq_, tp_ = w.ctx.ingest_read_meta(res["n_reads"])
            # the thresholds are global (bamlib.py:137-178): what each contig contributes are the query lengths over
            # its sampled windows, a few thousand integers
            samples[chrom] = bamlib.sample_qlens(ts, te, ql_, mq_, tp_, starts[chrom])

    def close_share():
        for w in resident.values():
            w.close()
        resident.clear()

    if group is None:
        ingest_share()
    else:
        # a rank that fails on one of its contigs (a record without cs, an unsorted file) still joins the collective
        # and every rank leaves with the same error
        err = None
        try:
            ingest_share()
        except Exception as e:                  # noqa: BLE001 -- handed to every rank, re-raised there
            err = e
            close_share()
        samples = {c: s for p_ in dist.share_or_raise(samples, err) for c, s in p_.items()}
    qlen_lower_limit, qlen_upper_limit, md_threshold = bamlib.thresholds_from_samples(samples, chrom_lst)
    if create_panel_of_normals:                                                             # caller.py:707-718
        (min_bq, min_gq, min_qv, min_mapq, min_trim, min_hap_count, min_sequence_identity, phase) = util.load_pon_params()
    if non_human_sample:                                                                    # caller.py:720-723
        germline_snv_prior, germline_indel_prior = vcflib.get_germline_priors(chrom_lst, ref_file, vcf_file, reference_sample)
    # the header call passes (max_mismatch_count, mismatch_window_size) into parameters named
    # (mismatch_window, max_mismatch_count): reproduced (caller.py:742-743 vs vcflib.py:167-168)
    vcf_header = vcflib.get_himut_vcf_header(
        bam_file, vcf_file, phased_vcf_file, region, region_list, tname2tsize, common_snps, panel_of_normals, min_qv,
        min_mapq, qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq, min_trim,
        max_mismatch_count, mismatch_window_size, md_threshold, min_ref_count, min_alt_count, min_hap_count, threads,
        somatic_snv_prior, germline_snv_prior, germline_indel_prior, phase, non_human_sample, reference_sample,
        create_panel_of_normals, version, out_file, bam.sample())
    chrom2tsbs_lst, chrom2tsbs_log, chrom2records = {}, {}, {}

    def scan(chrom, dev):
        get_somatic_substitutions(
            chrom, bam_file, common_snps, panel_of_normals, chrom2chunkloci_lst[chrom],
            ps2hbit.get(chrom, {}), ps2hpos.get(chrom, {}), ps2hetsnp.get(chrom, {}), min_qv, min_mapq,
            qlen_lower_limit, qlen_upper_limit, min_sequence_identity, min_gq, min_bq, min_trim,
            max_mismatch_count, mismatch_window_size, md_threshold, min_ref_count, min_alt_count, min_hap_count,
            somatic_snv_prior, germline_snv_prior, germline_indel_prior, phase, non_human_sample,
            create_panel_of_normals, chrom2tsbs_lst, chrom2tsbs_log, device=dev, resident_worker=resident[chrom],
            chrom2records=chrom2records)

    def scan_share():
        for chrom, dev in share:
            scan(chrom, dev)
            resident.pop(chrom).close()        # the contig's reads leave HBM

    if group is None:
        scan_share()
    else:
        # one process per GPU (torch.distributed.run): one exchange at the end brings every contig's record buffer and
        # counters to rank 0, which writes the files; before it the ranks agree that every scan went through
        rank, world, dev = group
        err = None
        try:
            scan_share()
        except Exception as e:                  # noqa: BLE001
            err = e
            close_share()
        dist.share_or_raise(None, err)
        res = dist.gather_contig_results({c: (chrom2records[c], chrom2tsbs_log[c]) for c in chrom2records},
                                         chrom_lst, rank, world)
        if rank != 0:
            dist.leave_group()
            return None, None
        chrom2records = {c: r for c, (r, _) in res.items()}
        chrom2tsbs_log = {c: l for c, (_, l) in res.items()}
    vcflib.dump_call_log(chrom_lst, chrom2tsbs_log, path=log_path)                         # caller.py:812-817
    vcflib.dump_records(out_file, vcf_header, chrom_lst, chrom2records, bool(phase))       # = dump_sbs / dump_phased_sbs
    print("himut single molecule somatic mutation detection took {} minutes".format((time.time() - t0) / 60))
    if group is not None:
        dist.leave_group()
    return chrom2records, chrom2tsbs_log
