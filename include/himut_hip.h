/*
 * himut_hip.h -- C ABI of libhimut_hip.so: the MI355X (gfx950) implementation of
 * himut's per-chromosome CCS pileup scan.
 *
 * The reference (sjin09/himut v1.0.4) has no plugin/FFI interface; its only seam
 * for this path is the Python worker
 *     himut.caller.get_somatic_substitutions      src/himut/caller.py:208-642
 * called once per contig by Pool.starmap (caller.py:805-808).  One himut_ctx
 * stands for one such worker bound to one GPU; the entry points below are what a
 * binding of that worker needs (INTEGRATION.md shows the ctypes stub):
 *
 *   worker argument / step (reference)                  entry point
 *   --------------------------------------------------  -------------------------
 *   16 scalar thresholds        caller.py:217-236       himut_set_params
 *   gtlib.init + log10 tables   gtlib.py:12-20,47-69    himut_set_gt_lut
 *   chunkloci_lst               caller.py:213,268       himut_set_chunks
 *   pon_sbs_set/common_snp_set  caller.py:245-262       himut_set_site_set
 *   phase_set2{hbit,hpos,hetsnp} caller.py:214-216      himut_set_phase
 *   alignments.fetch + BAM()    caller.py:299-300,
 *                               bamlib.py:14-32         himut_push_reads
 *   the body of the worker      caller.py:264-621       himut_run
 *   chrom2tsbs_lst[chrom]       caller.py:622-624       himut_get_records
 *   chrom2tsbs_log[chrom]       caller.py:625-641       himut_get_log
 *
 * Conventions: plain pointers and sizes only.  The caller owns every input
 * buffer and may free it when the call returns.  The library owns device
 * memory and the buffers returned by himut_get_records until the next
 * himut_run / himut_destroy.  Every function returns 0 on success or a
 * HIMUT_ERR_* code; himut_last_error() gives the message.  No C++ exception
 * crosses the boundary.  A context is single-threaded; contexts are independent
 * (one per GPU, driven by one host thread or process each).
 */
#ifndef HIMUT_HIP_H
#define HIMUT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIMUT_ABI_VERSION 2

typedef struct himut_ctx himut_ctx;

/* error codes (return values) */
enum {
    HIMUT_OK = 0,
    HIMUT_ERR_ARG = 1,          /* bad argument / call order */
    HIMUT_ERR_HIP = 2,          /* HIP runtime failure */
    HIMUT_ERR_CS = 3,           /* cs tag the reference's tokenizer (cslib.py:7-10) cannot split, or cs inconsistent
                                   with SEQ/CIGAR */
    HIMUT_ERR_BASE = 4,         /* KeyError in the reference: base outside ATGC (util.py:17) */
    HIMUT_ERR_BQ0 = 5,          /* ValueError in the reference: log10(0) for BQ 0 (gtlib.py:64) */
    HIMUT_ERR_CHUNK = 6,        /* chunk with start > end (pysam raises) */
    HIMUT_ERR_COVER = 7,        /* KeyError in tpos2qbase (haplib.py:51) */
    HIMUT_ERR_RESERVED8 = 8,    /* (never returned; kept so that the codes behind it do not move) */
    HIMUT_ERR_NOMEM = 9,
    HIMUT_ERR_DEPTH = 10        /* the contig's candidate columns need more than 2^32 column-store slots (or one
                                   256-position window holds more than 2^22 reads): split the contig's chunk list */
};

/* FILTER column values (caller.py:349-621, vcflib.py:189-209) */
enum {
    HIMUT_ST_PASS = 0, HIMUT_ST_LOWBQ = 1, HIMUT_ST_LOWGQ = 2, HIMUT_ST_INDEL = 3, HIMUT_ST_HET = 4,
    HIMUT_ST_HETALT = 5, HIMUT_ST_HOMALT = 6, HIMUT_ST_COMSNP = 7, HIMUT_ST_PON = 8, HIMUT_ST_LOWDEPTH = 9,
    HIMUT_ST_HIGHDEPTH = 10, HIMUT_ST_UNPHASED = 11
};

/* The scalar arguments of the worker (caller.py:217-236).  somatic_snv_prior and
 * germline_indel_prior are accepted by the reference but never read inside the
 * worker; germline_snv_prior enters through himut_set_gt_lut. */
typedef struct himut_params {
    int32_t min_qv;
    int32_t min_mapq;
    int32_t qlen_lower_limit;
    int32_t qlen_upper_limit;
    int32_t min_gq;
    int32_t min_bq;
    int32_t max_mismatch_count;
    int32_t mismatch_window_size;
    int32_t md_threshold;
    int32_t min_ref_count;
    int32_t min_alt_count;
    int32_t min_hap_count;
    int32_t phase;              /* 0 / 1 */
    int32_t reserved;
    double min_sequence_identity;
    double min_trim;
} himut_params;

/* The non-secondary alignments of one contig in BAM file order (coordinate
 * sorted).  Layout: himut_amd/readbatch.py.  seq is BAM 4-bit packed, bq raw
 * Phred, cs the concatenated cs:Z strings; qoff[i] (multiple of 32) is read i's
 * base offset into seq (in bases) and bq. */
typedef struct himut_read_batch {
    int64_t n_reads;
    const int32_t* tstart;      /* reference_start, 0-based */
    const int32_t* tend;        /* reference_end, exclusive (from CIGAR) */
    const int32_t* qstart;      /* query_alignment_start (leading soft clip) */
    const int32_t* qlen;        /* len(query_sequence) */
    const uint8_t* mapq;
    const uint16_t* flag;       /* SAM flag; 0x100 reads are skipped (bamlib.py:17) */
    const int32_t* qid;         /* index of the first read with the same query name */
    const int64_t* qoff;
    const int64_t* cs_off;      /* n_reads + 1 entries */
    const uint8_t* seq;
    const uint8_t* bq;
    const uint8_t* cs;
    int64_t seq_bytes;
    int64_t bq_bytes;
    int64_t cs_bytes;
} himut_read_batch;

/* One evaluated candidate, integers only: the host divides and formats exactly
 * as bamlib.py:181-219 / caller.py:174-192 / vcflib.py:820-1021 do. 64 bytes. */
typedef struct himut_record {
    int32_t tpos;               /* 1-based POS */
    int32_t chunk;              /* index of the chunk that evaluated it */
    int32_t phase_set;          /* chunk start for a phased PASS (caller.py:292,584), else -1 */
    int32_t gq;                 /* germ_gq (gtlib.py:138-174) */
    uint8_t ref, alt;           /* ASCII */
    uint8_t gt0, gt1;           /* germline genotype, reference allele first when het (gtlib.py:133-134) */
    uint8_t status;             /* HIMUT_ST_* */
    uint8_t gt_state;           /* 0 homref 1 het 2 hetalt 3 homalt */
    uint8_t flags;              /* internal; 0 in returned records */
    uint8_t pad;
    uint32_t counts[6];         /* A T G C ins del (util.py:14-20 order) at rpos = tpos - 1; ins = the reads with an insertion in
                                   front of the position (the reference counts insertion OPERATIONS, which is one more for a read
                                   whose cs holds two insertions in a row; nothing it prints depends on the number, only on != 0) */
    uint32_t bqsum[4];          /* sum of BQ per allele A T G C */
} himut_record;

/* Per-run figures for bench.py / DESIGN.md (times from hipEvents on the
 * context's stream, in milliseconds).  A stage time is 0 unless its events were
 * recorded: see himut_set_stage_timing. */
typedef struct himut_run_stats {
    double ms_total;
    double ms_parse;            /* k_parse_cs: cs decode, one wave per read (+ k_window_index and the fills beside it) */
    double ms_bqsum;            /* 0: the quality sum is taken inside k_stream_capture (kept for ABI layout) */
    double ms_hap;              /* k_read_hap (phase only) */
    double ms_emit;             /* k_propose (read filters, proposals -> mask), mask bit count + scan, k_mask_emit
                                   (+ sort when chunks are out of order): runs BEHIND the capture, which supplies
                                   the whole-read quality sums the read filter needs */
    double ms_index;            /* k_mark_positions (bitmap of substitution positions) + rank, column windows /
                                   offsets, column-store fill: runs in front of the capture */
    double ms_capture;          /* k_stream_capture: streams every read once, fills the column store, sums the
                                   qualities of every read.  After himut_run_normcounts: k_norm_quad, the sweep's
                                   dominant kernel, by itself (ms_parse = the decode, ms_index = the read pass,
                                   ms_eval = the whole position sweep) */
    double ms_eval;             /* k_eval_columns: counts, ordered likelihood sums, genotype, filters */
    double ms_finalize;         /* cross-chunk som_seen / counters / compaction */
    int64_t n_reads;
    int64_t read_bases;         /* sum of qlen */
    int64_t positions;          /* sum over chunks of (end - start + 1) */
    int64_t n_unique_positions; /* reserved */
    int64_t n_candidates;       /* evaluated candidates before the cross-chunk pass */
    int64_t n_records;
    int64_t column_slots;       /* column-store slots (unique candidate positions x reads in their windows) */
    int64_t reran;              /* 1 when the run was repeated with exact buffer sizes because a count exceeded the
                                   capacities kept from the previous run: ms_* then describe the second pass only
                                   and the wall time of himut_run covers both */
} himut_run_stats;

int himut_abi_version(void);
int himut_create(int device, himut_ctx** out);
void himut_destroy(himut_ctx* ctx);
const char* himut_last_error(const himut_ctx* ctx);

int himut_set_params(himut_ctx* ctx, const himut_params* p);
/* three tables of n_bq doubles indexed by BQ, and log10 priors in the order
 * homref, het, hetalt, homalt */
int himut_set_gt_lut(himut_ctx* ctx, const double* log_hom, const double* log_het, const double* log_err, int n_bq,
                     const double log_prior[4]);
int himut_set_chunks(himut_ctx* ctx, const int32_t* start, const int32_t* end, int64_t n_chunks);
/* which: 0 panel of normals, 1 common SNPs.  keys sorted ascending:
 * (pos1 << 4) | (ref << 2) | alt with A0 T1 G2 C3. */
int himut_set_site_set(himut_ctx* ctx, int which, const uint64_t* keys, int64_t n);
/* per chunk c (phase set keyed str(chunk_start), caller.py:292-295): hetSNPs
 * off[c]..off[c+1] with 1-based hpos, ASCII ref/alt (0 when not a single base)
 * and hbit '0'/'1'. */
int himut_set_phase(himut_ctx* ctx, const int64_t* off, const int32_t* hpos, const uint8_t* href,
                    const uint8_t* halt, const uint8_t* hbit, int64_t n_chunks);
int himut_push_reads(himut_ctx* ctx, const himut_read_batch* batch);
int himut_run(himut_ctx* ctx);
/* himut_run in two halves, for a caller that scans several contigs (a context each): begin queues the whole run and --
 * on any run but a context's first on its reads and chunks -- returns without waiting; end waits for the run's last
 * copy, checks it and makes the results available.  Runs of different contexts begun one after the other share the GPU
 * without the host in between.  A context must not be touched between its begin and its end. */
int himut_run_begin(himut_ctx* ctx);
int himut_run_end(himut_ctx* ctx);
int himut_get_records(himut_ctx* ctx, const himut_record** records, int64_t* n);
int himut_get_log(himut_ctx* ctx, int64_t out[15]);
int himut_get_stats(himut_ctx* ctx, himut_run_stats* out);
/* Which hipEvents himut_run records (each costs a barrier packet, i.e. a few microseconds of queue time):
 * 0 = run start / end only (ms_total), 1 = also around k_stream_capture (ms_capture; the default),
 * 2 = every stage of himut_run_stats.  No counterpart in the reference (it has no timers). */
int himut_set_stage_timing(himut_ctx* ctx, int level);

/* Device-resident copy of the result for the multi-GPU gather: number of
 * records, and a device-to-device copy into caller-provided device memory
 * (e.g. a torch tensor handed to RCCL). */
int himut_records_device(himut_ctx* ctx, const void** dev_ptr, int64_t* n);
int himut_copy_records_to_device(himut_ctx* ctx, void* dst_device, int64_t capacity_records);

/* ---- next row: the BAM ingest in front of the path (reference call sites caller.py:267,299:
 * pysam.AlignmentFile + alignments.fetch, and the record wrapper bamlib.BAM.__init__, bamlib.py:14-32).
 * An alternative to himut_push_reads: the contig's inflated BAM records go to HBM a window at a time and are parsed
 * there (CIGAR walk, tag scan, placement, byte copies: csrc/himut_ingest.h).  The caller inflates BGZF blocks straight
 * into one of the library's two pinned buffers (himut_ingest_buffer), lists where the records of the contig start
 * (rec_off: offset of each record's body, i.e. behind its block_size field, from byte `start` of the buffer, where the
 * window's nbytes begin) and which earlier record carries the same
 * read name (qid, as in himut_read_batch), and hands the window over; the copy of window k overlaps the inflate of
 * window k + 1 (two buffers: himut_ingest_wait(slot) returns once slot's bytes have left the host).  padded_bases =
 * sum of the records' l_seq rounded up to 32, tag_bytes = an upper bound of the cs text in the window (the bytes of
 * the records' auxiliary fields): what the library must have room for.  himut_ingest_end leaves the context as
 * himut_push_reads would.  The two pinned buffers belong to the PROCESS (pinning them takes tens of milliseconds): one
 * ingest may be open at a time -- himut_ingest_begin on a second context before the first's himut_ingest_end (or
 * himut_destroy) returns HIMUT_ERR_ARG.  libhimut_host.so's bam_stream_* (csrc/bam_ingest.cpp) is the host side that goes with it. */
typedef struct himut_ingest_result {
    int64_t n_reads, bases_padded, cs_bytes, read_bases;
    int64_t n_missing_cs;       /* records without a cs tag (the reference's get_tag("cs") raises KeyError, bamlib.py:32) */
    int64_t n_unsorted;         /* records in front of their predecessor: not coordinate sorted */
    int64_t n_malformed;
} himut_ingest_result;
int himut_ingest_begin(himut_ctx* ctx, int64_t inflated_bytes_bound, int64_t window_bytes);
void* himut_ingest_buffer(himut_ctx* ctx, int slot);
int himut_ingest_wait(himut_ctx* ctx, int slot);
int himut_ingest_window(himut_ctx* ctx, int slot, int64_t start, int64_t nbytes, const uint32_t* rec_off, const int32_t* qid,
                        int64_t n_rec, int64_t padded_bases, int64_t tag_bytes);
int himut_ingest_end(himut_ctx* ctx, int unique_qnames, himut_ingest_result* out);
/* per-read fields the host needs for bamlib.get_thresholds (bamlib.py:137-178); any pointer may be null */
int himut_ingest_read_meta(himut_ctx* ctx, int32_t* tstart, int32_t* tend, int32_t* qlen, uint8_t* mapq, uint8_t* tp);
/* the resident read batch back on the host (arrays of the caller, sized by the fields of `batch` on entry) */
int himut_download_reads(himut_ctx* ctx, himut_read_batch* batch, uint8_t* tp);

/* ---- next row: normcounts.get_callable_tricounts ---------------------------------
 * The worker's arguments (normcounts.py:206-241) map onto the same calls as the call path
 * (params, LUT, chunks, site sets, phase sets, reads) plus the contig's reference string:
 *   seq (str(refseq[chrom]), normcounts.py:504)          himut_set_reference
 *   the body of the worker (normcounts.py:243-402)       himut_run_normcounts
 *   chrom2{ccs,ref}_callable_tri2count, chrom2norm_log   himut_get_normcounts
 * seq is passed as the FASTA holds it (case matters: the reference skips positions whose base
 * is not an upper-case A/C/G/T).  cls[256] maps every byte of seq to a class id < n_classes
 * (one class per distinct byte, A C G T N always present, at most 32); the two histograms
 * are indexed (c0 * K + c1) * K + c2 over the class ids of the trinucleotide key.
 * alt_order[ref * 3 + i] = allele of list(base_set.difference(ref))[i] (normcounts.py:367): the
 * order python gives that set decides PoN/common precedence and ties, so the host supplies it.
 * log[14] in the order of chrom2norm_log (normcounts.py:404-419). */
int himut_set_reference(himut_ctx* ctx, const uint8_t* seq, int64_t len, const uint8_t cls[256], int n_classes);
int himut_run_normcounts(himut_ctx* ctx, const uint8_t alt_order[12], int non_human_sample);
int himut_get_normcounts(himut_ctx* ctx, int64_t* ccs_tri, int64_t* ref_tri, int64_t log[14]);
/* Test hook, no counterpart in the reference: which sweep himut_run_normcounts takes (0: k_norm_quad with its two lists,
 * the default; 1: k_norm_tile for the whole contig), the capacity of one part of the list of positions left to k_norm_dirty
 * (0: sized from the contig) and how many of a wave's pool slots may be handed out (0: all) -- the last two make the
 * fall-back paths run on small inputs.  Results never depend on any of them. */
int himut_debug_normcounts(himut_ctx* ctx, int sweep, int64_t dirty_list_cap, int pool_slots);
/* reflib.get_chrom_tricount (reflib.py:11-33) of the string given to himut_set_reference: out[first * 16 + centre * 4 +
 * last], letters A0 C1 G2 T3, purine centres already turned to the other strand (so 32 of the 64 bins fill). */
int himut_ref_tricounts(himut_ctx* ctx, int64_t out[64]);

/* ---- next row: mutlib.load_sbs96_counts / get_sbs96 (mutlib.py:1998-2018, 2058-2102) of the contig whose
 * string was given to himut_set_reference.  pos0 / ref / alt: the PASS bi-allelic single-base substitutions of that
 * contig as the VCF holds them (0-based position, ASCII).  out[sub * 16 + up * 4 + down] for the 96 classes (sub in the
 * order C>A C>G C>T T>A T>C T>G, A0 C1 G2 T3, purine references read on the other strand); out[96] = classes that
 * contain an N (the reference drops them), out[97] = classes outside the 96 without an N (KeyError in the reference),
 * out[98] = position + 1 behind the string (IndexError in the reference). */
int himut_sbs96_counts(himut_ctx* ctx, const int32_t* pos0, const uint8_t* ref, const uint8_t* alt, int64_t n, int64_t out[99]);

/* ---- next row: phaselib.get_edges (phaselib.py:16-67) --------------------------------
 * hpos / href: the contig's heterozygous SNPs (1-based position ascending, ASCII reference base), as
 * vcflib.load_hetsnps lists them.  For every primary read with mapq >= min_mapq and every ordered pair (i, j) of
 * the hetSNPs it spans whose base qualities are >= min_bq:  counts[(i * band + (j - i - 1)) * 4 + k] += 1 with
 * k = 0 cis1 (ref, ref), 1 cis2 (other, other), 2 trans1 (ref, other), 3 trans2 (other, ref).  band must be at least the
 * largest number of hetSNPs one read spans minus one (HIMUT_ERR_ARG otherwise). */
int himut_run_edges(himut_ctx* ctx, const int32_t* hpos, const uint8_t* href, int64_t n_het, int min_bq, int min_mapq,
                    int64_t band, uint32_t* counts);

/* Dense pile of [p0, p1) over ALL pushed reads (no chunk restriction):
 * counts[(p - p0) * 6 + a], bqsum[(p - p0) * 4 + b]  (caller.py:44-72). */
int himut_pile_counts(himut_ctx* ctx, int32_t p0, int32_t p1, uint32_t* counts, uint32_t* bqsum);

#ifdef __cplusplus
}
#endif
#endif
