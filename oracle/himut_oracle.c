/*
 * himut_oracle.c -- CPU restatement of the reference's per-chromosome SBS
 * caller.  TEST INFRASTRUCTURE ONLY: imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker.
 * The product path (himut_amd/, libhimut_hip.so) never links or calls it.
 *
 * It restates, function by function, what /root/reference does on the path
 *   caller.get_somatic_substitutions           src/himut/caller.py:208-642
 * in the same order and with the same data structures in spirit (a pile that is
 * rebuilt per chunk, per-position per-allele BQ lists kept in fetch order),
 * NOT in the tiled/LDS form the HIP kernels use, so that the two are
 * independent statements of the same algorithm.
 *
 * Pinned against the reference itself: the JSON fixtures under tests/golden/ were produced by
 * running the reference's own code under tests/golden/ref_harness.py; the
 * pysam.fetch and natsort boundaries are stand-ins there ("parity unpinned" at
 * those two third-party boundaries).
 *
 * Floating point: every sum below is a left-to-right double sum, as
 * Python <= 3.11's builtin sum() is.  Build with -ffp-contract=off and without
 * -ffast-math (oracle/Makefile does).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_CS 1          /* cs tag the reference's regex would not tokenise */
#define ORC_ERR_BASE 2        /* KeyError: base outside ATGC (util.py:17, caller.py:57) */
#define ORC_ERR_BQ0 3         /* ValueError: log10(0) for BQ 0 (gtlib.py:64-65) */
#define ORC_ERR_CHUNK 4       /* pysam: start > stop */
#define ORC_ERR_CAPACITY 5
#define ORC_ERR_COVER 6       /* KeyError in tpos2qbase (haplib.py:51) */
#define ORC_ERR_NOMEM 7
#define ORC_ERR_ARG 8         /* not restated (phased normcounts) / IndexError on the reference string */

enum {
    ST_PASS = 0, ST_LOWBQ = 1, ST_LOWGQ = 2, ST_INDEL = 3, ST_HET = 4, ST_HETALT = 5, ST_HOMALT = 6,
    ST_COMSNP = 7, ST_PON = 8, ST_LOWDEPTH = 9, ST_HIGHDEPTH = 10, ST_UNPHASED = 11
};
enum { GS_HOMREF = 0, GS_HET = 1, GS_HETALT = 2, GS_HOMALT = 3 };

typedef struct {
    int32_t tpos;      /* 1-based */
    int32_t chunk;
    int32_t phase_set; /* chunk start for a phased PASS, else -1 */
    int32_t gq;
    uint8_t ref, alt;  /* ASCII */
    uint8_t gt0, gt1;  /* germline genotype, reference allele first when het */
    uint8_t status;
    uint8_t gt_state;
    uint8_t flags;
    uint8_t pad;
    uint32_t counts[6]; /* A T G C ins del  (util.py:14-20 order) */
    uint32_t bqsum[4];  /* A T G C */
} orc_record;

typedef struct {
    int32_t min_qv, min_mapq, qlen_lower, qlen_upper, min_gq, min_bq;
    int32_t max_mismatch_count, mismatch_window, md_threshold;
    int32_t min_ref_count, min_alt_count, min_hap_count, phase, pad;
    double min_sequence_identity, min_trim;
} orc_params;

typedef struct {
    int64_t n;
    const int32_t *tstart, *tend, *qstart, *qlen;
    const uint8_t* mapq;
    const uint16_t* flag;
    const int32_t* qid;
    const int64_t *qoff, *cs_off;
    const uint8_t *seq, *bq, *cs;
} orc_reads;

/* one cs operation, reference cslib.cs2tuple (cslib.py:13-44) */
typedef struct {
    uint8_t state; /* 1 match 2 sub 3 ins 4 del */
    uint8_t ref, alt; /* sub only, upper-case ASCII */
    int32_t ref_len, alt_len;
    int64_t text; /* offset in cs of the op's payload (for '=' long form) */
} orc_op;

typedef struct {
    orc_op* ops;
    int32_t nops;
} orc_oplist;

static const char BASES[4] = {'A', 'T', 'G', 'C'}; /* util.py:14 base_lst */

static int base2idx(int ch) {
    switch (ch) {
        case 'A': return 0;
        case 'T': return 1;
        case 'G': return 2;
        case 'C': return 3;
    }
    return -1;
}

static const char NIB2CHAR[17] = "=ACMGRSVTWYHKDBN";

static int qbase(const orc_reads* R, int64_t r, int64_t q) {
    int64_t o = R->qoff[r] + q;
    uint8_t b = R->seq[o >> 1];
    return NIB2CHAR[(o & 1) ? (b & 15) : (b >> 4)];
}

static int isalpha_(int c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
static int upper_(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

/* cslib.cs2lst + cs2tuple (cslib.py:7-44). */
static int parse_cs(const orc_reads* R, int64_t r, orc_oplist* out) {
    const uint8_t* s = R->cs + R->cs_off[r];
    int64_t n = R->cs_off[r + 1] - R->cs_off[r];
    int32_t cap = (int32_t)(n / 2 + 2);
    orc_op* ops = (orc_op*)malloc(sizeof(orc_op) * (size_t)cap);
    if (!ops) return ORC_ERR_NOMEM;
    int32_t k = 0;
    int64_t i = 0;
    while (i < n) {
        int c = s[i];
        orc_op op;
        memset(&op, 0, sizeof(op));
        if (c == ':') {
            int64_t j = i + 1;
            int64_t v = 0;
            while (j < n && s[j] >= '0' && s[j] <= '9') { v = v * 10 + (s[j] - '0'); j++; }
            if (j == i + 1) { free(ops); return ORC_ERR_CS; }
            op.state = 1; op.ref_len = (int32_t)v; op.alt_len = (int32_t)v; op.text = -1;
            i = j;
        } else if (c == '*') {
            /* regex: \*[a-z][a-z] (lower case only) */
            if (i + 2 >= n || !(s[i + 1] >= 'a' && s[i + 1] <= 'z') || !(s[i + 2] >= 'a' && s[i + 2] <= 'z')) {
                free(ops); return ORC_ERR_CS;
            }
            op.state = 2; op.ref = (uint8_t)upper_(s[i + 1]); op.alt = (uint8_t)upper_(s[i + 2]);
            op.ref_len = 1; op.alt_len = 1;
            i += 3;
        } else if (c == '=' || c == '+' || c == '-') {
            int64_t j = i + 1;
            while (j < n && isalpha_(s[j])) j++;
            if (j == i + 1) { free(ops); return ORC_ERR_CS; }
            int32_t mlen = (int32_t)(j - i - 1);
            op.text = i + 1;
            if (c == '=') { op.state = 1; op.ref_len = mlen; op.alt_len = mlen; }
            else if (c == '+') { op.state = 3; op.ref_len = 0; op.alt_len = mlen; }
            else { op.state = 4; op.ref_len = mlen; op.alt_len = 0; }
            i = j;
        } else {
            free(ops);
            return ORC_ERR_CS;
        }
        ops[k++] = op;
    }
    out->ops = ops;
    out->nops = k;
    return ORC_OK;
}

/* base of a match op at offset i: short form slices qseq (cslib.py:28), long
 * form takes the cs text upper-cased (cslib.py:9,24) */
static int match_base(const orc_reads* R, int64_t r, const orc_op* op, int64_t qpos, int32_t i) {
    if (op->text >= 0) return upper_(R->cs[R->cs_off[r] + op->text + i]);
    return qbase(R, r, qpos + i);
}

/* ---------------- genotype likelihoods: gtlib.py:72-135 ---------------- */

typedef struct {
    const double* lut_hom;  /* log10(1 - eps(bq))           gtlib.py:64-65 */
    const double* lut_het;  /* log10(0.5 - eps(bq)/2)       gtlib.py:68-69 */
    const double* lut_err;  /* log10(eps(bq/3))             gtlib.py:60-61,93 */
    const double* log_prior; /* [homref, het, hetalt, homalt] log10 priors, gtlib.py:12-20,41-44 */
} orc_lut;

static const char GT_LST[10][3] = {"AA", "TA", "CA", "GA", "TT", "CT", "GT", "CC", "GC", "GG"}; /* gtlib.py:9 */

static int gt_state_of(int b1, int b2, int ref) { /* gtlib.py:23-38 */
    if (b1 == b2 && b2 == ref) return GS_HOMREF;
    if ((b1 == ref && b2 != ref) || (b1 != ref && b2 == ref)) return GS_HET;
    if (b1 != ref && b2 != ref && b1 != b2) return GS_HETALT;
    return GS_HOMALT;
}

typedef struct {
    uint8_t allele; /* 0..3 */
    uint8_t bq;
    int32_t read;
} pile_entry;

/* PLs of the ten genotypes for one column; entries are in fetch order. */
static int column_pls(const pile_entry* col, int32_t depth, int ref, const orc_lut* L, double pl[10], int st[10]) {
    for (int g = 0; g < 10; g++) {
        int b1 = GT_LST[g][0], b2 = GT_LST[g][1];
        int state = gt_state_of(b1, b2, ref);
        double gt_pD = 0.0;
        for (int bi = 0; bi < 4; bi++) {
            int base = BASES[bi];
            const double* lut;
            if (b1 == b2 && (base == b1 || base == b2)) lut = L->lut_hom;
            else if (b1 != b2 && (base == b1 || base == b2)) lut = L->lut_het;
            else lut = L->lut_err;
            double s = 0.0; /* sum([...]) starts from int 0 */
            for (int32_t k = 0; k < depth; k++) {
                if (col[k].allele != bi) continue;
                if (col[k].bq == 0) return ORC_ERR_BQ0;
                s = s + lut[col[k].bq];
            }
            gt_pD = gt_pD + s;
        }
        gt_pD = gt_pD + L->log_prior[state];
        pl[g] = -10.0 * gt_pD;
        st[g] = state;
    }
    return ORC_OK;
}

/* np.argsort on ten doubles with the scalar insertion sort of the numpy the
 * reference pins: stable, ties resolve to the lower index. */
static void argsort10(const double* v, int idx[10]) {
    for (int i = 0; i < 10; i++) idx[i] = i;
    for (int i = 1; i < 10; i++) {
        int x = idx[i];
        int j = i - 1;
        while (j >= 0 && v[idx[j]] > v[x]) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = x;
    }
}

/* ---------------- helpers for candidate filters ---------------- */

static int64_t bisect_left32(const int32_t* a, int64_t n, int32_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (a[m] < x) lo = m + 1; else hi = m; }
    return lo;
}
static int64_t bisect_right32(const int32_t* a, int64_t n, int32_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (x < a[m]) hi = m; else lo = m + 1; }
    return lo;
}
static int key_in(const uint64_t* a, int64_t n, uint64_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (a[m] < x) lo = m + 1; else hi = m; }
    return lo < n && a[lo] == x;
}

/* ---------------- read haplotype: haplib.py:46-83 ---------------- */

typedef struct {
    const int64_t* off;  /* [nchunks+1] into the arrays below; per chunk = per phase set */
    const int32_t* hpos; /* 1-based */
    const uint8_t* href; /* ASCII, 0 if the VCF ref is not a single base */
    const uint8_t* halt;
    const uint8_t* hbit; /* '0' / '1' */
} orc_phase;

/* cs2tpos2qbase look-up (cslib.py:153-170): base of read r at 1-based pos. */
static int base_at(const orc_reads* R, const orc_oplist* OL, int64_t r, int32_t pos1) {
    int64_t tpos = R->tstart[r];
    int64_t qpos = R->qstart[r];
    const orc_oplist* ol = &OL[r];
    int found = 0;
    for (int32_t k = 0; k < ol->nops; k++) {
        const orc_op* op = &ol->ops[k];
        if (op->state == 1) {
            if (pos1 >= tpos + 1 && pos1 <= tpos + op->ref_len) found = match_base(R, r, op, qpos, (int32_t)(pos1 - tpos - 1));
        } else if (op->state == 2) {
            if (pos1 == tpos + 1) found = op->alt;
        } else if (op->state == 4) {
            if (pos1 >= tpos + 1 && pos1 <= tpos + op->ref_len) found = '-';
        }
        tpos += op->ref_len;
        qpos += op->alt_len;
    }
    return found; /* 0: KeyError */
}

/* returns '0', '1', '.', or -1 on KeyError */
static int ccs_hap(const orc_reads* R, const orc_oplist* OL, int64_t r, const orc_phase* P, int64_t c) {
    const int32_t* hpos = P->hpos + P->off[c];
    int64_t nh = P->off[c + 1] - P->off[c];
    int64_t idx = bisect_right32(hpos, nh, R->tstart[r]);
    int64_t jdx = bisect_right32(hpos, nh, R->tend[r]);
    if (jdx - idx < 2) return '.';
    int all0 = 1, all1 = 1;
    for (int64_t k = idx; k < jdx; k++) {
        int64_t g = P->off[c] + k;
        int qb = base_at(R, OL, r, hpos[k]);
        if (qb == 0) return -1;
        int bit;
        if (P->href[g] && qb == P->href[g]) bit = '0';
        else if (P->halt[g] && qb == P->halt[g]) bit = '1';
        else bit = '-';
        int h0 = P->hbit[g];
        int h1 = (h0 == '0') ? '1' : (h0 == '1') ? '0' : '-';
        if (bit != h0) all0 = 0;
        if (bit != h1) all1 = 0;
    }
    if (all0) return '0';
    if (all1) return '1';
    return '.';
}

/* ---------------- the worker ---------------- */

typedef struct { int32_t tpos; uint8_t ref, alt; } cand_t;

static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->tpos != y->tpos) return x->tpos < y->tpos ? -1 : 1;
    if (x->ref != y->ref) return x->ref < y->ref ? -1 : 1;
    if (x->alt != y->alt) return x->alt < y->alt ? -1 : 1;
    return 0;
}

static void alt_string(const orc_record* r, char out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (r->status == ST_HETALT) { out[0] = (char)r->gt0; out[1] = ','; out[2] = (char)r->gt1; }
    else out[0] = (char)r->alt;
}

static int rec_cmp(const void* a, const void* b) {
    const orc_record* x = (const orc_record*)a; const orc_record* y = (const orc_record*)b;
    if (x->tpos != y->tpos) return x->tpos < y->tpos ? -1 : 1;
    if (x->ref != y->ref) return x->ref < y->ref ? -1 : 1;
    char ax[4], ay[4];
    alt_string(x, ax); alt_string(y, ay);
    int c = strcmp(ax, ay);
    if (c) return c;
    if (x->status != y->status) return x->status < y->status ? -1 : 1;
    if (x->gq != y->gq) return x->gq < y->gq ? -1 : 1;
    if (x->chunk != y->chunk) return x->chunk < y->chunk ? -1 : 1;
    return 0;
}

/* equality of the python tuple the reference builds (caller.py:351-620) */
static int rec_same_tuple(const orc_record* x, const orc_record* y) {
    char ax[4], ay[4];
    alt_string(x, ax); alt_string(y, ay);
    if (x->tpos != y->tpos || x->ref != y->ref || strcmp(ax, ay) || x->status != y->status || x->gq != y->gq ||
        x->phase_set != y->phase_set)
        return 0;
    if (memcmp(x->counts, y->counts, sizeof(x->counts))) return 0;
    /* alt_bq is derived from the sums of the printed alleles */
    if (x->status == ST_HETALT) {
        int p = base2idx(x->gt0), q = base2idx(x->gt1);
        return x->bqsum[p] == y->bqsum[p] && x->bqsum[q] == y->bqsum[q];
    }
    int a = base2idx(x->alt);
    return x->bqsum[a] == y->bqsum[a];
}

int orc_call(const orc_reads* R, const orc_params* P, const orc_lut* L, int64_t nchunks, const int32_t* cstart,
             const int32_t* cend, const uint64_t* pon, int64_t npon, const uint64_t* com, int64_t ncom,
             const orc_phase* PH, orc_record* out, int64_t cap, int64_t* nout, int64_t log[15], int64_t* n_candidates) {
    int rc = ORC_OK;
    const int64_t n = R->n;
    orc_oplist* OL = (orc_oplist*)calloc((size_t)(n > 0 ? n : 1), sizeof(orc_oplist));
    int32_t maxpos = 0;
    int32_t maxqid = 0;
    int64_t m_ccs = 0, m_sbs = 0, m_het = 0, m_hetalt = 0, m_homalt = 0, m_somrev = 0, m_homref = 0, m_uncall = 0,
            m_lowgq = 0, m_lowbq = 0, m_pon = 0, m_pop = 0, m_md = 0, m_ab = 0, m_som = 0;
    int64_t nrec = 0;
    uint8_t* som_seen = NULL;
    uint8_t* ccs_seen = NULL;
    if (!OL) return ORC_ERR_NOMEM;
    for (int64_t r = 0; r < n; r++) {
        if (R->flag[r] & 0x100) continue; /* bamlib.py:17 */
        rc = parse_cs(R, r, &OL[r]);
        if (rc) goto done;
        int64_t t = R->tstart[r];
        for (int32_t k = 0; k < OL[r].nops; k++) t += OL[r].ops[k].ref_len;
        if (t + 2 > maxpos) maxpos = (int32_t)(t + 2);
        if (R->tend[r] + 2 > maxpos) maxpos = R->tend[r] + 2;
        if (R->qid[r] > maxqid) maxqid = R->qid[r];
    }
    som_seen = (uint8_t*)calloc((size_t)maxpos + 8, 1);
    ccs_seen = (uint8_t*)calloc((size_t)maxqid + 8, 1);
    if (!som_seen || !ccs_seen) { rc = ORC_ERR_NOMEM; goto done; }

    for (int64_t c = 0; c < nchunks; c++) {
        const int32_t s = cstart[c], e = cend[c];
        if (s > e) { rc = ORC_ERR_CHUNK; goto done; }
        /* reads served by fetch(chrom, s, e): caller.py:299 */
        int64_t nfetch = 0;
        int64_t* fetch = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
        int32_t pmin = INT32_MAX, pmax = 0;
        for (int64_t r = 0; r < n; r++) {
            if (R->tstart[r] >= e) break; /* sorted */
            if (R->tend[r] <= s) continue;
            if (R->flag[r] & 0x100) continue;
            fetch[nfetch++] = r;
            if (R->tstart[r] < pmin) pmin = R->tstart[r];
            int64_t t = R->tstart[r];
            for (int32_t k = 0; k < OL[r].nops; k++) t += OL[r].ops[k].ref_len;
            if (t + 1 > pmax) pmax = (int32_t)(t + 1);
        }
        if (nfetch == 0) { free(fetch); continue; }
        const int64_t W = (int64_t)pmax - pmin + 1;
        uint32_t* counts = (uint32_t*)calloc((size_t)W * 6, sizeof(uint32_t)); /* caller.py:36 */
        int64_t* loff = (int64_t*)calloc((size_t)W + 1, sizeof(int64_t));
        cand_t* cands = NULL;
        int64_t ncand = 0, capcand = 0;
        /* pass A: list lengths per position */
        for (int64_t f = 0; f < nfetch; f++) {
            int64_t r = fetch[f];
            int64_t tpos = R->tstart[r];
            for (int32_t k = 0; k < OL[r].nops; k++) {
                const orc_op* op = &OL[r].ops[k];
                if (op->state == 1) for (int32_t i = 0; i < op->ref_len; i++) loff[tpos + i - pmin + 1]++;
                else if (op->state == 2) loff[tpos - pmin + 1]++;
                tpos += op->ref_len;
            }
        }
        for (int64_t w = 0; w < W; w++) loff[w + 1] += loff[w];
        pile_entry* ent = (pile_entry*)malloc(sizeof(pile_entry) * (size_t)(loff[W] > 0 ? loff[W] : 1));
        int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)W);
        memcpy(fill, loff, sizeof(int64_t) * (size_t)W);
        int8_t* haps = (int8_t*)malloc((size_t)nfetch);

        /* HOT LOOP 1: caller.py:299-322 */
        for (int64_t f = 0; f < nfetch && !rc; f++) {
            int64_t r = fetch[f];
            const orc_oplist* ol = &OL[r];
            /* update_allelecounts: caller.py:44-72 */
            int64_t tpos = R->tstart[r];
            int64_t qpos = R->qstart[r];
            int64_t match_count = 0, mismatch_count = 0;
            for (int32_t k = 0; k < ol->nops; k++) {
                const orc_op* op = &ol->ops[k];
                if (op->state == 1) {
                    for (int32_t i = 0; i < op->ref_len; i++) {
                        int b = base2idx(match_base(R, r, op, qpos, i));
                        if (b < 0) { rc = ORC_ERR_BASE; break; }
                        int64_t w = tpos + i - pmin;
                        counts[w * 6 + b]++;
                        pile_entry* pe = &ent[fill[w]++];
                        pe->allele = (uint8_t)b; pe->bq = R->bq[R->qoff[r] + qpos + i]; pe->read = (int32_t)r;
                    }
                    match_count += op->ref_len;
                } else if (op->state == 2) {
                    int b = base2idx(op->alt);
                    if (b < 0) { rc = ORC_ERR_BASE; break; }
                    int64_t w = tpos - pmin;
                    counts[w * 6 + b]++;
                    pile_entry* pe = &ent[fill[w]++];
                    pe->allele = (uint8_t)b; pe->bq = R->bq[R->qoff[r] + qpos]; pe->read = (int32_t)r;
                    mismatch_count += op->alt_len;
                } else if (op->state == 3) {
                    counts[(tpos - pmin) * 6 + 4]++;
                    mismatch_count += op->alt_len;
                } else {
                    for (int32_t j = 0; j < op->ref_len; j++) counts[(tpos + j - pmin) * 6 + 5]++;
                    mismatch_count += op->ref_len;
                }
                tpos += op->ref_len;
                qpos += op->alt_len;
            }
            if (rc) break;
            haps[f] = '.';
            if (P->phase) { /* caller.py:306-309 */
                int h = ccs_hap(R, OL, r, PH, c);
                if (h < 0) { rc = ORC_ERR_COVER; break; }
                haps[f] = (int8_t)h;
                if (!(h == '0' || h == '1')) continue;
            }
            /* read filters: caller.py:310-317 */
            int64_t bqs = 0;
            for (int32_t q = 0; q < R->qlen[r]; q++) bqs += R->bq[R->qoff[r] + q];
            double qv = (double)bqs / (double)R->qlen[r]; /* np.mean, bamlib.py:35 */
            if (qv < (double)P->min_qv) continue;
            if (R->mapq[r] < P->min_mapq) continue;
            double ident = (double)match_count / (double)(match_count + mismatch_count); /* bamlib.py:47-63 */
            if (ident < P->min_sequence_identity) continue;
            if (!(P->qlen_lower < R->qlen[r] && R->qlen[r] < P->qlen_upper)) continue;
            if (!ccs_seen[R->qid[r]]) { m_ccs++; ccs_seen[R->qid[r]] = 1; } /* caller.py:318-320 */
            /* get_tsbs_candidates: bamlib.py:69-86 ; cs2subindel: cslib.py:47-64 */
            int32_t nmis = 0;
            int32_t* mis = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ol->nops + 1));
            tpos = R->tstart[r];
            for (int32_t k = 0; k < ol->nops; k++) {
                const orc_op* op = &ol->ops[k];
                if ((op->state == 2 && op->ref != 'N') || op->state == 3 || op->state == 4) mis[nmis++] = (int32_t)(tpos + 1);
                tpos += op->ref_len;
            }
            const int32_t qlen = R->qlen[r];
            double trimmed_qstart = floor(P->min_trim * (double)qlen);      /* bamlib.py:226 */
            double trimmed_qend = ceil((1.0 - P->min_trim) * (double)qlen); /* bamlib.py:227 */
            tpos = R->tstart[r];
            qpos = R->qstart[r];
            for (int32_t k = 0; k < ol->nops; k++) {
                const orc_op* op = &ol->ops[k];
                if (op->state == 2 && op->ref != 'N') {
                    int32_t tp1 = (int32_t)(tpos + 1);
                    int ok = 1;
                    if (tp1 < maxpos && som_seen[tp1]) ok = 0;                          /* bamlib.py:77 */
                    if (ok && ((double)qpos < trimmed_qstart || (double)qpos > trimmed_qend)) ok = 0; /* :231-242 */
                    if (ok) { /* get_mismatch_range: bamlib.py:245-258 */
                        int64_t w = P->mismatch_window;
                        int64_t qs = qpos - w, qe = qpos + w, ur, dr;
                        if (qs < 0) { ur = w + qs; dr = w + (-qs); }
                        else if (qe > qlen) { ur = w + llabs(qe - qlen); dr = qlen - qpos; }
                        else { ur = w; dr = w; }
                        int64_t ms = tp1 - ur, me = tp1 + dr;
                        int64_t cnt = bisect_right32(mis, nmis, (int32_t)me) - bisect_left32(mis, nmis, (int32_t)ms) - 1;
                        if (cnt > P->max_mismatch_count) ok = 0;
                    }
                    if (ok) {
                        if (ncand == capcand) {
                            capcand = capcand ? capcand * 2 : 1024;
                            cands = (cand_t*)realloc(cands, sizeof(cand_t) * (size_t)capcand);
                        }
                        cands[ncand].tpos = tp1; cands[ncand].ref = op->ref; cands[ncand].alt = op->alt;
                        ncand++;
                    }
                }
                tpos += op->ref_len;
                qpos += op->alt_len;
            }
            free(mis);
        }

        /* set(candidates): caller.py:324 */
        if (!rc && ncand) {
            qsort(cands, (size_t)ncand, sizeof(cand_t), cand_cmp);
            int64_t u = 0;
            for (int64_t k = 0; k < ncand; k++)
                if (k == 0 || cand_cmp(&cands[k], &cands[u - 1]) != 0) cands[u++] = cands[k];
            ncand = u;
        }
        /* HOT LOOP 2: caller.py:324-621 */
        for (int64_t k = 0; k < ncand && !rc; k++) {
            const int32_t tpos1 = cands[k].tpos;
            const int ref = cands[k].ref, alt = cands[k].alt;
            if (!(s <= tpos1 && tpos1 <= e)) continue; /* is_chunk, caller.py:104-108 */
            m_sbs++;
            const int ri = base2idx(ref), ai = base2idx(alt);
            if (ri < 0 || ai < 0) { rc = ORC_ERR_BASE; break; }
            const int64_t rpos = (int64_t)tpos1 - 1;
            const int64_t w = rpos - pmin;
            uint32_t cz[6] = {0, 0, 0, 0, 0, 0};
            const pile_entry* col = NULL;
            int32_t depth_l = 0;
            if (w >= 0 && w < W) {
                memcpy(cz, &counts[w * 6], sizeof(cz));
                col = &ent[loff[w]];
                depth_l = (int32_t)(loff[w + 1] - loff[w]);
            }
            const uint32_t ins_count = cz[4], del_count = cz[5];
            const uint32_t read_depth = cz[0] + cz[1] + cz[2] + cz[3] + cz[5]; /* bamlib.py:213-219 */
            const uint32_t ref_count = cz[ri], alt_count = cz[ai];
            uint32_t bqsum[4] = {0, 0, 0, 0};
            int alt_hi = 0;
            for (int32_t q = 0; q < depth_l; q++) {
                bqsum[col[q].allele] += col[q].bq;
                if (col[q].allele == ai && col[q].bq >= P->min_bq) alt_hi++; /* caller.py:160-171 */
            }
            double pl[10];
            int st[10], order[10];
            rc = column_pls(col, depth_l, ref, L, pl, st);
            if (rc) break;
            argsort10(pl, order); /* gtlib.py:113-119 */
            double gqf = pl[order[1]] - pl[order[0]];
            int gq = gqf < 99.0 ? (int)gqf : 99;
            int g0 = GT_LST[order[0]][0], g1 = GT_LST[order[0]][1];
            int state = st[order[0]];
            if (g0 != ref && ((g0 == ref) + (g1 == ref)) == 1) { int t = g0; g0 = g1; g1 = t; } /* gtlib.py:133-134 */
            /* is_germ_gt: caller.py:111-147 */
            int germ = 0;
            if (state == GS_HET) germ = (g0 == ref && g1 == alt);
            else if (state == GS_HETALT) {
                uint32_t base_sum = cz[0] + cz[1] + cz[2] + cz[3];
                uint32_t base_counts = cz[base2idx(g0)] + cz[base2idx(g1)];
                germ = (base_sum == base_counts) && (alt == g0 || alt == g1);
            } else if (state == GS_HOMALT) germ = (ref_count == 0) && (g0 == alt && g1 == alt);
            else germ = (alt == g0);
            if (germ) {
                if (state == GS_HET) m_het++;
                else if (state == GS_HETALT) m_hetalt++;
                else if (state == GS_HOMALT) m_homalt++;
                continue;
            }
            if (tpos1 < maxpos) som_seen[tpos1] = 1; /* caller.py:347 */
            /* germ_gq: get_germ_gq is handed the 2-char genotype, so no base is
             * skipped and it equals gq (caller.py:348 vs gtlib.py:151) */
            int status;
            int32_t ps = -1;
            if (state == GS_HET) { m_somrev++; status = ST_HET; }
            else if (state == GS_HETALT) { m_somrev++; status = ST_HETALT; }
            else if (state == GS_HOMALT) { m_somrev++; status = ST_HOMALT; }
            else if (del_count != 0 || ins_count != 0) { m_uncall++; status = ST_INDEL; }
            else {
                m_homref++;
                uint64_t key = ((uint64_t)(uint32_t)tpos1 << 4) | ((uint64_t)ri << 2) | (uint64_t)ai;
                if (gq < P->min_gq) { m_lowgq++; status = ST_LOWGQ; }
                else if (alt_hi == 0) { m_lowbq++; status = ST_LOWBQ; }
                else if (key_in(pon, npon, key)) { m_pon++; status = ST_PON; }
                else if (key_in(com, ncom, key)) { m_pop++; status = ST_COMSNP; }
                else if (!((int64_t)ref_count >= P->min_ref_count && (int64_t)alt_count >= P->min_alt_count)) { m_ab++; status = ST_LOWDEPTH; }
                else if ((int64_t)read_depth > P->md_threshold) { m_md++; status = ST_HIGHDEPTH; }
                else if (P->phase) { /* caller.py:552-603 */
                    m_som++;
                    int64_t h0 = 0, h1 = 0;
                    int som0 = 0, som1 = 0;
                    for (int64_t j = 0; j < n; j++) { /* fetch(chrom, tpos, tpos + 1) */
                        if (R->tstart[j] >= tpos1 + 1) break;
                        if (R->tend[j] <= tpos1) continue;
                        if (R->flag[j] & 0x100) continue;
                        int in_wt = 0, in_alt = 0;
                        for (int32_t q = 0; q < depth_l; q++) {
                            if (R->qid[col[q].read] != R->qid[j]) continue;
                            if (col[q].allele == ri) in_wt = 1;
                            if (col[q].allele == ai) in_alt = 1;
                        }
                        if (in_wt) {
                            int h = ccs_hap(R, OL, j, PH, c);
                            if (h < 0) { rc = ORC_ERR_COVER; break; }
                            if (h == '0') h0++; else if (h == '1') h1++;
                        } else if (in_alt) {
                            int h = ccs_hap(R, OL, j, PH, c);
                            if (h < 0) { rc = ORC_ERR_COVER; break; }
                            if (h == '0') som0 = 1; else if (h == '1') som1 = 1;
                        }
                    }
                    if (rc) break;
                    if (h0 >= P->min_hap_count && h1 >= P->min_hap_count && (som0 + som1) == 1) { status = ST_PASS; ps = s; }
                    else status = ST_UNPHASED;
                } else { m_som++; status = ST_PASS; }
            }
            if (nrec >= cap) { rc = ORC_ERR_CAPACITY; break; }
            orc_record* o = &out[nrec++];
            memset(o, 0, sizeof(*o));
            o->tpos = tpos1; o->chunk = (int32_t)c; o->phase_set = ps; o->gq = gq;
            o->ref = (uint8_t)ref; o->alt = (uint8_t)alt; o->gt0 = (uint8_t)g0; o->gt1 = (uint8_t)g1;
            o->status = (uint8_t)status; o->gt_state = (uint8_t)state;
            memcpy(o->counts, cz, sizeof(cz));
            memcpy(o->bqsum, bqsum, sizeof(bqsum));
        }
        free(haps); free(fill); free(ent); free(loff); free(counts); free(cands); free(fetch);
        if (rc) goto done;
    }
    /* natsorted(list(set(...))): caller.py:622-624 */
    qsort(out, (size_t)nrec, sizeof(orc_record), rec_cmp);
    {
        int64_t u = 0;
        for (int64_t k = 0; k < nrec; k++) {
            if (u > 0 && rec_same_tuple(&out[k], &out[u - 1])) continue;
            out[u++] = out[k];
        }
        nrec = u;
    }
done:
    *nout = nrec;
    log[0] = m_ccs; log[1] = m_sbs; log[2] = m_het; log[3] = m_hetalt; log[4] = m_homalt; log[5] = m_somrev;
    log[6] = m_homref; log[7] = m_uncall; log[8] = m_lowgq; log[9] = m_lowbq; log[10] = m_pon; log[11] = m_pop;
    log[12] = m_md; log[13] = m_ab; log[14] = m_som;
    if (n_candidates) *n_candidates = m_sbs;
    for (int64_t r = 0; r < n; r++) free(OL[r].ops);
    free(OL); free(som_seen); free(ccs_seen);
    return rc;
}

/* ---------------- normcounts.get_callable_tricounts ---------------- */

/* PLs of the ten genotypes with one allele left out: gtlib.get_germ_gq (gtlib.py:138-174) */
static int column_pls_without(const pile_entry* col, int32_t depth, int skip_bi, const int st[10], const orc_lut* L,
                              double pl[10]) {
    for (int g = 0; g < 10; g++) {
        int b1 = GT_LST[g][0], b2 = GT_LST[g][1];
        double gt_pl = 0.0;
        for (int bi = 0; bi < 4; bi++) {
            if (bi == skip_bi) continue; /* gtlib.py:151-152 */
            int base = BASES[bi];
            const double* lut;
            if (b1 == b2 && (base == b1 || base == b2)) lut = L->lut_hom;
            else if (b1 != b2 && (base == b1 || base == b2)) lut = L->lut_het;
            else lut = L->lut_err;
            double s = 0.0;
            for (int32_t k = 0; k < depth; k++) {
                if (col[k].allele != bi) continue;
                if (col[k].bq == 0) return ORC_ERR_BQ0;
                s = s + lut[col[k].bq];
            }
            gt_pl = gt_pl + s;
        }
        gt_pl = gt_pl + L->log_prior[st[g]];
        pl[g] = gt_pl * -10.0; /* gtlib.py:164 */
    }
    return ORC_OK;
}

/* get_tri_context (normcounts.py:49-63) as class ids; cls maps a reference byte to its id, n_cls = 'N' */
static int64_t tri_index(const uint8_t* seq, int64_t len, int64_t pos, const uint8_t* cls, int K, int n_cls) {
    uint8_t t[3];
    if (pos - 1 < 0 || pos + 2 > len) { t[0] = t[1] = t[2] = 'N'; } /* seq[pos-1:pos+2] shorter than 3 (or empty for pos 0) */
    else {
        t[0] = seq[pos - 1]; t[1] = seq[pos]; t[2] = seq[pos + 1];
        if (t[1] == 'A' || t[1] == 'G') { /* purine: reverse complement, unknown letters -> N (mutlib.py:15) */
            uint8_t r[3];
            for (int i = 0; i < 3; i++) {
                uint8_t b = t[2 - i];
                r[i] = b == 'A' ? 'T' : b == 'T' ? 'A' : b == 'G' ? 'C' : b == 'C' ? 'G' : 'N';
            }
            t[0] = r[0]; t[1] = r[1]; t[2] = r[2];
        }
    }
    (void)n_cls;
    return ((int64_t)cls[t[0]] * K + cls[t[1]]) * K + cls[t[2]];
}

/* get_callable_tricounts (normcounts.py:206-421), with or without --phase.
 * refseq: the contig as the FASTA holds it (case preserved); cls[256]: byte -> class id (< K); alt_order[ri][0..2]:
 * list(base_set.difference(ref)) as alleles 0..3 for ref allele ri (python set order, supplied by the caller).
 * Outputs: ccs_tri/ref_tri [K*K*K] (added to), log[14]. */
int orc_normcounts(const orc_reads* R, const orc_params* P, const orc_lut* L, int64_t nchunks, const int32_t* cstart,
                   const int32_t* cend, const uint64_t* pon, int64_t npon, const uint64_t* com, int64_t ncom,
                   const uint8_t* refseq, int64_t reflen, const uint8_t* cls, int K, const uint8_t* alt_order,
                   int non_human_sample, const orc_phase* PH, int64_t* ccs_tri, int64_t* ref_tri, int64_t log[14]) {
    int rc = ORC_OK;
    const int64_t n = R->n;
    orc_oplist* OL = (orc_oplist*)calloc((size_t)(n > 0 ? n : 1), sizeof(orc_oplist));
    int32_t maxqid = 0;
    uint8_t* seen = NULL;
    int64_t m[14];
    memset(m, 0, sizeof(m));
    if (!OL) return ORC_ERR_NOMEM;
    if (P->phase && !PH) { free(OL); return ORC_ERR_ARG; }
    for (int64_t r = 0; r < n; r++) {
        if (R->flag[r] & 0x100) continue; /* bamlib.py:17, normcounts.py:291 */
        rc = parse_cs(R, r, &OL[r]);
        if (rc) goto done;
        if (R->qid[r] > maxqid) maxqid = R->qid[r];
    }
    seen = (uint8_t*)calloc((size_t)maxqid + 8, 1);
    if (!seen) { rc = ORC_ERR_NOMEM; goto done; }

    for (int64_t c = 0; c < nchunks && !rc; c++) {
        const int32_t s = cstart[c], e = cend[c];
        if (s > e) { rc = ORC_ERR_CHUNK; goto done; }
        /* fetch(chrom, s, e): normcounts.py:289 */
        int64_t nfetch = 0;
        int64_t* fetch = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
        int32_t pmin = INT32_MAX, pmax = 0;
        for (int64_t r = 0; r < n; r++) {
            if (R->tstart[r] >= e) break;
            if (R->tend[r] <= s) continue;
            if (R->flag[r] & 0x100) continue;
            fetch[nfetch++] = r;
            if (R->tstart[r] < pmin) pmin = R->tstart[r];
            int64_t t = R->tstart[r];
            for (int32_t k = 0; k < OL[r].nops; k++) t += OL[r].ops[k].ref_len;
            if (t + 1 > pmax) pmax = (int32_t)(t + 1);
        }
        if (nfetch == 0) { free(fetch); continue; }
        const int64_t W = (int64_t)pmax - pmin + 1;
        uint32_t* counts = (uint32_t*)calloc((size_t)W * 6, sizeof(uint32_t));
        int64_t* loff = (int64_t*)calloc((size_t)W + 1, sizeof(int64_t));
        uint32_t* callable = (uint32_t*)calloc((size_t)W, sizeof(uint32_t)); /* rpos2count, normcounts.py:286 */
        uint32_t* hapcnt = (uint32_t*)calloc((size_t)W * 2, sizeof(uint32_t)); /* rpos2hap2count for "0" and "1", :288 */
        for (int64_t f = 0; f < nfetch; f++) {
            int64_t r = fetch[f];
            int64_t tpos = R->tstart[r];
            for (int32_t k = 0; k < OL[r].nops; k++) {
                const orc_op* op = &OL[r].ops[k];
                if (op->state == 1) for (int32_t i = 0; i < op->ref_len; i++) loff[tpos + i - pmin + 1]++;
                else if (op->state == 2) loff[tpos - pmin + 1]++;
                tpos += op->ref_len;
            }
        }
        for (int64_t w = 0; w < W; w++) loff[w + 1] += loff[w];
        pile_entry* ent = (pile_entry*)malloc(sizeof(pile_entry) * (size_t)(loff[W] > 0 ? loff[W] : 1));
        int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)W);
        memcpy(fill, loff, sizeof(int64_t) * (size_t)W);

        for (int64_t f = 0; f < nfetch && !rc; f++) {
            int64_t r = fetch[f];
            const orc_oplist* ol = &OL[r];
            int hap = '.';
            if (P->phase) { /* normcounts.py:293-295 */
                hap = ccs_hap(R, OL, r, PH, c);
                if (hap < 0) { rc = ORC_ERR_COVER; break; }
            }
            /* update_allelecounts / update_phased_allelecounts: normcounts.py:113-172 */
            int64_t tpos = R->tstart[r];
            int64_t qpos = R->qstart[r];
            int64_t match_count = 0, mismatch_count = 0;
            for (int32_t k = 0; k < ol->nops; k++) {
                const orc_op* op = &ol->ops[k];
                if (op->state == 1) {
                    for (int32_t i = 0; i < op->ref_len; i++) {
                        int b = base2idx(match_base(R, r, op, qpos, i));
                        if (b < 0) { rc = ORC_ERR_BASE; break; }
                        int64_t w = tpos + i - pmin;
                        counts[w * 6 + b]++;
                        pile_entry* pe = &ent[fill[w]++];
                        pe->allele = (uint8_t)b; pe->bq = R->bq[R->qoff[r] + qpos + i]; pe->read = (int32_t)r;
                        if (hap == '0') hapcnt[w * 2]++; else if (hap == '1') hapcnt[w * 2 + 1]++;
                    }
                    match_count += op->ref_len;
                } else if (op->state == 2) {
                    int b = base2idx(op->alt);
                    if (b < 0) { rc = ORC_ERR_BASE; break; }
                    int64_t w = tpos - pmin;
                    counts[w * 6 + b]++;
                    pile_entry* pe = &ent[fill[w]++];
                    pe->allele = (uint8_t)b; pe->bq = R->bq[R->qoff[r] + qpos]; pe->read = (int32_t)r;
                    if (hap == '0') hapcnt[w * 2]++; else if (hap == '1') hapcnt[w * 2 + 1]++;
                    mismatch_count += op->alt_len;
                } else if (op->state == 3) {
                    counts[(tpos - pmin) * 6 + 4]++;
                    mismatch_count += op->alt_len;
                } else {
                    for (int32_t j = 0; j < op->ref_len; j++) counts[(tpos + j - pmin) * 6 + 5]++;
                    mismatch_count += op->ref_len;
                }
                tpos += op->ref_len;
                qpos += op->alt_len;
            }
            if (rc) break;
            if (P->phase && !(hap == '0' || hap == '1')) continue; /* normcounts.py:297-298 */
            /* read filters: normcounts.py:302-309 */
            int64_t bqs = 0;
            for (int32_t q = 0; q < R->qlen[r]; q++) bqs += R->bq[R->qoff[r] + q];
            double qv = (double)bqs / (double)R->qlen[r];
            if (qv < (double)P->min_qv) continue;
            if (R->mapq[r] < P->min_mapq) continue;
            double ident = (double)match_count / (double)(match_count + mismatch_count);
            if (ident < P->min_sequence_identity) continue;
            if (!(P->qlen_lower < R->qlen[r] && R->qlen[r] < P->qlen_upper)) continue;
            if (!seen[R->qid[r]]) { m[0]++; seen[R->qid[r]] = 1; } /* normcounts.py:310-312 */
            /* update_tri2count: normcounts.py:66-110 */
            int32_t nmis = 0;
            int32_t* mis = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ol->nops + 1));
            tpos = R->tstart[r];
            for (int32_t k = 0; k < ol->nops; k++) {
                const orc_op* op = &ol->ops[k];
                if ((op->state == 2 && op->ref != 'N') || op->state == 3 || op->state == 4) mis[nmis++] = (int32_t)(tpos + 1);
                tpos += op->ref_len;
            }
            const int32_t qlen = R->qlen[r];
            const double trimmed_qstart = floor(P->min_trim * (double)qlen);
            const double trimmed_qend = ceil((1.0 - P->min_trim) * (double)qlen);
            int64_t rpos = R->tstart[r];
            qpos = R->qstart[r];
            for (int32_t k = 0; k < ol->nops; k++) {
                const orc_op* op = &ol->ops[k];
                if (op->state == 1) {
                    /* get_mismatch_range(rpos, qpos, ...): 0-based rpos against the 1-based list, evaluated once
                     * per run and shifted by j (normcounts.py:84-90) */
                    int64_t w = P->mismatch_window;
                    int64_t qs = qpos - w, qe = qpos + w, ur, dr;
                    if (qs < 0) { ur = w + qs; dr = w + (-qs); }
                    else if (qe > qlen) { ur = w + llabs(qe - qlen); dr = qlen - qpos; }
                    else { ur = w; dr = w; }
                    const int64_t ms = rpos - ur, me = rpos + dr;
                    for (int32_t j = 0; j < op->ref_len; j++) {
                        int64_t cnt = bisect_right32(mis, nmis, (int32_t)(me + j)) - bisect_left32(mis, nmis, (int32_t)(ms + j));
                        if ((int)R->bq[R->qoff[r] + qpos + j] < P->min_bq) continue;
                        if (cnt > P->max_mismatch_count) continue;
                        if ((double)(qpos + j) < trimmed_qstart || (double)(qpos + j) > trimmed_qend) continue;
                        callable[rpos + j - pmin]++;
                    }
                } else if (op->state == 2) {
                    callable[rpos - pmin]++; /* the three filters are evaluated and ignored (normcounts.py:97-109) */
                }
                rpos += op->ref_len;
                qpos += op->alt_len;
            }
            free(mis);
        }

        /* positions of the chunk: normcounts.py:315-402 */
        for (int64_t rp = s; rp < e && !rc; rp++) {
            if (rp < 0 || rp >= reflen) { rc = ORC_ERR_ARG; break; } /* IndexError in the reference */
            const int ref = refseq[rp];
            const int64_t w = rp - pmin;
            const uint32_t tri_sum = (w >= 0 && w < W) ? callable[w] : 0;
            const int ridx = base2idx(ref);
            if (ridx < 0) continue;
            if (tri_sum == 0) continue;
            m[1] += tri_sum;
            if (P->phase && !((int64_t)hapcnt[w * 2] >= P->min_hap_count && (int64_t)hapcnt[w * 2 + 1] >= P->min_hap_count)) {
                m[2] += tri_sum; /* is_rpos_phased, normcounts.py:198-205,324-328 */
                continue;
            }
            const uint32_t* cz = &counts[w * 6];
            const pile_entry* col = &ent[loff[w]];
            const int32_t depth_l = (int32_t)(loff[w + 1] - loff[w]);
            const uint32_t ins_count = cz[4], del_count = cz[5];
            const uint32_t read_depth = cz[0] + cz[1] + cz[2] + cz[3] + cz[5];
            double pl[10];
            int st[10], order[10];
            rc = column_pls(col, depth_l, ref, L, pl, st);
            if (rc) break;
            argsort10(pl, order);
            double gqf = pl[order[1]] - pl[order[0]];
            int gq = gqf < 99.0 ? (int)gqf : 99;
            const int state = st[order[0]];
            if (state == GS_HET) { m[3] += tri_sum; continue; }
            if (state == GS_HETALT) { m[4] += tri_sum; continue; }
            if (state == GS_HOMALT) { m[5] += tri_sum; continue; }
            m[6] += tri_sum;
            if (del_count != 0 || ins_count != 0) { m[7] += tri_sum; continue; }
            if ((int64_t)read_depth > P->md_threshold) { m[8] += tri_sum; continue; }
            const uint32_t ref_count = cz[ridx];
            const int64_t tri = tri_index(refseq, reflen, rp, cls, K, 0);
            if (read_depth == ref_count) {
                if (gq < P->min_gq) { m[10] += tri_sum; continue; }
                if ((int64_t)ref_count < P->min_ref_count) { m[9] += tri_sum; continue; }
            } else {
                int alt_state = 0;
                uint32_t alt_counts[3];
                int n_seen = 0;
                for (int a = 0; a < 3; a++) { /* normcounts.py:369-385 */
                    const int aidx = alt_order[ridx * 3 + a];
                    const uint32_t alt_count = cz[aidx];
                    alt_counts[n_seen++] = alt_count;
                    if (alt_count == 0) continue;
                    const uint64_t key = ((uint64_t)(uint32_t)(rp + 1) << 4) | ((uint64_t)ridx << 2) | (uint64_t)aidx;
                    if (!non_human_sample && key_in(pon, npon, key)) { alt_state = 1; m[11] += tri_sum; break; }
                    if (!non_human_sample && key_in(com, ncom, key)) { alt_state = 1; m[12] += tri_sum; break; }
                }
                if (alt_state) continue;
                int best = 0; /* list.index(max(...)): first maximum */
                for (int a = 1; a < 3; a++) if (alt_counts[a] > alt_counts[best]) best = a;
                const int aidx = alt_order[ridx * 3 + best];
                double pl2[10];
                rc = column_pls_without(col, depth_l, aidx, st, L, pl2);
                if (rc) break;
                int order2[10];
                argsort10(pl2, order2);
                double g2 = pl2[order2[1]] - pl2[order2[0]];
                int gq2 = g2 < 99.0 ? (int)g2 : 99;
                if (gq2 < P->min_gq) { m[10] += tri_sum; continue; }
                const uint32_t alt_count = cz[aidx];
                if (!((int64_t)ref_count >= P->min_ref_count && (int64_t)alt_count >= P->min_alt_count)) { m[9] += tri_sum; continue; }
            }
            ref_tri[tri] += 1;
            ccs_tri[tri] += tri_sum;
            m[13] += tri_sum;
        }
        free(fill); free(ent); free(loff); free(counts); free(callable); free(hapcnt); free(fetch);
    }
done:
    /* chrom2norm_log order (normcounts.py:404-419): ccs, bases, unphased, het, hetalt, homalt, homref, uncallable,
     * md, ab, low_gq, pon, pop, callable */
    for (int k = 0; k < 14; k++) log[k] = m[k];
    for (int64_t r = 0; r < n; r++) free(OL[r].ops);
    free(OL); free(seen);
    return rc;
}

/* ---------------- phaselib.get_edges ---------------- */

/* For every primary read with mapq >= min_mapq: the hetSNPs with tstart < pos <= tend (phaselib.py:38-44), the
 * read's base and quality at each (cs2tpos2qbase, cslib.py:153-170: a deleted position has quality 0), and for every
 * ordered pair with both qualities >= min_bq one count in cis1 / cis2 / trans1 / trans2 (phaselib.py:48-67).
 * counts[(i * band + (j - i - 1)) * 4 + k] for the pair (i, j), i < j < i + 1 + band; returns ORC_ERR_CAPACITY if a
 * read sees hetSNPs further apart in the list than the band. */
int orc_edges(const orc_reads* R, int min_bq, int min_mapq, int64_t nhet, const int32_t* hpos, const uint8_t* href,
              int64_t band, uint32_t* counts) {
    int rc = ORC_OK;
    for (int64_t r = 0; r < R->n && !rc; r++) {
        if (R->flag[r] & 0x100) continue;            /* bamlib.py:17, phaselib.py:30 */
        if (R->mapq[r] < min_mapq) continue;
        int64_t idx = bisect_right32(hpos, nhet, R->tstart[r]);
        int64_t jdx = bisect_right32(hpos, nhet, R->tend[r]);
        if (jdx - idx < 2) continue;
        orc_oplist ol;
        rc = parse_cs(R, r, &ol);
        if (rc) break;
        orc_oplist* OL = (orc_oplist*)calloc((size_t)(r + 1), sizeof(orc_oplist)); /* base_at indexes by read */
        if (!OL) { free(ol.ops); return ORC_ERR_NOMEM; }
        OL[r] = ol;
        int64_t k = jdx - idx;
        int* state = (int*)malloc(sizeof(int) * (size_t)k);
        int* ok = (int*)malloc(sizeof(int) * (size_t)k);
        for (int64_t a = 0; a < k && !rc; a++) {
            const int32_t pos = hpos[idx + a];
            int qb = base_at(R, OL, r, pos);
            if (qb == 0) { rc = ORC_ERR_COVER; break; }
            int bq = 0;
            if (qb != '-') { /* quality of the base: walk again for the query offset */
                int64_t tpos = R->tstart[r], qpos = R->qstart[r];
                for (int32_t o = 0; o < ol.nops; o++) {
                    const orc_op* op = &ol.ops[o];
                    if ((op->state == 1 || op->state == 2) && pos >= tpos + 1 && pos <= tpos + op->ref_len)
                        bq = R->bq[R->qoff[r] + qpos + (pos - tpos - 1)];
                    tpos += op->ref_len;
                    qpos += op->alt_len;
                }
            }
            ok[a] = bq >= min_bq;
            state[a] = (qb == href[idx + a]) ? 0 : 1;
        }
        for (int64_t a = 0; a < k && !rc; a++) {
            if (!ok[a]) continue;
            for (int64_t b = a + 1; b < k; b++) {
                if (!ok[b]) continue;
                if (b - a - 1 >= band) { rc = ORC_ERR_CAPACITY; break; }
                int kk = (!state[a] && !state[b]) ? 0 : (state[a] && state[b]) ? 1 : (!state[a] && state[b]) ? 2 : 3;
                counts[((idx + a) * band + (b - a - 1)) * 4 + kk]++;
            }
        }
        free(state); free(ok); free(ol.ops); free(OL);
    }
    return rc;
}

/* ---------------- leaf entry points used by the golden-vector tests ---------------- */

/* cs2tuple of read r: fills state/ref_len/alt_len/ref/alt per op; returns nops or -err */
int orc_cs_ops(const orc_reads* R, int64_t r, int32_t cap, uint8_t* state, int32_t* ref_len, int32_t* alt_len,
               uint8_t* ref, uint8_t* alt) {
    orc_oplist ol;
    int rc = parse_cs(R, r, &ol);
    if (rc) return -rc;
    int32_t k;
    for (k = 0; k < ol.nops && k < cap; k++) {
        state[k] = ol.ops[k].state; ref_len[k] = ol.ops[k].ref_len; alt_len[k] = ol.ops[k].alt_len;
        ref[k] = ol.ops[k].ref; alt[k] = ol.ops[k].alt;
    }
    int n = ol.nops;
    free(ol.ops);
    return n;
}

/* Pile of all reads (no chunking): counts[(pos - p0) * 6 + a] for pos in [p0, p1). */
int orc_pile_counts(const orc_reads* R, int32_t p0, int32_t p1, uint32_t* counts, uint32_t* bqsum) {
    for (int64_t r = 0; r < R->n; r++) {
        if (R->flag[r] & 0x100) continue;
        orc_oplist ol;
        int rc = parse_cs(R, r, &ol);
        if (rc) return rc;
        int64_t tpos = R->tstart[r], qpos = R->qstart[r];
        for (int32_t k = 0; k < ol.nops; k++) {
            const orc_op* op = &ol.ops[k];
            if (op->state == 1 || op->state == 2) {
                for (int32_t i = 0; i < op->ref_len; i++) {
                    int64_t p = tpos + i;
                    int b = op->state == 1 ? base2idx(match_base(R, r, op, qpos, i)) : base2idx(op->alt);
                    if (b < 0) { free(ol.ops); return ORC_ERR_BASE; }
                    if (p >= p0 && p < p1) {
                        counts[(p - p0) * 6 + b]++;
                        bqsum[(p - p0) * 4 + b] += R->bq[R->qoff[r] + qpos + i];
                    }
                }
            } else if (op->state == 3) {
                if (tpos >= p0 && tpos < p1) counts[(tpos - p0) * 6 + 4]++;
            } else {
                for (int32_t j = 0; j < op->ref_len; j++)
                    if (tpos + j >= p0 && tpos + j < p1) counts[(tpos + j - p0) * 6 + 5]++;
            }
            tpos += op->ref_len;
            qpos += op->alt_len;
        }
        free(ol.ops);
    }
    return ORC_OK;
}

/* get_germ_gt on an explicit column (alleles 0..3 in fetch order). Returns gq,
 * writes gt chars (reference allele first) and the state. */
int orc_germ_gt(int ref, int32_t depth, const uint8_t* allele, const uint8_t* bq, const orc_lut* L, int* gt0,
                int* gt1, int* state, double pl_out[10]) {
    pile_entry* col = (pile_entry*)malloc(sizeof(pile_entry) * (size_t)(depth > 0 ? depth : 1));
    for (int32_t k = 0; k < depth; k++) { col[k].allele = allele[k]; col[k].bq = bq[k]; col[k].read = k; }
    double pl[10];
    int st[10], order[10];
    int rc = column_pls(col, depth, ref, L, pl, st);
    free(col);
    if (rc) return -rc;
    argsort10(pl, order);
    double gqf = pl[order[1]] - pl[order[0]];
    int gq = gqf < 99.0 ? (int)gqf : 99;
    int g0 = GT_LST[order[0]][0], g1 = GT_LST[order[0]][1];
    if (g0 != ref && ((g0 == ref) + (g1 == ref)) == 1) { int t = g0; g0 = g1; g1 = t; }
    *gt0 = g0; *gt1 = g1; *state = st[order[0]];
    if (pl_out) memcpy(pl_out, pl, sizeof(pl));
    return gq;
}

int orc_record_size(void) { return (int)sizeof(orc_record); }
