// Device code of libhimut_hip.so: hand-written HIP kernels for gfx950 (MI355X).
//
// Pipeline for one contig (himut_run in himut_hip.hip launches them in order):
//
//   k_bqsum            one wave per read: sum of BQ over the whole query (qv filter,
//                      caller.py:85-94 / bamlib.py:34-36)
//   k_parse_cs         one thread per read: cs tag -> gapless segments + mismatch
//                      positions + identity, chunk-independent read filters
//                      (cslib.py:7-64, bamlib.py:47-63, caller.py:310-317)
//   k_read_hap         (--phase) one thread per (chunk, read): haplib.py:46-83
//   k_emit_candidates  one thread per read: trim / mismatch-window filters
//                      (bamlib.py:69-86,222-282) -> per-chunk candidate bit mask
//   k_pileup_sweep     one workgroup per 256-position tile of a chunk: stages the
//                      base/BQ rows of every overlapping read through LDS, reduces the
//                      columns (caller.py:44-72, bamlib.py:181-219) and runs the
//                      genotyper + filter cascade on candidate columns
//                      (gtlib.py:72-174, caller.py:324-621)
//   k_record_keys / k_resolve_seen / k_finalize_flags / k_compact
//                      sorted order, cross-chunk som_seen (caller.py:244,347,
//                      bamlib.py:77), the 15 counters (caller.py:625-641), set()
//                      de-duplication (caller.py:622-624)
//
// Integer work plus a small fp64 tail; no MFMA.  Wave size 64 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "himut_hip.h"

namespace himut {

constexpr int TP = 256;  // reference positions per tile == threads per workgroup
constexpr int RB = 64;   // pile rows staged per LDS batch

// pile cell (LDS): bits 0-2 allele, bit 3 "an insertion precedes this position"
constexpr uint8_t CELL_OTHER = 4;  // query base outside ATGC (reference raises KeyError)
constexpr uint8_t CELL_DEL = 5;
constexpr uint8_t CELL_EMPTY = 7;
constexpr uint8_t CELL_INS = 8;

constexpr uint32_t SEG_DEL = 1;
constexpr uint32_t SEG_INS = 2;

constexpr uint8_t RF_SECONDARY = 1;
constexpr uint8_t RF_PASS = 2;

constexpr uint8_t REC_GERM = 1;        // dropped as germline (caller.py:338-345): counted, no record
constexpr uint8_t REC_SUPPRESSED = 2;  // tpos already in som_seen from an earlier chunk
constexpr uint8_t REC_DUP = 4;         // identical tuple (HetAltSite printed once)

constexpr uint8_t HAP_0 = 0, HAP_1 = 1, HAP_NONE = 2;

struct Seg {
    int32_t t0;      // first reference position (0-based)
    int32_t q0;      // query offset of the first base (soft clip included)
    int32_t len;     // reference length (0 for a trailing insertion marker)
    uint32_t flags;  // SEG_DEL | SEG_INS
};

struct Reads {
    int64_t n;
    const int32_t *tstart, *tend, *qstart, *qlen;
    const uint8_t* mapq;
    const uint16_t* flag;
    const int32_t* qid;
    const int64_t *qoff, *cs_off;
    const uint8_t *seq, *bq, *cs;
    const int32_t* prefmax_tend;  // running maximum of tend in file order
};

struct Derived {
    uint32_t* bqsum;
    int32_t* nseg;
    int32_t* nmis;
    Seg* segs;      // seg_base(r) = (cs_off[r] >> 1) + r
    int32_t* mis;   // same base; 1-based mismatch positions (cslib.py:54-62)
    uint8_t* rflag;
};

struct Chunks {
    int64_t n;
    const int32_t *start, *end;
    const int64_t* maskoff;  // prefix of (end - start + 1)
    const int64_t* tileoff;  // prefix of ceil((end - start + 1) / TP)
    const int32_t* s_start;  // starts sorted ascending
    const int32_t* s_idx;    // chunk index per sorted slot
    const int32_t* s_pmaxend;  // prefix maximum of end in sorted order
    const int64_t* rlo;      // first read with prefmax_tend > start
    const int64_t* rhi;      // first read with tstart >= end
    const int64_t* pairoff;  // prefix of (rhi - rlo)
};

struct Phase {
    const int64_t* off;
    const int32_t* hpos;
    const uint8_t *href, *halt, *hbit;
    uint8_t* hap;  // per (chunk, read) pair
};

struct Params {
    himut_params p;
    int32_t unique_qnames;
};

struct GtLut {
    double t[3][256];  // hom / het / err indexed by BQ
    double prior[4];   // homref het hetalt homalt
};

__device__ __forceinline__ int64_t seg_base(const Reads& R, int64_t r) { return (R.cs_off[r] >> 1) + r; }

__device__ __forceinline__ int nib_at(const uint8_t* seq, int64_t o) {
    uint8_t b = seq[o >> 1];
    return (o & 1) ? (b & 15) : (b >> 4);
}
// BAM nibble -> himut allele index A0 T1 G2 C3 (util.py:14-20), 4 otherwise
__device__ __forceinline__ int nib2allele(int n) { return (int)((0x4444444144424304ULL >> (4 * n)) & 15); }
__device__ __forceinline__ int nib2char(int n) { return "=ACMGRSVTWYHKDBN"[n]; }
__device__ __forceinline__ int allele2char(int a) { return "ATGC"[a & 3]; }
__device__ __forceinline__ int char2allele(int c) {
    return c == 'A' ? 0 : c == 'T' ? 1 : c == 'G' ? 2 : c == 'C' ? 3 : -1;
}
__device__ __forceinline__ int asc_rank(int a) { return a == 0 ? 0 : a == 3 ? 1 : a == 2 ? 2 : 3; }  // A<C<G<T
__device__ __forceinline__ int upper(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
__device__ __forceinline__ bool is_alpha(int c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

template <class T>
__device__ __forceinline__ int64_t lower_bound(const T* a, int64_t lo, int64_t hi, T x) {  // first a[i] >= x
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (a[m] < x) lo = m + 1; else hi = m; }
    return lo;
}
template <class T>
__device__ __forceinline__ int64_t upper_bound(const T* a, int64_t lo, int64_t hi, T x) {  // first a[i] > x
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (x < a[m]) hi = m; else lo = m + 1; }
    return lo;
}

__device__ __forceinline__ void set_err(int* err, int code) { atomicOr(err, 1 << code); }

// ---------------------------------------------------------------------------------------
// k_bqsum: one wave per read, 16-byte coalesced loads (qoff is a multiple of 32).
__global__ void __launch_bounds__(256) k_bqsum(Reads R, Derived D) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R.n) return;
    const uint8_t* base = R.bq + R.qoff[r];
    const int n = R.qlen[r];
    uint32_t sum = 0;
    for (int o = lane * 16; o < n; o += 64 * 16) {
        uint4 v = *reinterpret_cast<const uint4*>(base + o);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int rem = n - (o + 4 * k);
            uint32_t x = w[k];
            if (rem < 4) x = rem <= 0 ? 0u : (x & (0xffffffffu >> (8 * (4 - rem))));
            uint32_t s2 = (x & 0x00ff00ffu) + ((x >> 8) & 0x00ff00ffu);
            sum += (s2 & 0xffffu) + (s2 >> 16);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if (lane == 0) D.bqsum[r] = sum;
}

// ---------------------------------------------------------------------------------------
// cs tokenizer shared by k_parse_cs and k_emit_candidates.  Mirrors the regex of
// cslib.py:8  (:[0-9]+|\*[a-z][a-z]|[=\+\-][A-Za-z]+).
struct CsOp {
    int kind;     // ':' '*' '=' '+' '-' ; 0 on error
    int len;      // run length / letters
    int ref, alt; // '*' only, upper case
    int64_t text; // offset of the letters ('=' '+' '-')
};

__device__ __forceinline__ int64_t cs_next(const uint8_t* s, int64_t i, int64_t n, CsOp& op) {
    int c = s[i];
    op.kind = 0;
    if (c == ':') {
        int64_t j = i + 1;
        int64_t v = 0;
        while (j < n && s[j] >= '0' && s[j] <= '9') { v = v * 10 + (s[j] - '0'); j++; }
        if (j == i + 1) return n;
        op.kind = ':'; op.len = (int)v;
        return j;
    }
    if (c == '*') {
        if (i + 2 >= n) return n;
        int a = s[i + 1], b = s[i + 2];
        if (!(a >= 'a' && a <= 'z') || !(b >= 'a' && b <= 'z')) return n;
        op.kind = '*'; op.len = 1; op.ref = a - 32; op.alt = b - 32;
        return i + 3;
    }
    if (c == '=' || c == '+' || c == '-') {
        int64_t j = i + 1;
        while (j < n && is_alpha(s[j])) j++;
        if (j == i + 1) return n;
        op.kind = c; op.len = (int)(j - i - 1); op.text = i + 1;
        return j;
    }
    return n;
}

// k_parse_cs: thread per read.
__global__ void __launch_bounds__(256) k_parse_cs(Reads R, Derived D, Params P, int* err) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n) return;
    if (R.flag[r] & 0x100) { D.rflag[r] = RF_SECONDARY; D.nseg[r] = 0; D.nmis[r] = 0; return; }  // bamlib.py:17
    const uint8_t* s = R.cs + R.cs_off[r];
    const int64_t n = R.cs_off[r + 1] - R.cs_off[r];
    const int64_t sb = seg_base(R, r);
    Seg* segs = D.segs + sb;
    int32_t* mis = D.mis + sb;
    const int64_t qo = R.qoff[r];
    int64_t t = R.tstart[r], q = R.qstart[r];
    int ns = 0, nm = 0;
    int64_t match = 0, mism = 0;
    bool open = false, pending_ins = false;
    Seg cur = {0, 0, 0, 0};
    int bad = 0;
    int64_t i = 0;
    while (i < n) {
        CsOp op;
        i = cs_next(s, i, n, op);
        if (op.kind == 0) { bad = HIMUT_ERR_CS; break; }
        if (op.kind == ':' || op.kind == '=' || op.kind == '*') {
            if (!open) {
                cur.t0 = (int32_t)t; cur.q0 = (int32_t)q; cur.len = 0; cur.flags = pending_ins ? SEG_INS : 0;
                pending_ins = false; open = true;
            }
            if (op.kind == '*') {
                int qa = nib2allele(nib_at(R.seq, qo + q));
                if (char2allele(op.alt) < 0 || qa > 3) bad = HIMUT_ERR_BASE;          // caller.py:62
                else if (qa != char2allele(op.alt)) bad = HIMUT_ERR_CS;                // cs vs SEQ
                if (op.ref != 'N') {
                    if (char2allele(op.ref) < 0) bad = HIMUT_ERR_BASE;                 // bamlib.py:188
                    mis[nm++] = (int32_t)(t + 1);                                      // cslib.py:56-60
                }
                mism += 1;
            } else {
                if (op.kind == '=') {  // long form: letters must agree with SEQ
                    for (int k = 0; k < op.len; k++)
                        if (upper(s[op.text + k]) != nib2char(nib_at(R.seq, qo + q + k))) bad = HIMUT_ERR_CS;
                }
                match += op.len;
            }
            cur.len += op.len; t += op.len; q += op.len;
        } else {
            if (open) { segs[ns++] = cur; open = false; }
            if (op.kind == '+') {
                if (pending_ins) bad = HIMUT_ERR_CS;  // two insertions in a row: unsupported
                pending_ins = true;
                mis[nm++] = (int32_t)(t + 1);
                q += op.len; mism += op.len;
            } else {
                Seg d = {(int32_t)t, (int32_t)q, op.len, SEG_DEL | (pending_ins ? SEG_INS : 0u)};
                pending_ins = false;
                segs[ns++] = d;
                mis[nm++] = (int32_t)(t + 1);
                t += op.len; mism += op.len;
            }
        }
        if (bad) break;
    }
    if (open) segs[ns++] = cur;
    if (pending_ins) { Seg z = {(int32_t)t, (int32_t)q, 0, SEG_INS}; segs[ns++] = z; }
    if (!bad && (t != R.tend[r] || q > R.qlen[r])) bad = HIMUT_ERR_CS;  // cs inconsistent with CIGAR / SEQ
    if (bad) { set_err(err, bad); ns = 0; nm = 0; }
    D.nseg[r] = ns;
    D.nmis[r] = nm;
    // chunk-independent read filters, caller.py:310-317
    const int32_t qlen = R.qlen[r];
    bool pass = !bad;
    double qv = (double)D.bqsum[r] / (double)qlen;                    // bamlib.py:35
    if (qv < (double)P.p.min_qv) pass = false;
    if ((int)R.mapq[r] < P.p.min_mapq) pass = false;
    double ident = (double)match / (double)(match + mism);            // bamlib.py:61-62
    if (ident < P.p.min_sequence_identity) pass = false;
    if (!(P.p.qlen_lower_limit < qlen && qlen < P.p.qlen_upper_limit)) pass = false;
    D.rflag[r] = pass ? RF_PASS : 0;
}

// allele (0..3), CELL_OTHER, CELL_DEL or -1 (not covered) of read r at 0-based rpos
__device__ __forceinline__ int allele_at(const Reads& R, const Derived& D, int64_t r, int32_t rpos, int* nibout) {
    const Seg* segs = D.segs + seg_base(R, r);
    const int ns = D.nseg[r];
    for (int j = 0; j < ns; j++) {
        Seg sg = segs[j];
        if (rpos < sg.t0) break;
        if (rpos < sg.t0 + sg.len) {
            if (sg.flags & SEG_DEL) return CELL_DEL;
            int nb = nib_at(R.seq, R.qoff[r] + sg.q0 + (rpos - sg.t0));
            if (nibout) *nibout = nb;
            return nib2allele(nb);
        }
    }
    return -1;
}

// ---------------------------------------------------------------------------------------
// k_read_hap: thread per (chunk, read-in-window) pair; haplib.get_ccs_hap (haplib.py:61-83).
__global__ void __launch_bounds__(256) k_read_hap(Reads R, Derived D, Chunks C, Phase H, int64_t npairs, int* err) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npairs) return;
    const int64_t c = upper_bound(C.pairoff, (int64_t)0, C.n + 1, k) - 1;
    const int64_t r = C.rlo[c] + (k - C.pairoff[c]);
    uint8_t hap = HAP_NONE;
    const int32_t s = C.start[c], e = C.end[c];
    if (!(D.rflag[r] & RF_SECONDARY) && R.tstart[r] < e && R.tend[r] > s) {
        const int32_t* hpos = H.hpos;
        const int64_t a = H.off[c], b = H.off[c + 1];
        const int64_t idx = upper_bound(hpos, a, b, R.tstart[r]);  // bisect_right, haplib.py:68-69
        const int64_t jdx = upper_bound(hpos, a, b, R.tend[r]);
        if (jdx - idx >= 2) {
            bool all0 = true, all1 = true;
            const Seg* segs = D.segs + seg_base(R, r);
            const int ns = D.nseg[r];
            int j = 0;
            for (int64_t g = idx; g < jdx; g++) {
                const int32_t rpos = hpos[g] - 1;
                while (j < ns && rpos >= segs[j].t0 + segs[j].len) j++;
                int qb = 0;  // 0: not in tpos2qbase -> KeyError
                if (j < ns && rpos >= segs[j].t0) {
                    if (segs[j].flags & SEG_DEL) qb = '-';
                    else qb = nib2char(nib_at(R.seq, R.qoff[r] + segs[j].q0 + (rpos - segs[j].t0)));
                }
                if (qb == 0) { set_err(err, HIMUT_ERR_COVER); all0 = all1 = false; break; }
                int bit = '-';
                if (H.href[g] && qb == H.href[g]) bit = '0';        // haplib.py:52-57
                else if (H.halt[g] && qb == H.halt[g]) bit = '1';
                const int h0 = H.hbit[g];
                const int h1 = h0 == '0' ? '1' : (h0 == '1' ? '0' : '-');
                if (bit != h0) all0 = false;
                if (bit != h1) all1 = false;
            }
            hap = all0 ? HAP_0 : (all1 ? HAP_1 : HAP_NONE);
        }
    }
    H.hap[k] = hap;
}

// ---------------------------------------------------------------------------------------
// k_emit_candidates: thread per read that passed the read filters.  For every
// substitution that survives the trim and mismatch-window filters, sets the
// (ref, alt) bit of its position in the mask of every chunk that both contains
// tpos (caller.py:104-108,325) and fetched the read (caller.py:299).
__global__ void __launch_bounds__(256) k_emit_candidates(Reads R, Derived D, Chunks C, Phase H, Params P,
                                                         uint16_t* mask, uint8_t* ccs_flag,
                                                         unsigned long long* ncand, int* err) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n) return;
    if (!(D.rflag[r] & RF_PASS)) return;
    const int32_t ts = R.tstart[r], te = R.tend[r];
    const bool phase = P.p.phase != 0;
    // num_ccs (caller.py:318-320): the read is counted once it passes in any chunk that fetched it
    {
        int64_t hi = lower_bound(C.s_start, (int64_t)0, C.n, te);  // chunks with start < tend
        bool counted = false;
        for (int64_t j = hi - 1; j >= 0 && C.s_pmaxend[j] > ts && !counted; j--) {
            const int c = C.s_idx[j];
            if (C.end[c] > ts) {
                if (!phase) counted = true;
                else {
                    uint8_t h = H.hap[C.pairoff[c] + (r - C.rlo[c])];
                    if (h != HAP_NONE) counted = true;
                }
            }
        }
        if (counted) ccs_flag[R.qid[r]] = 1;
        else if (phase) { /* not phased anywhere: proposes nothing */ return; }
    }
    const uint8_t* s = R.cs + R.cs_off[r];
    const int64_t n = R.cs_off[r + 1] - R.cs_off[r];
    const int32_t* mis = D.mis + seg_base(R, r);
    const int nm = D.nmis[r];
    const int32_t qlen = R.qlen[r];
    const double trim_start = floor(P.p.min_trim * (double)qlen);        // bamlib.py:226
    const double trim_end = ceil((1.0 - P.p.min_trim) * (double)qlen);   // bamlib.py:227
    const int64_t w = P.p.mismatch_window_size;
    int64_t t = ts, q = R.qstart[r];
    int64_t i = 0;
    while (i < n) {
        CsOp op;
        i = cs_next(s, i, n, op);
        if (op.kind == 0) break;
        if (op.kind == ':' || op.kind == '=') { t += op.len; q += op.len; continue; }
        if (op.kind == '+') { q += op.len; continue; }
        if (op.kind == '-') { t += op.len; continue; }
        // substitution
        if (op.ref != 'N') {
            const int32_t tp1 = (int32_t)(t + 1);
            bool ok = !((double)q < trim_start || (double)q > trim_end);  // bamlib.py:231-242
            if (ok) {                                                       // bamlib.py:245-282
                int64_t qs = q - w, qe = q + w, ur, dr;
                if (qs < 0) { ur = w + qs; dr = w - qs; }
                else if (qe > qlen) { ur = w + (qe - qlen); dr = qlen - q; }
                else { ur = w; dr = w; }
                const int32_t ms = (int32_t)(tp1 - ur), me = (int32_t)(tp1 + dr);
                int64_t cnt = upper_bound(mis, (int64_t)0, (int64_t)nm, me) - lower_bound(mis, (int64_t)0, (int64_t)nm, ms) - 1;
                if (cnt > P.p.max_mismatch_count) ok = false;
            }
            if (ok) {
                const int bit = char2allele(op.ref) * 4 + char2allele(op.alt);
                int64_t hi = upper_bound(C.s_start, (int64_t)0, C.n, tp1);  // chunks with start <= tpos
                for (int64_t j = hi - 1; j >= 0 && C.s_pmaxend[j] >= tp1; j--) {
                    const int c = C.s_idx[j];
                    const int32_t cs_ = C.start[c], ce_ = C.end[c];
                    if (ce_ < tp1) continue;
                    if (!(ts < ce_ && te > cs_)) continue;  // not fetched by this chunk
                    if (phase) {
                        uint8_t h = H.hap[C.pairoff[c] + (r - C.rlo[c])];
                        if (h == HAP_NONE) continue;        // caller.py:306-309
                    }
                    uint16_t* cell = mask + C.maskoff[c] + (tp1 - cs_);
                    // 16-bit atomic OR through the containing aligned 32-bit word
                    uintptr_t addr = (uintptr_t)cell;
                    unsigned int* word = (unsigned int*)(addr & ~(uintptr_t)3);
                    const unsigned int sh = (addr & 2) ? 16u : 0u;
                    const unsigned int m = (1u << bit) << sh;
                    unsigned int old = atomicOr(word, m);
                    if (!(old & m)) atomicAdd(ncand, 1ULL);
                }
            }
        }
        t += 1; q += 1;
    }
}

// ---------------------------------------------------------------------------------------
// Candidate evaluation on one pile column (thread-level).

struct Column {
    uint32_t cnt[6];
    uint32_t bqs[4];
    uint32_t maxbq[4];
    double S[3][4];          // per allele: sum of log10 terms for hom / het / err (gtlib.py:84-93)
    unsigned long long h0, h1;  // 16-bit packed per-allele counts of rows with hap 0 / 1 that also cover rpos + 1
};

struct SiteSets {
    const uint64_t* pon; int64_t npon;
    const uint64_t* com; int64_t ncom;
};

__device__ __forceinline__ bool key_in(const uint64_t* a, int64_t n, uint64_t x) {
    int64_t k = lower_bound(a, (int64_t)0, n, x);
    return k < n && a[k] == x;
}

// genotype list of gtlib.py:9 in himut allele indices (A0 T1 G2 C3)
__device__ __constant__ const uint8_t GT_B1[10] = {0, 1, 3, 2, 1, 3, 2, 3, 2, 2};
__device__ __constant__ const uint8_t GT_B2[10] = {0, 0, 0, 0, 1, 1, 1, 3, 3, 2};

__device__ __forceinline__ int gt_state_of(int b1, int b2, int ref) {  // gtlib.py:23-38
    if (b1 == b2 && b2 == ref) return 0;
    if ((b1 == ref) != (b2 == ref)) return 1;
    if (b1 != b2) return 2;
    return 3;
}

__device__ void eval_candidate(const Column& col, int ref, int alt, int32_t tpos, int32_t chunk, int32_t cstart,
                               const double* prior, const Params& P, const SiteSets& S, himut_record* out,
                               unsigned long long* nrec, int64_t cap) {
    // ten PLs, gtlib.py:72-110
    double best = 0.0, second = 0.0;
    int ibest = -1;
    double pl[10];
#pragma unroll
    for (int g = 0; g < 10; g++) {
        const int b1 = GT_B1[g], b2 = GT_B2[g];
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double term;
            if (b1 == b2 && b == b1) term = col.S[0][b];
            else if (b1 != b2 && (b == b1 || b == b2)) term = col.S[1][b];
            else term = col.S[2][b];
            acc = acc + term;
        }
        acc = acc + prior[gt_state_of(b1, b2, ref)];
        pl[g] = -10.0 * acc;
    }
    // np.argsort with the scalar insertion sort: ties -> lower index (gtlib.py:113-119)
#pragma unroll
    for (int g = 0; g < 10; g++)
        if (ibest < 0 || pl[g] < best) { best = pl[g]; ibest = g; }
    bool have2 = false;
#pragma unroll
    for (int g = 0; g < 10; g++) {
        if (g == ibest) continue;
        if (!have2 || pl[g] < second) { second = pl[g]; have2 = true; }
    }
    const double gqf = second - best;
    const int gq = gqf < 99.0 ? (int)gqf : 99;
    int g0 = GT_B1[ibest], g1 = GT_B2[ibest];
    const int state = gt_state_of(g0, g1, ref);
    if (g0 != ref && ((g0 == ref) + (g1 == ref)) == 1) { int tmp = g0; g0 = g1; g1 = tmp; }  // gtlib.py:133-134

    const uint32_t* c = col.cnt;
    const uint32_t ref_count = c[ref], alt_count = c[alt];
    const uint32_t depth = c[0] + c[1] + c[2] + c[3] + c[5];  // bamlib.py:213-219
    bool germ;  // caller.py:111-147
    if (state == 1) germ = (g0 == ref && g1 == alt);
    else if (state == 2) germ = ((c[0] + c[1] + c[2] + c[3]) == (c[g0] + c[g1])) && (alt == g0 || alt == g1);
    else if (state == 3) germ = (ref_count == 0) && (g0 == alt && g1 == alt);
    else germ = (alt == g0);

    int status = 255;
    int32_t ps = -1;
    uint8_t flags = 0;
    if (germ) flags = REC_GERM;
    else if (state == 1) status = HIMUT_ST_HET;
    else if (state == 2) status = HIMUT_ST_HETALT;
    else if (state == 3) status = HIMUT_ST_HOMALT;
    else if (c[5] != 0 || c[4] != 0) status = HIMUT_ST_INDEL;
    else {
        const uint64_t key = ((uint64_t)(uint32_t)tpos << 4) | ((uint64_t)ref << 2) | (uint64_t)alt;
        if (gq < P.p.min_gq) status = HIMUT_ST_LOWGQ;
        else if ((int)col.maxbq[alt] < P.p.min_bq) status = HIMUT_ST_LOWBQ;   // caller.py:160-171
        else if (key_in(S.pon, S.npon, key)) status = HIMUT_ST_PON;
        else if (key_in(S.com, S.ncom, key)) status = HIMUT_ST_COMSNP;
        else if (!((int64_t)ref_count >= P.p.min_ref_count && (int64_t)alt_count >= P.p.min_alt_count)) status = HIMUT_ST_LOWDEPTH;
        else if ((int64_t)depth > P.p.md_threshold) status = HIMUT_ST_HIGHDEPTH;
        else if (P.p.phase) {  // caller.py:552-603 (unique query names: the voters are the pile's own rows)
            const int64_t h0 = (int64_t)((col.h0 >> (16 * ref)) & 0xffff), h1 = (int64_t)((col.h1 >> (16 * ref)) & 0xffff);
            const int som = (((col.h0 >> (16 * alt)) & 0xffff) ? 1 : 0) + (((col.h1 >> (16 * alt)) & 0xffff) ? 1 : 0);
            if (h0 >= P.p.min_hap_count && h1 >= P.p.min_hap_count && som == 1) { status = HIMUT_ST_PASS; ps = cstart; }
            else status = HIMUT_ST_UNPHASED;
        } else status = HIMUT_ST_PASS;
    }
    const unsigned long long idx = atomicAdd(nrec, 1ULL);
    if ((int64_t)idx >= cap) return;
    himut_record rec;
    rec.tpos = tpos; rec.chunk = chunk; rec.phase_set = ps; rec.gq = gq;
    rec.ref = (uint8_t)allele2char(ref); rec.alt = (uint8_t)allele2char(alt);
    rec.gt0 = (uint8_t)allele2char(g0); rec.gt1 = (uint8_t)allele2char(g1);
    rec.status = (uint8_t)status; rec.gt_state = (uint8_t)state; rec.flags = flags; rec.pad = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) rec.counts[k] = c[k];
#pragma unroll
    for (int k = 0; k < 4; k++) rec.bqsum[k] = col.bqs[k];
    const uint4* src = reinterpret_cast<const uint4*>(&rec);
    uint4* dst = reinterpret_cast<uint4*>(out + idx);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
}

struct SweepArgs {
    Reads R;
    Derived D;
    Chunks C;
    Phase H;
    Params P;
    SiteSets S;
    const GtLut* lut;
    const uint16_t* mask;
    himut_record* recs;
    unsigned long long* nrec;
    int64_t cap;
    int64_t n_tiles;
    uint32_t* dense_counts;  // DENSE only: [position][6]
    uint32_t* dense_bqsum;   // DENSE only: [position][4]
    unsigned long long* row_bases;
    int* err;
};

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give
// each XCD a contiguous range of tiles (neighbouring tiles share reads and
// metadata -> same L2).  Bijective for any n (speed only, never correctness).
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n) {
    const int64_t q = n >> 3, rm = n & 7, x = b & 7;
    const int64_t base = x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q;
    return base + (b >> 3);
}

template <bool DENSE>
__global__ void __launch_bounds__(TP) k_pileup_sweep(SweepArgs A) {
    __shared__ __align__(16) uint8_t s_cell[RB * TP];
    __shared__ __align__(16) uint8_t s_bq[RB * TP];
    __shared__ double s_lut[3 * 256];
    __shared__ int s_rows[TP];
    __shared__ int s_rowtend[RB];
    __shared__ uint8_t s_rowhap[RB];
    __shared__ int s_wcnt[TP / 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Reads& R = A.R;
    const Derived& D = A.D;
    const Chunks& C = A.C;
    const int64_t tile = xcd_remap(blockIdx.x, A.n_tiles);
    const int64_t c = upper_bound(C.tileoff, (int64_t)0, C.n + 1, tile) - 1;
    const int32_t cs_ = C.start[c], ce_ = C.end[c];
    const int64_t local = (tile - C.tileoff[c]) * TP;
    const int32_t p0 = (int32_t)(cs_ - 1 + local);
    const int32_t p1 = (int32_t)min((int64_t)p0 + TP, (int64_t)ce_);  // rpos in [p0, p1); last rpos of the chunk is end - 1
    const int32_t p = p0 + tid;
    const bool active = p < p1;
    const bool phase = A.P.p.phase != 0;
    uint32_t mask = 0;
    if (!DENSE && active) mask = A.mask[C.maskoff[c] + local + tid];
    const bool need = mask != 0;

    for (int i = tid; i < 3 * 256; i += TP) s_lut[i] = A.lut->t[i >> 8][i & 255];

    Column col;
#pragma unroll
    for (int k = 0; k < 6; k++) col.cnt[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { col.bqs[k] = 0; col.maxbq[k] = 0; col.S[0][k] = 0.0; col.S[1][k] = 0.0; col.S[2][k] = 0.0; }
    col.h0 = 0; col.h1 = 0;
    int bad = 0;
    unsigned long long staged = 0;

    // reads that can overlap the tile: tstart < p1 and (running max of) tend >= p0
    const int64_t rlo = C.rlo[c], rhi = C.rhi[c];
    const int64_t hi = lower_bound(R.tstart, rlo, rhi, p1);
    const int64_t lo = lower_bound(R.prefmax_tend, rlo, hi, p0);

    for (int64_t base = lo; base < hi; base += TP) {
        // ordered compaction of the rows of this tile (fetch rule of the chunk: caller.py:299)
        const int64_t r = base + tid;
        bool ok = false;
        if (r < hi) {
            const int32_t te = R.tend[r];
            ok = !(D.rflag[r] & RF_SECONDARY) && te >= p0 && te > cs_ && R.tstart[r] < ce_;
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, nrows = 0;
#pragma unroll
        for (int k = 0; k < TP / 64; k++) { if (k < wave) woff += s_wcnt[k]; nrows += s_wcnt[k]; }
        if (ok) s_rows[woff + __popcll(bal & ((1ULL << lane) - 1ULL))] = (int)(r - base);
        __syncthreads();

        for (int b0 = 0; b0 < nrows; b0 += RB) {
            const int nb = min(RB, nrows - b0);
            // ---- stage nb rows: each wave takes rows wave, wave+4, ...
            for (int i = wave; i < nb; i += TP / 64) {
                const int64_t rr = base + s_rows[b0 + i];
                const Seg* segs = D.segs + seg_base(R, rr);
                const int ns = D.nseg[rr];
                const int64_t qo = R.qoff[rr];
                if (lane == 0) {
                    s_rowtend[i] = R.tend[rr];
                    s_rowhap[i] = phase ? A.H.hap[C.pairoff[c] + (rr - rlo)] : HAP_NONE;
                }
                // segments of the read that touch [p0, p1)
                int sfirst = ns, slast = -1;
                for (int j0 = 0; j0 < ns; j0 += 64) {
                    const int j = j0 + lane;
                    bool hit = false;
                    if (j < ns) {
                        Seg sg = segs[j];
                        int32_t eend = sg.t0 + ((sg.flags & SEG_INS) ? max(sg.len, 1) : sg.len);
                        hit = sg.t0 < p1 && eend > p0;
                    }
                    const unsigned long long hb = __ballot(hit);
                    if (hb) {
                        if (sfirst == ns) sfirst = j0 + __ffsll((long long)hb) - 1;
                        slast = j0 + 63 - __clzll((long long)hb);
                    }
                }
#pragma unroll
                for (int k = 0; k < TP / 64; k++) {
                    const int x = lane + 64 * k;
                    const int32_t pp = p0 + x;
                    uint8_t cell = CELL_EMPTY, bqv = 0;
                    for (int j = sfirst; j <= slast; j++) {
                        const Seg sg = segs[j];
                        if (pp == sg.t0 && (sg.flags & SEG_INS)) cell |= CELL_INS;
                        if (pp >= sg.t0 && pp < sg.t0 + sg.len) {
                            if (sg.flags & SEG_DEL) cell = (cell & CELL_INS) | CELL_DEL;
                            else {
                                const int64_t o = qo + sg.q0 + (pp - sg.t0);
                                cell = (cell & CELL_INS) | (uint8_t)nib2allele(nib_at(R.seq, o));
                                bqv = R.bq[o];
                            }
                        }
                    }
                    s_cell[i * TP + x] = cell;
                    s_bq[i * TP + x] = bqv;
                }
            }
            __syncthreads();
            // ---- column pass: thread = position, rows in fetch order
            if (active) {
                for (int i = 0; i < nb; i++) {
                    const uint8_t cell = s_cell[i * TP + tid];
                    if (cell == CELL_EMPTY) continue;
                    if (cell & CELL_INS) col.cnt[4]++;
                    const int a = cell & 7;
                    if (a < 4) {
                        const uint32_t q = s_bq[i * TP + tid];
                        staged++;
                        bool vote = false; uint8_t hp = HAP_NONE;
                        if (phase && need) { vote = s_rowtend[i] > p + 1; hp = s_rowhap[i]; }
                        double vh = 0.0, vt = 0.0, ve = 0.0;
                        if (need) {
                            if (q == 0) bad |= 1 << HIMUT_ERR_BQ0;
                            vh = s_lut[q]; vt = s_lut[256 + q]; ve = s_lut[512 + q];
                        }
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            if (a == b) {
                                col.cnt[b]++; col.bqs[b] += q; col.maxbq[b] = max(col.maxbq[b], q);
                                if (need) {
                                    col.S[0][b] = col.S[0][b] + vh;
                                    col.S[1][b] = col.S[1][b] + vt;
                                    col.S[2][b] = col.S[2][b] + ve;
                                }
                            }
                        }
                        if (vote) {
                            if (hp == HAP_0) col.h0 += 1ULL << (16 * a);
                            else if (hp == HAP_1) col.h1 += 1ULL << (16 * a);
                        }
                    } else if (a == CELL_DEL) col.cnt[5]++;
                    else if (a == CELL_OTHER) bad |= 1 << HIMUT_ERR_BASE;
                }
            }
            __syncthreads();
        }
    }

    if (DENSE) {
        if (active) {
            const int64_t o = C.maskoff[c] + local + tid;
#pragma unroll
            for (int k = 0; k < 6; k++) A.dense_counts[o * 6 + k] = col.cnt[k];
#pragma unroll
            for (int k = 0; k < 4; k++) A.dense_bqsum[o * 4 + k] = col.bqs[k];
        }
    } else if (need) {
        const int32_t tpos = p + 1;
        for (int bit = 0; bit < 16; bit++) {
            if (!(mask & (1u << bit))) continue;
            eval_candidate(col, bit >> 2, bit & 3, tpos, (int32_t)c, cs_, A.lut->prior, A.P, A.S, A.recs, A.nrec, A.cap);
        }
    }
    if (bad) atomicOr(A.err, bad);
    // pile cells this workgroup staged (for the roofline figure)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) staged += __shfl_down(staged, d, 64);
    if (lane == 0 && staged) atomicAdd(A.row_bases, staged);
}

// ---------------------------------------------------------------------------------------
// finalisation

__global__ void __launch_bounds__(256) k_record_keys(const himut_record* recs, int64_t n, uint64_t* keys, uint32_t* vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const himut_record& r = recs[i];
    keys[i] = ((uint64_t)(uint32_t)r.tpos << 28) | ((uint64_t)(uint32_t)r.chunk << 4) |
              ((uint64_t)asc_rank(char2allele(r.ref)) << 2) | (uint64_t)asc_rank(char2allele(r.alt));
    vals[i] = (uint32_t)i;
}

// som_seen across chunks (caller.py:244,347; bamlib.py:77): a candidate whose
// tpos was already added by an EARLIER chunk is never proposed again.
__global__ void __launch_bounds__(256) k_resolve_seen(himut_record* recs, const uint64_t* keys, const uint32_t* vals, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t tp = keys[i] >> 28;
    if (i > 0 && (keys[i - 1] >> 28) == tp) return;  // not a group head
    if (i + 1 >= n || (keys[i + 1] >> 28) != tp) return;  // single record: nothing to resolve
    bool seen = false;
    int64_t j = i;
    while (j < n && (keys[j] >> 28) == tp) {
        const uint64_t ch = (keys[j] >> 4) & 0xffffff;
        bool nongerm = false;
        int64_t k = j;
        while (k < n && (keys[k] >> 28) == tp && ((keys[k] >> 4) & 0xffffff) == ch) {
            himut_record& rec = recs[vals[k]];
            if (seen) rec.flags |= REC_SUPPRESSED;
            else if (!(rec.flags & REC_GERM)) nongerm = true;
            k++;
        }
        if (!seen && nongerm) seen = true;
        j = k;
    }
}

// counters (caller.py:625-641) + output flags in sorted order
__global__ void __launch_bounds__(256) k_finalize_flags(himut_record* recs, const uint64_t* keys, const uint32_t* vals, int64_t n,
                                                        uint32_t* emit, unsigned long long* log) {
    __shared__ unsigned int s_log[16];
    if (threadIdx.x < 16) s_log[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        himut_record& rec = recs[vals[i]];
        uint32_t e = 0;
        if (!(rec.flags & REC_SUPPRESSED)) {
            atomicAdd(&s_log[1], 1u);  // num_sbs
            if (rec.flags & REC_GERM) {
                if (rec.gt_state == 1) atomicAdd(&s_log[2], 1u);
                else if (rec.gt_state == 2) atomicAdd(&s_log[3], 1u);
                else if (rec.gt_state == 3) atomicAdd(&s_log[4], 1u);
            } else {
                const int st = rec.status;
                if (st == HIMUT_ST_HET || st == HIMUT_ST_HETALT || st == HIMUT_ST_HOMALT) atomicAdd(&s_log[5], 1u);
                else if (st == HIMUT_ST_INDEL) atomicAdd(&s_log[7], 1u);
                else {
                    atomicAdd(&s_log[6], 1u);  // num_homref_sbs
                    if (st == HIMUT_ST_LOWGQ) atomicAdd(&s_log[8], 1u);
                    else if (st == HIMUT_ST_LOWBQ) atomicAdd(&s_log[9], 1u);
                    else if (st == HIMUT_ST_PON) atomicAdd(&s_log[10], 1u);
                    else if (st == HIMUT_ST_COMSNP) atomicAdd(&s_log[11], 1u);
                    else if (st == HIMUT_ST_HIGHDEPTH) atomicAdd(&s_log[12], 1u);
                    else if (st == HIMUT_ST_LOWDEPTH) atomicAdd(&s_log[13], 1u);
                    else atomicAdd(&s_log[14], 1u);  // num_som: PASS and Unphased (caller.py:553,605)
                }
                e = 1;
                if (st == HIMUT_ST_HETALT && i > 0) {
                    // set(): a HetAltSite tuple printed as ref / "a1,a2" is identical for every alt of the column
                    const uint64_t m = ~(uint64_t)3;
                    if ((keys[i - 1] & m) == (keys[i] & m)) {
                        const himut_record& prev = recs[vals[i - 1]];
                        if (prev.status == HIMUT_ST_HETALT && !(prev.flags & (REC_SUPPRESSED | REC_GERM))) { e = 0; rec.flags |= REC_DUP; }
                    }
                }
            }
        }
        emit[i] = e;
    }
    __syncthreads();
    if (threadIdx.x < 15 && s_log[threadIdx.x]) atomicAdd(&log[threadIdx.x], (unsigned long long)s_log[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_compact(const himut_record* recs, const uint32_t* vals, const uint32_t* emit,
                                                 const uint32_t* pos, int64_t n, himut_record* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !emit[i]) return;
    const uint4* src = reinterpret_cast<const uint4*>(recs + vals[i]);
    uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    b.y &= 0xff00ffffu;  // flags byte (offset 22) -> 0
    uint4* dst = reinterpret_cast<uint4*>(out + pos[i]);
    dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
}

__global__ void __launch_bounds__(256) k_count_flags(const uint8_t* flags, int64_t n, unsigned long long* out) {
    unsigned int local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) local += flags[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, (unsigned long long)local);
}

}  // namespace himut
