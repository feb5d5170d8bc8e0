// Device code of libhimut_hip.so: hand-written HIP kernels for gfx950 (MI355X).
//
// Pipeline for one contig (himut_run in himut_hip.hip launches them in order; DESIGN.md sections 2-4):
//
//   k_parse_cs         one wave per read, a wave-parallel tokenizer: cs tag -> gapless segments + mismatch list +
//                      identity (cslib.py:7-64, bamlib.py:47-63); sets the bitmap of column positions (the
//                      substitutions of the reads that pass the filters known so far); checks the cs bases against
//                      SEQ; on its way every wave stores its share of the EMPTY column store
//   k_read_hap         (--phase) sixteen lanes per (chunk, read), a lane per hetSNP: haplib.py:46-83
//   k_window_index     per 256-position block: the range of reads that can cover it (once per pushed batch)
//   k_block_sums / k_block_table3
//                      per 256-position block the number of column positions and of column-store slots, their
//                      prefix sums (first rank, slot offset) and the block table
//   k_stream_capture   one wave per read, a software pipeline over windows of 2048 query bases: streams the read's
//                      qualities and packed bases once (16-byte coalesced loads, two windows ahead), sums the
//                      qualities (np.mean of the whole query, bamlib.py:34-36), enumerates the column positions
//                      under each window from the bitmap, one per lane, and drops their (allele, BQ) cells into the
//                      column store; then the read's proposals (propose_read): read filters of caller.py:310-317,
//                      trim / mismatch-window filters (bamlib.py:69-86,222-282); every surviving (chunk, tpos, ref,
//                      alt) sets its bit in a per-chunk mask (the set() of caller.py:324)
//   k_scan_small / k_mask_emit
//                      the set bits of the mask, enumerated in (chunk, tpos, ref, alt) order = the candidate list;
//                      radix-sorted afterwards only when the chunks are not already in coordinate order
//   k_eval_columns     one THREAD per candidate column: allele counts, BQ sums, the genotype likelihood sums added in
//                      fetch order exactly as the reference's python sum() does, genotype, filter cascade
//                      (caller.py:44-72,324-621, bamlib.py:181-219, gtlib.py:72-174)
//   k_finalize_flags / k_run_totals / k_compact
//                      cross-chunk som_seen (caller.py:244,347, bamlib.py:77), the 15 counters (caller.py:625-641),
//                      set() de-duplication (caller.py:622-624), the totals, the emitted records to the front
//   k_pile_dense       dense per-position pile (counts + BQ sums for EVERY position of a range): LDS-staged base/BQ
//                      tiles, one workgroup per tile.  Used by himut_pile_counts.
//
// Integer work plus a small fp64 tail; no MFMA.  Wave size 64 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "himut_hip.h"

namespace himut {

constexpr int WIN_SHIFT = 8;  // k_window_index granularity: 256 reference positions
constexpr int CHUNK_HINT_SHIFT = 14;  // chunk look-up hint granularity: 16 kb

// k_pile_dense geometry: tile width, LDS row batch, threads
constexpr int PD_TP = 512, PD_RB = 56, PD_NT = 256;

// pile cell: bits 0-2 allele, bit 3 "an insertion precedes this position"
constexpr uint8_t CELL_OTHER = 4;  // query base outside ATGC (reference raises KeyError)
constexpr uint8_t CELL_DEL = 5;
constexpr uint8_t CELL_EMPTY = 7;
constexpr uint8_t CELL_INS = 8;

constexpr uint32_t SEG_DEL = 1;
constexpr uint32_t SEG_INS = 2;

constexpr uint8_t RF_SECONDARY = 1;
constexpr uint8_t RF_PASS = 2;
constexpr uint8_t RF_IDENT_OK = 4;
constexpr uint8_t RF_LONGCS = 8;

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
    const uint8_t* nonacgt;       // per read: SEQ holds a base outside ATGC somewhere (k_flag_bases, once per pushed batch)
};

struct Derived {
    uint32_t* bqsum;
    int32_t* nseg;
    int32_t* nmis;
    Seg* segs;       // seg_base(r) = (cs_off[r] >> 1) + r
    int32_t* mis;    // same base; 1-based mismatch positions (cslib.py:54-62)
    uint32_t* mq;    // same base; per mismatch: qpos << 5 | (substitution ? 16 | ref << 2 | alt : 0)
    uint8_t* rflag;
    struct ReadMeta* meta;
    int32_t* nnsub;  // substitutions whose reference base is N (cslib.py:54-56 keeps them out of the mismatch list): their
                     // query offsets sit in mq[] from the TOP of the read's slots downwards, mq[top - k] = qpos << 5 | 8
};

// everything a pile row needs about its read, in one 32-byte load
struct ReadMeta {
    int32_t tstart, tend;
    int32_t nseg;
    uint32_t flags;   // RF_*
    int64_t segbase;
    int64_t qoff;
};

// one chunk, in the order of the sorted starts
struct ChunkRec {
    int32_t start, end;
    int32_t idx;        // chunk index
    int32_t pmaxend;    // running maximum of end up to and including this one
    int64_t maskoff;    // first mask cell
    int64_t pairbase;   // pairoff - rlo: + read index = the (chunk, read) pair
};

// what k_mask_emit needs about a mask tile (MASK_TILE_CELLS cells): the chunk of its first cell
struct MaskTile {
    int32_t ck0;      // chunk of the tile's first cell
    int32_t start0;   // its start
    int64_t off0;     // its first cell
    int64_t off1;     // first cell of the next chunk
    int64_t pad;
};

struct Chunks {
    int64_t n;
    const ChunkRec* rec;       // sorted by start
    const MaskTile* mtile;     // per mask tile
    const int32_t *start, *end;
    const int64_t* maskoff;    // prefix of (end - start + 1)
    const int32_t* s_start;    // starts sorted ascending
    const int32_t* s_idx;      // chunk index per sorted slot
    const int32_t* s_pmaxend;  // prefix maximum of end in sorted order
    const int64_t* rlo;        // first read with prefmax_tend > start
    const int64_t* rhi;        // first read with tstart >= end
    const int64_t* pairoff;    // prefix of (rhi - rlo)
    const int32_t* hint;       // hint[p >> CHUNK_HINT_SHIFT] = number of sorted starts <= (p >> SHIFT) << SHIFT
    int64_t nhint;
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

// one proposed (chunk, tpos, ref, alt)
struct Cand {
    int32_t tpos;        // 1-based
    uint32_t chunk_bit;  // chunk << 4 | (ref << 2 | alt)
};

__device__ __forceinline__ int64_t seg_base(const Reads& R, int64_t r) { return (R.cs_off[r] >> 1) + r; }

__device__ __forceinline__ int nib_at(const uint8_t* seq, int64_t o) {
    uint8_t b = seq[o >> 1];
    return (o & 1) ? (b & 15) : (b >> 4);
}
// BAM nibble -> himut allele index A0 T1 G2 C3 (util.py:14-20), 4 otherwise
__device__ __forceinline__ int nib2allele(int n) { return (int)((0x4444444144424304ULL >> (4 * n)) & 15); }
__device__ __forceinline__ int nib2char(int n) { return "=ACMGRSVTWYHKDBN"[n]; }
__device__ __forceinline__ int allele2char(int a) { return (int)((0x43475441u >> (8 * (a & 3))) & 255); }  // "ATGC"
__device__ __forceinline__ int char2allele(int c) {
    return c == 'A' ? 0 : c == 'T' ? 1 : c == 'G' ? 2 : c == 'C' ? 3 : -1;
}
__device__ __forceinline__ int asc_rank(int a) { return a == 0 ? 0 : a == 3 ? 1 : a == 2 ? 2 : 3; }  // A<C<G<T
__device__ __forceinline__ int upper(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
__device__ __forceinline__ bool is_alpha(int c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

// ---------------------------------------------------------------------------------------
// k_flag_bases: which reads hold a base outside ATGC anywhere in SEQ -- the reference's pile raises KeyError on one that is
// aligned (caller.py:57, util.py:17), wherever it sits in a fetched read.  The packed bases are read once per pushed batch
// (like the window index, the answer depends on the reads only); a flagged read -- CCS reads do not carry N -- is then
// looked at base by base, per run, by the kernel that knows what is aligned and what is fetched (aligned_bases_ok).
// One wave per read, 32 bases a lane and step; a BAM code is one of A C G T exactly when it has one bit set.
__global__ void __launch_bounds__(256) k_flag_bases(int64_t n, const int64_t* qoff, const int32_t* qlen, const uint8_t* seq, uint8_t* out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t qo = __builtin_amdgcn_readfirstlane((int)(qoff[r] >> 32)) * 4294967296ll + (uint32_t)__builtin_amdgcn_readfirstlane((int)qoff[r]);
    const int32_t ql = __builtin_amdgcn_readfirstlane(qlen[r]);
    uint32_t badw = 0;
    for (int32_t o = lane * 32; o < ql; o += 2048) {
        const uint4 v = *reinterpret_cast<const uint4*>(seq + ((qo + o) >> 1));      // (a read's bases start at a multiple of 32)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = w[k];
            const int left = ql - (o + 8 * k);                   // bases of the word inside the read (byte i: bases 2i, 2i + 1, high nibble first)
            if (left <= 0) continue;
            uint32_t pc = x - ((x >> 1) & 0x55555555u);
            pc = (pc & 0x33333333u) + ((pc >> 2) & 0x33333333u);   // bits set, per nibble
            uint32_t d = pc ^ 0x11111111u;
            if (left < 8) {                                      // nibbles behind the read's last base do not count
                uint32_t keep = 0;
                for (int b = 0; b < left; b++) keep |= 0xfu << (8 * (b >> 1) + ((b & 1) ? 0 : 4));
                d &= keep;
            }
            badw |= d;
        }
    }
    const unsigned long long any = __ballot(badw != 0);
    if (lane == 0) out[r] = any ? 1 : 0;
}

// every aligned base of read r (the gapless segments that are not deletions) is one of ATGC?  For the lanes [l0, l0 + nl) of
// a wave together; the answer is valid in every one of them.  (A flagged read only: k_flag_bases.)
__device__ __forceinline__ bool aligned_bases_ok(const Reads& R, const Seg* segs, int ns, int64_t qo, int l, int nl) {
    bool ok = true;
    for (int j = 0; j < ns; j++) {
        const Seg g = segs[j];
        if ((g.flags & SEG_DEL) || g.len <= 0) continue;
        for (int32_t i = l; i < g.len; i += nl) if (nib2allele(nib_at(R.seq, qo + g.q0 + i)) > 3) ok = false;
    }
    return ok;
}

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

// wave-uniform values belong in scalar registers: everything computed from them then runs on the scalar unit
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int64_t uni(int64_t v) {
    return ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// value of lane l, l wave-uniform
__device__ __forceinline__ int lane_val(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// slot reservation for the lanes that reach this point together: one atomic per wave
__device__ __forceinline__ unsigned long long wave_reserve(unsigned long long* counter) {
    const unsigned long long act = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)act) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(act));
    base = __shfl(base, leader, 64);
    return base + (unsigned long long)__popcll(act & ((1ULL << lane) - 1ULL));
}

// ---------------------------------------------------------------------------------------
// bq_issue / bq_finish: the wave streams its read's qualities with 16-byte coalesced loads, eight in flight
// (np.mean of the whole query, bamlib.py:34-36).  k_parse_cs issues the first eight rows before it decodes
// the cs tag and sums after it.
constexpr int BQ_AHEAD = 4;   // rows of 1 KB issued ahead of the decode

struct BqAhead {            // the first rows of a read's qualities, in flight while the wave decodes its cs tag
    uint4 v[BQ_AHEAD];
    const uint8_t* base;   // this lane's 16 bytes of row 0
    int n, npre;           // quality bytes of the read; whole 1 KB rows among the BQ_AHEAD (the others were clamped loads)
};

__device__ __forceinline__ void bq_issue(const Reads& R, int64_t r, int lane, BqAhead& A) {
    const uint8_t* row0 = R.bq + uni(R.qoff[r]);
    A.base = row0 + lane * 16;
    A.n = uni(R.qlen[r]);
    A.npre = min(BQ_AHEAD, A.n >> 10);
#pragma unroll
    for (int k = 0; k < BQ_AHEAD; k++)                   // a row past the whole ones: reload the read's first bytes (always there)
        A.v[k] = *reinterpret_cast<const uint4*>(k < A.npre ? A.base + k * 1024 : row0);
}

__device__ __forceinline__ void bq_finish(const BqAhead& A, int64_t r, int lane, uint32_t* bqsum) {
    const uint8_t* base = A.base;
    const int n = A.n;
    uint32_t sum = 0;
    const int nfull = n & ~1023;                 // whole 1 KB steps: four byte sums per lane and step
#define BQ_ADD(V) do { sum = __builtin_amdgcn_sad_u8(V.x, 0u, sum); sum = __builtin_amdgcn_sad_u8(V.y, 0u, sum); \
        sum = __builtin_amdgcn_sad_u8(V.z, 0u, sum); sum = __builtin_amdgcn_sad_u8(V.w, 0u, sum); } while (0)
#pragma unroll
    for (int k = 0; k < BQ_AHEAD; k++) if (k < A.npre) BQ_ADD(A.v[k]);
    int o = A.npre * 1024;
    for (; o + 8192 <= nfull; o += 8192) {       // eight loads in flight per lane
        const uint4 a = *reinterpret_cast<const uint4*>(base + o);
        const uint4 b = *reinterpret_cast<const uint4*>(base + o + 1024);
        const uint4 c = *reinterpret_cast<const uint4*>(base + o + 2048);
        const uint4 d = *reinterpret_cast<const uint4*>(base + o + 3072);
        const uint4 e = *reinterpret_cast<const uint4*>(base + o + 4096);
        const uint4 f = *reinterpret_cast<const uint4*>(base + o + 5120);
        const uint4 g = *reinterpret_cast<const uint4*>(base + o + 6144);
        const uint4 h = *reinterpret_cast<const uint4*>(base + o + 7168);
        BQ_ADD(a); BQ_ADD(b); BQ_ADD(c); BQ_ADD(d); BQ_ADD(e); BQ_ADD(f); BQ_ADD(g); BQ_ADD(h);
    }
    for (; o + 4096 <= nfull; o += 4096) {       // four loads in flight per lane
        const uint4 a = *reinterpret_cast<const uint4*>(base + o);
        const uint4 b = *reinterpret_cast<const uint4*>(base + o + 1024);
        const uint4 c = *reinterpret_cast<const uint4*>(base + o + 2048);
        const uint4 d = *reinterpret_cast<const uint4*>(base + o + 3072);
        BQ_ADD(a); BQ_ADD(b); BQ_ADD(c); BQ_ADD(d);
    }
    for (; o < nfull; o += 1024) {
        const uint4 a = *reinterpret_cast<const uint4*>(base + o);
        BQ_ADD(a);
    }
#undef BQ_ADD
    if (nfull + lane * 16 < n) {                 // the last, partial step: bytes behind the read are masked off
        const uint4 v = *reinterpret_cast<const uint4*>(base + nfull);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rem = n - (nfull + lane * 16 + 4 * k);
            uint32_t x = w[k];
            if (rem < 4) x = rem <= 0 ? 0u : (x & (0xffffffffu >> (8 * (4 - rem))));
            sum = __builtin_amdgcn_sad_u8(x, 0u, sum);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if (lane == 0) bqsum[r] = sum;
}

// ---------------------------------------------------------------------------------------
// k_parse_cs: one WAVE per read, a wave-parallel cs tokenizer.
//
// The tag is consumed 1 KB per step, 16 bytes per lane.  Operation starts are the bytes
// ':' '*' '+' '-' '=' (the alternatives of the reference's regex, cslib.py:8; payload bytes
// are digits or letters and can never be one of them), found by every lane in its own 16
// bytes and compacted into an LDS list with one wave scan.  An operation is complete when
// the NEXT start (or the end of the tag) is known, so lane k takes operation k of the
// list: kind = its first byte, payload = the bytes up to the next start.  Reference and
// query offsets are wave prefix sums of the per-operation advances; the mismatch list
// (cslib.py:47-64) and the gapless segments are written at offsets that come from two more
// scans.  Only a handful of scalars (running offsets, the open aligned run, the unfinished
// last operation) carry from one step to the next.

constexpr int PB = 1024;  // cs bytes per step

// inclusive wave scans on the DPP network: four shifts inside each row of 16 lanes, then
// the row totals are carried across with the two row broadcasts
__device__ __forceinline__ int wave_incl_add(int v, int) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int) {
    constexpr int lowest = -0x7fffffff - 1;
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x143, 0xc, 0xf, false));
    return v;
}
// inclusive count of the lanes up to and including this one for which p holds
__device__ __forceinline__ int wave_rank_incl(bool p) {
    const unsigned long long b = __ballot(p);
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u)) + (p ? 1 : 0);
}
__device__ __forceinline__ bool cs_is_start(int c) { return c == ':' || c == '*' || c == '+' || c == '-' || c == '='; }
__device__ __forceinline__ bool cs_is_digit(int c) { return c >= '0' && c <= '9'; }
// byte-parallel classification of four ASCII bytes; results carry 0x80 in the bytes that qualify
__device__ __forceinline__ uint32_t cs_eq_bytes(uint32_t w, uint32_t c) {          // bytes equal to c (bytes < 0x80)
    return ~((w ^ (c * 0x01010101u)) + 0x7f7f7f7fu) & 0x80808080u;
}
__device__ __forceinline__ uint32_t cs_ge_bytes(uint32_t w, uint32_t c) {          // bytes >= c (bytes < 0x80, c >= 1)
    return (w + (0x80u - c) * 0x01010101u) & 0x80808080u;
}
__device__ __forceinline__ uint32_t cs_start_bytes(uint32_t w) {
    return cs_eq_bytes(w, ':') | cs_eq_bytes(w, '*') | cs_eq_bytes(w, '+') | cs_eq_bytes(w, '-') | cs_eq_bytes(w, '=');
}
__device__ __forceinline__ uint32_t cs_payload_bytes(uint32_t w) {                 // digits and letters
    return (cs_ge_bytes(w, '0') & ~cs_ge_bytes(w, '9' + 1)) | (cs_ge_bytes(w, 'A') & ~cs_ge_bytes(w, 'Z' + 1)) |
           (cs_ge_bytes(w, 'a') & ~cs_ge_bytes(w, 'z' + 1));
}
__device__ __forceinline__ uint32_t cs_pack4(uint32_t f) {                         // 0x80 flags of bytes 0..3 -> bits 0..3
    const uint32_t g = f >> 7;
    return (g | (g >> 7) | (g >> 14) | (g >> 21)) & 15u;
}


// WITH_BQ: the wave also issues the first rows of its read's qualities (bq_issue) behind the first KB of the tag,
// decodes the tag while they are in flight, then sums the qualities (bq_finish): waves in that phase are bound by
// HBM, waves in the decode by VALU, and a CU holds both kinds at any time.  No caller asks for it any more: the call
// path takes the sum from k_stream_capture and normcounts from k_callable, which stream the qualities anyway, and
// the edge counts and the dense pile never looked at it (0.24 ms a contig each).
//
// posbits (call path; else null): the bitmap of reference positions at which a column must be captured = every
// substitution of every read that passes the filters known before the qualities have been streamed (identity,
// mapq, qlen: caller.py:312-317).  A superset of the candidate positions -- the whole-read quality mean
// (caller.py:310), the trim and mismatch-window filters and the chunk rules only take proposals away (k_propose,
// which runs behind the capture and knows the mean by then) -- and nearly equal to them.  The wave keeps the
// positions of its read in LDS and sets the bits once the identity is known (a read with more substitutions than
// the list holds sets them as it goes: a superset is all that is asked for); the atomics cost the decode nothing,
// it is bound by instruction issue.
constexpr int MARK_CAP = 128;

// The decode is bound by latency (a chain of three memory round trips per wave times the waves a CU holds), and what
// limits the waves is the scalar registers: the values the wave keeps uniform.  256-thread workgroups are admitted
// per CU up to 800 / (ceil(sgpr / 16) * 16 + 16) (MI355X_MICROARCH.md): 80 scalars give 8, the 106 the compiler takes
// unasked give 6 -- the cap costs a few spills to vector lanes and is worth a tenth of the kernel.
#ifndef HIMUT_PARSE_SGPR
#define HIMUT_PARSE_SGPR 80
#endif
template <bool WITH_BQ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(HIMUT_PARSE_SGPR)))
k_parse_cs(Reads R, Derived D, Params P, int* err, uint8_t* ccs, uint32_t* posbits,
                                                  int64_t nposwords, uint4* fill, int64_t fill16, int fill_per) {
    __shared__ __align__(16) uint8_t s_txt[4][32 + PB + 32];   // 32 bytes of the previous step, then this step
    __shared__ uint16_t s_start[4][PB + 8];                      // operation starts, relative to the step (an operation carried
                                                                 // over from an earlier step keeps its start in a register)
    __shared__ int32_t s_mark[4][MARK_CAP];                      // substitution positions of the read (0-based)
    const int tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6);
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= R.n) return;
    // The column store must be EMPTY before the capture writes into it (fill: its fill16 16-byte pieces, or null).  The
    // decode is bound by latency and leaves the memory system idle: every wave stores its share, fill_per pieces per lane,
    // and neither a fill between the decode and the capture nor a second stream is needed.
    if (fill) {
        const uint4 e = make_uint4(0x00070007u, 0x00070007u, 0x00070007u, 0x00070007u);   // CELL_EMPTY
        const int64_t f0 = r * 64 * fill_per;
        for (int k = 0; k < fill_per; k++) { const int64_t o = f0 + 64 * k + lane; if (o < fill16) fill[o] = e; }
    }
    if (lane == 0) ccs[r] = 0;               // the flag k_propose raises for a read that may propose (num_ccs)
    const int64_t cs0 = uni(R.cs_off[r]);
    const int64_t sb = (cs0 >> 1) + r;
    ReadMeta M;
    M.tstart = uni(R.tstart[r]); M.tend = uni(R.tend[r]); M.nseg = 0; M.flags = 0; M.segbase = sb; M.qoff = uni(R.qoff[r]);
    if (uni((int)R.flag[r]) & 0x100) {  // bamlib.py:17
        if (lane == 0) {
            M.flags = RF_SECONDARY;
            D.rflag[r] = RF_SECONDARY; D.nseg[r] = 0; D.nmis[r] = 0; D.nnsub[r] = 0; D.meta[r] = M;
        }
        return;
    }
    const int n = (int)(uni(R.cs_off[r + 1]) - cs0);
    const uint8_t* cs = R.cs + cs0;
    uint8_t* txt = s_txt[wv];
    uint16_t* starts = s_start[wv];
    Seg* segs = D.segs + sb;
    int32_t* mis = D.mis + sb;
    uint32_t* mq = D.mq + sb;
    const int32_t qlen = uni(R.qlen[r]);
    int32_t* marks = s_mark[wv];
    int nmark = 0;
    int nN = 0;                                                          // substitutions with an N reference base
    const int64_t top = (uni(R.cs_off[r + 1]) >> 1) - (cs0 >> 1);        // the read's last slot (an operation takes >= 2 bytes)
    const bool mark = posbits != nullptr && !(uni((int)R.mapq[r]) < P.p.min_mapq) &&
                      (P.p.qlen_lower_limit < qlen && qlen < P.p.qlen_upper_limit);       // caller.py:312-317
    auto flush_marks = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nmark; i += 64) {
            const int32_t p = marks[i];
            if (p >= 0 && (int64_t)(p >> 5) < nposwords) atomicOr(posbits + (p >> 5), 1u << (p & 31));
        }
        nmark = 0;
        __builtin_amdgcn_wave_barrier();
    };
    // wave-uniform running state
    int t = M.tstart, q = uni(R.qstart[r]);
    int ns = 0, nm = 0, bad = 0;
    int match = 0, mism = 0;         // per lane; summed over the wave after the last step (a read's reference span fits 31 bits)
    bool have_carry = false; int carry_start = 0, carry_kind = 0;     // unfinished last operation of the previous step
    bool aligned_open = false; int run_t0 = 0, run_q0 = 0; bool run_ins = false;  // the aligned run still growing
    int last_kind = 0;                                                 // kind of the last finished operation
    bool has_long = false;

    // The text is loaded one step ahead, always exactly one load per step at an address that depends on no loaded
    // data (clamped to the last step): with loads inside branches the compiler waits for every load in flight at
    // the join, and the first rows of the qualities are in flight here (bq_issue) while the tag is decoded.
    const int last_base = ((max(n, 1) - 1) / PB) * PB;
    uint4 vcur;
    __builtin_memcpy(&vcur, cs + 16 * lane, 16);                   // himut_push_reads leaves 2 KB of slack behind the text
    BqAhead Q;
    if constexpr (WITH_BQ) bq_issue(R, r, lane, Q);
    for (int base = 0; base < n; base += PB) {
        const int nb = min(PB, n - base);
        // ---- text of this step into LDS, behind the last 32 bytes of the previous step (still in LDS)
        const uint4 v = vcur;
        uint4 pb = make_uint4(0, 0, 0, 0);
        if (lane < 2 && base > 0) pb = *reinterpret_cast<const uint4*>(txt + PB + 16 * lane);
        *reinterpret_cast<uint4*>(txt + 32 + 16 * lane) = v;
        if (lane < 2) *reinterpret_cast<uint4*>(txt + 16 * lane) = pb;
        __builtin_memcpy(&vcur, cs + min(base + PB, last_base) + 16 * lane, 16);
        __builtin_amdgcn_wave_barrier();
        // ---- operation starts in this lane's 16 bytes, four bytes at a time in the registers they came in.  A byte's
        // class comes from two 16-entry tables looked up with v_perm_b32, one by its high nibble (which row of the ASCII
        // table), one by its low nibble (which rows that column belongs to): bit 0 / 1 = an operation's first byte
        // (* + - in row 2, : = in row 3), bits 2..4 = payload (digits, upper case, lower case)
        uint32_t stf[4];                                             // 0x80 in the bytes that start an operation
        {
            const uint32_t words[4] = {v.x, v.y, v.z, v.w};
            const int nhere = min(max(nb - 16 * lane, 0), 16);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w = words[k];
                const int nv = min(max(nhere - 4 * k, 0), 4);
                const uint32_t inside = nv >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1u));   // bytes of the tag
                const uint32_t rh = __builtin_amdgcn_perm(0x10081008u, 0x06010000u, (w >> 4) & 0x07070707u);
                const uint32_t lo = w & 0x0f0f0f0fu, l7 = lo & 0x07070707u;
                const uint32_t la = __builtin_amdgcn_perm(0x1c1c1c1cu, 0x1c1c1c14u, l7), lb = __builtin_amdgcn_perm(0x08080b08u, 0x091b1c1cu, l7);
                const uint32_t rl = __builtin_amdgcn_perm(lb, la, 0x03020100u | ((lo >> 1) & 0x04040404u));
                const uint32_t r = rh & rl, ascii = ~w & 0x80808080u;
                stf[k] = ((r & 0x03030303u) + 0x7f7f7f7fu) & ascii & inside;
                const uint32_t ok = ((r & 0x1f1f1f1fu) + 0x7f7f7f7fu) & ascii;
                if (~ok & inside) bad = HIMUT_ERR_CS;                // a byte the reference's pattern has no place for
            }
        }
        const int cnt = __popc(stf[0]) + __popc(stf[1]) + __popc(stf[2]) + __popc(stf[3]);
        const int incl = wave_incl_add(cnt, lane);
        const int total = lane_val(incl, 63);
        const int off0 = have_carry ? 1 : 0;
        {
            int w = off0 + incl - cnt;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t mk = stf[k];
                while (mk) { const int i = __ffs((int)mk) - 1; mk &= mk - 1; starts[w++] = (uint16_t)(16 * lane + 4 * k + (i >> 3)); }
            }
        }
        const int m = off0 + total;
        const bool last_block = base + PB >= n;
        if (last_block && lane == 0) starts[m] = (uint16_t)(n - base);
        const int nops = last_block ? m : m - 1;
        __builtin_amdgcn_wave_barrier();
        if (base == 0 && (m == 0 || starts[0] != 0)) bad = HIMUT_ERR_CS;   // the tag does not begin with an operation
        if (__ballot(bad != 0)) break;
        // ---- operations, 64 per round
        for (int k0 = 0; k0 < nops; k0 += 64) {
            const int k = k0 + lane;
            const bool valid = k < nops;
            int s = 0, e = 0, kind = 0, len = 0, dt = 0, dq = 0, ref = 0, alt = 0;
            if (valid) {
                s = (k == 0 && have_carry) ? carry_start : base + (int)starts[k];
                e = base + (int)starts[k + 1];
                kind = (k == 0 && have_carry) ? carry_kind : (int)txt[32 + (s - base)];
                len = e - s - 1;
                if (kind == ':') {
                    if (len < 1 || len > 9) bad = HIMUT_ERR_CS;
                    else {
                        int v = 0;
                        for (int i = 0; i < len; i++) {
                            const int c = txt[32 + (s + 1 + i - base)];
                            if (!cs_is_digit(c)) bad = HIMUT_ERR_CS;
                            v = v * 10 + (c - '0');
                        }
                        dt = v; dq = v;
                    }
                } else if (kind == '*') {
                    const int a = txt[32 + (s + 1 - base)], b = (len >= 2) ? (int)txt[32 + (s + 2 - base)] : 0;
                    if (len != 2 || !(a >= 'a' && a <= 'z') || !(b >= 'a' && b <= 'z')) bad = HIMUT_ERR_CS;   // \*[a-z][a-z]
                    ref = a - 32; alt = b - 32;
                    dt = 1; dq = 1;
                } else {
                    if (len < 1) bad = HIMUT_ERR_CS;
                    if (kind == '=') { dt = len; dq = len; }
                    else if (kind == '+') dq = len;
                    else dt = len;
                }
            }
            const bool indel = valid && (kind == '+' || kind == '-');
            const bool sub = valid && kind == '*';
            // reference / query offset of every operation
            const int it = wave_incl_add(dt, lane), iq = wave_incl_add(dq, lane);
            const int tk = t + it - dt, qk = q + iq - dq;           // at the operation
            const int ta = t + it, qa = q + iq;                     // after it
            // previous operation's kind, previous indel in this round
            int prev_kind = __shfl_up(kind, 1, 64);
            if (lane == 0) prev_kind = last_kind;
            const int pidx = wave_incl_max(indel ? lane : -1, lane);
            int Pk = __shfl_up(pidx, 1, 64);
            if (lane == 0) Pk = -1;
            const int src = Pk < 0 ? 0 : Pk;
            const int p_ta = __shfl(ta, src, 64), p_qa = __shfl(qa, src, 64), p_kind = __shfl(kind, src, 64);
            // segments: an indel closes the aligned run before it; a deletion is a segment of its own
            int nseg_here = 0;
            Seg sg_run = {0, 0, 0, 0}, sg_del = {0, 0, 0, 0};
            bool run_before = false;
            if (indel) {
                run_before = Pk >= 0 ? (lane - 1 - Pk) > 0 : (aligned_open || lane > 0);
                if (run_before) {
                    const int rt0 = Pk >= 0 ? p_ta : (aligned_open ? run_t0 : t), rq0 = Pk >= 0 ? p_qa : (aligned_open ? run_q0 : q);
                    const bool rins = Pk >= 0 ? (p_kind == '+') : (aligned_open ? run_ins : (last_kind == '+'));
                    sg_run.t0 = rt0; sg_run.q0 = rq0; sg_run.len = tk - rt0; sg_run.flags = rins ? SEG_INS : 0u;
                    nseg_here++;
                }
                if (kind == '-') {
                    sg_del.t0 = tk; sg_del.q0 = qk; sg_del.len = len; sg_del.flags = SEG_DEL | (prev_kind == '+' ? SEG_INS : 0u);
                    nseg_here++;
                }
                // (an insertion straight behind an insertion -- the tokenizer splits "+a+cg" in two, cslib.py:7-10; an aligner
                //  writes one -- is one more mismatch entry at the same position and no segment of its own: the position
                //  still carries "an insertion precedes", and nothing the reference prints depends on how many)
            }
            const int iseg = wave_rank_incl(nseg_here >= 1) + wave_rank_incl(nseg_here == 2);
            if (nseg_here) {
                int w = ns + iseg - nseg_here;
                if (run_before) segs[w++] = sg_run;
                if (kind == '-') segs[w] = sg_del;
            }
            // mismatch list (cslib.py:54-62): substitutions with a non-N reference base, all indels
            int aa = 0, ra = 0, seq_nib = -1;
            if (sub) {
                aa = char2allele(alt);
                if (aa < 0) bad = HIMUT_ERR_BASE;                    // caller.py:62
                if (ref != 'N') { ra = char2allele(ref); if (ra < 0) bad = HIMUT_ERR_BASE; }   // bamlib.py:188
                // the base cs names must be the base SEQ holds (caller.py:62 takes it from cs, the pile from SEQ).  One
                // random sector of SEQ per substitution: here it is fetched beside a decode that is bound by
                // instruction issue, not by memory
                if (ref != 'N' && !bad) {
                    if (qk < 0 || qk >= qlen) bad = HIMUT_ERR_CS;
                    else seq_nib = nib_at(R.seq, M.qoff + qk);       // looked at when the round is over: the load has the round to arrive
                }
            }
            const bool ismis = indel || (sub && ref != 'N');
            const int imis = wave_rank_incl(ismis);
            if (ismis) {
                const int w = nm + imis - 1;
                mis[w] = tk + 1;
                mq[w] = sub ? (((uint32_t)qk << 5) | 16u | ((uint32_t)(ra & 3) << 2) | (uint32_t)(aa & 3)) : ((uint32_t)qk << 5);
            }
            {   // a substitution whose reference base is N: no mismatch entry, but normcounts counts its base
                const bool nsb = sub && ref == 'N';
                const unsigned long long nb = __ballot(nsb);
                if (nb) {
                    if (nsb) {
                        const int64_t k = nN + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(nb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nb, 0u));
                        mq[top - k] = ((uint32_t)qk << 5) | 8u;
                    }
                    nN += __popcll(nb);
                }
            }
            if (mark) {
                const bool mk = sub && ref != 'N';
                const unsigned long long mb = __ballot(mk);
                if (mb) {
                    if (nmark + __popcll(mb) > MARK_CAP) flush_marks();
                    if (mk) marks[nmark + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u))] = tk;
                    nmark += __popcll(mb);
                }
            }
            // identity counts (bamlib.py:47-63)
            match += (valid && (kind == ':' || kind == '=')) ? dt : 0;
            mism += sub ? 1 : (indel ? len : 0);
            if (__ballot(valid && kind == '=')) has_long = true;
            // ---- carry the round's end state
            const int nvalid = min(64, nops - k0);
            const unsigned long long ib = __ballot(indel);
            t = lane_val(ta, nvalid - 1); q = lane_val(qa, nvalid - 1);
            ns += lane_val(iseg, 63); nm += lane_val(imis, 63);
            if (ib) {
                const int L = 63 - __clzll((long long)ib);
                run_t0 = lane_val(ta, L); run_q0 = lane_val(qa, L); run_ins = lane_val(kind, L) == '+';
                aligned_open = L < nvalid - 1;
            } else if (!aligned_open) {
                // the run opens at the first operation of this round
                run_t0 = lane_val(tk, 0); run_q0 = lane_val(qk, 0); run_ins = last_kind == '+';
                aligned_open = true;
            }
            last_kind = lane_val(kind, nvalid - 1);
            if (seq_nib >= 0) {                                    // the substitutions' bases against SEQ
                const int qa = nib2allele(seq_nib);
                if (qa > 3) bad = HIMUT_ERR_BASE;
                else if (qa != aa) bad = HIMUT_ERR_CS;
            }
            if (__ballot(bad != 0)) break;
        }
        if (__ballot(bad != 0)) break;
        if (!last_block) {
            if (m > 0) {
                have_carry = true;
                carry_start = (m == 1 && off0 == 1) ? carry_start : base + (int)starts[m - 1];
                carry_kind = (m == 1 && off0 == 1) ? carry_kind : (int)txt[32 + (carry_start - base)];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (WITH_BQ) bq_finish(Q, r, lane, D.bqsum);
    // a reduction of the lanes' error codes
    if (__ballot(bad != 0)) bad = lane_val(wave_incl_max(bad, lane), 63);      // (rare; the code with the highest number)
    if (!bad) {
        // end of the tag: close the open run; an insertion at the very end is a marker segment
        if (lane == 0) {
            if (aligned_open) { Seg z = {run_t0, run_q0, t - run_t0, run_ins ? SEG_INS : 0u}; segs[ns] = z; }
            else if (last_kind == '+') { Seg z = {t, q, 0, SEG_INS}; segs[ns] = z; }
        }
        if (aligned_open || last_kind == '+') ns++;
        if (t != M.tend || q > qlen) bad = HIMUT_ERR_CS;   // cs inconsistent with CIGAR / SEQ
    }
    const long long match_all = (long long)lane_val(wave_incl_add(match, lane), 63), mism_all = (long long)lane_val(wave_incl_add(mism, lane), 63);
    // identity filter (bamlib.py:47-63, caller.py:314); the other read filters follow behind the capture
    if (!bad && match_all + mism_all == 0) bad = HIMUT_ERR_CS;     // an empty tag: ZeroDivisionError in the reference (bamlib.py:62)
    const double ident = (double)match_all / (double)(match_all + mism_all);
    const bool ident_ok = !bad && !(ident < P.p.min_sequence_identity);
    if (mark && ident_ok && nmark > 0) flush_marks();
    if (lane == 0) {
        if (bad) { set_err(err, bad); ns = 0; nm = 0; }
        uint8_t fl = 0;
        if (ident_ok) fl = RF_IDENT_OK;
        if (has_long) fl |= RF_LONGCS;
        M.nseg = ns; M.flags = fl;
        D.nseg[r] = ns;
        D.nmis[r] = nm;
        D.nnsub[r] = bad ? 0 : nN;
        D.rflag[r] = fl;
        D.meta[r] = M;
    }
}

// long-form tags ('=' with the matched bases spelled out): the letters must be the bases
// SEQ holds, because the pile takes match bases from SEQ (cslib.py:24 takes them from cs).
// Thread per flagged read; minimap2 --cs=short (what himut asks for) never gets here.
__global__ void __launch_bounds__(256) k_check_longcs(Reads R, Derived D, int* err) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n || !(D.rflag[r] & RF_LONGCS)) return;
    const uint8_t* cs = R.cs + R.cs_off[r];
    const int64_t n = R.cs_off[r + 1] - R.cs_off[r];
    int64_t q = R.qstart[r];
    const int64_t qo = R.qoff[r];
    int64_t i = 0;
    while (i < n) {
        const int c = cs[i];
        int64_t j = i + 1;
        while (j < n && !cs_is_start(cs[j])) j++;
        const int64_t len = j - i - 1;
        if (c == ':') { int64_t v = 0; for (int64_t k = i + 1; k < j; k++) v = v * 10 + (cs[k] - '0'); q += v; }
        else if (c == '*') q += 1;
        else if (c == '+') q += len;
        else if (c == '=') {
            for (int64_t k = 0; k < len; k++)
                if (upper(cs[i + 1 + k]) != nib2char(nib_at(R.seq, qo + q + k))) { set_err(err, HIMUT_ERR_CS); return; }
            q += len;
        }
        i = j;
    }
}

// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// k_read_hap: sixteen lanes per (chunk, read-in-window) pair; haplib.get_ccs_hap (haplib.py:61-83).
__global__ void __launch_bounds__(256) k_read_hap(Reads R, Derived D, Chunks C, Phase H, int* err) {
    // chunk = blockIdx.y; sixteen lanes per (chunk, read) pair, a lane per heterozygous SNP of the chunk's phase set under the
    // read: each finds its segment by a search of its own and fetches its base, instead of one thread walking up to forty
    // of them in turn
    const int64_t c = blockIdx.y;
    const int64_t kin = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;       // the pair's place in the chunk's window
    const int gl = threadIdx.x & 15;
    const int64_t p0 = C.pairoff[c];
    if (kin >= C.pairoff[c + 1] - p0) return;
    const int64_t k = p0 + kin;
    const int64_t r = C.rlo[c] + kin;
    uint8_t hap = HAP_NONE;
    const int32_t s = C.start[c], e = C.end[c];
    const int32_t ts = R.tstart[r], te = R.tend[r];
    if (!(D.rflag[r] & RF_SECONDARY) && ts < e && te > s) {
        const int32_t* hpos = H.hpos;
        const int64_t a = H.off[c], b = H.off[c + 1];
        // bisect_right of the read's start and end among the set's positions (haplib.py:68-69): counted, sixteen at a time
        int n_le_s = 0, n_le_e = 0;
        for (int64_t g = a + gl; g < b; g += 16) { const int32_t hp = hpos[g]; n_le_s += hp <= ts ? 1 : 0; n_le_e += hp <= te ? 1 : 0; }
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) { n_le_s += __shfl_xor(n_le_s, d, 16); n_le_e += __shfl_xor(n_le_e, d, 16); }
        const int64_t idx = a + n_le_s, jdx = a + n_le_e;
        if (jdx - idx >= 2) {
            bool all0 = true, all1 = true, uncovered = false;
            const Seg* segs = D.segs + seg_base(R, r);
            const int ns = D.nseg[r];
            const int64_t qo = R.qoff[r];
            for (int64_t g = idx + gl; g < jdx; g += 16) {
                const int32_t rpos = hpos[g] - 1;
                // the first segment that ends behind rpos (segments are in order and do not overlap)
                int lo = 0, hi = ns;
                while (lo < hi) { const int m = (lo + hi) >> 1; if (rpos >= segs[m].t0 + segs[m].len) lo = m + 1; else hi = m; }
                int qb = 0;  // 0: not in tpos2qbase -> KeyError
                if (lo < ns) {
                    const Seg sg = segs[lo];
                    if (rpos >= sg.t0) {
                        if (sg.flags & SEG_DEL) qb = '-';
                        else qb = nib2char(nib_at(R.seq, qo + sg.q0 + (rpos - sg.t0)));
                    }
                }
                if (qb == 0) { uncovered = true; continue; }
                int bit = '-';
                if (H.href[g] && qb == H.href[g]) bit = '0';        // haplib.py:52-57
                else if (H.halt[g] && qb == H.halt[g]) bit = '1';
                const int h0 = H.hbit[g];
                const int h1 = h0 == '0' ? '1' : (h0 == '1' ? '0' : '-');
                if (bit != h0) all0 = false;
                if (bit != h1) all1 = false;
            }
            // the pair's sixteen lanes agree (they sit in one row of the wave)
            int v0 = all0 ? 1 : 0, v1 = all1 ? 1 : 0, vu = uncovered ? 1 : 0;      // (every lane takes part in every exchange)
#pragma unroll
            for (int d = 8; d > 0; d >>= 1) {
                v0 &= __shfl_xor(v0, d, 16);
                v1 &= __shfl_xor(v1, d, 16);
                vu |= __shfl_xor(vu, d, 16);
            }
            all0 = v0 != 0; all1 = v1 != 0; uncovered = vu != 0;
            if (uncovered) { if (gl == 0) set_err(err, HIMUT_ERR_COVER); }
            else hap = all0 ? HAP_0 : (all1 ? HAP_1 : HAP_NONE);
        }
    }
    if (gl == 0) H.hap[k] = hap;
}

// ---------------------------------------------------------------------------------------
// propose_read: the proposals of one read.  Applies the read filters (caller.py:310-317) and, for every substitution that
// survives the trim and mismatch-window filters (bamlib.py:69-86,222-282) and every chunk that both contains tpos
// (caller.py:104-108,325) and fetched the read (caller.py:299), sets the (ref, alt) bit of the position in that
// chunk's mask -- the set() of caller.py:324.  The candidates are enumerated from the mask afterwards.  The proposal
// that sets a mask bit for the first time also counts it in its tile of 8192 mask cells: the scan of those counts is
// where k_mask_emit puts each tile's candidates.
//
// It is the tail of k_stream_capture's wave: the wave that has streamed a read knows the read's quality sum, the last
// thing the read filters were waiting for, and what follows is a chain of dependent look-ups (chunk hint -> chunk
// records -> mismatch entries -> atomics) that costs a wave microseconds of latency and next to no bandwidth -- as a
// kernel of its own that chain, times the number of waves a chip holds, was 0.1 ms; behind a bandwidth-bound stream
// the other waves of the CU fill the time.
constexpr int EMIT_MAXC = 4;   // chunks of one read kept in registers
constexpr int PROP_TAB_BITS = 11;   // entries of a recent-proposal table (a workgroup's, where one is kept)
constexpr int MASK_TILE_SHIFT = 13;
constexpr int MASK_TILE_CELLS = 1 << MASK_TILE_SHIFT;

// The proposals of ONE read, by the sixteen lanes of a DPP row (gl = lane in the row, gsh = the row's first lane in
// the wave); the row takes the mismatch entries e0, e0 + estride, ...  (k_stream_capture calls it with the four rows of
// the read's wave: each row works out the read's chunks for itself, the rows share the entries.)  bqs: the sum of the
// read's qualities.  s_seen: the workgroup's recent-proposal table, or null.
__device__ __forceinline__ void propose_read(const Reads& R, const Derived& D, const Chunks& C, const Phase& H, const Params& P,
                                             const int64_t r, const uint32_t bqs, const int e0, const int estride, const int gl,
                                             const int gsh, uint32_t* mask, uint32_t* tilecnt, uint8_t* ccs_flag,
                                             unsigned long long* s_seen) {
    // the (ref, alt) bit of a mask cell: 16 mask bits per position, two positions per 32-bit word
    auto propose = [&](const int64_t cell, const int bit) {
        if (s_seen) {
            const unsigned long long key = ((unsigned long long)cell << 4) | (unsigned long long)bit;
            const uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - PROP_TAB_BITS));
            if (atomicExch(&s_seen[h], key) == key) return;
        }
        const uint32_t m = (1u << bit) << ((cell & 1) ? 16 : 0);
        if (!(atomicOr(mask + (cell >> 1), m) & m)) atomicAdd(tilecnt + (cell >> MASK_TILE_SHIFT), 1u);
    };
    // cross-lane operations below stay inside the group: shuffles of width 16, the group's 16 bits of a ballot;
    // every branch around them is decided per read, i.e. uniform in the group
#define GBALLOT(P) ((uint32_t)((__ballot(P) >> gsh) & 0xffffull))
    const ReadMeta M = D.meta[r];
    const int32_t qlen = R.qlen[r];
    const int mapq = R.mapq[r];
    const int nm = D.nmis[r];
    const int32_t qid = R.qid[r];
    if (M.flags & RF_SECONDARY) return;
    const bool phase = P.p.phase != 0;
    bool live = (M.flags & RF_IDENT_OK) != 0;                             // bamlib.py:47-63, caller.py:314
    // np.mean(bq) < min_qv (bamlib.py:35, caller.py:310) in integers: for integers S, n, k the rounded quotient fl(S / n) is
    // below k exactly when S < k n (a quotient below k is below it by at least 1 / n, far more than half an ulp; rounding
    // is monotonic), and an empty query (0 / 0, not below anything) passes either way
    if (P.p.min_qv > 0 && (unsigned long long)bqs < (unsigned long long)P.p.min_qv * (unsigned long long)(uint32_t)qlen) live = false;
    if (mapq < P.p.min_mapq) live = false;
    if (!(P.p.qlen_lower_limit < qlen && qlen < P.p.qlen_upper_limit)) live = false;
    const int32_t ts = M.tstart, te = M.tend;

    // the chunks that fetched this read (start < tend and end > tstart): 15 records of the
    // sorted table around the look-up hint (and the one in front of them), one per lane, decided with one ballot
    int nc = 0;
    bool overflow = false;
    int64_t hi = 0;
    int32_t cc[EMIT_MAXC], cst[EMIT_MAXC], cen[EMIT_MAXC];
    int64_t cmo[EMIT_MAXC];
#pragma unroll
    for (int k = 0; k < EMIT_MAXC; k++) { cc[k] = -1; cst[k] = 0; cen[k] = -1; cmo[k] = 0; }
    if (live && C.n > 0) {
        const int64_t h0 = C.hint[min((int64_t)(te > 0 ? te : 0) >> CHUNK_HINT_SHIFT, C.nhint - 1)];
        const int64_t wlo = max(h0 - 8, (int64_t)0);
        const int64_t j = wlo - 1 + gl;                 // lane 0: the record in front of the window
        ChunkRec rec;
        rec.start = 0x7fffffff; rec.end = -0x7fffffff - 1; rec.idx = -1; rec.pmaxend = -0x7fffffff - 1; rec.maskoff = 0; rec.pairbase = 0;
        if (j >= 0 && j < C.n) rec = C.rec[j];
        // complete when no chunk in front of the window reaches the read and none behind it starts inside
        const bool front_ok = __shfl(rec.pmaxend, 0, 16) <= ts;
        const bool back_ok = wlo + 15 >= C.n || GBALLOT(gl >= 1 && rec.start >= te) != 0u;
        bool take = gl >= 1 && rec.idx >= 0 && rec.start < te && rec.end > ts;
        if (take && phase && H.hap[rec.pairbase + r] == HAP_NONE) take = false;     // caller.py:306-309
        const uint32_t tk = GBALLOT(take);
        nc = __popc(tk);
        if (!front_ok || !back_ok || nc > EMIT_MAXC) {
            // unusual chunk tables (deep nesting, a read across many chunks): walk the table per entry
            overflow = true;
            hi = h0;
            while (hi < C.n && C.rec[hi].start < te) hi++;
            nc = 0;
            for (int64_t jj = hi - 1; jj >= 0 && C.rec[jj].pmaxend > ts; jj--) {
                const ChunkRec q = C.rec[jj];
                if (q.end <= ts) continue;
                if (phase && H.hap[q.pairbase + r] == HAP_NONE) continue;
                nc++;
            }
        } else {
            uint32_t rest = tk;
#pragma unroll
            for (int k = 0; k < EMIT_MAXC; k++) {
                if (rest) {
                    const int src = __ffs((int)rest) - 1;
                    rest &= rest - 1;
                    cc[k] = __shfl(rec.idx, src, 16); cst[k] = __shfl(rec.start, src, 16); cen[k] = __shfl(rec.end, src, 16);
                    cmo[k] = ((int64_t)__shfl((int)(rec.maskoff >> 32), src, 16) << 32) | (uint32_t)__shfl((int)rec.maskoff, src, 16);
                }
            }
        }
        // num_ccs (caller.py:318-320): counted once it passes in any chunk that fetched it
        if (nc == 0) live = false;
        else if (gl == 0) ccs_flag[qid] = 1;
    } else live = false;

    if (!live) return;         // (the cs-vs-SEQ check of every substitution is k_parse_cs's)
    const int32_t* mis = D.mis + M.segbase;
    const uint32_t* mq = D.mq + M.segbase;
    const double trim_start = floor(P.p.min_trim * (double)qlen);        // bamlib.py:226
    const double trim_end = ceil((1.0 - P.p.min_trim) * (double)qlen);   // bamlib.py:227
    const int64_t w = P.p.mismatch_window_size;
    for (int e = e0; e < nm; e += estride) {
        const uint32_t v = mq[e];
        const int32_t tp1 = mis[e];
        const int32_t mprev = e > 0 ? mis[e - 1] : -0x7fffffff - 1;
        const int32_t mnext = e + 1 < nm ? mis[e + 1] : 0x7fffffff;
        if (!(v & 16u)) continue;                                             // substitutions only
        const int64_t q = v >> 5;
        if ((double)q < trim_start || (double)q > trim_end) continue;          // bamlib.py:231-242
        {                                                                     // bamlib.py:245-282
            int64_t qs = q - w, qe = q + w, ur, dr;
            if (qs < 0) { ur = w + qs; dr = w - qs; }
            else if (qe > qlen) { ur = w + (qe - qlen); dr = qlen - q; }
            else { ur = w; dr = w; }
            const int32_t ms = (int32_t)(tp1 - ur), me = (int32_t)(tp1 + dr);
            // bisect_right(me) - bisect_left(ms) - 1 on the sorted list; the entry itself is inside [ms, me]
            int lo = e, up = e + 1;
            if (mprev >= ms) { lo = e - 1; while (lo > 0 && mis[lo - 1] >= ms) lo--; }
            if (mnext <= me) { up = e + 2; while (up < nm && mis[up] <= me) up++; }
            if ((int64_t)(up - lo) - 1 > P.p.max_mismatch_count) continue;
        }
        const int bit = (int)(v & 15u);
        if (!overflow) {
#pragma unroll
            for (int k = 0; k < EMIT_MAXC; k++)
                if (cc[k] >= 0 && cst[k] <= tp1 && tp1 <= cen[k]) propose(cmo[k] + (tp1 - cst[k]), bit);
        } else {
            for (int64_t jj = hi - 1; jj >= 0 && C.rec[jj].pmaxend > ts; jj--) {
                const ChunkRec qr = C.rec[jj];
                if (qr.end <= ts || !(qr.start <= tp1 && tp1 <= qr.end)) continue;
                if (phase && H.hap[qr.pairbase + r] == HAP_NONE) continue;
                propose(qr.maskoff + (tp1 - qr.start), bit);
            }
        }
    }
#undef GBALLOT
}

// ---------------------------------------------------------------------------------------
// The candidate list = the set bits of the mask (16 bits per cell: one per (ref, alt)).  Cells
// with a bit are few (one in ~150 at 30x) and all of them sit at column positions, so the two
// sweeps read the position bitmap (k_parse_cs) instead of the mask and look only at the cells of
// marked positions: bits per tile of 8192 cells (32 cells per thread), then -- after a scan of
// the tile counts -- the candidates themselves, in mask order: chunk, position, then (ref, alt)
// in ASCII order.  The sort key of a candidate == the sort key of its record: (tpos, chunk, ref,
// alt), natsorted order of the reference's tuples (caller.py:622).  The emit sweep leaves the
// mask zeroed for the next run.

// where a thread of the sweeps stands in the chunk list
struct CellCursor {
    int64_t ck, cbeg, cend;   // chunk of the cell, its first cell, first cell of the next chunk
    int32_t cstart;           // the chunk's start (tpos of cell cbeg)
};

// Summary word of the mask cells [32 i, 32 i + 32): bit b set when the cell's position (rpos = tpos - 1) is a
// column position.  cur: the chunk of the tile's first cell on entry, of cell 32 i on return.
__device__ __forceinline__ uint32_t cell_summary(const Chunks& C, const uint32_t* posbits, int64_t i, int64_t ncells, CellCursor& cur) {
    const int64_t c0 = i * 32;
    if (c0 >= ncells) return 0u;
    while (c0 >= cur.cend) { cur.ck++; cur.cbeg = cur.cend; cur.cend = C.maskoff[cur.ck + 1]; cur.cstart = C.start[cur.ck]; }
    if (c0 + 32 <= cur.cend) {                       // the 32 cells lie in one chunk: 32 consecutive positions
        const int64_t rp = (int64_t)cur.cstart + (c0 - cur.cbeg) - 1;
        if (rp < 0) return posbits[0] << 1;          // cell of tpos 0 (a chunk that starts at 0): no such position
        const uint32_t lo = posbits[rp >> 5], hi = posbits[(rp >> 5) + 1];
        const int sh = (int)(rp & 31);
        return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
    }
    uint32_t w = 0;
    CellCursor t = cur;
    for (int b = 0; b < 32 && c0 + b < ncells; b++) {
        const int64_t c = c0 + b;
        while (c >= t.cend) { t.ck++; t.cbeg = t.cend; t.cend = C.maskoff[t.ck + 1]; t.cstart = C.start[t.ck]; }
        const int64_t rp = (int64_t)t.cstart + (c - t.cbeg) - 1;
        if (rp >= 0 && ((posbits[rp >> 5] >> (rp & 31)) & 1u)) w |= 1u << b;
    }
    return w;
}

// cap: capacity of cands / keys (the host may have sized them before the count was known: nothing is
// written past it, and the true count lands in *total for the host to compare with cap)
// tilecnt: the tile counts k_propose made (their scan is tileoff); zeroed here for the next run, like the mask.
__global__ void __launch_bounds__(256) k_mask_emit(const uint32_t* posbits, int64_t ncells, uint16_t* mask16, const uint32_t* tileoff,
                                                   Chunks C, Cand* cands, uint64_t* keys, int64_t cap, unsigned long long* total,
                                                   uint32_t* tilecnt) {
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = 0;
    // the chunk of the tile's first cell comes with the tile; cells may run into the next chunks
    const MaskTile mt = C.mtile[blockIdx.x];
    CellCursor cur = {mt.ck0, mt.off0, mt.off1, mt.start0};
    uint32_t w = cell_summary(C, posbits, i, ncells, cur);
    int c = 0;
    for (uint32_t x = w; x; x &= x - 1) c += __popc((uint32_t)mask16[i * 32 + (__ffs((int)x) - 1)]);
    const int incl = wave_incl_add(c, lane);
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        *total = (unsigned long long)tileoff[blockIdx.x] + (unsigned long long)(s_w[0] + s_w[1] + s_w[2] + s_w[3]);
    if (!c) return;
    int64_t slot = (int64_t)tileoff[blockIdx.x] + (incl - c);
#pragma unroll
    for (int k = 0; k < 4; k++) if (k < wv) slot += s_w[k];
    int64_t ck = cur.ck;
    int64_t cbeg = cur.cbeg, cend_ = cur.cend;
    int32_t cstart = cur.cstart;
    while (w) {
        const int b = __ffs((int)w) - 1; w &= w - 1;
        const int64_t cell = i * 32 + b;
        const uint32_t m = mask16[cell];
        if (!m) continue;
        mask16[cell] = 0;
        while (cell >= cend_) { ck++; cbeg = cend_; cend_ = C.maskoff[ck + 1]; cstart = C.start[ck]; }
        const int32_t tpos = cstart + (int32_t)(cell - cbeg);
#pragma unroll
        for (int rk = 0; rk < 16; rk++) {   // (ref, alt) in ASCII order A C G T = alleles 0 3 2 1
            const int ra = (0x1230 >> (4 * (rk >> 2))) & 15, aa = (0x1230 >> (4 * (rk & 3))) & 15;
            const int bit = (ra << 2) | aa;
            if ((m >> bit) & 1u) {
                if (slot < cap) {
                    Cand cd; cd.tpos = tpos; cd.chunk_bit = ((uint32_t)ck << 4) | (uint32_t)bit;
                    cands[slot] = cd;
                    keys[slot] = ((uint64_t)(uint32_t)tpos << 28) | ((uint64_t)ck << 4) | (uint64_t)rk;
                }
                slot++;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_window_index: per block of 256 reference positions, the range of reads that can
// cover a position of the block (reads are coordinate sorted).
__global__ void __launch_bounds__(256) k_window_index(Reads R, int64_t nblk, int32_t* winlo, int32_t* winhi) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const int32_t p0 = (int32_t)(b << WIN_SHIFT), p1 = (int32_t)((b + 1) << WIN_SHIFT);
    const int64_t hi = lower_bound(R.tstart, (int64_t)0, R.n, p1);           // reads with tstart < p1
    // a read ending exactly at p0 still belongs: a trailing insertion is counted at tend (caller.py:66-67)
    int64_t lo = lower_bound(R.prefmax_tend, (int64_t)0, hi, p0);            // running max of tend >= p0
    while (lo < hi && R.tend[lo] < p0) lo++;                                 // ... and the first read that really reaches the block
    winlo[b] = (int32_t)lo;
    winhi[b] = (int32_t)hi;
}

// 16 BAM nibbles (base j at bits 4j..4j+3) -> 16 pile cells (himut allele index, 4 = not ATGC)
__device__ __forceinline__ uint64_t nib16_to_cells(uint64_t x) {
    const uint64_t m = 0x1111111111111111ULL;
    const uint64_t n0 = x & m, n1 = (x >> 1) & m, n2 = (x >> 2) & m, n3 = (x >> 3) & m;  // A C G T one-hot bits
    const uint64_t sum = n0 + n1 + n2 + n3;
    const uint64_t inv = ((sum >> 1) | (sum >> 2) | ~sum) & m;     // not exactly one bit set
    uint64_t code = (n3 | n1) | ((n2 | n1) << 1);                  // T,C -> bit0 ; G,C -> bit1
    code = (code & ~(inv * 3)) | (inv << 2);
    return code;
}

// ---------------------------------------------------------------------------------------
// The column store.  Candidate columns live at the UNIQUE reference positions that carry
// a candidate (several chunks / alts can share one).  All positions of a 256-position block
// share one read window [lo, lo + n) (k_window_index), so a block with cnt candidate
// positions owns n * cnt slots, read-major: the slot of (read r, unique position u) is
// boff[b] + (r - lo) * cnt + (u - ufirst[b]).  Neighbouring positions of one read are
// neighbours in memory, which is what lets the dense sweep read its columns coalesced.
// One 16-bit slot per (read of the window, position), walked in fetch order:
//   bits 0-2 cell (0-3 allele A T G C, 4 base outside ATGC, 5 deletion, 7 not in the pile)
//   bit 3    an insertion precedes the position
//   bit 4    unused
//   bits 8-15 base quality
// k_stream_capture fills it while streaming every read once with coalesced loads;
// k_eval_columns consumes it, one thread per candidate.

struct BlockTab {     // one per 256 reference positions
    int32_t lo;       // first read of the window
    uint32_t ncnt;    // bits 0-21: reads in the window = slots per column; bits 22-31: candidate positions in the block
    uint32_t boff;    // slot offset of the block's first column
    uint32_t ufirst;  // unique-position rank of the block's first candidate position
};
constexpr uint32_t BT_N_MASK = (1u << 22) - 1u;

struct PosIndex {
    const uint32_t* bits;    // bit rpos set: some candidate sits at rpos
    const uint32_t* rank;    // exclusive prefix popcount per 32-bit word, nwords + 1 entries
    int64_t nwords;
    const BlockTab* bt;
    int64_t nblk;
};

// Rank of a position among the column positions = the block's first rank (BlockTab) + the set bits of the block's
// eight bitmap words in front of it (one 32-byte sector).
__device__ __forceinline__ uint32_t pos_rank_in_block(const uint32_t* bits, int32_t rpos) {
    const uint4* wp = reinterpret_cast<const uint4*>(bits + (((int64_t)rpos >> 8) << 3));
    const uint4 a = wp[0], b = wp[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const int wi = (rpos >> 5) & 7;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t m = k < wi ? 0xffffffffu : (k == wi ? ((1u << (rpos & 31)) - 1u) : 0u);
        c += (uint32_t)__popc(w[k] & m);
    }
    return c;
}
__device__ __forceinline__ uint32_t pos_rank(const PosIndex& X, int32_t rpos) {
    return X.bt[rpos >> 8].ufirst + pos_rank_in_block(X.bits, rpos);
}

// Column positions and column-store slots per 256-position block, taken straight from the bitmap and the read
// windows; their two prefix sums are each block's first rank and its slot offset.
struct BlockCount {
    const uint32_t* bits;
    const int32_t *winlo, *winhi;
    __host__ __device__ uint2 operator()(int64_t b) const {
        const uint4* wp = reinterpret_cast<const uint4*>(bits + (b << 3));
        const uint4 x = wp[0], y = wp[1];
        const uint32_t cnt = (uint32_t)(__builtin_popcount(x.x) + __builtin_popcount(x.y) + __builtin_popcount(x.z) + __builtin_popcount(x.w) +
                                        __builtin_popcount(y.x) + __builtin_popcount(y.y) + __builtin_popcount(y.z) + __builtin_popcount(y.w));
        const unsigned long long s = ((unsigned long long)cnt * (unsigned long long)(uint32_t)(winhi[b] - winlo[b]) + 15ULL) & ~15ULL;
        return make_uint2(cnt, (uint32_t)s);      // every block starts on a 32-byte boundary of the column store
    }
};
struct PlusU2 {
    __host__ __device__ uint2 operator()(const uint2& a, const uint2& b) const { return make_uint2(a.x + b.x, a.y + b.y); }
};

// ---- the same index without a library scan.  rocPRIM's scan is two launches (state initialisation, look-back scan)
// of ~14 us for a quarter of a million elements, most of it launch and drain; here the prefix sums are two plain
// kernels: per-workgroup totals, then every workgroup adds up the (at most 1024) totals in front of it and scans its
// own blocks.  A thread takes `per` consecutive blocks (1 unless the contig has more than 2^18 blocks).
__device__ __forceinline__ void block_counts(const BlockCount& F, int64_t b, uint32_t& cnt, uint32_t& nr, unsigned long long& slots) {
    const uint4* wp = reinterpret_cast<const uint4*>(F.bits + (b << 3));
    const uint4 x = wp[0], y = wp[1];
    cnt = (uint32_t)(__popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w) + __popc(y.x) + __popc(y.y) + __popc(y.z) + __popc(y.w));
    nr = (uint32_t)(F.winhi[b] - F.winlo[b]);
    slots = ((unsigned long long)cnt * (unsigned long long)nr + 15ULL) & ~15ULL;   // every block starts on a 32-byte boundary of the column store
}
// sums of (a, b) over the workgroup (256 threads), the same value in every thread
__device__ __forceinline__ void wg_sum2(uint32_t& a, unsigned long long& b, uint32_t* s_a, unsigned long long* s_b) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_a[wv] = a; s_b[wv] = b; }
    __syncthreads();
    a = s_a[0] + s_a[1] + s_a[2] + s_a[3];
    b = s_b[0] + s_b[1] + s_b[2] + s_b[3];
}
__global__ void __launch_bounds__(256) k_block_sums(BlockCount F, int64_t nblk, int per, uint4* part) {
    __shared__ uint32_t s_a[4];
    __shared__ unsigned long long s_b[4];
    const int64_t b0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * per;
    uint32_t c = 0;
    unsigned long long s = 0;
    for (int k = 0; k < per; k++)
        if (b0 + k < nblk) { uint32_t cnt, nr; unsigned long long sl; block_counts(F, b0 + k, cnt, nr, sl); c += cnt; s += sl; }
    wg_sum2(c, s, s_a, s_b);
    if (threadIdx.x == 0) part[blockIdx.x] = make_uint4(c, 0u, (uint32_t)s, (uint32_t)(s >> 32));
}
// BlockTab (first rank, slot offset) of every block; also leaves the slot offsets / counts as plain arrays (the run's
// totals).  err: HIMUT_ERR_DEPTH when the column store of the contig needs more than 2^32 slots or a window holds more
// than 2^22 reads.
__global__ void __launch_bounds__(256) k_block_table3(BlockCount F, int64_t nblk, int per, const uint4* part, BlockTab* bt,
                                                      uint32_t* blkoff, uint32_t* blkslots, int* err) {
    __shared__ uint32_t s_a[4];
    __shared__ unsigned long long s_b[4];
    // what the workgroups in front of this one hold
    uint32_t pc = 0;
    unsigned long long ps = 0;
    for (int w = threadIdx.x; w < (int)blockIdx.x; w += 256) {
        const uint4 v = part[w];
        pc += v.x; ps += (unsigned long long)v.z | ((unsigned long long)v.w << 32);
    }
    wg_sum2(pc, ps, s_a, s_b);
    // this thread's blocks, then the threads in front of it
    const int64_t b0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * per;
    uint32_t tc = 0;
    unsigned long long ts = 0;
    for (int k = 0; k < per; k++)
        if (b0 + k < nblk) { uint32_t cnt, nr; unsigned long long sl; block_counts(F, b0 + k, cnt, nr, sl); tc += cnt; ts += sl; }
    uint32_t ic = tc;
    unsigned long long is = ts;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t uc = __shfl_up(ic, d, 64);
        const unsigned long long us = __shfl_up(is, d, 64);
        if (lane >= d) { ic += uc; is += us; }
    }
    __syncthreads();
    if (lane == 63) { s_a[wv] = ic; s_b[wv] = is; }
    __syncthreads();
    uint32_t rc = pc + ic - tc;
    unsigned long long rs = ps + is - ts;
    for (int w = 0; w < wv; w++) { rc += s_a[w]; rs += s_b[w]; }
    bool deep = false;
    for (int k = 0; k < per; k++) {
        const int64_t b = b0 + k;
        if (b >= nblk) break;
        uint32_t cnt, nr;
        unsigned long long sl;
        block_counts(F, b, cnt, nr, sl);
        BlockTab t;
        t.lo = F.winlo[b]; t.ncnt = nr | (cnt << 22); t.boff = (uint32_t)rs; t.ufirst = rc;
        bt[b] = t;
        blkoff[b] = (uint32_t)rs; blkslots[b] = (uint32_t)sl;
        if (nr > BT_N_MASK || rs + sl > 0xffffffffULL) deep = true;
        rc += cnt; rs += sl;
    }
    if (deep) set_err(err, HIMUT_ERR_DEPTH);
}

// exclusive prefix sums of a few thousand counts by one workgroup of 1024 threads (the candidate counts of the mask
// tiles), 16 k at a time through LDS: coalesced loads, eight or sixteen consecutive counts per thread, coalesced stores
constexpr int SCAN_SMALL_CHUNK = 16384;
__global__ void __launch_bounds__(1024) k_scan_small(const uint32_t* in, uint32_t* out, int n) {
    __shared__ uint32_t s_v[SCAN_SMALL_CHUNK];
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    for (int c0 = 0; c0 < n; c0 += SCAN_SMALL_CHUNK) {
        const int m = min(SCAN_SMALL_CHUNK, n - c0);
        for (int i = threadIdx.x; i < m; i += 1024) s_v[i] = in[c0 + i];
        __syncthreads();
        const int per = (m + 1023) / 1024, i0 = (int)threadIdx.x * per;
        uint32_t t = 0;
        for (int k = 0; k < per; k++) if (i0 + k < m) t += s_v[i0 + k];
        uint32_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t run = s_carry + inc - t;
        for (int w = 0; w < wv; w++) run += s_w[w];
        for (int k = 0; k < per; k++) if (i0 + k < m) { const uint32_t v = s_v[i0 + k]; s_v[i0 + k] = run; run += v; }
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 1024) out[c0 + i] = s_v[i];
        if (threadIdx.x == 1023) s_carry = run;          // (the last thread's running sum is the chunk's total + carry)
        __syncthreads();
    }
}

// The kernels behind the candidate count take it from device memory (*n_dev, clamped to the capacity the
// grid was sized for): the host need not have seen it yet.
__device__ __forceinline__ int64_t dev_count(const unsigned long long* n_dev, int64_t cap) {
    const unsigned long long n = *n_dev;
    return n < (unsigned long long)cap ? (int64_t)n : cap;
}




struct CaptureArgs {
    Reads R;
    Derived D;
    PosIndex X;
    uint16_t* colstore;
    int64_t nslots;
    int64_t r_begin, r_end;      // reads of this launch
    uint32_t* bqsum;             // call path: per read, the sum of the qualities of the whole query (bamlib.py:34-36)
    const int* err;              // a device error raised by an earlier kernel of the run: nothing is captured then
    // call path: the read's proposals follow its stream (propose_read); mask null = none
    Chunks C;
    Phase H;
    Params P;
    uint32_t *mask, *tilecnt;
    uint8_t* ccs_flag;
};

constexpr int CLQ = 512;    // marked positions a wave keeps at a time: 16-bit offsets from the list's first word
constexpr int CSG = 63;     // segments per LDS window (+ 1 sentinel = one per lane)
constexpr int CWQ = 2048;   // query bases per window
constexpr int CPL = 256;    // reference positions per lane when the bitmap is enumerated (eight words): 64 lanes take 16 k
                            // positions per pass, a read's span in one or two
constexpr int CPD = 2;      // windows in flight (register sets, at most 4); the loop is unrolled by it

// One wave per read; a software pipeline over windows of 2048 query bases.
//
// Window k is [c_k, c_k + 2048) of the query.  The segment list turns it into a reference
// range [tA_k, tB_k) (a deleted position goes with the base that follows it); those ranges
// tile the read's span.  The read's marked positions (the bitmap of column positions under its span) are listed once,
// 16 k positions a turn, as 16-bit offsets in LDS (refill).  Every turn of the loop
//   * stores the window's bytes -- 2 KB of qualities + 1 KB of packed bases, loaded two
//     turns earlier (CPD) with 16-byte coalesced loads -- in LDS,
//   * issues the same four loads for window k + CPD (three for the bytes, one for the block table under it): the
//     addresses depend only on the segment list, never on loaded data, so nothing in the loop waits on a
//     load younger than CPD turns and the count of loads in flight is the same every turn,
//   * takes the list's next entries below tB_k, one per lane: segment from a cursor that walks the list once, query
//     offset, base and quality out of the window in LDS, slot from the block table.
// Every byte of the read is fetched exactly once; stores return nothing.
#ifndef HIMUT_CAP_WAVES
#define HIMUT_CAP_WAVES 7
#endif
__device__ __forceinline__ void capture_wave(const CaptureArgs& A) {
    __shared__ __align__(16) uint8_t s_bq[4][CWQ];
    __shared__ __align__(16) uint8_t s_sq[4][CWQ / 2];
    __shared__ __align__(16) int4 s_seg[4][CSG + 1];
    __shared__ __align__(16) uint32_t s_bt[4][64];        // 16 block-table entries
    __shared__ uint16_t s_list[4][CLQ];
    const int tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6);
    const Reads& R = A.R;
    const PosIndex& X = A.X;
    const int64_t r64 = A.r_begin + (int64_t)blockIdx.x * 4 + wv;
    if (r64 >= A.r_end || uni(*A.err)) return;
    const int32_t r = (int32_t)r64;
    const ReadMeta Mv = A.D.meta[r];
    const int32_t qlen = uni(R.qlen[r]);
    if (uni(Mv.flags) & RF_SECONDARY) return;
    if (uni(Mv.nseg) <= 0) {                 // nothing aligned (an empty cs tag): only the quality sum is wanted
        {
            if (A.bqsum) {
                const uint8_t* q = R.bq + uni(Mv.qoff);
                uint32_t sum = 0;
                for (int32_t i = lane; i < qlen; i += 64) sum += q[i];
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
                if (lane == 0) A.bqsum[r] = sum;
                if (A.mask) propose_read(R, A.D, A.C, A.H, A.P, r, uni(sum), lane, 64, lane & 15, lane & 48, A.mask, A.tilecnt, A.ccs_flag, nullptr);
            }
        }
        return;
    }
    const Seg* gsegs = A.D.segs + uni(Mv.segbase);
    const int ns = uni(Mv.nseg);
    const int64_t qo = uni(Mv.qoff);
    const int32_t tstart = uni(Mv.tstart), tend = uni(Mv.tend);
    uint8_t* wbq = s_bq[wv];
    uint8_t* wsq = s_sq[wv];
    int4* lseg = s_seg[wv];
    uint32_t* lbt = s_bt[wv];
    uint16_t* list = s_list[wv];
    const bool all_lds = ns <= CSG;     // the whole segment list fits the LDS window

    // the windows start at query offset 0, soft clip included: the quality mean of the read filter is over the
    // whole query (a window in front of the aligned part has an empty reference range)
    const int32_t c0 = 0;
    const int nwin = ((max(qlen, 1) - 1 - c0) >> 11) + 1;
    uint32_t qsum = 0;     // this lane's share of the sum of the read's qualities
    const int32_t qpad = (qlen + 31) & ~31;

    // a load inside a rarely taken branch is waited for inside that branch, so that the join behind it
    // does not have to wait for every load in flight
#define CAP_LANDED4(V) asm volatile("" : "+v"(V.x), "+v"(V.y), "+v"(V.z), "+v"(V.w))
#define CAP_LANDED2(V) asm volatile("" : "+v"(V.x), "+v"(V.y))
    // the four loads of one window (addresses clamped into the read, so every lane always loads)
#define CAP_ISSUE(BA, BB, SQ, BT, K, TA) do { \
        const int32_t _c = c0 + min((K), nwin - 1) * CWQ; \
        const int32_t _qa = min(_c + lane * 16, qpad - 16), _qb = min(_c + 1024 + lane * 16, qpad - 16), _qs = min(_c + lane * 32, qpad - 32); \
        BA = *reinterpret_cast<const uint4*>(R.bq + qo + _qa); \
        BB = *reinterpret_cast<const uint4*>(R.bq + qo + _qb); \
        SQ = *reinterpret_cast<const uint4*>(R.seq + ((qo + _qs) >> 1)); \
        BT = reinterpret_cast<const uint32_t*>(X.bt + min((int64_t)((TA) >> 8) + (lane >> 2), X.nblk - 1))[lane & 3]; \
    } while (0)
#define CAP_SEGWIN() do { if (lane <= nw) { int4 z = make_int4(0x7fffffff, 0, 0, 0); if (jb + lane < ns) z = *reinterpret_cast<const int4*>(gsegs + jb + lane); \
        lseg[lane] = z; } } while (0)

    // segment window [jb, jb + nw) plus a sentinel that carries the next segment's start
    int jb = 0, nw = min(ns, CSG);
    CAP_SEGWIN();
    uint4 ba0, bb0, sq0, ba1, bb1, sq1, ba2, bb2, sq2, ba3, bb3, sq3;
    uint32_t bt0r, bt1r, bt2r, bt3r;
    CAP_ISSUE(ba0, bb0, sq0, bt0r, 0, tstart);
    // rank of the first column position at or behind tstart: the block's first rank + the bits in front of tstart
    const uint32_t rk0 = uni(X.bt[min((int64_t)(tstart >> 8), X.nblk - 1)].ufirst) +
                         uni(pos_rank_in_block(X.bits, (int32_t)min((int64_t)tstart, (X.nblk << 8) - 1) & ~31));
    __builtin_amdgcn_wave_barrier();

    // end of window kk in reference coordinates: the first position whose query offset is >= c
    int jq = 0;   // no segment before jq ends behind the last boundary asked for
    auto window_end = [&](int kk) -> int32_t {
        if (kk >= nwin - 1) return tend + 1;
        const int32_t c = c0 + (kk + 1) * CWQ;
        while (true) {
            int4 sg = make_int4(tend + 1, 0x7ffffff0, 0, (int)SEG_DEL);        // behind the list: ends nowhere
            if (jq + lane < ns) {
                if (all_lds) sg = lseg[jq + lane];
                else { sg = *reinterpret_cast<const int4*>(gsegs + jq + lane); CAP_LANDED4(sg); }
            }
            const bool isrun = !((uint32_t)sg.w & SEG_DEL) && sg.z > 0;
            const int32_t qend = isrun ? sg.y + sg.z : sg.y + 1;
            const unsigned long long bal = __ballot(qend > c);
            if (bal) {
                const int src = __ffsll((long long)bal) - 1;
                const int32_t t0 = lane_val(sg.x, src), q0 = lane_val(sg.y, src);
                const bool inside = lane_val((int)(isrun && sg.y < c), src) != 0;
                jq += src;
                return inside ? t0 + (c - q0) : t0;
            }
            jq += 64;
        }
    };
    // ta[i] = first reference position of window k + i
    int32_t ta[5];
    ta[0] = tstart;
    ta[1] = window_end(0);
    CAP_ISSUE(ba1, bb1, sq1, bt1r, 1, ta[1]);
    ta[2] = window_end(1);
    if constexpr (CPD > 2) { CAP_ISSUE(ba2, bb2, sq2, bt2r, 2, ta[2]); ta[3] = window_end(2); }
    if constexpr (CPD > 3) { CAP_ISSUE(ba3, bb3, sq3, bt3r, 3, ta[3]); ta[4] = window_end(3); }
    // ---- the read's marked positions, taken from the bitmap 16 k positions at a time (a read's span in one or two turns)
    //      into a list the windows then consume in order: list[i] = position - lbase, entries i_list .. n_list - 1 not yet
    //      used, all the marked positions of [.., covB) are in it or done with; u_list = the rank of list[0] among the
    //      contig's marked positions (the column's number: the block's first rank + what lies in front of tstart in it)
    int32_t covB = tstart, lbase = tstart & ~31;
    uint32_t u_list = rk0;
    int n_list = 0, i_list = 0;
    bool first_fill = true;
    auto refill = [&]() {
        u_list += (uint32_t)n_list; n_list = 0; i_list = 0;
        lbase = covB & ~31;
        for (int pass = 0; pass < 4 && covB <= tend; pass++) {          // (offsets stay below 2^16)
            const int32_t s0 = covB & ~31;
            const int64_t w0 = ((int64_t)s0 >> 5) + 8 * lane;
            uint32_t w[8];
            {
                uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
                if (w0 + 8 <= X.nwords + 2) {                               // (nwords + 2 words are allocated)
                    __builtin_memcpy(&a, X.bits + w0, 16);
                    __builtin_memcpy(&b, X.bits + w0 + 4, 16);
                } else {
                    uint32_t t[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) t[i] = (w0 + i < X.nwords) ? X.bits[w0 + i] : 0u;
                    a = make_uint4(t[0], t[1], t[2], t[3]); b = make_uint4(t[4], t[5], t[6], t[7]);
                }
                w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
            }
            if (first_fill) {       // what the block's rank does not count yet: the bits of tstart's word in front of it
                u_list += (uint32_t)__popc((uint32_t)lane_val((int)w[0], 0) & ((1u << (tstart & 31)) - 1u));
                first_fill = false;
            }
            // positions in front of covB (they lie in the first word of lane 0) and behind tend (the read's last turn) are not
            // the read's
            const int32_t p_lane = s0 + CPL * lane;
            if (lane == 0) w[0] &= 0xffffffffu << (covB - s0);
            const int32_t keep = tend + 1 - p_lane;                        // this lane's leading positions that are the read's
            if (__ballot(keep < CPL)) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int32_t ki = keep - 32 * i;
                    if (ki < 32) w[i] &= ki <= 0 ? 0u : ((1u << ki) - 1u);
                }
            }
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) cnt += __popc(w[i]);
            const int incl = wave_incl_add(cnt, lane);
            // the lanes whose positions still fit the list, from lane 0 on
            const unsigned long long fitb = __ballot(n_list + incl <= CLQ);
            const int L = fitb == ~0ULL ? 64 : (int)__builtin_ctzll(~fitb);
            if (lane < L) {
                int at = n_list + incl - cnt;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    uint32_t b = w[i];
                    const int32_t off = p_lane + 32 * i - lbase;
                    while (b) { const int k = __ffs((int)b) - 1; b &= b - 1; list[at++] = (uint16_t)(off + k); }
                }
            }
            if (L > 0) n_list += lane_val(incl, L - 1);
            covB = s0 + CPL * L;
            if (L < 64) break;                                              // the list is full up to here
        }
        if (covB > tend) covB = tend + 1;
        __builtin_amdgcn_wave_barrier();
    };
    int jcur = 0;          // segment cursor of the candidate walk

    auto window = [&](uint4& ba, uint4& bb, uint4& sq, uint32_t& btv, const int k) {
        const int32_t tA = ta[0], tB = ta[1];
        const int32_t cq = c0 + k * CWQ;
        // ---- this window's bytes and tables -> LDS; its registers take window k + 2
        *reinterpret_cast<uint4*>(wbq + lane * 16) = ba;
        *reinterpret_cast<uint4*>(wbq + 1024 + lane * 16) = bb;
        *reinterpret_cast<uint4*>(wsq + lane * 16) = sq;
        lbt[lane] = btv;                                   // lane = entry * 4 + field
        {
            // np.mean(bq_int_lst) of caller.py:310 / bamlib.py:34-36: the bytes are here anyway.  A window behind the
            // read (k >= nwin) holds a reloaded copy of the last one; bytes behind qlen are masked off
            if (k < nwin) {
                if (cq + CWQ <= qlen) {
                    qsum = __builtin_amdgcn_sad_u8(ba.x, 0u, qsum); qsum = __builtin_amdgcn_sad_u8(ba.y, 0u, qsum);
                    qsum = __builtin_amdgcn_sad_u8(ba.z, 0u, qsum); qsum = __builtin_amdgcn_sad_u8(ba.w, 0u, qsum);
                    qsum = __builtin_amdgcn_sad_u8(bb.x, 0u, qsum); qsum = __builtin_amdgcn_sad_u8(bb.y, 0u, qsum);
                    qsum = __builtin_amdgcn_sad_u8(bb.z, 0u, qsum); qsum = __builtin_amdgcn_sad_u8(bb.w, 0u, qsum);
                } else {
                    const uint32_t wa[4] = {ba.x, ba.y, ba.z, ba.w}, wb_[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int ra = qlen - (cq + lane * 16 + 4 * i), rb = ra - 1024;   // bytes of the word inside the read
                        uint32_t xa = wa[i], xb = wb_[i];
                        if (ra < 4) xa = ra <= 0 ? 0u : (xa & (0xffffffffu >> (8 * (4 - ra))));
                        if (rb < 4) xb = rb <= 0 ? 0u : (xb & (0xffffffffu >> (8 * (4 - rb))));
                        qsum = __builtin_amdgcn_sad_u8(xa, 0u, qsum);
                        qsum = __builtin_amdgcn_sad_u8(xb, 0u, qsum);
                    }
                }
            }
        }
        CAP_ISSUE(ba, bb, sq, btv, k + CPD, ta[CPD]);
        const int32_t ta_next = window_end(k + CPD);
        __builtin_amdgcn_wave_barrier();
        const int64_t btb = tA >> 8;
        // one candidate per lane: segment, cell value, slot, store
        auto batch = [&](const bool act, const uint32_t rpos, const uint32_t u, const int32_t rlast) {
            // the segment that holds rpos = the last one that starts at or before it.  Candidates come
            // in reference order, so one cursor walks the list once per read (wave-uniform LDS reads)
            int4 sg = make_int4(0, 0, 0, 0);   // t0, q0, len, flags
            {
                int j = jcur;
                while (j < ns) {
                    if (j < jb || j >= jb + nw) {    // lists longer than the LDS window: reload it from j
                        jb = j; nw = min(ns - jb, CSG);
                        __builtin_amdgcn_wave_barrier();
                        CAP_SEGWIN();
                        __builtin_amdgcn_wave_barrier();
                    }
                    const int4 t = lseg[j - jb];
                    if (uni(t.x) > rlast) break;
                    if (act && (int32_t)rpos >= t.x) sg = t;
                    j++;
                }
                jcur = max(j - 1, jcur);
            }
            if (act) {
                const int32_t d = (int32_t)rpos - sg.x;
                const uint32_t insb = (d == 0 && ((uint32_t)sg.w & SEG_INS)) ? CELL_INS : 0u;
                bool store = false;
                uint32_t val = 0;
                if ((uint32_t)sg.w & SEG_DEL) { if (d < sg.z) { store = true; val = CELL_DEL | insb; } }
                else if (sg.z == 0) { if (d == 0 && insb) { store = true; val = CELL_EMPTY | CELL_INS; } }
                else if (d < sg.z) {
                    store = true;
                    const int32_t q = sg.y + d;
                    const int32_t o = (q - cq) & (CWQ - 1);       // inside the window by construction
                    const uint32_t qv = wbq[o], sb = wsq[o >> 1];
                    const int nib = (q & 1) ? (int)(sb & 15u) : (int)(sb >> 4);
                    val = insb | (uint32_t)nib2allele(nib) | (qv << 8);
                }
                if (store) {
                    const int64_t bi = (int64_t)(rpos >> 8) - btb;
                    uint4 t;
                    if (bi >= 0 && bi < 16) t = *reinterpret_cast<const uint4*>(lbt + 4 * bi);
                    else { t = *reinterpret_cast<const uint4*>(X.bt + min((int64_t)(rpos >> 8), X.nblk - 1)); CAP_LANDED4(t); }
                    // BlockTab: x = lo, y = n | cnt << 22, z = boff, w = ufirst
                    const int64_t slot = (int64_t)t.z + (int64_t)(r - (int32_t)t.x) * (int64_t)(t.y >> 22) + (int64_t)(u - t.w);
                    if ((uint64_t)slot < (uint64_t)A.nslots) A.colstore[slot] = (uint16_t)val;
                }
            }
        };
        // ---- the marked positions of [tA, tB): the next entries of the list, a lane each (a window holds about twenty);
        //      the list is filled again where the window reaches beyond what it covers
        while (true) {
            const int32_t hiP = min(tB, covB);
            int32_t e = 0x7fffffff;
            if (i_list + lane < n_list) e = lbase + (int32_t)list[i_list + lane];
            const bool act = e < hiP;
            const int bn = (int)__popcll(__ballot(act));                     // (the list is in order: the first bn lanes)
            if (bn) {
                batch(act, (uint32_t)e, u_list + (uint32_t)(i_list + lane), lane_val(e, bn - 1));
                i_list += bn;
            }
            if (bn == 64) continue;
            if (tB <= covB || covB > tend) break;
            __builtin_amdgcn_wave_barrier();
            refill();
        }
#pragma unroll
        for (int i = 0; i < CPD; i++) ta[i] = ta[i + 1];
        ta[CPD] = ta_next;
    };

    // CPD windows per trip, each with its own registers; the last ones of a trip may lie behind the
    // read (empty range): it still issues its loads, so the number in flight never depends on the path
    for (int k = 0; k < nwin; k += CPD) {
        window(ba0, bb0, sq0, bt0r, k);
        window(ba1, bb1, sq1, bt1r, k + 1);
        if constexpr (CPD > 2) window(ba2, bb2, sq2, bt2r, k + 2);
        if constexpr (CPD > 3) window(ba3, bb3, sq3, bt3r, k + 3);
    }
    {
        if (A.bqsum) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) qsum += __shfl_down(qsum, d, 64);
            if (lane == 0) A.bqsum[r] = qsum;
            // a read with a base outside ATGC somewhere (k_flag_bases): KeyError in the reference if the base is aligned and
            // some chunk fetches the read (caller.py:57,299)
            if (R.nonacgt && uni((int)R.nonacgt[r])) {
                int64_t lo_ = 0, hi_ = A.C.n;
                while (lo_ < hi_) { const int64_t m_ = (lo_ + hi_) >> 1; if (A.C.rec[m_].start < tend) lo_ = m_ + 1; else hi_ = m_; }
                if (lo_ > 0 && A.C.rec[lo_ - 1].pmaxend > tstart && __ballot(!aligned_bases_ok(R, gsegs, ns, qo, lane, 64)))
                    set_err(const_cast<int*>(A.err), HIMUT_ERR_BASE);
            }
            // the read filters have their last input: this read's proposals
            if (A.mask) propose_read(R, A.D, A.C, A.H, A.P, r, uni(qsum), lane, 64, lane & 15, lane & 48, A.mask, A.tilecnt, A.ccs_flag, nullptr);
        }
    }
#undef CAP_ISSUE
#undef CAP_SEGWIN
#undef CAP_LANDED4
#undef CAP_LANDED2
}

__global__ void __launch_bounds__(256, HIMUT_CAP_WAVES) k_stream_capture(CaptureArgs A) { capture_wave(A); }

// ---------------------------------------------------------------------------------------
// k_eval_columns: one THREAD per candidate column: walks the column's slots in fetch
// order -- allele counts, BQ sums (caller.py:44-72, bamlib.py:181-219), the ordered
// likelihood sums of gtlib.py:72-96 -- then genotypes and filters (caller.py:324-621).

struct SiteSets {
    const uint64_t* pon; int64_t npon;
    const uint64_t* com; int64_t ncom;
    const uint32_t* posbits;  // bit tpos set when either set holds a key at that position
    int64_t nposbits;         // number of valid bits
};

__device__ __forceinline__ bool key_in(const uint64_t* a, int64_t n, uint64_t x) {
    int64_t k = lower_bound(a, (int64_t)0, n, x);
    return k < n && a[k] == x;
}

// genotype list of gtlib.py:9 in himut allele indices (A0 T1 G2 C3):
// AA TA CA GA TT CT GT CC GC GG
#define HIMUT_GT_B1(g) ((0x2232312310ULL >> (4 * (g))) & 15)
#define HIMUT_GT_B2(g) ((0x2331110000ULL >> (4 * (g))) & 15)

__device__ __forceinline__ int gt_state_of(int b1, int b2, int ref) {  // gtlib.py:23-38
    if (b1 == b2 && b2 == ref) return 0;
    if ((b1 == ref) != (b2 == ref)) return 1;
    if (b1 != b2) return 2;
    return 3;
}

struct EvalArgs {
    Params P;
    SiteSets S;
    const GtLut* lut;
    const Cand* cands;       // sorted by record key
    int64_t ncand;           // capacity the grid covers; the count itself is *ncand_dev
    const unsigned long long* ncand_dev;
    Reads R;
    Derived D;
    Chunks C;
    Phase H;
    PosIndex X;
    const uint16_t* colstore;
    int64_t nslots;          // capacity of colstore
    himut_record* recs;      // record j belongs to candidate j
    int* err;
};

#ifndef HIMUT_EVAL_WAVES
#define HIMUT_EVAL_WAVES 4
#endif
#ifndef HIMUT_EVAL_BATCH
#define HIMUT_EVAL_BATCH 8
#endif
template <bool PHASE>
__global__ void __launch_bounds__(256, HIMUT_EVAL_WAVES) k_eval_columns(EvalArgs A) {
    __shared__ double s_lut[3 * 256];
    __shared__ double s_prior[4];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * 256; i += 256) s_lut[i] = A.lut->t[i >> 8][i & 255];
    if (tid < 4) s_prior[tid] = A.lut->prior[tid];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + tid;
    if (j >= dev_count(A.ncand_dev, A.ncand) || *A.err) return;
    const Cand cd = A.cands[j];
    const int32_t tpos = cd.tpos;
    const int chunk = (int)(cd.chunk_bit >> 4);
    const int ref = (int)((cd.chunk_bit >> 2) & 3), alt = (int)(cd.chunk_bit & 3);
    const int32_t rpos = tpos - 1;
    const uint32_t u = pos_rank(A.X, rpos);
    const BlockTab bt = A.X.bt[rpos >> 8];
    const uint32_t n = bt.ncnt & BT_N_MASK, stride = bt.ncnt >> 22;
    const int32_t lo = bt.lo;
    const uint16_t* col = A.colstore + ((int64_t)bt.boff + (int64_t)(u - bt.ufirst));
    // a column past the capacity: the store was sized before the slot count was known and the host runs again
    if (n && (int64_t)bt.boff + (int64_t)(u - bt.ufirst) + (int64_t)(n - 1) * (int64_t)stride >= A.nslots) return;
    const int min_bq = A.P.p.min_bq;
    const int32_t cs_ = A.C.start[chunk];
    // Only next to the chunk start can a read lie in the pile of the position without having
    // been fetched by the chunk (caller.py:299: fetch needs reference_end > chunk_start).
    const bool edge = rpos <= cs_;

    uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
    uint32_t bqs[4] = {0, 0, 0, 0};
    double S[3][4];
#pragma unroll
    for (int b = 0; b < 4; b++) { S[0][b] = 0.0; S[1][b] = 0.0; S[2][b] = 0.0; }
    uint32_t ref_count = 0, alt_count = 0, alt_hi = 0, h0_ref = 0, h1_ref = 0, som0 = 0, som1 = 0;
    double R0 = 0.0, R1 = 0.0, R2 = 0.0, A0 = 0.0, A1 = 0.0, A2 = 0.0;    // the reference / alternative allele's three sums
    uint32_t Rq = 0, Aq = 0;                                              // and their quality sums
    int bad = 0;
    constexpr int EB = HIMUT_EVAL_BATCH;
    for (uint32_t i0 = 0; i0 < n; i0 += EB) {    // EB slots in flight: their addresses do not depend on each other
      uint32_t vv[EB];
#pragma unroll
      for (int k = 0; k < EB; k++) vv[k] = (i0 + k < n) ? (uint32_t)col[(int64_t)(i0 + k) * stride] : (uint32_t)CELL_EMPTY;
#pragma unroll
      for (int k = 0; k < EB; k++) {
        const uint32_t i = i0 + k;
        const uint32_t v = vv[k];
        const uint32_t cell = v & 7u;
        if ((v & 15u) == CELL_EMPTY) continue;
        int32_t tend = 0;
        if (edge || PHASE) {
            tend = A.R.tend[lo + (int32_t)i];
            if (edge && !(tend > cs_)) continue;   // not fetched by this chunk
        }
        if (v & CELL_INS) cnt[4]++;
        if (cell < 4) {
            const uint32_t q = v >> 8;
            if (q == 0) bad |= 1 << HIMUT_ERR_BQ0;                     // gtlib.py:64
            const double vh = s_lut[q], vt = s_lut[256 + q], ve = s_lut[512 + q];
            // sums in fetch order per allele (gtlib.py:84-93).  Nearly every cell is the candidate's reference or
            // alternative allele: those two have accumulators of their own; the four-way update runs for the rest
            if ((int)cell == ref) {
                ref_count++; Rq += q;
                R0 = R0 + vh; R1 = R1 + vt; R2 = R2 + ve;
            } else if ((int)cell == alt) {
                alt_count++; Aq += q; if ((int)q >= min_bq) alt_hi++;      // caller.py:160-171
                A0 = A0 + vh; A1 = A1 + vt; A2 = A2 + ve;
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if ((int)cell == b) {
                        cnt[b]++; bqs[b] += q;
                        S[0][b] = S[0][b] + vh;
                        S[1][b] = S[1][b] + vt;
                        S[2][b] = S[2][b] + ve;
                    }
                }
            }
            if (PHASE) {
                uint32_t hp = HAP_NONE;   // the vote counts reads that also cover rpos + 1 (caller.py:558)
                if (tend > rpos + 1 && ((int)cell == ref || (int)cell == alt))
                    hp = A.H.hap[A.C.pairoff[chunk] + ((int64_t)(lo + (int32_t)i) - A.C.rlo[chunk])];
                if ((int)cell == ref) { if (hp == HAP_0) h0_ref++; else if (hp == HAP_1) h1_ref++; }   // caller.py:562-564
                if ((int)cell == alt) { if (hp == HAP_0) som0 = 1; else if (hp == HAP_1) som1 = 1; }   // caller.py:565-569
            }
        } else if (cell == CELL_DEL) cnt[5]++;
        else if (cell == CELL_OTHER) bad |= 1 << HIMUT_ERR_BASE;       // caller.py:57
      }
    }

    // the two alleles' accumulators back into their places (ref != alt; nothing else touched those entries)
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if (b == ref) { cnt[b] = ref_count; bqs[b] = Rq; S[0][b] = R0; S[1][b] = R1; S[2][b] = R2; }
        if (b == alt) { cnt[b] = alt_count; bqs[b] = Aq; S[0][b] = A0; S[1][b] = A1; S[2][b] = A2; }
    }
    // ten PLs, gtlib.py:72-110; np.argsort with the scalar insertion sort: ties -> lower index (gtlib.py:113-119)
    double best = 0.0, second = 0.0;
    int ibest = 0;
#pragma unroll
    for (int g = 0; g < 10; g++) {
        const int b1 = (int)HIMUT_GT_B1(g), b2 = (int)HIMUT_GT_B2(g);
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double term;
            if (b1 == b2 && b == b1) term = S[0][b];
            else if (b1 != b2 && (b == b1 || b == b2)) term = S[1][b];
            else term = S[2][b];
            acc = acc + term;
        }
        acc = acc + s_prior[gt_state_of(b1, b2, ref)];
        const double pl = -10.0 * acc;
        if (g == 0) { best = pl; ibest = 0; }
        else if (pl < best) { second = best; best = pl; ibest = g; }
        else if (g == 1 || pl < second) second = pl;
    }
    const double gqf = second - best;
    const int gq = gqf < 99.0 ? (int)gqf : 99;
    int g0 = (int)HIMUT_GT_B1(ibest), g1 = (int)HIMUT_GT_B2(ibest);
    const int state = gt_state_of(g0, g1, ref);
    if (g0 != ref && ((g0 == ref) + (g1 == ref)) == 1) { int tmp = g0; g0 = g1; g1 = tmp; }  // gtlib.py:133-134

    const uint32_t depth = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[5];  // bamlib.py:213-219
    bool germ;  // caller.py:111-147
    if (state == 1) germ = (g0 == ref && g1 == alt);
    else if (state == 2) {
        uint32_t c0 = 0, c1 = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) { if (g0 == b) c0 = cnt[b]; if (g1 == b) c1 = cnt[b]; }
        germ = ((cnt[0] + cnt[1] + cnt[2] + cnt[3]) == (c0 + c1)) && (alt == g0 || alt == g1);
    } else if (state == 3) germ = (ref_count == 0) && (g0 == alt && g1 == alt);
    else germ = (alt == g0);

    int status = 255;
    int32_t ps = -1;
    uint32_t flags = 0;
    if (germ) flags = REC_GERM;
    else if (state == 1) status = HIMUT_ST_HET;
    else if (state == 2) status = HIMUT_ST_HETALT;
    else if (state == 3) status = HIMUT_ST_HOMALT;
    else if (cnt[5] != 0 || cnt[4] != 0) status = HIMUT_ST_INDEL;
    else {
        const uint64_t key = ((uint64_t)(uint32_t)tpos << 4) | ((uint64_t)ref << 2) | (uint64_t)alt;
        if (gq < A.P.p.min_gq) status = HIMUT_ST_LOWGQ;
        else if (alt_hi == 0) status = HIMUT_ST_LOWBQ;
        else {
            const SiteSets& St = A.S;
            const bool site_maybe = (int64_t)tpos < St.nposbits && ((St.posbits[tpos >> 5] >> (tpos & 31)) & 1u);
            if (site_maybe && key_in(St.pon, St.npon, key)) status = HIMUT_ST_PON;
            else if (site_maybe && key_in(St.com, St.ncom, key)) status = HIMUT_ST_COMSNP;
            else if (!((int64_t)ref_count >= A.P.p.min_ref_count && (int64_t)alt_count >= A.P.p.min_alt_count)) status = HIMUT_ST_LOWDEPTH;
            else if ((int64_t)depth > A.P.p.md_threshold) status = HIMUT_ST_HIGHDEPTH;
            else if (PHASE) {  // caller.py:552-603 (unique query names: the voters are the pile's own rows)
                if (!A.P.unique_qnames) {
                    // Alignments that share a query name (supplementary alignments: the README asks for -F 0x900, the
                    // reference does not insist).  The vote goes by NAME: every alignment over the base behind the
                    // candidate (fetch(chrom, tpos, tpos + 1), caller.py:558) whose name is among the names of the
                    // column's reference-allele reads votes with ITS haplotype, else one whose name is among the
                    // alternative-allele reads' names gives its haplotype to the candidate.
                    h0_ref = h1_ref = som0 = som1 = 0;
                    const int32_t p1 = rpos + 1;
                    const BlockTab b1 = A.X.bt[min((int64_t)(p1 >> 8), A.X.nblk - 1)];
                    const uint32_t n1 = b1.ncnt & BT_N_MASK;
                    for (uint32_t jj = 0; jj < n1; jj++) {
                        const int64_t jr = (int64_t)b1.lo + jj;
                        if (!(A.R.tstart[jr] <= p1 && A.R.tend[jr] > p1) || (A.D.rflag[jr] & RF_SECONDARY)) continue;
                        const int32_t qj = A.R.qid[jr];
                        bool in_wt = false, in_alt = false;
                        for (uint32_t i = 0; i < n; i++) {
                            const uint32_t v = (uint32_t)col[(int64_t)i * stride];
                            const int cv = (int)(v & 7u);
                            if ((v & 15u) == CELL_EMPTY || (cv != ref && cv != alt)) continue;
                            if (edge && !(A.R.tend[lo + (int32_t)i] > cs_)) continue;      // not fetched by this chunk
                            if (A.R.qid[lo + (int32_t)i] != qj) continue;
                            if (cv == ref) in_wt = true; else in_alt = true;
                        }
                        if (!in_wt && !in_alt) continue;
                        uint32_t hp = HAP_NONE;       // an alignment the chunk did not fetch spans none of its hetSNPs
                        if (jr >= A.C.rlo[chunk] && jr < A.C.rhi[chunk]) hp = A.H.hap[A.C.pairoff[chunk] + (jr - A.C.rlo[chunk])];
                        if (in_wt) { if (hp == HAP_0) h0_ref++; else if (hp == HAP_1) h1_ref++; }
                        else { if (hp == HAP_0) som0 = 1; else if (hp == HAP_1) som1 = 1; }
                    }
                }
                if ((int64_t)h0_ref >= A.P.p.min_hap_count && (int64_t)h1_ref >= A.P.p.min_hap_count && (som0 + som1) == 1) {
                    status = HIMUT_ST_PASS; ps = cs_;
                } else status = HIMUT_ST_UNPHASED;
            } else status = HIMUT_ST_PASS;
        }
    }
    uint4 w0, w1, w2, w3;
    w0.x = (uint32_t)tpos; w0.y = (uint32_t)chunk; w0.z = (uint32_t)ps; w0.w = (uint32_t)gq;
    w1.x = (uint32_t)allele2char(ref) | ((uint32_t)allele2char(alt) << 8) | ((uint32_t)allele2char(g0) << 16) | ((uint32_t)allele2char(g1) << 24);
    w1.y = (uint32_t)(status & 255) | ((uint32_t)state << 8) | (flags << 16);
    w1.z = cnt[0]; w1.w = cnt[1];
    w2.x = cnt[2]; w2.y = cnt[3]; w2.z = cnt[4]; w2.w = cnt[5];
    w3.x = bqs[0]; w3.y = bqs[1]; w3.z = bqs[2]; w3.w = bqs[3];
    uint4* dst = reinterpret_cast<uint4*>(A.recs + j);
    dst[0] = w0; dst[1] = w1; dst[2] = w2; dst[3] = w3;
    if (bad) atomicOr(A.err, bad);
}

// ---------------------------------------------------------------------------------------
// k_pile_dense: counts + BQ sums at every position of the given chunks, through
// LDS-staged base / BQ tiles (one workgroup per PD_TP positions).

struct TileInfo {
    int32_t chunk;
    int32_t p0;     // first rpos of the tile
    int32_t npos;   // positions in the tile
    int32_t nwin;   // reads in the candidate window [lo, lo + nwin)
    int64_t lo;
    int64_t outbase;  // index of the tile's first position in the output arrays
};

template <int TP>
__global__ void __launch_bounds__(256) k_tile_index(Reads R, Chunks C, const int64_t* tileoff, int64_t n_tiles, TileInfo* out) {
    const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= n_tiles) return;
    const int64_t c = upper_bound(tileoff, (int64_t)0, C.n + 1, tile) - 1;
    const int32_t cs_ = C.start[c], ce_ = C.end[c];
    const int64_t local = (tile - tileoff[c]) * TP;
    TileInfo t;
    t.chunk = (int32_t)c;
    t.p0 = (int32_t)(cs_ - 1 + local);
    const int32_t p1 = (int32_t)min((int64_t)t.p0 + TP, (int64_t)ce_);  // last rpos of the chunk is end - 1
    t.npos = p1 - t.p0;
    const int64_t rlo = C.rlo[c], rhi = C.rhi[c];
    const int64_t hi = lower_bound(R.tstart, rlo, rhi, p1);                // reads with tstart < p1
    const int64_t lo = lower_bound(R.prefmax_tend, rlo, hi, t.p0);         // running max of tend >= p0
    t.lo = lo;
    t.nwin = (int32_t)(hi - lo);
    t.outbase = C.maskoff[c] + local;
    out[tile] = t;
}

struct DenseArgs {
    Reads R;
    Derived D;
    Chunks C;
    const TileInfo* tiles;
    int64_t n_tiles;
    uint32_t* counts;  // [position][6]
    uint32_t* bqsum;   // [position][4]
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

// One piece = the part of one gapless segment (or deletion) of a read that falls
// into the tile.  x0/x1 are tile-local positions, qa the query offset of x0.
struct Piece {
    int32_t x0, x1;
    int32_t qa;
    uint32_t flags;  // SEG_DEL, SEG_INS (insertion in front of position x0)
};
constexpr int MAXP = 4;

template <int TP, int RB, int NT>
__global__ void __launch_bounds__(NT) k_pile_dense(DenseArgs A) {
    constexpr int PPL = TP / 64;        // positions per lane when a wave stages one row
    constexpr int NW = NT / 64;
    constexpr int DPT = TP / NT;        // positions per thread in the column pass
    static_assert(PPL == 8, "tile width must be 512");
    __shared__ __align__(16) uint8_t s_bq[RB * TP];
    __shared__ __align__(16) uint8_t s_cell[RB * TP / 2];
    __shared__ __align__(16) Piece s_piece[RB * MAXP];
    __shared__ int64_t s_rowqo[RB];
    __shared__ int s_rowread[RB];
    __shared__ uint8_t s_rownp[RB];
    __shared__ uint16_t s_rows[NT];
    __shared__ int s_wcnt[NW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Reads& R = A.R;
    const Derived& D = A.D;
    const TileInfo T = A.tiles[xcd_remap(blockIdx.x, A.n_tiles)];
    const int32_t p0 = T.p0, p1 = T.p0 + T.npos;
    const int32_t cs_ = A.C.start[T.chunk], ce_ = A.C.end[T.chunk];

    uint32_t dcnt[DPT][6];
    uint32_t dbq[DPT][4];
#pragma unroll
    for (int d = 0; d < DPT; d++) {
#pragma unroll
        for (int k = 0; k < 6; k++) dcnt[d][k] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) dbq[d][k] = 0;
    }
    int bad = 0;

    for (int64_t base = T.lo; base < T.lo + T.nwin; base += NT) {
        // ---- rows of the tile, in fetch order (fetch rule of the chunk: caller.py:299)
        const int64_t r = base + tid;
        bool ok = false;
        if (r < T.lo + T.nwin) {
            const int32_t te = R.tend[r];
            ok = !(D.rflag[r] & RF_SECONDARY) && te >= p0 && te > cs_ && R.tstart[r] < ce_;
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, nrows = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) { if (k < wave) woff += s_wcnt[k]; nrows += s_wcnt[k]; }
        if (ok) s_rows[woff + __popcll(bal & ((1ULL << lane) - 1ULL))] = (uint16_t)tid;
        __syncthreads();

        for (int b0 = 0; b0 < nrows; b0 += RB) {
            const int nb = min(RB, nrows - b0);
            // ---- row setup: one thread per row turns the read's segments into tile pieces
            if (tid < nb) {
                const int64_t rr = base + s_rows[b0 + tid];
                const int ns = D.nseg[rr];
                const Seg* segs = D.segs + seg_base(R, rr);
                s_rowqo[tid] = R.qoff[rr];
                s_rowread[tid] = (int)(rr - T.lo);
                int np = 0;
                auto add_seg = [&](const Seg& sg) {
                    const int32_t eend = sg.t0 + ((sg.flags & SEG_INS) ? max(sg.len, 1) : sg.len);
                    if (sg.t0 < p1 && eend > p0) {
                        if (np < MAXP) {
                            Piece pc;
                            pc.x0 = max(sg.t0, p0) - p0;
                            pc.x1 = max(min(sg.t0 + sg.len, p1) - p0, pc.x0);
                            pc.qa = sg.q0 + (p0 + pc.x0 - sg.t0);
                            pc.flags = (sg.flags & SEG_DEL) | (((sg.flags & SEG_INS) && sg.t0 >= p0) ? SEG_INS : 0u);
                            s_piece[tid * MAXP + np] = pc;
                        }
                        np++;
                    }
                };
                if (ns > 0) {
                    // the first eight segments with independent loads (one memory latency), the rest in a loop
                    Seg first[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) first[j] = segs[min(j, ns - 1)];
#pragma unroll
                    for (int j = 0; j < 8; j++) if (j < ns) add_seg(first[j]);
                    for (int j = 8; j < ns; j++) {
                        const Seg sg = segs[j];
                        if (sg.t0 >= p1) break;
                        add_seg(sg);
                    }
                }
                s_rownp[tid] = (uint8_t)min(np, 255);
            }
            __syncthreads();
            // ---- staging: one wave per row; every lane assembles its PPL positions in registers
            for (int i = wave; i < nb; i += NW) {
                const int gx = lane * PPL;
                uint64_t cell = 0x7777777777777777ULL;  // CELL_EMPTY everywhere (low PPL nibbles used)
                uint64_t bqw = 0;
                const int np = s_rownp[i];
                const int64_t qo = s_rowqo[i];
                if (np <= MAXP) {
                    for (int k = 0; k < np; k++) {
                        const Piece pc = s_piece[i * MAXP + k];
                        const int a = max(pc.x0, gx) - gx, b = min(pc.x1, gx + PPL) - gx;
                        if ((pc.flags & SEG_INS) && pc.x0 >= gx && pc.x0 < gx + PPL) cell |= 8ULL << (4 * (pc.x0 - gx));
                        const bool cov = a < b;
                        const int aa = cov ? a : 0, bb = cov ? b : 0;
                        const uint64_t nm = ((1ULL << (4 * bb)) - 1ULL) & ~((1ULL << (4 * aa)) - 1ULL);  // nibbles [a, b)
                        if (pc.flags & SEG_DEL) {
                            cell = (cell & ~(nm & 0x7777777777777777ULL)) | (nm & 0x5555555555555555ULL);
                        } else {
                            // loads are issued unconditionally (clamped address) so that they go out together
                            const int64_t o = qo + pc.qa + (cov ? (gx + a - pc.x0) : 0);
                            const uint8_t* sp = R.seq + (o >> 1);
                            uint32_t w32;
                            __builtin_memcpy(&w32, sp, 4);
                            const uint32_t extra = sp[4];
                            uint64_t l;
                            __builtin_memcpy(&l, R.bq + o, 8);
                            uint64_t w = w32;
                            w = ((w & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((w >> 4) & 0x0f0f0f0f0f0f0f0fULL);
                            if (o & 1) w = (w >> 4) | ((uint64_t)(extra >> 4) << 28);
                            const uint64_t codes = nib16_to_cells(w) << (4 * aa);
                            if (codes & nm & 0x4444444444444444ULL) bad |= 1 << HIMUT_ERR_BASE;  // caller.py:57
                            cell = (cell & ~(nm & 0x7777777777777777ULL)) | (codes & nm);
                            const uint64_t bm = ((bb >= 8) ? ~0ULL : ((1ULL << (8 * bb)) - 1ULL)) & ~((1ULL << (8 * aa)) - 1ULL);
                            bqw |= (l << (8 * aa)) & bm;
                        }
                    }
                } else {
                    // rare: more than MAXP pieces in one tile -> position-by-position from the segment list
                    const int64_t rr = T.lo + s_rowread[i];
                    const Seg* segs = D.segs + seg_base(R, rr);
                    const int ns = D.nseg[rr];
                    for (int j = 0; j < PPL; j++) {
                        const int32_t pp = p0 + gx + j;
                        for (int q = 0; q < ns; q++) {
                            const Seg sg = segs[q];
                            if (sg.t0 > pp) break;
                            if (pp == sg.t0 && (sg.flags & SEG_INS)) cell |= 8ULL << (4 * j);
                            if (pp < sg.t0 + sg.len) {
                                uint64_t code;
                                if (sg.flags & SEG_DEL) code = CELL_DEL;
                                else {
                                    const int64_t o = qo + sg.q0 + (pp - sg.t0);
                                    code = (uint64_t)nib2allele(nib_at(R.seq, o));
                                    bqw |= (uint64_t)R.bq[o] << (8 * j);
                                }
                                cell = (cell & ~(7ULL << (4 * j))) | (code << (4 * j));
                            }
                        }
                    }
                }
                *reinterpret_cast<uint32_t*>(&s_cell[i * (TP / 2) + lane * 4]) = (uint32_t)cell;
                *reinterpret_cast<uint64_t*>(&s_bq[i * TP + lane * 8]) = bqw;
            }
            __syncthreads();
            // ---- column pass: thread = position, rows in fetch order
#pragma unroll
            for (int d = 0; d < DPT; d++) {
                const int x = tid + d * NT;
                if (x < T.npos) {
                    for (int i = 0; i < nb; i++) {
                        const uint32_t cb = (s_cell[i * (TP / 2) + (x >> 1)] >> (4 * (x & 1))) & 15;
                        if (cb == CELL_EMPTY) continue;
                        if (cb & CELL_INS) dcnt[d][4]++;
                        const int a = cb & 7;
                        if (a < 4) {
                            const uint32_t q = s_bq[i * TP + x];
#pragma unroll
                            for (int b = 0; b < 4; b++) if (a == b) { dcnt[d][b]++; dbq[d][b] += q; }
                        } else if (a == CELL_DEL) dcnt[d][5]++;
                        else if (a == CELL_OTHER) bad |= 1 << HIMUT_ERR_BASE;
                    }
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 0; d < DPT; d++) {
        const int x = tid + d * NT;
        if (x < T.npos) {
            const int64_t o = T.outbase + x;
#pragma unroll
            for (int k = 0; k < 6; k++) A.counts[o * 6 + k] = dcnt[d][k];
#pragma unroll
            for (int k = 0; k < 4; k++) A.bqsum[o * 4 + k] = dbq[d][k];
        }
    }
    if (bad) atomicOr(A.err, bad);
}

// ---------------------------------------------------------------------------------------
// finalisation

// som_seen across chunks (caller.py:244,347; bamlib.py:77): a candidate whose tpos was already added by an EARLIER
// chunk is never proposed again.  Whether record i is suppressed: the records of one tpos are neighbours, ordered by
// chunk; a chunk's records are suppressed when a chunk in front of it kept a non-germline candidate there.  Nearly
// every record is alone at its tpos (two key loads); the others replay their group from its head.
__device__ __forceinline__ bool seen_in_earlier_chunk(const himut_record* recs, const uint64_t* keys, const uint32_t* vals, int64_t i,
                                                      int64_t n) {
    const uint64_t tp = keys[i] >> 28;
    const bool first = i == 0 || (keys[i - 1] >> 28) != tp;
    if (first) return false;                       // the group's first chunk (or a record alone) has nothing in front
    const uint64_t mych = (keys[i] >> 4) & 0xffffff;
    int64_t j = i;
    while (j > 0 && (keys[j - 1] >> 28) == tp) j--;
    bool seen = false;
    while (j < n && (keys[j] >> 28) == tp) {
        const uint64_t ch = (keys[j] >> 4) & 0xffffff;
        if (ch == mych) break;
        bool nongerm = false;
        int64_t k = j;
        while (k < n && (keys[k] >> 28) == tp && ((keys[k] >> 4) & 0xffffff) == ch) {
            if (!(recs[vals ? vals[k] : (uint32_t)k].flags & REC_GERM)) nongerm = true;
            k++;
        }
        if (nongerm) seen = true;
        j = k;
    }
    return seen;
}

// counters (caller.py:625-641) + output flags in sorted order
// emit[] is written for the whole capacity (zeros past the count), so that its scan can run over the capacity.
// Counters: one ballot per counter and wave, one shared-memory add per wave, one global add per block.
// Counter 0 (num_ccs, caller.py:318-320) = the reads the proposals flagged: ccs[0 .. nreads).  Slot 15 of a
// workgroup's partial counters: how many of its records are emitted (k_run_totals turns those into offsets).
__global__ void __launch_bounds__(256) k_finalize_flags(himut_record* recs, const uint64_t* keys, const uint32_t* vals,
                                                        const unsigned long long* n_dev, int64_t cap, uint32_t* emit,
                                                        uint32_t* logpart, const uint8_t* ccs, int64_t nreads) {
    __shared__ unsigned int s_log[16];
    if (threadIdx.x < 16) s_log[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    {
        unsigned int f = 0;
        for (int64_t j = i; j < nreads; j += (int64_t)gridDim.x * 256) f += ccs[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) f += __shfl_xor(f, d, 64);
        if ((threadIdx.x & 63) == 0 && f) atomicAdd(&s_log[0], f);
    }
    const int64_t n = dev_count(n_dev, cap);
    if (i >= n && i < cap) emit[i] = 0;
    int slot = -1, slot2 = -1;     // the counters this record adds to (besides num_sbs, slot 1)
    bool counted = false, emitted = false;
    if (i < n) {
        himut_record& rec = recs[vals ? vals[i] : (uint32_t)i];
        uint32_t e = 0;
        const bool suppressed = seen_in_earlier_chunk(recs, keys, vals, i, n);
        if (!suppressed) {
            counted = true;  // num_sbs
            if (rec.flags & REC_GERM) {
                if (rec.gt_state == 1) slot = 2;
                else if (rec.gt_state == 2) slot = 3;
                else if (rec.gt_state == 3) slot = 4;
            } else {
                const int st = rec.status;
                if (st == HIMUT_ST_HET || st == HIMUT_ST_HETALT || st == HIMUT_ST_HOMALT) slot = 5;
                else if (st == HIMUT_ST_INDEL) slot = 7;
                else {
                    slot = 6;  // num_homref_sbs
                    if (st == HIMUT_ST_LOWGQ) slot2 = 8;
                    else if (st == HIMUT_ST_LOWBQ) slot2 = 9;
                    else if (st == HIMUT_ST_PON) slot2 = 10;
                    else if (st == HIMUT_ST_COMSNP) slot2 = 11;
                    else if (st == HIMUT_ST_HIGHDEPTH) slot2 = 12;
                    else if (st == HIMUT_ST_LOWDEPTH) slot2 = 13;
                    else slot2 = 14;  // num_som: PASS and Unphased (caller.py:553,605)
                }
                e = 1;
                if (st == HIMUT_ST_HETALT && i > 0) {
                    // set(): a HetAltSite tuple printed as ref / "a1,a2" is identical for every alt of the column
                    const uint64_t m = ~(uint64_t)3;
                    if ((keys[i - 1] & m) == (keys[i] & m)) {
                        const himut_record& prev = recs[vals ? vals[i - 1] : (uint32_t)(i - 1)];
                        // (the neighbour is of this record's tpos and chunk: suppressed or not like this one, i.e. not)
                        if (prev.status == HIMUT_ST_HETALT && !(prev.flags & REC_GERM)) { e = 0; rec.flags |= REC_DUP; }
                    }
                }
            }
        }
        emit[i] = e;
        emitted = e != 0;
    }
    const int lane = threadIdx.x & 63;
    {
        const unsigned long long b = __ballot(counted);
        if (lane == 0 && b) atomicAdd(&s_log[1], (unsigned int)__popcll(b));
        const unsigned long long be = __ballot(emitted);
        if (lane == 0 && be) atomicAdd(&s_log[15], (unsigned int)__popcll(be));
    }
    if (__ballot(slot >= 0 || slot2 >= 0)) {
#pragma unroll
        for (int k = 2; k < 15; k++) {
            const unsigned long long b = __ballot(slot == k || slot2 == k);
            if (lane == 0 && b) atomicAdd(&s_log[k], (unsigned int)__popcll(b));
        }
    }
    __syncthreads();
    // per-workgroup partial counters (one global atomic per workgroup and counter would be ~10^4 atomics on two cache
    // lines, which cost more than the rest of this kernel): the last workgroup adds them up
    if (threadIdx.x < 16) logpart[(int64_t)blockIdx.x * 16 + threadIdx.x] = s_log[threadIdx.x];
}

// The emitted records, in order, to the front of `out`: where a workgroup's records begin comes from k_run_totals
// (wgoff), the place inside the workgroup from a ballot per wave.
__global__ void __launch_bounds__(256) k_compact(const himut_record* recs, const uint32_t* vals, const uint32_t* emit,
                                                 const uint32_t* wgoff, const unsigned long long* n_dev, int64_t cap, himut_record* out) {
    __shared__ uint32_t s_w[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool e = i < dev_count(n_dev, cap) && emit[i] != 0;
    const unsigned long long b = __ballot(e);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_w[wv] = (uint32_t)__popcll(b);
    __syncthreads();
    if (!e) return;
    uint32_t pos = wgoff[blockIdx.x] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; w++) pos += s_w[w];
    const uint4* src = reinterpret_cast<const uint4*>(recs + (vals ? vals[i] : (uint32_t)i));
    uint4 a = src[0], bb = src[1], c = src[2], d = src[3];
    bb.y &= 0xff00ffffu;  // flags byte (offset 22) -> 0
    uint4* dst = reinterpret_cast<uint4*>(out + pos);
    dst[0] = a; dst[1] = bb; dst[2] = c; dst[3] = d;
}

__global__ void __launch_bounds__(256) k_count_flags(const uint8_t* flags, int64_t n, unsigned long long* out) {
    unsigned int local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) local += flags[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, (unsigned long long)local);
}

// the totals the host reads once at the end of a run: the 15 counters (sums of k_finalize_flags' per-workgroup
// partials), records out and where each workgroup's emitted records begin (the prefix sums of slot 15 of the
// partials), column slots (the end of the last block)
__global__ void __launch_bounds__(1024) k_run_totals(int64_t cap, const uint32_t* blkoff, const uint32_t* blkslots, int64_t nblk,
                                                     unsigned long long* nrec, unsigned long long* nslots, const uint32_t* logpart,
                                                     int64_t nparts, unsigned long long* log, uint32_t* wgoff) {
    __shared__ unsigned long long s_log[16];
    __shared__ uint32_t s_w[16];
    if (threadIdx.x < 16) s_log[threadIdx.x] = 0;
    __syncthreads();
    // thread = (row phase, quarter of the 16 counters): 256 rows of partials per pass, 16-byte loads, a few passes
    const int q = threadIdx.x & 3;
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
    for (int64_t b = threadIdx.x >> 2; b < nparts; b += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(logpart + b * 16 + q * 4);
        a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
    }
    // lanes with the same quarter: xor-shuffles over the row-phase bits of the lane index
#pragma unroll
    for (int d = 4; d < 64; d <<= 1) {
        a0 += __shfl_xor(a0, d, 64); a1 += __shfl_xor(a1, d, 64); a2 += __shfl_xor(a2, d, 64); a3 += __shfl_xor(a3, d, 64);
    }
    if ((threadIdx.x & 63) < 4) {
        atomicAdd(&s_log[q * 4 + 0], a0); atomicAdd(&s_log[q * 4 + 1], a1);
        atomicAdd(&s_log[q * 4 + 2], a2); atomicAdd(&s_log[q * 4 + 3], a3);
    }
    // exclusive prefix sums of the workgroups' emitted-record counts
    const int per = (int)((nparts + 1023) / 1024);
    const int64_t p0 = (int64_t)threadIdx.x * per;
    uint32_t t = 0;
    for (int k = 0; k < per; k++) if (p0 + k < nparts) t += logpart[(p0 + k) * 16 + 15];
    uint32_t inc = t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t run = inc - t;
    for (int w = 0; w < wv; w++) run += s_w[w];
    for (int k = 0; k < per; k++) if (p0 + k < nparts) { wgoff[p0 + k] = run; run += logpart[(p0 + k) * 16 + 15]; }
    if (threadIdx.x < 15) log[threadIdx.x] = s_log[threadIdx.x];
    if (threadIdx.x == 0) {
        *nrec = cap > 0 ? s_log[15] : 0ull;
        *nslots = nblk > 0 ? (unsigned long long)blkoff[nblk - 1] + blkslots[nblk - 1] : 0ull;
    }
}

}  // namespace himut
