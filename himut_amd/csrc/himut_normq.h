// The position sweep of normcounts.get_callable_tricounts (src/himut/normcounts.py:113-140,315-402) in two kernels.
//
//   k_norm_plan   a wave per four consecutive tiles of 256 positions of a chunk, a LANE per read of the window index under them:
//                 which gapless pieces of which reads lie over each tile, written as a list of ITEMS in read order (where a piece's
//                 bases sit in the arrays, its first position and length, whether it is a deletion, whether an insertion
//                 precedes it, the read's haplotype).  Everything here is a chain of dependent look-ups -- window index,
//                 read header, segment starts, segment -- and every chain is short and independent of every other, so the
//                 chip hides them behind each other.
//   k_norm_quad   a wave per 256 positions again, FOUR consecutive positions per lane (columns P0 .. P0 + 3, P0 = base +
//                 4 * lane).  It reads its items with one load and goes through them in order.  An item that spans all
//                 256 positions -- nine in ten -- is three loads at a scalar base plus a fixed per-lane offset: a dword of
//                 four qualities (a 256-byte row per wave), a dword that holds the four packed bases, sixteen bits of the
//                 callable bit array.  The reference allele's three ordered fp64 sums of the lane's four columns are four
//                 independent chains (nq_pair: scheduled by hand); a cell of another allele -- one in a thousand -- sends
//                 the lanes that hold one down a side path that adds to a small POOL of accumulators in LDS (sixty-four of
//                 a wave's 256 columns may own one: nine doubles and four counts each).  An item that covers a part of the
//                 positions rides in the same batches of four under a per-lane mask of cells.  At the end of a column: nothing but the
//                 reference allele in it makes the ten genotype sums four numbers and the kernel classifies the position
//                 itself; the others go to k_norm_dirty's list (himut_norm.h).  A wave whose pool runs out, or whose
//                 positions have more items than the plan holds, leaves them to k_norm_tile through a list of tiles.
// Same counts as k_norm_tile, bit for bit.
#pragma once

#include "himut_norm.h"

namespace himut {

#ifndef HIMUT_NQ_NB
#define HIMUT_NQ_NB 4            // pieces whose loads are issued together
#endif
#ifndef HIMUT_NQ_OCC
#define HIMUT_NQ_OCC 4           // waves per SIMD asked of the register allocator
#endif
#ifndef HIMUT_NQ_Q
#define HIMUT_NQ_Q 2             // workgroups per XCD class and chunk at least (a contig of few chunks gets more: do_normcounts)
#endif
constexpr int NQ_WAVES = 4;
constexpr int NQ_COLS = 256;                       // positions per wave
constexpr int NQ_WG_COLS = NQ_WAVES * NQ_COLS;     // positions per workgroup and step
constexpr int NQ_SLOTS = 64;                       // pool of other-allele accumulators per wave
constexpr int NQ_Q = HIMUT_NQ_Q;

typedef const __attribute__((address_space(1))) uint8_t* nq_g8;
// loads at a 64-bit base (wave-uniform where the caller keeps it so) plus a 32-bit per-lane offset; no alignment assumed
__device__ __forceinline__ uint32_t nq_ld32(uint64_t base, uint32_t off) {
    return *reinterpret_cast<const __attribute__((address_space(1), aligned(1))) uint32_t*>(reinterpret_cast<nq_g8>(base) + off);
}
__device__ __forceinline__ uint32_t nq_ld16(uint64_t base, uint32_t off) {
    return *reinterpret_cast<const __attribute__((address_space(1), aligned(1))) uint16_t*>(reinterpret_cast<nq_g8>(base) + off);
}
__device__ __forceinline__ uint32_t nq_spread4(uint32_t b) {           // bits 0..3 -> bit 0 of bytes 0..3
    return ((b & 15u) * 0x00204081u) & 0x01010101u;
}
__device__ __forceinline__ uint32_t nq_zero_bytes(uint32_t w) {        // non-zero iff one of the four bytes is zero
    return (w - 0x01010101u) & ~w & 0x80808080u;
}
// a lane's cells (bits 0..3: positions P0 .. P0 + 3) of a piece of len positions from tlo on
__device__ __forceinline__ uint32_t nq_cells(int32_t P0, int32_t tlo, int32_t len) {
    const int32_t jlo = min(max(tlo - P0, 0), 4), jhi = min(max(tlo + len - P0, 0), 4);
    return jhi > jlo ? (((1u << (jhi - jlo)) - 1u) << jlo) : 0u;
}
__device__ __forceinline__ uint32_t nq_nibbles(uint32_t c4) {          // cell j -> the nibble at bits 12 - 4 j
    return (c4 & 1u ? 0xf000u : 0u) | (c4 & 2u ? 0x0f00u : 0u) | (c4 & 4u ? 0x00f0u : 0u) | (c4 & 8u ? 0x000fu : 0u);
}
__device__ __forceinline__ uint64_t nq_lane64(int64_t v, int l) {
    return ((uint64_t)(uint32_t)lane_val((int)(v >> 32), l) << 32) | (uint32_t)lane_val((int)v, l);
}

// What the sweep wants to know about a reference position, worked out once per himut_set_reference (k_ref_codes): bits 0-1
// the allele (A0 T1 G2 C3) of the letter with its case folded, bit 2 "that letter is one of ACGT", bit 3 "the letter is an
// upper-case ACGT" (normcounts.py:320: only those positions are classified), bits 4-8 the pyrimidine trinucleotide bin of
// NORM_TRIBINS, bit 9 "the bin is valid" (the position and both neighbours inside the string, all three upper-case ACGT).
constexpr uint32_t NQR_FOLD_OK = 4, NQR_CLS = 8, NQR_BIN_SHIFT = 4, NQR_BIN_OK = 512;
__global__ void __launch_bounds__(256) k_ref_codes(const uint8_t* seq, int64_t len, uint16_t* out, int64_t n_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t code = 0;
        if (i < len) {
            const int c = seq[i], cu = c & 0xdf;
            const int al = cu == 'A' ? 0 : cu == 'T' ? 1 : cu == 'G' ? 2 : cu == 'C' ? 3 : -1;
            if (al >= 0) code |= (uint32_t)al | NQR_FOLD_OK;
            if (char2allele(c) >= 0) code |= NQR_CLS;
            if (i - 1 >= 0 && i + 2 <= len) {
                int t0 = seq[i - 1], t1 = c, t2 = seq[i + 1];
                if (t1 == 'A' || t1 == 'G') {
                    const int a0 = t2, a2 = t0;
                    t0 = a0 == 'A' ? 'T' : a0 == 'T' ? 'A' : a0 == 'G' ? 'C' : a0 == 'C' ? 'G' : 'N';
                    t1 = t1 == 'A' ? 'T' : 'C';
                    t2 = a2 == 'A' ? 'T' : a2 == 'T' ? 'A' : a2 == 'G' ? 'C' : a2 == 'C' ? 'G' : 'N';
                }
                auto acgt = [](int x) { return x == 'A' ? 0 : x == 'C' ? 1 : x == 'G' ? 2 : x == 'T' ? 3 : -1; };
                const int i0 = acgt(t0), i2 = acgt(t2);
                if (i0 >= 0 && i2 >= 0 && (t1 == 'C' || t1 == 'T'))
                    code |= NQR_BIN_OK | ((uint32_t)(i0 * 8 + (t1 == 'T' ? 4 : 0) + i2) << NQR_BIN_SHIFT);
            }
        }
        out[i] = (uint16_t)code;
    }
}

// per-wave pool of accumulators for the alleles that are not the reference's
struct NqPool {
    double S[9][NQ_SLOTS];      // [table * 3 + slot of the allele][pool slot]
    uint32_t cnt[4][NQ_SLOTS];  // [allele][pool slot]
    uint32_t n;                 // slots handed out
    uint32_t pad;
};

// The update of two neighbouring columns (J, J + 1) with one read's cells, scheduled by hand: the compiler, short of
// registers, waited for each of the twelve table values of an item by itself.  xi: a nibble of zeros (bits 12 - 4j ..) = the
// cell is the reference allele's; qv: the qualities (byte j); lut: LDS address of the three tables (257 doubles each, entry
// 256 = +0.0: what a cell adds that is not the reference allele's -- the sums stay bit for bit what they were).  The six
// values are asked for together and added as they arrive.
template <int J>
__device__ __forceinline__ void nq_pair(double& r0a, double& r1a, double& r2a, uint32_t& na, double& r0b, double& r1b,
                                        double& r2b, uint32_t& nb, uint32_t qv, uint32_t xi, uint32_t lut) {
    uint32_t a0, a1, t0;
    double d0, d1, d2, d3, d4, d5;
    asm volatile(
        "v_and_b32 %[t0], %[ma], %[xi]\n\t"
        "v_bfe_u32 %[a0], %[qv], %[sa], 8\n\t"
        "v_cmp_eq_u32 vcc, 0, %[t0]\n\t"
        "v_cndmask_b32 %[a0], %[c256], %[a0], vcc\n\t"
        "v_addc_co_u32 %[na], vcc, 0, %[na], vcc\n\t"
        "v_lshl_add_u32 %[a0], %[a0], 3, %[lut]\n\t"
        "ds_read_b64 %[d0], %[a0]\n\t"
        "ds_read_b64 %[d1], %[a0] offset:2056\n\t"
        "ds_read_b64 %[d2], %[a0] offset:4112\n\t"
        "v_and_b32 %[t0], %[mb], %[xi]\n\t"
        "v_bfe_u32 %[a1], %[qv], %[sb], 8\n\t"
        "v_cmp_eq_u32 vcc, 0, %[t0]\n\t"
        "v_cndmask_b32 %[a1], %[c256], %[a1], vcc\n\t"
        "v_addc_co_u32 %[nb], vcc, 0, %[nb], vcc\n\t"
        "v_lshl_add_u32 %[a1], %[a1], 3, %[lut]\n\t"
        "ds_read_b64 %[d3], %[a1]\n\t"
        "ds_read_b64 %[d4], %[a1] offset:2056\n\t"
        "ds_read_b64 %[d5], %[a1] offset:4112\n\t"
        "s_waitcnt lgkmcnt(5)\n\t"
        "v_add_f64 %[r0a], %[r0a], %[d0]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_add_f64 %[r1a], %[r1a], %[d1]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[r2a], %[r2a], %[d2]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_add_f64 %[r0b], %[r0b], %[d3]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_add_f64 %[r1b], %[r1b], %[d4]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_f64 %[r2b], %[r2b], %[d5]"
        : [r0a] "+v"(r0a), [r1a] "+v"(r1a), [r2a] "+v"(r2a), [na] "+v"(na), [r0b] "+v"(r0b), [r1b] "+v"(r1b), [r2b] "+v"(r2b),
          [nb] "+v"(nb), [a0] "=&v"(a0), [a1] "=&v"(a1), [t0] "=&v"(t0), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2),
          [d3] "=&v"(d3), [d4] "=&v"(d4), [d5] "=&v"(d5)
        : [qv] "v"(qv), [xi] "v"(xi), [lut] "s"(lut), [c256] "v"(256u), [ma] "n"(0xf000 >> (4 * J)), [mb] "n"(0x0f00 >> (4 * J)), [sa] "n"(8 * J),
          [sb] "n"(8 * J + 8)
        : "vcc");
}

// one gapless piece of one read over a tile's positions
struct NqItem {
    int64_t kq;        // the cell at position P is query base kq + P of the arrays (MATCH)
    int32_t tlo;       // first position of the piece inside the tile
    uint16_t len;      // positions
    uint16_t flags;    // NQI_*
};
constexpr uint32_t NQI_TYPE = 3, NQI_MATCH = 0, NQI_DEL = 1, NQI_INSONLY = 2;   // bases / deleted positions / only the mark of an insertion
constexpr uint32_t NQI_INS = 4;                        // an insertion precedes the piece's first position
constexpr uint32_t NQI_HAP_SHIFT = 3;                  // two bits: HAP_0, HAP_1, HAP_NONE
constexpr int NQ_ITEMS = 128;                          // items per tile the plan has room for
constexpr uint32_t NQ_PLAN_OVER = 0xffffffffu;         // count of a tile that has more

// ---------------------------------------------------------------------------------------
// k_norm_plan: tile k of chunk blockIdx.y is positions [cs + 256 k, cs + 256 k + 256) below the chunk's end; its items go
// to items[(chunk * tpc + k) * NQ_ITEMS ..], their number to counts[chunk * tpc + k].  A wave plans NQ_PLAN_TILES
// consecutive tiles: the reads over them are nearly the same, and a lane fetches its read's header and first segment starts
// once for all of them.
#ifndef HIMUT_NQ_PLAN_TILES
#define HIMUT_NQ_PLAN_TILES 4
#endif
constexpr int NQ_PLAN_TILES = HIMUT_NQ_PLAN_TILES;
template <bool PHASE>
__global__ void __launch_bounds__(256) k_norm_plan(NormArgs A, Derived D, const int32_t* winlo, const int32_t* winhi, int64_t nblk,
                                                   int64_t tpc, NqItem* items, uint32_t* counts, NormRedo* redo,
                                                   unsigned int* nredo, unsigned int redo_cap) {
    const int lane = threadIdx.x & 63, wv = uni((int)(threadIdx.x >> 6));
    const int chunk = blockIdx.y;
    const int32_t cs_ = A.C.start[chunk], ce_ = A.C.end[chunk];
    const int64_t k_first = ((int64_t)blockIdx.x * 4 + wv) * NQ_PLAN_TILES;
    const int64_t first64 = (int64_t)cs_ + k_first * NQ_COLS;
    if (first64 >= ce_) return;
    const int32_t span_end = (int32_t)min(first64 + (int64_t)NQ_PLAN_TILES * NQ_COLS, (int64_t)ce_);
    constexpr bool phase = PHASE;
    const int64_t pairbase = phase ? A.C.pairoff[chunk] - A.C.rlo[chunk] : 0;
    const int64_t b0 = min(max(first64, (int64_t)0) >> WIN_SHIFT, nblk - 1), b1 = min(((int64_t)span_end - 1) >> WIN_SHIFT, nblk - 1);
    const int32_t lo = uni(winlo[b0]), hi = uni(winhi[max(b1, b0)]);
    uint32_t total[NQ_PLAN_TILES];
    bool over[NQ_PLAN_TILES];
#pragma unroll
    for (int t = 0; t < NQ_PLAN_TILES; t++) { total[t] = 0; over[t] = false; }
    for (int32_t r0 = lo; r0 < hi; r0 += 64) {
        const int nb = min(64, hi - r0);
        bool live_any = false;
        ReadMeta M;
        M.tstart = 0; M.tend = 0; M.nseg = 0; M.flags = RF_SECONDARY; M.segbase = 0; M.qoff = 0;
        uint32_t hp = HAP_NONE;
        // the starts of the read's first eight segments in one round trip: where a tile begins among them, without a search
        constexpr int NP = 8;
        int32_t tp[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) tp[q] = 0x7fffffff;
        if (lane < nb) {
            M = D.meta[r0 + lane];
            live_any = !(M.flags & RF_SECONDARY) && M.nseg > 0 && M.tstart < span_end && (int64_t)M.tend >= first64;
            if (live_any) {
#pragma unroll
                for (int q = 0; q < NP; q++) tp[q] = D.segs[M.segbase + min(q, M.nseg - 1)].t0;
                if (phase && M.tstart < ce_ && M.tend > cs_) hp = A.H.hap[pairbase + r0 + lane];   // fetched by the chunk
            }
        }
#pragma unroll
        for (int t = 0; t < NQ_PLAN_TILES; t++) {
            const int64_t base64 = first64 + (int64_t)t * NQ_COLS;
            if (base64 >= ce_ || over[t]) continue;
            const int32_t base = (int32_t)base64;
            const int32_t tile_end = (int32_t)min(base64 + NQ_COLS, (int64_t)ce_);
            const int64_t k = k_first + t;
            NqItem* out = items + (chunk * tpc + k) * NQ_ITEMS;
            const bool live = live_any && M.tstart < tile_end && M.tend >= base;
            int j0 = 0;
            if (live) {
                // the last segment that starts at or before the tile; the binary search only where it lies beyond the eight
                int a = 0;
#pragma unroll
                for (int q = 0; q < NP; q++) a += (q < M.nseg && tp[q] <= base) ? 1 : 0;
                if (a == NP && M.nseg > NP) {
                    int e = M.nseg;
                    while (a < e) { const int mm = (a + e) >> 1; if (D.segs[M.segbase + mm].t0 <= base) a = mm + 1; else e = mm; }
                }
                j0 = max(a - 1, 0);
            }
            // the pieces of this lane's read inside the tile; pass 0 counts them, pass 1 writes them behind the rows in front.
            // The first NG segments from j0 on come in one round trip and serve both passes (a tile seldom holds more of one read).
            constexpr int NG = 3;
            Seg gs[NG];
#pragma unroll
            for (int q = 0; q < NG; q++) { gs[q].t0 = 0x7fffffff; gs[q].q0 = 0; gs[q].len = 0; gs[q].flags = 0; }
            if (live) {
#pragma unroll
                for (int q = 0; q < NG; q++) if (j0 + q < M.nseg) gs[q] = D.segs[M.segbase + j0 + q];
            }
            uint32_t at = 0, batch = 0;
            for (int pass = 0; pass < 2; pass++) {
                uint32_t n = 0;
                auto piece = [&](const Seg& g) {
                    const int32_t span = g.len > 0 ? g.len : ((g.flags & SEG_INS) ? 1 : 0);      // a trailing insertion marks one position
                    int32_t tlo = max(g.t0, base);
                    const int32_t thi = (int32_t)min((int64_t)g.t0 + span, (int64_t)tile_end);
                    // a read this chunk did not fetch (normcounts.py:289) adds nothing: only its trailing insertion can reach in
                    if (!(M.tend > cs_)) tlo = max(tlo, cs_ + 1);
                    if (tlo >= thi) return;
                    if (pass == 1 && total[t] + at + n < (uint32_t)NQ_ITEMS) {
                        NqItem it;
                        it.kq = M.qoff + g.q0 - (int64_t)g.t0;
                        it.tlo = tlo;
                        it.len = (uint16_t)(thi - tlo);
                        uint32_t f = (g.flags & SEG_DEL) ? NQI_DEL : g.len == 0 ? NQI_INSONLY : NQI_MATCH;
                        if ((g.flags & SEG_INS) && tlo == g.t0) f |= NQI_INS;
                        f |= hp << NQI_HAP_SHIFT;
                        it.flags = (uint16_t)f;
                        out[total[t] + at + n] = it;
                    }
                    n++;
                };
                if (live) {
                    bool more = true;
#pragma unroll
                    for (int q = 0; q < NG; q++) {
                        if (more && (j0 + q >= M.nseg || gs[q].t0 >= tile_end)) more = false;
                        if (more) piece(gs[q]);
                    }
                    if (more)
                        for (int j = j0 + NG; j < M.nseg; j++) {
                            const Seg g = D.segs[M.segbase + j];
                            if (g.t0 >= tile_end) break;
                            piece(g);
                        }
                }
                if (pass == 0) {
                    const uint32_t incl = (uint32_t)wave_incl_add((int)n, lane);
                    at = incl - n;
                    batch = (uint32_t)lane_val((int)incl, 63);
                    if (total[t] + batch > (uint32_t)NQ_ITEMS) {        // more than the plan holds: the tile goes to k_norm_tile
                        if (lane == 0) {
                            counts[chunk * tpc + k] = NQ_PLAN_OVER;
                            const unsigned int w = atomicAdd(nredo, 1u);
                            if (w < redo_cap) { NormRedo z; z.chunk = chunk; z.base = base; redo[w] = z; }
                        }
                        over[t] = true;
                        break;
                    }
                }
            }
            if (!over[t]) total[t] += batch;
        }
    }
#pragma unroll
    for (int t = 0; t < NQ_PLAN_TILES; t++)
        if (first64 + (int64_t)t * NQ_COLS < ce_ && !over[t] && lane == 0) counts[chunk * tpc + k_first + t] = total[t];
}

// packed per-lane column flags
constexpr uint32_t NQF_CLS = 0, NQF_RAL = 4, NQF_ZERO = 12, NQF_INDEL = 16, NQF_OVER = 20;

template <bool PHASE>
__global__ void __launch_bounds__(NQ_WAVES * 64, HIMUT_NQ_OCC)
k_norm_quad(NormArgs A, const uint32_t* __restrict__ callable, int64_t nbases, const uint16_t* __restrict__ refcode,
            const NqItem* __restrict__ items, const uint32_t* __restrict__ counts, int64_t tpc, int64_t tiles_per_class,
            NormDirty* dirty, uint32_t* dcount, int64_t dirty_cap, int* dirty_over, NormRedo* redo, unsigned int* nredo,
            unsigned int redo_cap, unsigned int pool_limit) {
    __shared__ double s_lut[3 * 257];         // three tables of 256 qualities + a zero entry each (index 256)
    __shared__ unsigned int s_log[16];
    __shared__ unsigned long long s_bins[32];  // per pyrimidine trinucleotide: callable bases | positions << 32
    __shared__ NqPool s_pool[NQ_WAVES];
    __shared__ int s_bad;                     // a base outside ATGC was seen (the reference raises KeyError)
    __shared__ unsigned int s_ndirty;         // positions this workgroup has left to k_norm_dirty
    const int tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6);
    for (int i = tid; i < 3 * 256; i += NQ_WAVES * 64) s_lut[(i >> 8) * 257 + (i & 255)] = A.lut->t[i >> 8][i & 255];
    if (tid < 3) s_lut[tid * 257 + 256] = 0.0;
    if (tid < 16) s_log[tid] = 0;
    if (tid < 32) s_bins[tid] = 0;
    if (tid == 0) { s_bad = 0; s_ndirty = 0; }
    __syncthreads();
    const uint32_t lut = uni((uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)s_lut);
    const double pr0 = A.lut->prior[0], pr1 = A.lut->prior[1], pr2 = A.lut->prior[2], pr3 = A.lut->prior[3];   // (uniform: scalar registers)
    NqPool& pool = s_pool[wv];
    const int chunk = blockIdx.y;
    // this workgroup's part of the list of positions left to k_norm_dirty: filled from its start, the counter in LDS until the end
    const int64_t dregion = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int32_t cs_ = A.C.start[chunk], ce_ = A.C.end[chunk];
    constexpr bool phase = PHASE;
    const Reads& R = A.R;
    const int32_t md_thr = A.P.p.md_threshold, min_gq = A.P.p.min_gq, min_ref = A.P.p.min_ref_count, min_hap = A.P.p.min_hap_count;
    int bad = 0;
    const uint32_t o_q = 4u * (uint32_t)lane;                   // this lane's offset into a spanning piece's 256 qualities
    constexpr int NB = HIMUT_NQ_NB;
    static_assert(NB == 4, "the switches below are written for four places");
    // workgroup tiles per XCD class, by THIS chunk's length (chunks differ under --phase: with the longest chunk's figure the
    // short ones would sit on the first few XCDs only)
    const int64_t per = min(tiles_per_class, (((int64_t)(ce_ - cs_) + NQ_WG_COLS - 1) / NQ_WG_COLS + 7) / 8);
    for (int64_t t = blockIdx.x >> 3; t < per; t += (int64_t)(gridDim.x >> 3)) {         // the tile mapping of k_norm_tile
        const int64_t tile = ((int64_t)(blockIdx.x & 7) * per + t) * NQ_WAVES + wv;       // this wave's 256 positions
        const int64_t base64 = (int64_t)cs_ + tile * NQ_COLS;
        if (base64 >= ce_) { if (base64 - NQ_COLS * wv >= ce_) break; continue; }          // (this wave's part lies behind the chunk)
        const uint32_t n_items = uni(counts[chunk * tpc + tile]);
        if (n_items == NQ_PLAN_OVER) continue;                                            // (k_norm_plan listed it for k_norm_tile)
        if (n_items > (uint32_t)NQ_ITEMS) { bad |= 1 << HIMUT_ERR_ARG; continue; }
        const NqItem* plan = items + (chunk * tpc + tile) * NQ_ITEMS;
        const int32_t base = (int32_t)base64;
        const int32_t P0 = base + 4 * lane;
        // ---- the columns: which of the four exist, and what k_ref_codes knows about their letters
        uint32_t valid4 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t rp = (int64_t)P0 + j;
            if (rp < ce_) {
                if (rp < 0 || rp >= A.reflen) bad |= 1 << HIMUT_ERR_ARG;                   // IndexError in the reference
                else valid4 |= 1u << j;
            }
        }
        uint64_t codes = 0;                                      // four 16-bit codes
        if (valid4) __builtin_memcpy(&codes, refcode + P0, 8);   // (the array is padded behind the string with codes of 0)
        // packed per column j: ref4 the BAM code to match (bits 12 - 4j ..) and, sixteen bits up, a never-equal mark; fl the
        // flags: "an upper-case ATGC letter: the position is classified", the allele (two bits), a zero quality seen, an
        // insertion or deletion seen, and bit NQF_OVER: the pool ran out
        uint32_t ref4 = 0, fl = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t c = ((valid4 >> j) & 1u) ? (uint32_t)(codes >> (16 * j)) & 0xffffu : 0u;
            // a column whose letter is not one of ATGC is never classified; its cells are matched against the upper-case
            // letter so that they stay on the common path (a base outside ATGC is still found: it equals no letter)
            const uint32_t nib = (c & NQR_FOLD_OK) ? ((0x2481u >> (4 * (c & 3u))) & 15u) : 0u;     // A 1, T 8, G 4, C 2
            ref4 |= nib << (12 - 4 * j);
            if (nib == 0) ref4 |= 0x10000u << (12 - 4 * j);
            if (c & NQR_CLS) fl |= (1u << (NQF_CLS + j)) | ((c & 3u) << (NQF_RAL + 2 * j));
        }
        // ---- per-column state
        double R0[4], R1[4], R2[4];
        uint32_t nref[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { R0[j] = 0.0; R1[j] = 0.0; R2[j] = 0.0; nref[j] = 0; }
        // callable bases per column, a byte each: a tile has at most NQ_ITEMS = 128 pieces, so a byte holds the count (and the
        // counts of the reads of either haplotype, --phase, likewise)
        static_assert(NQ_ITEMS <= 255, "byte counters per column");
        uint32_t tri4b = 0;
        uint32_t slotmap = 0xffffffffu;        // pool slot per column (255: none)
        uint32_t h0b = 0, h1b = 0;             // (phase) the reads of either haplotype with a base in the column
        if (lane == 0) pool.n = 0;
        __builtin_amdgcn_wave_barrier();

        // rare: the cells of one item that are not the reference allele's -- another allele goes to the pool, a base outside
        // ATGC and a zero quality are noted.  qk / nk: the item's qualities and BAM codes, ck: which of the four cells are there
        auto rare_item = [&](uint32_t qk, uint32_t nk, uint32_t ck) {
            const uint32_t keep = (ck & 1u ? 0xf000u : 0u) | (ck & 2u ? 0x0f00u : 0u) | (ck & 4u ? 0x00f0u : 0u) | (ck & 8u ? 0x000fu : 0u);
            // zero qualities among the cells, byte by byte (exact), to the columns' flags
            const uint32_t zb = ~(((qk & 0x7f7f7f7fu) + 0x7f7f7f7fu) | qk) & 0x80808080u;
            fl |= (cs_pack4(zb) & ck) << NQF_ZERO;
            uint32_t x = ((nk ^ ref4) | (ref4 >> 16)) & keep;
            while (x) {                                                     // (a lane has one such cell, seldom two)
                const int hb = 31 - __builtin_clz(x);
                const int j = 3 - (hb >> 2);
                x &= ~(0xfu << (12 - 4 * j));
                const uint32_t q = (qk >> (8 * j)) & 255u;
                const uint32_t cell = (uint32_t)nib2allele((int)((nk >> (12 - 4 * j)) & 15u));
                if (cell > 3) { s_bad = 1 << HIMUT_ERR_BASE; continue; }
                if (!((fl >> (NQF_CLS + j)) & 1u) || q == 0u) continue;       // (never classified / ends at the classification)
                uint32_t slot = (slotmap >> (8 * j)) & 255u;
                if (slot == 255u) {
                    slot = atomicAdd(&pool.n, 1u);
                    if (slot >= pool_limit) { fl |= 1u << NQF_OVER; continue; }
#pragma unroll
                    for (int z = 0; z < 9; z++) pool.S[z][slot] = 0.0;
#pragma unroll
                    for (int z = 0; z < 4; z++) pool.cnt[z][slot] = 0;
                    slotmap = (slotmap & ~(255u << (8 * j))) | (slot << (8 * j));
                }
                const uint32_t rj = (fl >> (NQF_RAL + 2 * j)) & 3u;
                const uint32_t a_ = min(cell - (cell > rj ? 1u : 0u), 2u);   // allele c sits in slot c - (c > ref)
                // (the seven reads first, then the writes: written as three updates in a row the compiler keeps them in order)
                const double s0 = pool.S[a_][slot], s1 = pool.S[3 + a_][slot], s2 = pool.S[6 + a_][slot];
                const double l0 = s_lut[q], l1 = s_lut[257 + q], l2 = s_lut[514 + q];
                const uint32_t c0 = pool.cnt[cell][slot];
                pool.S[a_][slot] = s0 + l0;
                pool.S[3 + a_][slot] = s1 + l1;
                pool.S[6 + a_][slot] = s2 + l2;
                pool.cnt[cell][slot] = c0 + 1u;
            }
        };

        for (uint32_t i0 = 0; i0 < n_items; i0 += 64) {
            // ---- a lane per item
            const int nb = (int)min(64u, n_items - i0);
            int64_t kq = 0;
            int32_t tlo = 0;
            uint32_t lf = 0;                                  // len | flags << 16
            if (lane < nb) {
                const int4 raw = *reinterpret_cast<const int4*>(plan + i0 + lane);
                kq = ((int64_t)raw.y << 32) | (uint32_t)raw.x; tlo = raw.z; lf = (uint32_t)raw.w;
            }
            bool is_item = lane < nb;
            // (a piece whose bases lie outside the arrays, or outside the tile, cannot come out of k_norm_plan; acting on one
            //  would be a memory fault, so it is looked for)
            if (is_item && (tlo < base || (int64_t)tlo + (lf & 0xffffu) > (int64_t)base + NQ_COLS || (lf & 0xffffu) == 0 ||
                            (((lf >> 16) & NQI_TYPE) == NQI_MATCH && (kq + tlo < 0 || kq + tlo + (int64_t)(lf & 0xffffu) > nbases)))) {
                bad |= 1 << HIMUT_ERR_ARG;
                is_item = false;
            }
            const uint32_t ifl = lf >> 16;
            const bool is_match = is_item && (ifl & NQI_TYPE) == NQI_MATCH;
            const bool full = is_match && tlo == base && (lf & 0xffffu) == (uint32_t)NQ_COLS;
            // ---- what only sets flags, in any order: an insertion in front of a piece is counted at the piece's first
            //      position, a deletion at each of its positions
            {
                uint64_t mf = __ballot(is_item && ((ifl & NQI_INS) || (ifl & NQI_TYPE) == NQI_DEL));
                while (mf) {
                    const int gl = (int)__builtin_ctzll(mf);
                    mf &= mf - 1;
                    const int32_t g_tlo = lane_val(tlo, gl);
                    const uint32_t g_lf = (uint32_t)lane_val((int)lf, gl);
                    const uint32_t c4 = nq_cells(P0, g_tlo, (int32_t)(g_lf & 0xffffu)) & valid4;
                    const uint32_t first = (P0 <= g_tlo && g_tlo < P0 + 4) ? (1u << (g_tlo - P0)) : 0u;
                    if ((g_lf >> 16) & NQI_INS) fl |= (c4 & first) << NQF_INDEL;
                    if (((g_lf >> 16) & NQI_TYPE) == NQI_DEL) fl |= c4 << NQF_INDEL;
                }
            }
            // ---- the pieces with bases, in read order, NB at a time with their loads issued together.  A piece's first cell
            //      of the tile is query base kq + base of the arrays whether or not the piece reaches that far: addresses are
            //      32-bit distances from a base of the batch (K0: a multiple of 8 at least 256 in front of the nearest piece's
            //      first base), so that a load takes a scalar base, set up once per batch, and a per-lane offset.  A piece
            //      that covers a part of the positions (an indel or a read end inside them, one in ten) goes the same way
            //      under a mask of cells; its lanes outside the piece load the piece's first base and use nothing of it.
            //      (A piece further than 2^31 from K0, which 64 reads of a window do not give, or one that starts within the
            //      arrays' first bytes, sends the tile to k_norm_tile.)
            uint64_t kmin = is_match ? (uint64_t)(kq + tlo) : ~0ull;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(kmin >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)kmin, d, 64);
                kmin = min(kmin, o);
            }
            const int64_t K0 = uni((int64_t)((kmin == ~0ull ? 0ull : kmin) & ~7ull)) - NQ_COLS;      // (the same in every lane: scalar registers)
            const uint64_t dk64 = (uint64_t)(kq + base - K0);
            const bool far = is_match && (dk64 >= (1ull << 31) || (!full && kq + tlo < 8));
            if (__ballot(far)) { fl |= 1u << NQF_OVER; break; }
            const uint32_t dk = is_match ? (uint32_t)dk64 : 0u;
            const uint64_t s_bq = (uint64_t)((int64_t)(uint64_t)R.bq + K0), s_sq = (uint64_t)((int64_t)(uint64_t)R.seq + (K0 >> 1)),
                           s_cb = (uint64_t)((int64_t)(uint64_t)callable + (K0 >> 3));
            const uint64_t m_full = __ballot(full);
            uint64_t m = __ballot(is_match);
            while (m) {
                // cnt pieces sit in the LAST cnt of the NB places (a place in front of them asks for the first piece's words once
                // more, so that the loads stand in the text unconditionally); the updates are entered at the place of the first
                // piece: every one stands in the text once and runs unconditionally from its entry on
                const int cnt = min(NB, (int)__builtin_popcountll(m));
                uint32_t qv[NB], n4[NB], cb[NB], dd[NB], ifk[NB];
                int32_t gt[NB], gn[NB];             // (scalar) a partial piece's first position and length; gn = 0: the piece spans the tile
                {
                    const int first = (int)__builtin_ctzll(m);
#pragma unroll
                    for (int k = 0; k < NB; k++) {
                        int l = first;
                        if (k >= NB - cnt) { l = (int)__builtin_ctzll(m); m &= m - 1; }
                        dd[k] = (uint32_t)lane_val((int)dk, l);
                        ifk[k] = phase ? (uint32_t)lane_val((int)ifl, l) : 0u;
                        // (a scalar base and a 32-bit offset per lane; the offset is made opaque here so that its widening to
                        //  64 bits is not hoisted out of the loop, where the load would lose the form)
                        uint32_t tq = dd[k] + o_q;
                        gt[k] = 0; gn[k] = 0;
                        if (!((m_full >> l) & 1ull)) {
                            gt[k] = lane_val(tlo, l);
                            gn[k] = (int32_t)((uint32_t)lane_val((int)lf, l) & 0xffffu);
                            if (!(nq_cells(P0, gt[k], gn[k]) & valid4)) tq = dd[k] + (uint32_t)(gt[k] - base);
                        }
                        asm volatile("" : "+v"(tq));
                        qv[k] = nq_ld32(s_bq, tq);
                        n4[k] = nq_ld32(s_sq, tq >> 1);
                        cb[k] = nq_ld16(s_cb, tq >> 3);
                    }
                }
                uint32_t rare = 0;
#define NQ_UPDATE(k) { \
                    n4[k] = (__builtin_bswap32(n4[k]) >> ((dd[k] & 1u) ? 12u : 16u)) & 0xffffu; \
                    uint32_t cbk = (cb[k] >> ((dd[k] + o_q) & 7u)) & 15u;          /* (the sixteen bits begin at a byte) */ \
                    uint32_t x = (n4[k] ^ ref4) | (ref4 >> 16);                /* a nibble of zeros: the reference allele */ \
                    uint32_t xr = x & 0xffffu, zq = qv[k], hadd = 0x01010101u; \
                    if (gn[k]) {                                                /* a part of the positions: the cells outside add nothing */ \
                        const uint32_t c4 = nq_cells(P0, gt[k], gn[k]) & valid4; \
                        const uint32_t keep = nq_nibbles(c4); \
                        xr = x & keep; x |= ~keep; cbk &= c4; \
                        zq |= ~(nq_spread4(c4) * 255u); \
                        hadd = nq_spread4(c4); \
                    } \
                    if (phase) { \
                        const uint32_t hap = (ifk[k] >> NQI_HAP_SHIFT) & 3u; \
                        if (hap == HAP_0) h0b += hadd; \
                        else if (hap == HAP_1) h1b += hadd; \
                        else cbk = 0;                                        /* the read carries no haplotype in this chunk: no bit of it counts */ \
                    } \
                    tri4b += nq_spread4(cbk); \
                    nq_pair<0>(R0[0], R1[0], R2[0], nref[0], R0[1], R1[1], R2[1], nref[1], qv[k], x, lut); \
                    nq_pair<2>(R0[2], R1[2], R2[2], nref[2], R0[3], R1[3], R2[3], nref[3], qv[k], x, lut); \
                    if (xr | nq_zero_bytes(zq)) rare |= 1u << (k); }
                switch (cnt) {
                    case 4: NQ_UPDATE(0) [[fallthrough]];
                    case 3: NQ_UPDATE(1) [[fallthrough]];
                    case 2: NQ_UPDATE(2) [[fallthrough]];
                    default: NQ_UPDATE(3)
                }
#undef NQ_UPDATE
                if (__builtin_expect(__ballot(rare != 0) != 0, 0)) {
#pragma unroll 1
                    for (int k = NB - cnt; k < NB; k++) {
                        if (!__ballot((rare >> k) & 1u)) continue;
                        uint32_t qk = qv[0], nk = n4[0];
                        int32_t tk = gt[0], lk = gn[0];
#pragma unroll
                        for (int kk = 1; kk < NB; kk++) if (k == kk) { qk = qv[kk]; nk = n4[kk]; tk = gt[kk]; lk = gn[kk]; }
                        if ((rare >> k) & 1u) rare_item(qk, nk, lk ? (nq_cells(P0, tk, lk) & valid4) : 15u);
                    }
                }
            }
        }
        // ---- the pool ran out somewhere in the wave: the 256 positions go to k_norm_tile as they are
        if (__ballot((fl >> NQF_OVER) & 1u)) {
            if (lane == 0) {
                const unsigned int at = atomicAdd(nredo, 1u);
                if (at < redo_cap) { NormRedo z; z.chunk = chunk; z.base = base; redo[at] = z; }
            }
            continue;
        }
        // ---- the positions' classes (normcounts.py:317-402), in the order of the general text (NORM_CLASSIFY); straight-line
        //      for the lane's four columns, what is rare behind a test of the whole wave.  The counters nearly every position
        //      adds to are summed over the lane's columns and the wave first
        uint32_t w1 = 0, w2 = 0, w6 = 0, w13 = 0;
#ifndef NQ_CLS_UNROLL
#define NQ_CLS_UNROLL 0
#endif
#if NQ_CLS_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int j = 0; j < 4; j++) {
            constexpr bool rot = !NQ_CLS_UNROLL;
            const int jj = rot ? 0 : j;
            const uint32_t code = (uint32_t)(codes >> (rot ? 0 : 16 * j)) & 0xffffu;
            const uint32_t tri_sum = (tri4b >> (rot ? 0 : 8 * j)) & 255u;
            const bool cls = ((fl >> (NQF_CLS + (rot ? 0 : j))) & 1u) && tri_sum != 0;
            uint32_t h0 = 0, h1 = 0;
            if (phase) { h0 = (h0b >> (rot ? 0 : 8 * j)) & 255u; h1 = (h1b >> (rot ? 0 : 8 * j)) & 255u; }
            const bool hapfail = phase && cls && !((int32_t)h0 >= min_hap && (int32_t)h1 >= min_hap);
            const bool q0 = cls && !hapfail && ((fl >> (NQF_ZERO + (rot ? 0 : j))) & 1u);
            bad |= q0 ? (1 << HIMUT_ERR_BQ0) : 0;
            const bool open = cls && !hapfail && !q0;
            const uint32_t slot = (slotmap >> (rot ? 0 : 8 * j)) & 255u;
            // Nothing but the reference allele in the column: the ten genotype sums are four numbers (an allele that was not
            // seen adds +0.0 to a sum, which leaves it bit for bit what it was).  When hom-ref is the smallest by itself it
            // is the genotype and the quality is the gap to the smallest of the rest; any other outcome, and any column
            // with another allele, goes to k_norm_dirty.
            const double pa = -10.0 * (R0[jj] + pr0), pb = -10.0 * (R1[jj] + pr1);
            const double pc_ = -10.0 * (R2[jj] + pr2), pd = -10.0 * (R2[jj] + pr3);
            const double nxt = fmin(pb, fmin(pc_, pd));
            const bool mine = open && slot == 255u && pa < nxt;
            const double gqf = nxt - pa;
            const int gq = (gqf < 99.0) ? (int)gqf : 99;
            const bool indel = (fl >> (NQF_INDEL + (rot ? 0 : j))) & 1u;
            // (depths beyond 2^31 do not occur; the thresholds are 32-bit)
            const int slotn = indel ? 7 : (int32_t)nref[jj] > md_thr ? 8 : gq < min_gq ? 10 : (int32_t)nref[jj] < min_ref ? 9 : 13;
            w1 += (mine || hapfail) ? tri_sum : 0u;
            w2 += hapfail ? tri_sum : 0u;
            w6 += mine ? tri_sum : 0u;
            const bool call = mine && slotn == 13;
            w13 += call ? tri_sum : 0u;
            if (__ballot(mine && !call)) { if (mine && !call) atomicAdd(&s_log[slotn], tri_sum); }
            // the trinucleotide bins of a callable position (row 13 of norm.log): the bin is k_ref_codes', one LDS word for
            // both of its counters; a context with a letter outside upper-case ACGT takes NORM_TRIBINS' general way
            if (call && (code & NQR_BIN_OK)) atomicAdd(&s_bins[(code >> NQR_BIN_SHIFT) & 31u], (unsigned long long)tri_sum | (1ull << 32));
            if (__ballot(call && !(code & NQR_BIN_OK))) {
                if (call && !(code & NQR_BIN_OK)) {
                    const int64_t rpos = (int64_t)P0 + j;
                    int t0 = 'N', t1 = 'N', t2 = 'N';
                    if (rpos - 1 >= 0 && rpos + 2 <= A.reflen) {
                        t0 = (int)A.refseq[rpos - 1]; t1 = (int)A.refseq[rpos]; t2 = (int)A.refseq[rpos + 1];
                        if (t1 == 'A' || t1 == 'G') {
                            const int a0 = t2, a2 = t0;
                            t0 = a0 == 'A' ? 'T' : a0 == 'T' ? 'A' : a0 == 'G' ? 'C' : a0 == 'C' ? 'G' : 'N';
                            t1 = t1 == 'A' ? 'T' : 'C';
                            t2 = a2 == 'A' ? 'T' : a2 == 'T' ? 'A' : a2 == 'G' ? 'C' : a2 == 'C' ? 'G' : 'N';
                        }
                    }
                    const int64_t k = ((int64_t)A.cls[t0] * A.K + A.cls[t1]) * A.K + A.cls[t2];
                    atomicAdd(&A.ccs_tri[k], (unsigned long long)tri_sum);
                    atomicAdd(&A.ref_tri[k], 1ULL);
                }
            }
            // a position left to k_norm_dirty: the next places in this workgroup's part of the list (one LDS atomic per wave)
            const bool left = open && !mine;
            const uint64_t lm = __ballot(left);
            if (lm) {
                uint32_t at0 = 0;
                if (lane == 0) at0 = atomicAdd(&s_ndirty, (unsigned int)__builtin_popcountll(lm));
                const uint32_t at = uni(at0) + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                if (left) {
                    if ((int64_t)at < dirty_cap) {
                        NormDirty* d = dirty + (dregion * dirty_cap + at);
                        d->rpos = (int64_t)P0 + j; d->nref = nref[jj]; d->tri_sum = tri_sum; d->n_ins = indel ? 1u : 0u; d->n_del = 0; d->h0 = h0; d->h1 = h1;
                        d->R[0] = R0[jj]; d->R[1] = R1[jj]; d->R[2] = R2[jj];
                        if (slot != 255u) {
#pragma unroll
                            for (int k = 0; k < 4; k++) d->cnt[k] = pool.cnt[k][slot];
#pragma unroll
                            for (int k = 0; k < 9; k++) d->S[k] = pool.S[k][slot];
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; k++) d->cnt[k] = 0;
#pragma unroll
                            for (int k = 0; k < 9; k++) d->S[k] = 0.0;
                        }
                    } else *dirty_over = 1;          // more of them than there is room for: the host repeats the contig with k_norm_tile
                }
            }
            if (rot) {                                       // the next column into place
                R0[0] = R0[1]; R0[1] = R0[2]; R0[2] = R0[3]; R1[0] = R1[1]; R1[1] = R1[2]; R1[2] = R1[3];
                R2[0] = R2[1]; R2[1] = R2[2]; R2[2] = R2[3];
                nref[0] = nref[1]; nref[1] = nref[2]; nref[2] = nref[3]; tri4b >>= 8;
                if (phase) { h0b >>= 8; h1b >>= 8; }
                fl = (fl >> 1) & 0x77007u;                   // (the one-bit fields move down; the alleles are not looked at here)
                slotmap >>= 8; codes >>= 16;
            }
        }
        {
            const uint32_t s1 = (uint32_t)lane_val(wave_incl_add((int)w1, lane), 63);
            const uint32_t s6 = (uint32_t)lane_val(wave_incl_add((int)w6, lane), 63);
            const uint32_t s13 = (uint32_t)lane_val(wave_incl_add((int)w13, lane), 63);
            uint32_t s2 = 0;
            if (phase) s2 = (uint32_t)lane_val(wave_incl_add((int)w2, lane), 63);
            if (lane == 0) {
                if (s1) atomicAdd(&s_log[1], s1);
                if (s2) atomicAdd(&s_log[2], s2);
                if (s6) atomicAdd(&s_log[6], s6);
                if (s13) atomicAdd(&s_log[13], s13);
            }
        }
        __builtin_amdgcn_wave_barrier();           // (the pool is handed out anew by the next tile)
    }
    __syncthreads();
    if (tid == 0) dcount[dregion] = s_ndirty;      // (what was asked for: k_norm_dirty stops at the part's room, the host sizes a second pass by it)
    if (tid < 14 && s_log[tid]) atomicAdd(&A.log[tid], (unsigned long long)s_log[tid]);
    if (tid < 32 && s_bins[tid]) {
        const int cl[4] = {A.cA, A.cC, A.cG, A.cT};
        const int64_t k = ((int64_t)cl[tid >> 3] * A.K + ((tid & 4) ? A.cT : A.cC)) * A.K + cl[tid & 3];
        atomicAdd(&A.ccs_tri[k], s_bins[tid] & 0xffffffffull);
        atomicAdd(&A.ref_tri[k], s_bins[tid] >> 32);
    }
    if (tid == 0 && s_bad) bad |= s_bad;
    if (bad) atomicOr(A.err, bad);
}

}  // namespace himut
