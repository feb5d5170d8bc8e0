// k_norm_quad: the position sweep of normcounts.get_callable_tricounts (src/himut/normcounts.py:113-140,315-402) with FOUR
// consecutive reference positions per lane.
//
// A wave owns 256 consecutive positions of a chunk (a lane the columns P0 .. P0 + 3, P0 = base + 4 * lane), a workgroup of
// four waves 1024.  The reads of the window index under the 256 positions are the wave's rows, prepared 64 at a time with
// a LANE per ROW (header, the segment that reaches the positions, whether that one gapless segment spans all 256) and
// then taken in read order.  A spanning row is 256 consecutive query bases from K on, so a row is three loads at a scalar
// base plus a fixed per-lane offset: a dword of four qualities (a 256-byte transaction per wave), a dword that holds the
// four packed bases, sixteen bits of the callable bit array.  The reference allele's three ordered fp64 sums of the lane's
// four columns are four independent chains; a cell of another allele -- one in a thousand -- sends the row's lanes that
// hold one down a side path that adds to a small POOL of accumulators in LDS (thirty-two columns of a wave's 256 may own
// one; nine doubles and four counts each), so the kernel's LDS is the tables plus 11 KB and the registers decide the
// occupancy.  A row with an indel or a read end inside the 256 positions is taken segment by segment with the same update
// under a per-lane mask of cells.  At the end of a column: nothing but the reference allele in it makes the ten genotype
// sums four numbers and the kernel classifies the position itself; the others go to k_norm_dirty's list (himut_norm.h).
// A wave whose pool runs out (more than thirty-two columns with another allele among 256) leaves its 256 positions to
// k_norm_tile through a list of tiles.  Same counts as k_norm_tile, bit for bit.
#pragma once

#include "himut_norm.h"

namespace himut {

#ifndef HIMUT_NQ_NB
#define HIMUT_NQ_NB 4            // spanning rows whose loads are issued together
#endif
#ifndef HIMUT_NQ_OCC
#define HIMUT_NQ_OCC 6           // waves per SIMD asked of the register allocator
#endif
#ifndef HIMUT_NQ_Q
#define HIMUT_NQ_Q 8             // workgroups per XCD class and chunk (neighbouring tiles: the mapping of k_norm_tile)
#endif
constexpr int NQ_WAVES = 4;
constexpr int NQ_COLS = 256;                       // positions per wave
constexpr int NQ_WG_COLS = NQ_WAVES * NQ_COLS;     // positions per workgroup and step
constexpr int NQ_SLOTS = 32;                       // pool of other-allele accumulators per wave
constexpr int NQ_Q = HIMUT_NQ_Q;

struct NormRedo { int32_t chunk, base; };          // 256 positions from `base` of chunk `chunk`, left to k_norm_tile

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
__device__ __forceinline__ uint64_t nq_lane64(int64_t v, int l) {
    return ((uint64_t)(uint32_t)lane_val((int)(v >> 32), l) << 32) | (uint32_t)lane_val((int)v, l);
}

// per-wave pool of accumulators for the alleles that are not the reference's
struct NqPool {
    double S[9][NQ_SLOTS];      // [table * 3 + slot of the allele][pool slot]
    uint32_t cnt[4][NQ_SLOTS];  // [allele][pool slot]
    uint32_t n;                 // slots handed out
    uint32_t pad;
};

template <bool PHASE>
__global__ void __launch_bounds__(NQ_WAVES * 64, HIMUT_NQ_OCC)
k_norm_quad(NormArgs A, Derived D, const uint32_t* __restrict__ callable, const int32_t* winlo, const int32_t* winhi, int64_t nblk,
            int64_t tiles_per_class, NormDirty* dirty, unsigned long long* dcount, int64_t dirty_cap, int* dirty_over,
            NormRedo* redo, unsigned int* nredo, unsigned int redo_cap) {
    __shared__ double s_lut[3 * 257];         // three tables of 256 qualities + a zero entry each (index 256)
    __shared__ double s_prior[4];
    __shared__ unsigned int s_log[16];
    __shared__ unsigned int s_ccs[32], s_ref[32];
    __shared__ NqPool s_pool[NQ_WAVES];
    __shared__ int s_bad;                     // a base outside ATGC was seen (the reference raises KeyError)
    const int tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6);
    for (int i = tid; i < 3 * 256; i += NQ_WAVES * 64) s_lut[(i >> 8) * 257 + (i & 255)] = A.lut->t[i >> 8][i & 255];
    if (tid < 3) s_lut[tid * 257 + 256] = 0.0;
    if (tid < 4) s_prior[tid] = A.lut->prior[tid];
    if (tid < 16) s_log[tid] = 0;
    if (tid < 32) { s_ccs[tid] = 0; s_ref[tid] = 0; }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    NqPool& pool = s_pool[wv];
    const int chunk = blockIdx.y;
    const int64_t dregion = (int64_t)((blockIdx.x + blockIdx.y * gridDim.x) & (NORM_DIRTY_REGIONS - 1));   // this workgroup's part of the list
    const int32_t cs_ = A.C.start[chunk], ce_ = A.C.end[chunk];
    constexpr bool phase = PHASE;
    const int64_t pairbase = phase ? A.C.pairoff[chunk] - A.C.rlo[chunk] : 0;
    const Reads& R = A.R;
    int bad = 0;
    // per-lane load offsets of a spanning row (kept in vector registers: the loads then take a scalar base + this offset)
    uint32_t o_q = 4u * (uint32_t)lane, o_s = 2u * (uint32_t)lane, o_b = (uint32_t)lane >> 1;
    const uint32_t bsh_lane = 4u * ((uint32_t)lane & 1u);
    constexpr int NB = HIMUT_NQ_NB;
    const int64_t per = tiles_per_class;
    for (int64_t t = blockIdx.x >> 3; t < per; t += (int64_t)(gridDim.x >> 3)) {         // the tile mapping of k_norm_tile
        const int64_t tile = (int64_t)(blockIdx.x & 7) * per + t;
        const int64_t base = (int64_t)cs_ + tile * NQ_WG_COLS + NQ_COLS * wv;             // this wave's 256 positions
        if (base >= ce_) { if ((int64_t)cs_ + tile * NQ_WG_COLS >= ce_) break; continue; }   // (this wave's part lies behind the chunk)
        const int64_t P0l = base + 4 * lane;
        const int32_t P0 = (int32_t)P0l;
        // ---- the columns: which of the four exist, their reference letters and the two beside them
        uint32_t valid4 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t rp = P0l + j;
            if (rp < ce_) {
                if (rp < 0 || rp >= A.reflen) bad |= 1 << HIMUT_ERR_ARG;                   // IndexError in the reference
                else valid4 |= 1u << j;
            }
        }
        uint32_t rl_lo = 0x4e4e4e4eu, rl_hi = 0x4e4e4e4eu;       // the letters at P0 - 1 .. P0 + 2 and P0 + 3 .. P0 + 6 ("N" outside)
        if (valid4) {
            if (P0l >= 1 && P0l + 7 <= A.reflen) {                // (the array has slack behind it, but its end is the contig's)
                rl_lo = nq_ld32((uint64_t)A.refseq, (uint32_t)(P0l - 1));
                rl_hi = nq_ld32((uint64_t)A.refseq, (uint32_t)(P0l + 3));
            } else {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int64_t p = P0l - 1 + k;
                    const uint32_t c = (p >= 0 && p < A.reflen) ? (uint32_t)A.refseq[p] : 0u;   // 0: outside the string
                    if (k < 4) lo |= c << (8 * k); else hi |= c << (8 * (k - 4));
                }
                rl_lo = lo; rl_hi = hi;
            }
        }
        const uint32_t letters = (rl_lo >> 8) | (rl_hi << 24);   // the four columns' own letters
        int ref[4];
        uint32_t ref4 = 0, force4 = 0, cls4 = 0;                 // packed BAM codes (column j in bits 12 - 4j ..), never-equal marks, classifiable columns
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = (int)((letters >> (8 * j)) & 255u);
            ref[j] = ((valid4 >> j) & 1u) ? char2allele(c) : -1;
            // a column whose letter is not one of ATGC is never classified; its cells are matched against the upper-case
            // letter so that they stay on the common path (a base outside ATGC is still found: it equals no letter)
            const int cu = c & 0xdf;
            const uint32_t nib = cu == 'A' ? 1u : cu == 'C' ? 2u : cu == 'G' ? 4u : cu == 'T' ? 8u : 0u;
            ref4 |= nib << (12 - 4 * j);
            if (nib == 0) force4 |= 1u << (12 - 4 * j);
            if (ref[j] >= 0) cls4 |= 1u << j;
        }
        // ---- per-column state
        double R0[4], R1[4], R2[4];
        uint32_t nref[4], tri[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { R0[j] = 0.0; R1[j] = 0.0; R2[j] = 0.0; nref[j] = 0; tri[j] = 0; }
        uint32_t tri4b = 0;                    // callable bases of the running batch, a byte per column
        uint32_t indel4 = 0, zero4 = 0;        // columns with an insertion or a deletion; with a zero quality
        uint32_t slotmap = 0xffffffffu;        // pool slot per column (255: none)
        uint32_t h0g[4], h1g[4];               // (phase) haplotype votes of the rows that do not span
        uint32_t h0b = 0, h1b = 0;
        uint32_t h0_span = 0, h1_span = 0;     // (phase) spanning rows per haplotype: the same for every column
        if (phase) {
#pragma unroll
            for (int j = 0; j < 4; j++) { h0g[j] = 0; h1g[j] = 0; }
        }
        bool over = false;                     // the pool ran out: the tile goes to k_norm_tile
        if (lane == 0) pool.n = 0;
        __builtin_amdgcn_wave_barrier();

        // one cell of another allele than the reference's (or a base outside ATGC), column J, BAM code NIB, quality Q
        auto alt_cell = [&](int J, uint32_t nibv, uint32_t q) {
            const uint32_t cell = (uint32_t)nib2allele((int)nibv);
            if (cell > 3) { s_bad = 1 << HIMUT_ERR_BASE; return; }
            if (!((cls4 >> J) & 1u) || q == 0) return;                      // (never classified / ends at the classification)
            uint32_t slot = (slotmap >> (8 * J)) & 255u;
            if (slot == 255u) {
                slot = atomicAdd(&pool.n, 1u);
                if (slot >= (uint32_t)NQ_SLOTS) { over = true; return; }
#pragma unroll
                for (int k = 0; k < 9; k++) pool.S[k][slot] = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) pool.cnt[k][slot] = 0;
                slotmap = (slotmap & ~(255u << (8 * J))) | (slot << (8 * J));
            }
            const uint32_t a_ = min(cell - ((int)cell > ref[J] ? 1u : 0u), 2u);
            pool.S[a_][slot] = pool.S[a_][slot] + s_lut[q];
            pool.S[3 + a_][slot] = pool.S[3 + a_][slot] + s_lut[257 + q];
            pool.S[6 + a_][slot] = pool.S[6 + a_][slot] + s_lut[514 + q];
            pool.cnt[cell][slot] = pool.cnt[cell][slot] + 1u;
        };
        // The update of four cells of one read: qualities qv (byte j = column j), BAM codes n4 (column j in bits 12 - 4j ..),
        // callable bits cb (bit j), cm = which of the four are cells of this read at all (15 for a spanning row)
        auto update4 = [&](uint32_t qv, uint32_t n4, uint32_t cb, uint32_t cm, bool full) {
            uint32_t x = (n4 ^ ref4) | force4;                               // a nibble of zeros: the reference allele
            if (!full) {
                const uint32_t keep = (cm & 1u ? 0xf000u : 0u) | (cm & 2u ? 0x0f00u : 0u) | (cm & 4u ? 0x00f0u : 0u) | (cm & 8u ? 0x000fu : 0u);
                x &= keep;
                cb &= cm;
            }
            tri4b += nq_spread4(cb);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool isref = ((x >> (12 - 4 * j)) & 15u) == 0u && (full || ((cm >> j) & 1u));
                const uint32_t q = (qv >> (8 * j)) & 255u;
                const uint32_t e = isref ? q : 256u;                        // the zero row for everything but the reference allele
                nref[j] += isref ? 1u : 0u;
                R0[j] = R0[j] + s_lut[e]; R1[j] = R1[j] + s_lut[257 + e]; R2[j] = R2[j] + s_lut[514 + e];
            }
            uint32_t zq = nq_zero_bytes(qv);
            if (!full && zq) {                                               // (only the zero bytes of cells that are there)
                zq = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) if (((cm >> j) & 1u) && ((qv >> (8 * j)) & 255u) == 0u) zq |= 1u;
            }
            if (__builtin_expect((x | zq) != 0u, 0)) {                       // rare: another allele, a base outside ATGC, a zero quality
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (!full && !((cm >> j) & 1u)) continue;
                    const uint32_t q = (qv >> (8 * j)) & 255u;
                    if (q == 0u) zero4 |= 1u << j;
                    if ((x >> (12 - 4 * j)) & 15u) alt_cell(j, (n4 >> (12 - 4 * j)) & 15u, q);
                }
            }
        };

        const int64_t b0 = min(max(base, (int64_t)0) >> WIN_SHIFT, nblk - 1), b1 = min((base + NQ_COLS - 1) >> WIN_SHIFT, nblk - 1);
        const int32_t lo = uni(winlo[b0]), hi = uni(winhi[b1]);
        for (int32_t r0 = lo; r0 < hi; r0 += 64) {
            // ---- a lane per row: header, the last segment that starts at or before the positions, the spanning test
            const int nb = min(64, hi - r0);
            ReadMeta M;
            M.tstart = 0; M.tend = 0; M.nseg = 0; M.flags = RF_SECONDARY; M.segbase = 0; M.qoff = 0;
            if (lane < nb) M = D.meta[r0 + lane];
            const bool live_row = lane < nb && !(M.flags & RF_SECONDARY) && M.nseg > 0 && M.tstart < base + NQ_COLS && M.tend >= base;
            int j0 = 0;
            int4 sg0 = make_int4(0x7fffffff, 0, 0, 0);
            if (live_row) {
                // the starts of the read's first eight segments in one round trip; the binary search, a round trip a step, only
                // where the segment lies further on
                constexpr int NP = 8;
                int32_t tp[NP];
#pragma unroll
                for (int k = 0; k < NP; k++) tp[k] = D.segs[M.segbase + min(k, M.nseg - 1)].t0;
                int a = 0;
#pragma unroll
                for (int k = 0; k < NP; k++) a += (k < M.nseg && tp[k] <= (int32_t)base) ? 1 : 0;       // (starts ascend: the first `a` of them)
                if (a == NP && M.nseg > NP) {
                    int e = M.nseg;
                    while (a < e) { const int m = (a + e) >> 1; if (D.segs[M.segbase + m].t0 <= (int32_t)base) a = m + 1; else e = m; }
                }
                j0 = max(a - 1, 0);
                sg0 = *reinterpret_cast<const int4*>(D.segs + M.segbase + j0);
            }
            // the bases of a batch's reads lie side by side in the arrays: a row's first base as a distance from the first
            // read's first base (a row further away than 2^31, which does not happen, takes the general path)
            const int64_t Kb = (int64_t)nq_lane64(M.qoff, 0);
            const int64_t K0 = M.qoff + sg0.y + ((int32_t)base - sg0.x);
            const int64_t dK = K0 - Kb;
            const bool whole = live_row && !((uint32_t)sg0.w & SEG_DEL) && sg0.x <= (int32_t)base &&
                               (int64_t)sg0.x + sg0.z >= base + NQ_COLS && dK >= 0 && dK < ((int64_t)1 << 31);
            const int dqv = whole ? (int)dK : 0;
            uint32_t hp = HAP_NONE;
            if (phase && live_row && M.tstart < ce_ && M.tend > cs_) hp = A.H.hap[pairbase + r0 + lane];   // fetched by the chunk
            const uint64_t m_live = __ballot(live_row), m_whole = __ballot(whole);
            const uint64_t m_slow = m_live & ~m_whole;
            const uint64_t m_hnone = phase ? __ballot(hp != HAP_0 && hp != HAP_1) : 0;
            // counts that do not depend on the order: an insertion in front of a spanning segment that starts with the
            // positions is counted at the first of them; every cell of a spanning row is a base of its haplotype
            const uint64_t m_ins0 = __ballot(whole && sg0.x == (int32_t)base && ((uint32_t)sg0.w & SEG_INS));
            if (lane == 0 && m_ins0) indel4 |= 1u;
            if (phase) {
                h0_span += (uint32_t)__builtin_popcountll(__ballot(whole && hp == HAP_0));
                h1_span += (uint32_t)__builtin_popcountll(__ballot(whole && hp == HAP_1));
            }
            // ---- the live rows in read order
            uint64_t m = m_live;
            while (m) {
                // do NB spanning rows follow each other?  (live rows in front of the next row of the other kind)
                const uint64_t ms = m & m_slow;
                const uint64_t front = ms ? (m & ((ms & (0 - ms)) - 1)) : m;
                asm volatile("" : "+v"(o_q), "+v"(o_s), "+v"(o_b));
                if (__builtin_popcountll(front) >= NB) {
                    uint32_t qv[NB], sv[NB], bv[NB], sh_s[NB], sh_b[NB];
                    bool hn[NB];
#pragma unroll
                    for (int k = 0; k < NB; k++) {
                        const int l = (int)__builtin_ctzll(m);
                        m &= m - 1;
                        const uint64_t Kr = (uint64_t)Kb + (uint32_t)lane_val(dqv, l);
                        sh_s[k] = (Kr & 1u) ? 12u : 16u;
                        sh_b[k] = (uint32_t)(Kr & 7u);
                        hn[k] = phase && ((m_hnone >> l) & 1);
                        qv[k] = nq_ld32((uint64_t)R.bq + Kr, o_q);
                        sv[k] = nq_ld32((uint64_t)R.seq + (Kr >> 1), o_s);
                        bv[k] = nq_ld16((uint64_t)callable + (Kr >> 3), o_b);
                    }
#pragma unroll
                    for (int k = 0; k < NB; k++) {
                        const uint32_t n4 = (__builtin_bswap32(sv[k]) >> sh_s[k]) & 0xffffu;
                        uint32_t cb = (bv[k] >> (sh_b[k] + bsh_lane)) & 15u;
                        if (phase && hn[k]) cb = 0;                          // the read carries no haplotype in this chunk
                        update4(qv[k], n4, cb, 15u, true);
                    }
                    continue;
                }
                const int l0 = (int)__builtin_ctzll(m);
                m &= m - 1;
                if ((m_whole >> l0) & 1) {                               // a spanning row by itself
                    const uint64_t Kr = (uint64_t)Kb + (uint32_t)lane_val(dqv, l0);
                    const uint32_t q1 = nq_ld32((uint64_t)R.bq + Kr, o_q);
                    const uint32_t s1 = nq_ld32((uint64_t)R.seq + (Kr >> 1), o_s);
                    const uint32_t b1_ = nq_ld16((uint64_t)callable + (Kr >> 3), o_b);
                    const uint32_t n4 = (__builtin_bswap32(s1) >> ((Kr & 1u) ? 12u : 16u)) & 0xffffu;
                    uint32_t cb = (b1_ >> ((uint32_t)(Kr & 7u) + bsh_lane)) & 15u;
                    if (phase && ((m_hnone >> l0) & 1)) cb = 0;
                    update4(q1, n4, cb, 15u, true);
                    continue;
                }
                // ---- the general row: segment by segment from the cursor on, the same update under a mask of cells
                {
                    const int ns = lane_val(M.nseg, l0), jf = lane_val(j0, l0);
                    const uint64_t segbase = nq_lane64(M.segbase, l0), qoff = nq_lane64(M.qoff, l0);
                    const int32_t tend_r = lane_val(M.tend, l0);
                    uint32_t hps = HAP_NONE;
                    if (phase) hps = (uint32_t)lane_val((int)hp, l0);
                    const bool hap_ok = !phase || hps == HAP_0 || hps == HAP_1;
                    for (int j = jf; j < ns; j++) {
                        int32_t t0, q0, len;
                        uint32_t fl;
                        if (j == jf) {                                   // (the row vector holds it)
                            t0 = lane_val(sg0.x, l0); q0 = lane_val(sg0.y, l0); len = lane_val(sg0.z, l0); fl = (uint32_t)lane_val(sg0.w, l0);
                        } else {
                            const Seg g = D.segs[segbase + j];
                            t0 = uni(g.t0); q0 = uni(g.q0); len = uni(g.len); fl = uni(g.flags);
                        }
                        if (t0 >= base + NQ_COLS) break;
                        const int32_t span = len > 0 ? len : ((fl & SEG_INS) ? 1 : 0);      // a trailing insertion marks one position
                        // this lane's cells of the segment: columns jlo .. jhi - 1
                        const int32_t jlo = min(max(t0 - P0, 0), 4), jhi = min(max(t0 + span - P0, 0), 4);
                        uint32_t cm = jhi > jlo ? (((1u << (jhi - jlo)) - 1u) << jlo) : 0u;
                        // a read this chunk did not fetch (normcounts.py:289) adds nothing: only its trailing insertion can reach in
                        if (!(tend_r > cs_)) {
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) if (P0 + jj <= cs_) cm &= ~(1u << jj);
                        }
                        cm &= valid4;
                        // an insertion in front of the segment is counted at its first position
                        if ((fl & SEG_INS) && P0 <= t0 && t0 < P0 + 4 && ((cm >> (t0 - P0)) & 1u)) indel4 |= 1u << (t0 - P0);
                        if (fl & SEG_DEL) { indel4 |= cm; continue; }
                        if (len == 0 || !__ballot(cm != 0)) continue;
                        if (cm) {
                            // the first of the lane's cells is query base off (from the read's first): loads from there on, brought
                            // to the columns' places (nothing in front of the segment is touched: the first read has nothing there)
                            const uint32_t off = (uint32_t)(q0 + (P0 + jlo - t0));
                            const uint32_t qraw = nq_ld32((uint64_t)R.bq + qoff, off);
                            const uint64_t ks = qoff + off;                  // absolute base index
                            const uint32_t sraw = nq_ld32((uint64_t)R.seq, (uint32_t)(ks >> 1));
                            const uint32_t braw = nq_ld16((uint64_t)callable, (uint32_t)(ks >> 3));
                            const uint32_t qv = qraw << (8 * jlo);
                            const uint32_t n4 = ((__builtin_bswap32(sraw) >> ((ks & 1u) ? 12u : 16u)) & 0xffffu) >> (4 * jlo);
                            uint32_t cb = ((braw >> (uint32_t)(ks & 7u)) & 15u) << jlo;
                            if (!hap_ok) cb = 0;
                            update4(qv, n4, cb, cm, false);
                            if (phase) {
                                if (hps == HAP_0) h0b += nq_spread4(cm);
                                else if (hps == HAP_1) h1b += nq_spread4(cm);
                            }
                        }
                    }
                }
            }
            // ---- the batch's byte counters into the columns' words
#pragma unroll
            for (int j = 0; j < 4; j++) tri[j] += (tri4b >> (8 * j)) & 255u;
            tri4b = 0;
            if (phase) {
#pragma unroll
                for (int j = 0; j < 4; j++) { h0g[j] += (h0b >> (8 * j)) & 255u; h1g[j] += (h1b >> (8 * j)) & 255u; }
                h0b = 0; h1b = 0;
            }
        }
        // ---- the pool ran out somewhere in the wave: the 256 positions go to k_norm_tile as they are
        if (__ballot(over)) {
            if (lane == 0) {
                const unsigned int at = atomicAdd(nredo, 1u);
                if (at < redo_cap) { NormRedo z; z.chunk = chunk; z.base = (int32_t)base; redo[at] = z; }
            }
            continue;
        }
        // ---- the positions' classes (normcounts.py:317-402), in the order of the general text (NORM_CLASSIFY).  The counters
        //      nearly every position adds to are summed over the lane's columns and the wave first
        uint32_t w1 = 0, w2 = 0, w6 = 0, w13 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t rpos = P0l + j;
            const uint32_t tri_sum = tri[j];
            const bool cls = ((cls4 >> j) & 1u) && tri_sum != 0;
            uint32_t h0 = 0, h1 = 0;
            if (phase) { h0 = h0_span + h0g[j]; h1 = h1_span + h1g[j]; }
            const bool hapfail = phase && cls && !((int64_t)h0 >= A.P.p.min_hap_count && (int64_t)h1 >= A.P.p.min_hap_count);
            const bool q0 = cls && !hapfail && ((zero4 >> j) & 1u);
            if (q0) bad |= 1 << HIMUT_ERR_BQ0;
            const bool open = cls && !hapfail && !q0;
            const uint32_t slot = (slotmap >> (8 * j)) & 255u;
            // Nothing but the reference allele in the column: the ten genotype sums are four numbers (an allele that was not
            // seen adds +0.0 to a sum, which leaves it bit for bit what it was).  When hom-ref is the smallest by itself it
            // is the genotype and the quality is the gap to the smallest of the rest; any other outcome, and any column
            // with another allele, goes to k_norm_dirty.
            const double pa = -10.0 * (R0[j] + s_prior[0]), pb = -10.0 * (R1[j] + s_prior[1]);
            const double pc = -10.0 * (R2[j] + s_prior[2]), pd = -10.0 * (R2[j] + s_prior[3]);
            const double nxt = fmin(pb, fmin(pc, pd));
            const bool mine = open && slot == 255u && pa < nxt;
            const double gqf = nxt - pa;
            const int gq = (gqf < 99.0) ? (int)gqf : 99;
            const bool indel = (indel4 >> j) & 1u;
            int slotn = 13;
            if (indel) slotn = 7;
            else if ((int64_t)nref[j] > A.P.p.md_threshold) slotn = 8;
            else if (gq < A.P.p.min_gq) slotn = 10;
            else if ((int64_t)nref[j] < A.P.p.min_ref_count) slotn = 9;
            w1 += (mine || hapfail) ? tri_sum : 0u;
            w2 += hapfail ? tri_sum : 0u;
            w6 += mine ? tri_sum : 0u;
            w13 += (mine && slotn == 13) ? tri_sum : 0u;
            if (mine && slotn != 13) atomicAdd(&s_log[slotn], tri_sum);
            if (mine && slotn == 13) {
                const int refc = (int)((letters >> (8 * j)) & 255u);
                const uint64_t six = (uint64_t)rl_lo | ((uint64_t)rl_hi << 32);
                NORM_TRIBINS((int)((six >> (8 * j)) & 255u), (int)((six >> (8 * (j + 2))) & 255u))
            }
            if (open && !mine) {
                // a place in the workgroup's region of the list: one atomic per wave and column index
                const int64_t at = (int64_t)wave_reserve(dcount + dregion * 16);
                if (at < dirty_cap) {
                    NormDirty d;
                    d.rpos = rpos; d.nref = nref[j]; d.tri_sum = tri_sum; d.n_ins = indel ? 1u : 0u; d.n_del = 0; d.h0 = h0; d.h1 = h1;
#pragma unroll
                    for (int k = 0; k < 4; k++) d.cnt[k] = slot != 255u ? pool.cnt[k][slot] : 0u;
                    d.R[0] = R0[j]; d.R[1] = R1[j]; d.R[2] = R2[j];
#pragma unroll
                    for (int k = 0; k < 9; k++) d.S[k] = slot != 255u ? pool.S[k][slot] : 0.0;
                    dirty[dregion * dirty_cap + at] = d;
                } else *dirty_over = 1;              // more of them than there is room for: the host repeats the contig with k_norm_tile
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
    if (tid < 14 && s_log[tid]) atomicAdd(&A.log[tid], (unsigned long long)s_log[tid]);
    if (tid < 32 && (s_ccs[tid] || s_ref[tid])) {
        const int cl[4] = {A.cA, A.cC, A.cG, A.cT};
        const int64_t k = ((int64_t)cl[tid >> 3] * A.K + ((tid & 4) ? A.cT : A.cC)) * A.K + cl[tid & 3];
        atomicAdd(&A.ccs_tri[k], (unsigned long long)s_ccs[tid]);
        atomicAdd(&A.ref_tri[k], (unsigned long long)s_ref[tid]);
    }
    if (tid == 0 && s_bad) bad |= s_bad;
    if (bad) atomicOr(A.err, bad);
}

}  // namespace himut
