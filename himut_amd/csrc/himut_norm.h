// Device code of the normcounts sweep: himut's `normcounts.get_callable_tricounts`
// (src/himut/normcounts.py:206-421, with or without --phase) on the machinery of the call path.
//
// Every reference position of the chunks is genotyped from its pile column and, when callable, binned by
// trinucleotide context.  A pile cell is the 16-bit value of the call path (CELL_* code, CELL_INS, BQ) with one
// more bit: "this base counts as callable for its read" (update_tri2count, normcounts.py:66-110).
//
//   k_read_live    sixteen lanes per read: the read filters but the mean quality (normcounts.py:302-309), cs-vs-SEQ check
//   k_callable     wave per read: one bit per query base (the mismatch-window / trim / BQ rules); the mean-quality filter
//                  (it reads every quality anyway), num_ccs
//   k_norm_plan, k_norm_quad (himut_normq.h)
//                  the sweep in use: per 256 positions the list of the reads' gapless pieces over them, then a wave per 256
//                  positions, four columns per lane, no cells in memory; classifies the columns that hold nothing but the
//                  reference allele, lists the others
//   k_norm_dirty   lane per listed position: the general classification (ten PLs twice, PoN / common look-ups)
//   k_norm_tile    round 2's first sweep: workgroup per 256-position tile of a chunk, the cells of the reads over the tile
//                  built in LDS, 48 rows at a time, every thread down its column.  It takes the tiles k_norm_quad leaves
//                  alone (from a list) and the whole contig when a list was too short; all of them evaluate a position with
//                  the same text (NORM_* macros).
#pragma once

#include "himut_kernels.h"

namespace himut {

// ---------------------------------------------------------------------------------------
// phased runs: a read counts in a chunk only if it carries haplotype 0 or 1 there (normcounts.py:293-298)
__global__ void __launch_bounds__(256) k_pair_ccs(Chunks C, Phase H, Reads R, const uint8_t* live, int64_t npairs,
                                                  uint8_t* ccs_flag) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npairs) return;
    const int64_t c = upper_bound(C.pairoff, (int64_t)0, C.n + 1, k) - 1;
    const int64_t r = C.rlo[c] + (k - C.pairoff[c]);
    if (live[r] && H.hap[k] != HAP_NONE) ccs_flag[R.qid[r]] = 1;
}

__global__ void __launch_bounds__(256) k_read_live(Reads R, Derived D, Chunks C, Params P, uint8_t* live, uint8_t* ccs_flag,
                                                   int* err) {
    // sixteen lanes per read: the substitution check runs down the mismatch list sixteen entries at a time (a noisy
    // read has hundreds); the rest is lane 0 of the group
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int gl = threadIdx.x & 15;
    if (r >= R.n) return;
    const ReadMeta M = D.meta[r];
    uint8_t lv = 0;
    if (!(M.flags & RF_SECONDARY)) {
        // substitutions: the base cs names must be the base SEQ holds (the pile takes it from SEQ)
        const int nm = D.nmis[r];
        const uint32_t* mq = D.mq + M.segbase;
        int bad = 0;
        for (int e = gl; e < nm; e += 16) {
            const uint32_t v = mq[e];
            if (v & 16u) {
                const int qa = nib2allele(nib_at(R.seq, M.qoff + (v >> 5)));
                if (qa > 3) bad = HIMUT_ERR_BASE;
                else if (qa != (int)(v & 3u)) bad = HIMUT_ERR_CS;
            }
        }
        if (bad) set_err(err, bad);
        // a read with a base outside ATGC somewhere (k_flag_bases): KeyError in the reference's pile if the base is aligned
        // and some chunk fetches the read (normcounts.py:289,117; every fetched read is piled, whatever its filters say)
        if (R.nonacgt && R.nonacgt[r] && M.nseg > 0) {
            int64_t lo = 0, hi = C.n;
            while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (C.rec[m].start < M.tend) lo = m + 1; else hi = m; }
            if (lo > 0 && C.rec[lo - 1].pmaxend > M.tstart && !aligned_bases_ok(R, D.segs + M.segbase, M.nseg, M.qoff, gl, 16))
                set_err(err, HIMUT_ERR_BASE);
        }
        if (gl != 0) return;
        const int32_t qlen = R.qlen[r];
        bool ok = (M.flags & RF_IDENT_OK) != 0;
        // (the mean quality, the last of the read filters, is k_callable's: it is the kernel that reads every quality)
        if ((int)R.mapq[r] < P.p.min_mapq) ok = false;
        if (!(P.p.qlen_lower_limit < qlen && qlen < P.p.qlen_upper_limit)) ok = false;
        if (ok) {
            // fetched by some chunk: start < tend and end > tstart (normcounts.py:289)
            int64_t lo = 0, hi = C.n;
            while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (C.rec[m].start < M.tend) lo = m + 1; else hi = m; }
            ok = lo > 0 && C.rec[lo - 1].pmaxend > M.tstart;
        }
        if (ok) lv = 1;
    }
    if (gl == 0) live[r] = lv;
}

// ---------------------------------------------------------------------------------------
// k_callable: one wave per read; a word of the bit array = 32 query bases.
// A match base counts when its quality is at least min_bq, it is not trimmed, and the
// number of mismatch-list entries in its window is at most max_mismatch_count; the window
// is the one get_mismatch_range gives for the START of the base's cs match operation,
// shifted along (normcounts.py:84-90), in 0-based coordinates against the 1-based list.
// A substitution always counts (its three tests are evaluated and ignored, :97-109).
//
// Away from every mismatch entry the window test cannot fail and a base's bit is its quality test and its trim test, so
// the kernel runs in two passes.  Pass A streams the read's qualities once (their sum decides the read filter's mean
// quality, normcounts.py:302, bamlib.py:34-36: it is the kernel that reads every quality anyway), 32 bases a lane, and
// writes quality-and-trim words for the whole read.  Pass B takes the words that lie within two windows of a mismatch entry --
// marked beforehand in a bitmap, from the segment boundaries (indels) and the substitutions, a twentieth of the words -- 64
// at a time, a lane each, with the exact rule.  A read that fails a filter gets zeros: it is piled, not counted; a read that
// stays is counted (num_ccs: ccs_flag, unless the run is phased -- k_pair_ccs then).
constexpr int CAL_NM = 256;     // mismatch entries kept in LDS per wave
constexpr int CAL_BM = 64;      // dwords of the bitmap of words to redo: reads of up to 65,536 bases (a longer one: every word exact)
constexpr int CAL_LIST = 128;   // words gathered before a round of pass B

// A read's segment and mismatch lists, in LDS when they fit there (nearly always) or in memory: two types with
// address-space-qualified pointers, chosen once per wave.  (A choice per access -- `k < CAL_NM ? lds[k] : mem[k]` through
// generic pointers -- is compiled into a select between the two POINTERS and a flat load; the chain of those in the
// binary searches made pass B more than half of this kernel's time.)
typedef int cal_v4i __attribute__((ext_vector_type(4)));
struct CalListsLds {
    const __attribute__((address_space(3))) int32_t* mis;
    const __attribute__((address_space(3))) uint32_t* mq;
    const __attribute__((address_space(3))) cal_v4i* seg;
    __device__ __forceinline__ int32_t MIS(int k) const { return mis[k]; }
    __device__ __forceinline__ uint32_t MQ(int k) const { return mq[k]; }
    __device__ __forceinline__ int4 SEG(int j) const { const cal_v4i v = seg[j]; return make_int4(v.x, v.y, v.z, v.w); }
};
struct CalListsMem {
    const __attribute__((address_space(1))) int32_t* mis;
    const __attribute__((address_space(1))) uint32_t* mq;
    const __attribute__((address_space(1))) cal_v4i* seg;
    __device__ __forceinline__ int32_t MIS(int k) const { return mis[k]; }
    __device__ __forceinline__ uint32_t MQ(int k) const { return mq[k]; }
    __device__ __forceinline__ int4 SEG(int j) const { const cal_v4i v = seg[j]; return make_int4(v.x, v.y, v.z, v.w); }
};

// What the tests of one read need (wave-uniform).
struct CalRead {
    const __attribute__((address_space(1))) uint8_t* bq;      // the read's qualities
    __attribute__((address_space(1))) uint32_t* words;        // the read's words of the bit array
    const __attribute__((address_space(1))) uint32_t* nq_top; // N-reference substitutions: query offsets at nq_top[-k] >> 5, ascending
    int32_t qlen, qstart, ns, nm, nN;
    int32_t w, maxmm, min_bq, min_bq_c, trim_lo, trim_hi;

    __device__ __forceinline__ int32_t NQ(int k) const { return (int32_t)(nq_top[-k] >> 5); }

    // the quality and trim tests of the 32 bases from qa on, given their qualities (bytes behind the read are masked by the trim)
    __device__ __forceinline__ uint32_t quality_trim(const uint32_t (&bw)[8], int32_t qa) const {
        uint32_t okq = 0;    // four bytes at a time: the low seven bits compared, bit 7 by itself
        const uint32_t hb = (bw[0] | bw[1] | bw[2] | bw[3] | bw[4] | bw[5] | bw[6] | bw[7]) & 0x80808080u;   // a quality of 128 or more among them
        if (min_bq <= 127 && !hb) {
#pragma unroll
            for (int k = 0; k < 8; k++) okq |= cs_pack4(cs_ge_bytes(bw[k], (uint32_t)min_bq_c)) << (4 * k);
        } else if (min_bq <= 127) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                okq |= cs_pack4(cs_ge_bytes(bw[k] & 0x7f7f7f7fu, (uint32_t)min_bq_c) | (bw[k] & 0x80808080u)) << (4 * k);
        } else if (min_bq <= 255) {                           // (a threshold above 127: only a quality of 128 or more can pass)
#pragma unroll
            for (int k = 0; k < 8; k++)
                okq |= cs_pack4((min_bq > 128 ? cs_ge_bytes(bw[k] & 0x7f7f7f7fu, (uint32_t)(min_bq - 128)) : 0x80808080u) &
                                (bw[k] & 0x80808080u)) << (4 * k);
        }
        if (min_bq <= 0) okq = ~0u;
        // not trimmed: trim_lo <= q <= trim_hi (the two bounds are whole numbers), and q < qlen
        const int32_t lo_q = max(trim_lo, 0), hi_q = min(trim_hi, qlen - 1);
        const int32_t a0 = max(lo_q - qa, 0), a1 = min(hi_q - qa, 31);
        return okq & ((a0 <= a1) ? ((a1 - a0 >= 31 ? ~0u : ((1u << (a1 - a0 + 1)) - 1u)) << a0) : 0u);
    }

    // Which words lie within two windows of a mismatch entry: around every segment boundary (an indel: from the last base in
    // front of it to the first base behind it) and every substitution, in query coordinates.  (A base and an entry that are
    // D reference positions apart with no indel between them are D query bases apart; with indels between them the nearest
    // of those is an entry itself and no further from the base.)  2 w + 2 either way: a base's window reaches w reference
    // positions to each side, up to 2 w to one side where its match operation starts near an end of the read, one more for
    // the 0-based positions against the 1-based list.
    template <class L>
    __device__ __forceinline__ void mark_words(const L& ls, uint32_t* bm, int lane) const {
        const int32_t reach = 2 * w + 2;
        auto mark = [&](int32_t qlo, int32_t qhi) {
            qlo = max(qlo, 0); qhi = min(qhi, qlen - 1);
            for (int32_t wd = qlo >> 5; wd <= (qhi >> 5); wd++) atomicOr(&bm[wd >> 5], 1u << (wd & 31));
        };
        for (int j = lane - 1; j + 1 < ns; j += 64) {             // boundary between segment j and j + 1 (j = -1: in front of the first)
            const int4 sb = ls.SEG(j + 1);
            if (j < 0) { if ((uint32_t)sb.w & SEG_INS) mark(qstart - reach, sb.y + reach); continue; }
            const int4 sa = ls.SEG(j);
            const int32_t qa_end = sa.y + ((((uint32_t)sa.w & SEG_DEL) || sa.z <= 0) ? 0 : sa.z);
            mark(qa_end - 1 - reach, sb.y + reach);
        }
        for (int k = lane; k < nm; k += 64) { const uint32_t v = ls.MQ(k); if (v & 16u) { const int32_t q = (int32_t)(v >> 5); mark(q - reach, q + reach); } }
        for (int k = lane; k < nN; k += 64) { const int32_t q = NQ(k); mark(q - reach, q + reach); }
    }

    // Pass B: the word whose first base is qa, the exact way.
    template <class L>
    __device__ __forceinline__ void exact_word(const L& ls, int32_t qa) const {
        auto lower = [&](int32_t x) { int lo = 0, hi = nm; while (lo < hi) { const int m = (lo + hi) >> 1; if (ls.MIS(m) < x) lo = m + 1; else hi = m; } return lo; };
        auto upper = [&](int32_t x) { int lo = 0, hi = nm; while (lo < hi) { const int m = (lo + hi) >> 1; if (x < ls.MIS(m)) hi = m; else lo = m + 1; } return lo; };
        uint32_t bw[8];
        {
            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
            const v4u b0 = *reinterpret_cast<const __attribute__((address_space(1))) v4u*>(bq + qa);
            const v4u b1 = *reinterpret_cast<const __attribute__((address_space(1))) v4u*>(bq + qa + 16);
            bw[0] = b0.x; bw[1] = b0.y; bw[2] = b0.z; bw[3] = b0.w; bw[4] = b1.x; bw[5] = b1.y; bw[6] = b1.z; bw[7] = b1.w;
        }
        const uint32_t okq = quality_trim(bw, qa);
        uint32_t word = 0;
        // the first segment that reaches behind qa (segments are in query order)
        int jc = 0;
        {
            int lo = 0, hi = ns;
            while (lo < hi) {
                const int m = (lo + hi) >> 1;
                const int4 sg = ls.SEG(m);
                const int32_t qspan = (((uint32_t)sg.w & SEG_DEL) || sg.z <= 0) ? 0 : sg.z;
                if (sg.y + qspan <= qa) lo = m + 1; else hi = m;
            }
            jc = lo;
        }
        for (int j = jc; j < ns; j++) {
            const int4 sg = ls.SEG(j);
            if (sg.y >= qa + 32) break;
            if (((uint32_t)sg.w & SEG_DEL) || sg.z <= 0) continue;
            const int32_t a = max(sg.y, qa), b = min(sg.y + sg.z, qa + 32);     // query overlap
            if (a >= b) continue;
            const int32_t tlo = sg.x + (a - sg.y), thi = sg.x + (b - 1 - sg.y);  // 0-based reference positions
            uint32_t nsub_bits = 0;      // N-reference substitutions among these bases; the last one in front of them
            int32_t nsub_prev = -1;
            for (int k = 0; k < nN; k++) {
                const int32_t q = NQ(k);
                if (q >= a && q < b) nsub_bits |= 1u << (q - qa);
                else if (q >= sg.y && q < a) nsub_prev = q;
            }
            // any list entry that could fall into a window of these bases?
            const int k0 = lower(tlo - 2 * w - 1);
            if (k0 >= nm || ls.MIS(k0) > thi + 2 * w + 1) {
                word |= (okq & (((b - a) >= 32 ? ~0u : ((1u << (b - a)) - 1u)) << (a - qa))) | nsub_bits;
                continue;
            }
            // the few entries near these bases, in registers (value, query offset | substitution flag); a lane with more
            // than NE nearby takes the exact loop over its bases below
            constexpr int NE = 4;
            int32_t ev[NE];
            uint32_t eq[NE];
            int ne = 0;
#pragma unroll
            for (int i = 0; i < NE; i++) {
                ev[i] = 0x7fffffff; eq[i] = 0;
                if (k0 + i < nm) { const int32_t m = ls.MIS(k0 + i); if (m <= thi + 2 * w + 1) { ev[i] = m; eq[i] = ls.MQ(k0 + i); ne = i + 1; } }
            }
            const bool overflow = k0 + NE < nm && ls.MIS(k0 + NE) <= thi + 2 * w + 1;   // more than NE entries nearby
            // start of the match operation the first base belongs to: behind the previous substitution of this
            // segment, else the segment start
            int32_t osq = sg.y;
            {
                // the last entry at or in front of the first base: counted among the entries in registers (those in front of
                // them lie further back still), a search only when all NE of them do and there are more
                int le = 0;
#pragma unroll
                for (int i = 0; i < NE; i++) le += (ev[i] <= tlo) ? 1 : 0;
                const int kp = (le == NE && overflow) ? lower(tlo + 1) - 1 : k0 + le - 1;
                if (kp >= 0) {
                    uint32_t pv;
                    if (kp < k0 || kp >= k0 + NE) pv = ls.MQ(kp);
                    else { pv = eq[0];
#pragma unroll
                        for (int i = 1; i < NE; i++) if (kp - k0 == i) pv = eq[i]; }
                    const int32_t pq = (int32_t)(pv >> 5);
                    if ((pv & 16u) && pq >= sg.y && pq < a) osq = pq + 1;
                }
                if (nsub_prev >= 0) osq = max(osq, nsub_prev + 1);
            }
            // "How many entries see this base" as a sum of bit ranges: bit-sliced counters instead of a loop over the
            // bases.  A match operation that starts at least w bases from either end of the read has the window (w, w)
            // (bamlib.py:245-258) -- every operation that starts among these 32 bases, given where the word lies; the one
            // that started in front of them (osq) may begin within w of the read's start and has (osq, 2 w - osq) then: it
            // holds the bases up to the first substitution among them.
            // (Where none of the bases passes its quality and trim tests -- the trimmed ends of the read -- the counts decide
            // nothing and the same code gives the substitutions' bits.)
            const uint32_t span = ((b - a) >= 32 ? ~0u : ((1u << (b - a)) - 1u)) << (a - qa);
            if (!overflow && ((qa >= w && (int64_t)qa + 32 + w <= (int64_t)qlen) || !(okq & span))) {
                const int32_t shift = sg.y - sg.x - qa;           // bit of reference position t = t + shift
                const int32_t ur0 = min(osq, w), dr0 = 2 * w - ur0;
                uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, sub = 0;
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    const int32_t sq = (int32_t)(eq[i] >> 5);
                    if (i < ne && (eq[i] & 16u) && sq >= a && sq < b && ev[i] == sg.x + (sq - sg.y) + 1) sub |= 1u << (sq - qa);
                }
                const uint32_t starts = sub | nsub_bits;          // behind each of these a new operation begins
                const uint32_t old_op = (ur0 == w) ? 0u : starts ? ((1u << __builtin_ctz(starts)) - 1u) : ~0u;
                auto bits_of = [](int lo, int hi) -> uint32_t {
                    lo = max(lo, 0); hi = min(hi, 31);
                    return lo <= hi ? ((hi - lo) >= 31 ? ~0u : ((1u << (hi - lo + 1)) - 1u)) << lo : 0u;
                };
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    if (i >= ne) continue;
                    uint32_t rr = bits_of(ev[i] - w + shift, ev[i] + w + shift);         // base t sees entry e: e - dr <= t <= e + ur
                    if (old_op) rr = (rr & ~old_op) | (bits_of(ev[i] - dr0 + shift, ev[i] + ur0 + shift) & old_op);
                    const uint32_t k0_ = c0 & rr; c0 ^= rr;
                    const uint32_t k1_ = c1 & k0_; c1 ^= k0_;
                    const uint32_t k2_ = c2 & k1_; c2 ^= k1_;
                    c3 ^= k2_;
                }
                uint32_t gt = 0;                                   // bases seen by more than maxmm entries
                if (maxmm < 8) {
                    uint32_t same = ~0u;
                    const uint32_t planes[4] = {c0, c1, c2, c3};
#pragma unroll
                    for (int pbit = 3; pbit >= 0; pbit--) {
                        const uint32_t kb = ((maxmm >> pbit) & 1) ? ~0u : 0u;
                        gt |= same & planes[pbit] & ~kb;
                        same &= ~(planes[pbit] ^ kb);
                    }
                }
                word |= (span & ((okq & ~gt) | sub)) | nsub_bits;
                continue;
            }
            for (int32_t q = a; q < b; q++) {
                const int32_t t = sg.x + (q - sg.y);
                const int bit = q - qa;
                bool is_sub = false;
#pragma unroll
                for (int i = 0; i < NE; i++) if (ev[i] == t + 1 && (eq[i] & 16u) && (int32_t)(eq[i] >> 5) == q) is_sub = true;
                if (overflow && !is_sub)
                    for (int kk = lower(t + 1); kk < nm && ls.MIS(kk) == t + 1; kk++) { const uint32_t v = ls.MQ(kk); if ((v & 16u) && (int32_t)(v >> 5) == q) is_sub = true; }
                if (is_sub || ((nsub_bits >> bit) & 1u)) { word |= 1u << bit; osq = q + 1; continue; }
                int64_t qs = (int64_t)osq - w, qe = (int64_t)osq + w, ur, dr;      // bamlib.py:245-258
                if (qs < 0) { ur = w + qs; dr = w - qs; }
                else if (qe > qlen) { ur = w + (qe - qlen); dr = qlen - osq; }
                else { ur = w; dr = w; }
                int cnt = 0;
                if (!overflow) {
#pragma unroll
                    for (int i = 0; i < NE; i++) cnt += (ev[i] >= t - ur && ev[i] <= t + dr) ? 1 : 0;
                } else cnt = upper((int32_t)(t + dr)) - lower((int32_t)(t - ur));
                if (cnt <= maxmm && ((okq >> bit) & 1u)) word |= 1u << bit;
            }
        }
        words[qa >> 5] = word;
    }
};

__global__ void __launch_bounds__(256) k_callable(Reads R, Derived D, Params P, uint8_t* live, uint32_t* cbits, uint8_t* ccs_flag) {
    __shared__ int32_t s_mis[4][CAL_NM];
    __shared__ uint32_t s_mq[4][CAL_NM];
    __shared__ __align__(16) int4 s_seg[4][64];
    __shared__ uint32_t s_bm[4][CAL_BM];
    __shared__ uint16_t s_list[4][CAL_LIST];
    const int lane = threadIdx.x & 63, wv = uni((int)(threadIdx.x >> 6));
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= R.n) return;
    const int64_t qo = uni(R.qoff[r]);
    const int32_t qlen = uni(R.qlen[r]);
    if (!uni((int)live[r])) {                                  // no base of it counts (every read's words are written: the array is not cleared first)
        for (int32_t o = lane * 32; o < qlen; o += 2048) cbits[(qo + o) >> 5] = 0;
        return;
    }
    const ReadMeta Mv = D.meta[r];
    const int ns = uni(Mv.nseg);
    const int64_t segbase = uni(Mv.segbase);
    const int nm = uni(D.nmis[r]);
    const Seg* gsegs = D.segs + segbase;
    const int32_t* gmis = D.mis + segbase;
    const uint32_t* gmq = D.mq + segbase;
    uint32_t* bm = s_bm[wv];
    uint16_t* list = s_list[wv];
    const bool in_lds = nm <= CAL_NM && ns <= 64;              // (the wave's choice, made once)
    if (in_lds) {
        for (int k = lane; k < nm; k += 64) { s_mis[wv][k] = gmis[k]; s_mq[wv][k] = gmq[k]; }
        if (lane < ns) s_seg[wv][lane] = *reinterpret_cast<const int4*>(gsegs + lane);
    }
    bm[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    CalListsLds ll;
    ll.mis = (const __attribute__((address_space(3))) int32_t*)s_mis[wv];
    ll.mq = (const __attribute__((address_space(3))) uint32_t*)s_mq[wv];
    ll.seg = (const __attribute__((address_space(3))) cal_v4i*)s_seg[wv];
    CalListsMem lm;
    lm.mis = (const __attribute__((address_space(1))) int32_t*)gmis;
    lm.mq = (const __attribute__((address_space(1))) uint32_t*)gmq;
    lm.seg = (const __attribute__((address_space(1))) cal_v4i*)gsegs;
    CalRead cr;
    cr.bq = (const __attribute__((address_space(1))) uint8_t*)(R.bq + qo);
    cr.words = (__attribute__((address_space(1))) uint32_t*)(cbits + (qo >> 5));
    // substitutions with an N reference base (rare): not in the mismatch list, their bases always count, and each one
    // starts a new match operation behind it (normcounts.py:75-110 walks the cs operations).  Query offsets, ascending.
    cr.nN = uni(D.nnsub[r]);
    cr.nq_top = (const __attribute__((address_space(1))) uint32_t*)(gmq + ((uni(R.cs_off[r + 1]) >> 1) - (uni(R.cs_off[r]) >> 1)));
    cr.qlen = qlen; cr.qstart = uni(R.qstart[r]); cr.ns = ns; cr.nm = nm;
    cr.w = P.p.mismatch_window_size; cr.maxmm = P.p.max_mismatch_count;
    cr.min_bq = P.p.min_bq; cr.min_bq_c = min(max(cr.min_bq, 1), 127);
    cr.trim_lo = (int32_t)floor(P.p.min_trim * (double)qlen);
    cr.trim_hi = (int32_t)ceil((1.0 - P.p.min_trim) * (double)qlen);
    const int32_t nwords = (qlen + 31) >> 5;
    const bool big = nwords > CAL_BM * 32;                      // a read the bitmap does not hold: every word the exact way

    if (!big) { if (in_lds) cr.mark_words(ll, bm, lane); else cr.mark_words(lm, bm, lane); }
    __builtin_amdgcn_wave_barrier();

    // ---- pass A: the qualities once; quality-and-trim words everywhere
    uint32_t qsum = 0;   // this lane's share of the sum of the read's qualities (from offset 0: the mean is over the whole query)
    {
        const int32_t q_last = (max(qlen, 1) - 1) & ~31;       // (addresses clamped into the read)
        uint4 p0, p1;
        {
            const int32_t qn = min(lane * 32, q_last);
            p0 = *reinterpret_cast<const uint4*>(R.bq + qo + qn);
            p1 = *reinterpret_cast<const uint4*>(R.bq + qo + qn + 16);
        }
        for (int32_t c0 = 0; c0 < qlen; c0 += 2048) {
            const int32_t qa = c0 + lane * 32;
            const uint4 b0 = p0, b1 = p1;
            {                                                   // the next window's qualities are asked for before this one is worked on
                const int32_t qn = min(c0 + 2048 + lane * 32, q_last);
                p0 = *reinterpret_cast<const uint4*>(R.bq + qo + qn);
                p1 = *reinterpret_cast<const uint4*>(R.bq + qo + qn + 16);
            }
            if (qa < qlen) {
                const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                if (qa + 32 <= qlen) {
#pragma unroll
                    for (int k = 0; k < 8; k++) qsum = __builtin_amdgcn_sad_u8(bw[k], 0u, qsum);
                } else {                                          // the read's last bases: the bytes behind them are padding
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int rb = qlen - (qa + 4 * k);       // bytes of the word inside the read
                        const uint32_t x = rb >= 4 ? bw[k] : rb <= 0 ? 0u : (bw[k] & (0xffffffffu >> (8 * (4 - rb))));
                        qsum = __builtin_amdgcn_sad_u8(x, 0u, qsum);
                    }
                }
                cbits[(qo + qa) >> 5] = cr.quality_trim(bw, qa);
            }
        }
    }

    // ---- pass B: the words with a mismatch entry nearby (every word of a read the bitmap does not hold), gathered into a
    //      list and worked off 64 at a time, a lane each.  One place in the code for each of the two kinds of lists.
    const int ndw = (nwords + 31) >> 5;
    for (int32_t next = 0;;) {                                  // next: the bitmap dword (big: the word) to go on from
        int nl = 0;
        if (big) {
            nl = min(CAL_LIST, nwords - next);
            for (int i = lane; i < nl; i += 64) list[i] = (uint16_t)i;      // (offsets from `next`: a word number may not fit 16 bits)
        } else {
            for (; next < ndw; next += 2) {
                const uint64_t bits2 = (uint64_t)uni(bm[next]) | ((uint64_t)(next + 1 < ndw ? uni(bm[next + 1]) : 0u) << 32);
                if (!bits2) continue;
                const int here = __builtin_popcountll(bits2);
                if (nl + here > CAL_LIST) break;
                if ((bits2 >> lane) & 1ull) {
                    const int rank = __builtin_popcountll(bits2 & ((1ull << lane) - 1ull));
                    const int32_t wd = next * 32 + lane;
                    list[nl + rank] = (uint16_t)min(wd, nwords - 1);
                }
                nl += here;
            }
        }
        if (nl <= 0) break;
        __builtin_amdgcn_wave_barrier();
        const int32_t wbase = big ? next : 0;
        if (in_lds) { for (int i0 = 0; i0 < nl; i0 += 64) if (i0 + lane < nl) cr.exact_word(ll, (wbase + (int32_t)list[i0 + lane]) * 32); }
        else { for (int i0 = 0; i0 < nl; i0 += 64) if (i0 + lane < nl) cr.exact_word(lm, (wbase + (int32_t)list[i0 + lane]) * 32); }
        __builtin_amdgcn_wave_barrier();
        if (big) next += nl;
    }
    // ---- the mean quality (np.mean of the whole query against min_qv, as in k_read_live's other tests)
    const uint32_t qtot = (uint32_t)lane_val(wave_incl_add((int)qsum, lane), 63);
    if ((double)qtot / (double)qlen < (double)P.p.min_qv) {
        if (lane == 0) live[r] = 0;
        for (int32_t o = lane * 32; o < qlen; o += 2048) cbits[(qo + o) >> 5] = 0;     // no base of it counts
    } else if (!P.p.phase && lane == 0) ccs_flag[R.qid[r]] = 1;
}

// ---------------------------------------------------------------------------------------
struct NormArgs {
    Params P;
    SiteSets S;
    const GtLut* lut;
    Reads R;
    Chunks C;
    Phase H;
    PosIndex X;
    const uint16_t* colstore;
    const uint8_t* refseq;       // the contig as the FASTA holds it
    int64_t reflen;
    uint8_t cls[256];            // byte -> class id
    int K;                       // number of classes
    int cA, cC, cG, cT;          // class ids of the four bases
    uint8_t alt_order[12];       // [ref allele][0..2]: list(base_set.difference(ref)) as alleles
    int non_human;
    int64_t p_lo, p_hi;          // positions of this pass
    unsigned long long* ccs_tri; // [K^3]
    unsigned long long* ref_tri;
    unsigned long long* log;     // [14]
    int* err;
};

// The per-position state and the classification of a position are shared by k_norm_tile and k_norm_dirty as text, so that
// both compile to the same register-resident code (normcounts.py:243-402; gtlib.py:72-174 for the sums).
#define NORM_POS_STATE() \
        uint32_t cnt[6] = {0, 0, 0, 0, 0, 0}; \
        double S[3][4]; \
_Pragma("unroll") \
        for (int b = 0; b < 4; b++) { S[0][b] = 0.0; S[1][b] = 0.0; S[2][b] = 0.0; } \
        double R0 = 0.0, R1 = 0.0, R2 = 0.0; \
        uint32_t nref = 0; \
        uint32_t tri_sum = 0, h0 = 0, h1 = 0; \
        bool bq0 = false;

// The tile sweep's form of the update.  (TH, TT, TE) are the table values of the cell's quality if the cell is the
// reference allele and +0.0 otherwise (the caller reads entry 256, a zero row, for those), so the reference allele's
// sums need no select; the four small counters that every cell may touch travel as 16-bit fields of two words
// (ACC1: insertions | deletions << 16, ACC2: reference bases | callable bases << 16).
#define NORM_CELL_Z(V, USE, ISREF, HP, TH, TT, TE, ACC1, ACC2) \
            { \
                const uint32_t cell = (V) & 7u; \
                const bool use_ = (USE), isref_ = (ISREF); \
                const bool base_ = use_ && cell < 4; \
                const uint32_t q = (V) >> 8; \
                bq0 = bq0 || (base_ && q == 0); \
                R0 = R0 + (TH); R1 = R1 + (TT); R2 = R2 + (TE); \
                if (use_ && !isref_ && cell <= CELL_OTHER) {        /* rare: another allele, or a base outside ATGC */ \
                    if (cell == CELL_OTHER) bad |= 1 << HIMUT_ERR_BASE; \
                    const double vh = s_lut[q], vt = s_lut[257 + q], ve = s_lut[514 + q]; \
_Pragma("unroll") \
                    for (int b = 0; b < 4; b++) { \
                        if ((int)cell == b) { \
                            cnt[b]++; \
                            S[0][b] = S[0][b] + vh; \
                            S[1][b] = S[1][b] + vt; \
                            S[2][b] = S[2][b] + ve; \
                        } \
                    } \
                } \
                uint32_t callable_ = base_ ? (((V) >> 4) & 1u) : 0u; \
                if (phase) { \
                    const uint32_t hp = (HP); \
                    h0 += (base_ && hp == HAP_0) ? 1u : 0u; \
                    h1 += (base_ && hp == HAP_1) ? 1u : 0u; \
                    if (hp != HAP_0 && hp != HAP_1) callable_ = 0; \
                } \
                ACC1 += (use_ ? (((V) >> 3) & 1u) : 0u) | ((use_ && cell == CELL_DEL) ? 0x10000u : 0u); \
                ACC2 += (isref_ ? 1u : 0u) | (callable_ << 16); \
            }

#define NORM_CLASSIFY() \
        if (ref < 0 || tri_sum == 0) continue; \
_Pragma("unroll") \
        for (int b = 0; b < 4; b++) if (b == ref) { cnt[b] = nref; S[0][b] = R0; S[1][b] = R1; S[2][b] = R2; } \
        if (phase && !((int64_t)h0 >= A.P.p.min_hap_count && (int64_t)h1 >= A.P.p.min_hap_count)) { \
            atomicAdd(&s_log[1], tri_sum); \
            atomicAdd(&s_log[2], tri_sum); \
            continue; \
        } \
        if (bq0) { bad |= 1 << HIMUT_ERR_BQ0; continue; } \
        int slot = 1; \
        double best = 0.0, second = 0.0; \
        int ibest = 0; \
_Pragma("unroll") \
        for (int g = 0; g < 10; g++) { \
            const int b1 = (int)HIMUT_GT_B1(g), b2 = (int)HIMUT_GT_B2(g); \
            double acc = 0.0; \
_Pragma("unroll") \
            for (int b = 0; b < 4; b++) { \
                double term; \
                if (b1 == b2 && b == b1) term = S[0][b]; \
                else if (b1 != b2 && (b == b1 || b == b2)) term = S[1][b]; \
                else term = S[2][b]; \
                acc = acc + term; \
            } \
            acc = acc + s_prior[gt_state_of(b1, b2, ref)]; \
            const double pl = -10.0 * acc; \
            if (g == 0) { best = pl; ibest = 0; } \
            else if (pl < best) { second = best; best = pl; ibest = g; } \
            else if (g == 1 || pl < second) second = pl; \
        } \
        const double gqf = second - best; \
        const int gq = gqf < 99.0 ? (int)gqf : 99; \
        const int state = gt_state_of((int)HIMUT_GT_B1(ibest), (int)HIMUT_GT_B2(ibest), ref); \
        const uint32_t depth = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[5]; \
        uint32_t ref_count = 0; \
_Pragma("unroll") \
        for (int b = 0; b < 4; b++) if (b == ref) ref_count = cnt[b]; \
        if (state == 1) slot = 3; \
        else if (state == 2) slot = 4; \
        else if (state == 3) slot = 5; \
        else { \
            atomicAdd(&s_log[6], tri_sum); \
            if (cnt[5] != 0 || cnt[4] != 0) slot = 7; \
            else if ((int64_t)depth > A.P.p.md_threshold) slot = 8; \
            else if (depth == ref_count) { \
                if (gq < A.P.p.min_gq) slot = 10; \
                else if ((int64_t)ref_count < A.P.p.min_ref_count) slot = 9; \
                else slot = 13; \
            } else { \
                bool filtered = false; \
                uint32_t ac[3] = {0, 0, 0}; \
                const int32_t tpos = (int32_t)rpos + 1; \
                const SiteSets& St = A.S; \
                const bool site_maybe = !A.non_human && (int64_t)tpos < St.nposbits && ((St.posbits[tpos >> 5] >> (tpos & 31)) & 1u); \
_Pragma("unroll") \
                for (int a = 0; a < 3; a++) { \
                    if (filtered) continue; \
                    const int aidx = A.alt_order[ref * 3 + a]; \
                    uint32_t c = 0; \
_Pragma("unroll") \
                    for (int b = 0; b < 4; b++) if (b == aidx) c = cnt[b]; \
                    ac[a] = c; \
                    if (c == 0 || !site_maybe) continue; \
                    const uint64_t key = ((uint64_t)(uint32_t)tpos << 4) | ((uint64_t)ref << 2) | (uint64_t)aidx; \
                    if (key_in(St.pon, St.npon, key)) { filtered = true; slot = 11; } \
                    else if (key_in(St.com, St.ncom, key)) { filtered = true; slot = 12; } \
                } \
                if (!filtered) { \
                    int bi = 0; \
                    if (ac[1] > ac[bi]) bi = 1; \
                    if (ac[2] > ac[bi]) bi = 2; \
                    const int aidx = A.alt_order[ref * 3 + bi]; \
                    double b2best = 0.0, b2second = 0.0; \
_Pragma("unroll") \
                    for (int g = 0; g < 10; g++) { \
                        const int b1 = (int)HIMUT_GT_B1(g), b2 = (int)HIMUT_GT_B2(g); \
                        double acc = 0.0; \
_Pragma("unroll") \
                        for (int b = 0; b < 4; b++) { \
                            if (b == aidx) continue; \
                            double term; \
                            if (b1 == b2 && b == b1) term = S[0][b]; \
                            else if (b1 != b2 && (b == b1 || b == b2)) term = S[1][b]; \
                            else term = S[2][b]; \
                            acc = acc + term; \
                        } \
                        acc = acc + s_prior[gt_state_of(b1, b2, ref)]; \
                        const double pl = acc * -10.0; \
                        if (g == 0) b2best = pl; \
                        else if (pl < b2best) { b2second = b2best; b2best = pl; } \
                        else if (g == 1 || pl < b2second) b2second = pl; \
                    } \
                    const double g2 = b2second - b2best; \
                    const int gq2 = g2 < 99.0 ? (int)g2 : 99; \
                    uint32_t alt_count = ac[bi]; \
                    if (gq2 < A.P.p.min_gq) slot = 10; \
                    else if (!((int64_t)ref_count >= A.P.p.min_ref_count && (int64_t)alt_count >= A.P.p.min_alt_count)) slot = 9; \
                    else slot = 13; \
                } \
            } \
        } \
        NORM_TALLY(slot)

// log counters and the trinucleotide bins of a classified position (SLOT: its row of norm.log)
#define NORM_TALLY(SLOT) \
        atomicAdd(&s_log[1], tri_sum); \
        atomicAdd(&s_log[(SLOT)], tri_sum); \
        if ((SLOT) == 13) { \
            NORM_TRIBINS(A.refseq[rpos - 1], A.refseq[rpos + 1]) \
        }

// the trinucleotide bins of a callable position (row 13 of norm.log); PREV, NEXT: the reference's letters beside it
#define NORM_TRIBINS(PREV, NEXT) \
            int t0 = 'N', t1 = 'N', t2 = 'N'; \
            if (rpos - 1 >= 0 && rpos + 2 <= A.reflen) { \
                t0 = (PREV); t1 = refc; t2 = (NEXT); \
                if (t1 == 'A' || t1 == 'G') { \
                    const int a0 = t2, a2 = t0; \
                    t0 = a0 == 'A' ? 'T' : a0 == 'T' ? 'A' : a0 == 'G' ? 'C' : a0 == 'C' ? 'G' : 'N'; \
                    t1 = t1 == 'A' ? 'T' : 'C'; \
                    t2 = a2 == 'A' ? 'T' : a2 == 'T' ? 'A' : a2 == 'G' ? 'C' : a2 == 'C' ? 'G' : 'N'; \
                } \
            } \
            auto acgt = [](int c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }; \
            const int i0 = acgt(t0), i2 = acgt(t2); \
            if (i0 >= 0 && i2 >= 0 && (t1 == 'C' || t1 == 'T')) { \
                const int k = i0 * 8 + (t1 == 'T' ? 4 : 0) + i2; \
                atomicAdd(&s_ccs[k], tri_sum); \
                atomicAdd(&s_ref[k], 1u); \
            } else { \
                const int64_t k = ((int64_t)A.cls[t0] * A.K + A.cls[t1]) * A.K + A.cls[t2]; \
                atomicAdd(&A.ccs_tri[k], (unsigned long long)tri_sum); \
                atomicAdd(&A.ref_tri[k], 1ULL); \
            }

// ---------------------------------------------------------------------------------------
// k_norm_tile: the same sweep without a column store.  A workgroup takes tiles of 256 positions of one chunk; the
// reads that can cover a tile (the window index of its one or two 256-position blocks) are its rows.  Rows are
// handled NT_ROWS at a time: the four waves build the rows' cells in LDS -- a lane owns four consecutive positions
// of a row, finds them in the read's segment list, and loads the four qualities, the four packed bases and the
// four callable bits with one unaligned load each -- then every thread runs down its own column in read order.
// Nothing is written to HBM except the counters.
#ifndef HIMUT_NT_Q
#define HIMUT_NT_Q 16
#endif
#ifndef HIMUT_NT_WAVES
#define HIMUT_NT_WAVES 5
#endif
#ifndef HIMUT_NT_NSG
#define HIMUT_NT_NSG 2
#endif
constexpr int NT_Q = HIMUT_NT_Q;         // workgroups per XCD class and chunk (each strides over its class's tiles)
constexpr int NT_ROWS = 48;
constexpr int NT_RPW = NT_ROWS / 4;    // rows per wave and batch

struct NormRedo { int32_t chunk, base; };          // 256 positions from `base` of chunk `chunk`, left to k_norm_tile by k_norm_quad

__global__ void __launch_bounds__(256, HIMUT_NT_WAVES) k_norm_tile(NormArgs A, Derived D, const uint32_t* callable, const int32_t* winlo,
                                                   const int32_t* winhi, int64_t nblk, int64_t tiles_per_class, const NormRedo* redo,
                                                   const unsigned int* nredo, unsigned int redo_cap) {
    __shared__ double s_lut[3 * 257];         // three tables of 256 qualities + a zero entry each (index 256)
    __shared__ double s_prior[4];
    __shared__ unsigned int s_log[16];
    __shared__ unsigned int s_ccs[32], s_ref[32];
    __shared__ __align__(16) uint16_t s_cells[NT_ROWS][256];
    __shared__ int32_t s_tend[NT_ROWS];
    __shared__ uint32_t s_hap[NT_ROWS];
    // with a list (launched behind k_norm_quad): the tiles that kernel left alone, any workgroup any tile; a list that did
    // not hold them all is the host's business (it repeats the contig without the list)
    const int64_t nlist = redo ? (int64_t)min(*nredo, redo_cap) : 0;
    if (redo && nlist == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6);
    for (int i = tid; i < 3 * 256; i += 256) s_lut[(i >> 8) * 257 + (i & 255)] = A.lut->t[i >> 8][i & 255];
    if (tid < 3) s_lut[tid * 257 + 256] = 0.0;
    if (tid < 4) s_prior[tid] = A.lut->prior[tid];
    if (tid < 16) s_log[tid] = 0;
    if (tid < 32) { s_ccs[tid] = 0; s_ref[tid] = 0; }
    __syncthreads();
    const bool phase = A.P.p.phase != 0;
    const Reads& R = A.R;
    int bad = 0;
    // Which tile: workgroups are dealt round-robin over the eight XCDs (each with an L2 of its own), so the workgroups
    // b, b + 8, b + 16 ... that run side by side on one XCD take NEIGHBOURING tiles of the chunk: a read's bytes at a
    // tile boundary (its rows are 256 + 128 bytes at arbitrary offsets, i.e. partial 128-byte lines at both ends) are
    // then asked for twice within microseconds and the second time come out of that L2.  gridDim.x is a multiple of 8.
    // (Speed only: any mapping gives the same counts.)
    // A workgroup goes through several such tiles (its counters go to memory once): XCD class r = blockIdx.x & 7 owns the
    // tiles [r * per, (r + 1) * per) of the chunk and its NT_Q workgroups take NT_Q neighbouring ones per step.
    // (per: by THIS chunk's length -- with the longest chunk's figure a short chunk would sit on the first XCDs only)
    int64_t per = tiles_per_class;
    if (!redo) {
        const int64_t span = (int64_t)A.C.end[blockIdx.y] - (int64_t)A.C.start[blockIdx.y];
        per = min(per, ((span + 255) / 256 + 7) / 8);
    }
    const int64_t wg = redo ? (int64_t)blockIdx.x + (int64_t)blockIdx.y * gridDim.x : (int64_t)(blockIdx.x >> 3);
    const int64_t wgs = redo ? (int64_t)gridDim.x * gridDim.y : (int64_t)(gridDim.x >> 3);
    for (int64_t t = wg; t < (redo ? nlist : per); t += wgs) {
        const int chunk = redo ? redo[t].chunk : (int)blockIdx.y;
        const int32_t cs_ = A.C.start[chunk], ce_ = A.C.end[chunk];
        const int64_t pairbase = phase ? A.C.pairoff[chunk] - A.C.rlo[chunk] : 0;
        const int64_t tile = (int64_t)(blockIdx.x & 7) * per + t;
        const int64_t base = redo ? (int64_t)redo[t].base : (int64_t)cs_ + tile * 256;
        if (base >= ce_) { if (redo) continue; break; }           // the same for every thread
        const int64_t rpos = base + tid;
        bool valid = rpos < ce_;
        if (valid && (rpos < 0 || rpos >= A.reflen)) { bad |= 1 << HIMUT_ERR_ARG; valid = false; }   // IndexError in the reference
        const int refc = valid ? (int)A.refseq[rpos] : 'N';
        const int ref = char2allele(refc);
        const bool edge = rpos <= cs_;
        const bool any_edge = base <= cs_;                       // only the chunk's first tile has such positions
        NORM_POS_STATE()
        uint32_t acc1 = 0, acc2 = 0;                              // 16-bit fields: insertions | deletions, reference | callable bases
        // rows: the reads of the window index of the blocks under the tile
        const int64_t b0 = min(max(base, (int64_t)0) >> WIN_SHIFT, nblk - 1), b1 = min((base + 255) >> WIN_SHIFT, nblk - 1);
        const int32_t lo = winlo[b0], hi = winhi[b1];
        const int32_t P0 = (int32_t)base + 4 * lane;              // this lane's four positions of every row
        for (int32_t r0 = lo; r0 < hi; r0 += NT_ROWS) {
            const int nb = min(NT_ROWS, hi - r0);
            // ---- the wave's rows, one per lane for the part that is a chain of dependent loads: read header, first
            //      segment that reaches the tile (binary search), the four segments from there on
            const int myrow = wv + 4 * lane;                      // rows wv, wv + 4, ... of the batch
            const bool rowlane = lane < NT_RPW && myrow < nb;
            ReadMeta M;
            M.tstart = 0; M.tend = 0; M.nseg = 0; M.flags = RF_SECONDARY; M.segbase = 0; M.qoff = 0;
            if (rowlane) M = D.meta[r0 + myrow];
            const bool live_row = rowlane && !(M.flags & RF_SECONDARY) && M.nseg > 0 && M.tstart < base + 256 && M.tend >= base;
            int j0 = 0;
            if (live_row) {                                       // last segment that starts at or before the tile
                int a = 0, e = M.nseg;
                while (a < e) { const int m = (a + e) >> 1; if (D.segs[M.segbase + m].t0 <= (int32_t)base) a = m + 1; else e = m; }
                j0 = max(a - 1, 0);
            }
            constexpr int NSG = HIMUT_NT_NSG;          // segments of a row kept in registers (the rest, rarely wanted, come from memory)
            int4 sg[NSG];
#pragma unroll
            for (int k = 0; k < NSG; k++) {
                sg[k] = make_int4(0x7fffffff, 0, 0, 0);
                if (live_row && j0 + k < M.nseg) sg[k] = *reinterpret_cast<const int4*>(D.segs + M.segbase + j0 + k);
            }
            if (rowlane) {
                s_tend[myrow] = M.tend;
                uint32_t hp = HAP_NONE;
                if (phase && live_row && M.tstart < ce_ && M.tend > cs_) hp = A.H.hap[pairbase + r0 + myrow];   // fetched by the chunk
                s_hap[myrow] = hp;
            }
            // ---- one row at a time, four positions per lane
            for (int l = 0; l < NT_RPW; l++) {
                const int row = wv + 4 * l;
                if (row >= nb) break;
                const bool rlive = lane_val((int)live_row, l) != 0;
                uint32_t cell[4] = {CELL_EMPTY, CELL_EMPTY, CELL_EMPTY, CELL_EMPTY};
                // nearly every row: one gapless segment spans the whole tile -- four bases straight from the three loads
                const int32_t f_t0 = lane_val(sg[0].x, l), f_len = lane_val(sg[0].z, l);
                const bool whole = rlive && !((uint32_t)lane_val(sg[0].w, l) & SEG_DEL) && f_t0 <= (int32_t)base &&
                                   (int64_t)f_t0 + f_len >= base + 256;
                if (whole) {
                    const int64_t qoff = ((int64_t)lane_val((int)(M.qoff >> 32), l) << 32) | (uint32_t)lane_val((int)M.qoff, l);
                    const int64_t K = qoff + lane_val(sg[0].y, l) + (P0 - f_t0);
                    uint32_t qv, sb;
                    __builtin_memcpy(&qv, R.bq + K, 4);
                    __builtin_memcpy(&sb, R.seq + (K >> 1), 4);
                    const uint64_t cw = (uint64_t)callable[K >> 5] | ((uint64_t)callable[(K >> 5) + 1] << 32);
                    const uint32_t cb = (uint32_t)(cw >> (K & 31));
                    // base K + y sits in byte (K + y) >> 1, high half when K + y is even: bring the four nibbles to bits 0..15
                    const uint32_t sw = __builtin_bswap32(sb);                 // bytes in nibble order
                    const uint32_t n4 = (K & 1) ? (sw >> 12) & 0xffffu : sw >> 16;   // base y at bits 12 - 4y .. 15 - 4y
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const int nib = (int)((n4 >> (12 - 4 * x)) & 15u);
                        cell[x] = (uint32_t)nib2allele(nib) | (((qv >> (8 * x)) & 0xffu) << 8) | (((cb >> x) & 1u) << 4);
                    }
                    if (f_t0 == (int32_t)base && ((uint32_t)lane_val(sg[0].w, l) & SEG_INS) && lane == 0) cell[0] |= CELL_INS;
                } else if (rlive) {
                    const int ns = lane_val(M.nseg, l), jf = lane_val(j0, l);
                    const int64_t segbase = ((int64_t)lane_val((int)(M.segbase >> 32), l) << 32) | (uint32_t)lane_val((int)M.segbase, l);
                    const int64_t qoff = ((int64_t)lane_val((int)(M.qoff >> 32), l) << 32) | (uint32_t)lane_val((int)M.qoff, l);
                    for (int j = jf; j < ns; j++) {
                        int4 sv;
                        const int k = j - jf;
                        if (k < NSG) {
                            int4 c = sg[0];
#pragma unroll
                            for (int kk = 1; kk < NSG; kk++) if (k == kk) c = sg[kk];
                            sv = make_int4(lane_val(c.x, l), lane_val(c.y, l), lane_val(c.z, l), lane_val(c.w, l));
                        } else {
                            const Seg g = D.segs[segbase + j];
                            sv = make_int4(uni(g.t0), uni(g.q0), uni(g.len), uni((int)g.flags));
                        }
                        const int32_t t0 = sv.x, q0 = sv.y, len = sv.z;
                        const uint32_t fl = (uint32_t)sv.w;
                        if (t0 >= base + 256) break;
                        const int32_t span = len > 0 ? len : ((fl & SEG_INS) ? 1 : 0);      // a trailing insertion marks one position
                        const int32_t a = max(P0, t0), e = min(P0 + 4, t0 + span);
                        if (a >= e) continue;
                        if (fl & SEG_DEL) {
#pragma unroll
                            for (int x = 0; x < 4; x++)
                                if (P0 + x >= a && P0 + x < e) cell[x] = CELL_DEL | ((P0 + x == t0 && (fl & SEG_INS)) ? CELL_INS : 0u);
                        } else if (len == 0) {
#pragma unroll
                            for (int x = 0; x < 4; x++) if (P0 + x == t0) cell[x] = CELL_EMPTY | CELL_INS;
                        } else {
                            // up to four consecutive query bases from K on: qualities, packed bases (high nibble first) and
                            // callable bits, each with one unaligned load (the buffers carry slack behind the last read)
                            const int64_t K = qoff + q0 + (a - t0);
                            uint32_t qv, sb;
                            __builtin_memcpy(&qv, R.bq + K, 4);
                            __builtin_memcpy(&sb, R.seq + (K >> 1), 4);
                            const uint64_t cw = (uint64_t)callable[K >> 5] | ((uint64_t)callable[(K >> 5) + 1] << 32);
                            const uint32_t cb = (uint32_t)(cw >> (K & 31));
                            // nibble of base K + y: byte (K + y) >> 1, high half when K + y is even
                            const uint32_t odd = (uint32_t)(K & 1);
#pragma unroll
                            for (int x = 0; x < 4; x++) {
                                const int y = P0 + x - a;                                   // index among the loaded bases
                                if (y >= 0 && P0 + x < e) {
                                    const uint32_t kk = (uint32_t)y + odd;                  // nibble index from the first loaded byte
                                    const uint32_t byte = (sb >> (8 * (kk >> 1))) & 0xffu;
                                    const int nib = (kk & 1) ? (int)(byte & 15u) : (int)(byte >> 4);
                                    uint32_t val = (uint32_t)nib2allele(nib) | (((qv >> (8 * y)) & 0xffu) << 8) | (((cb >> y) & 1u) << 4);
                                    if (P0 + x == t0 && (fl & SEG_INS)) val |= CELL_INS;
                                    cell[x] = val;
                                }
                            }
                        }
                    }
                }
                uint2 packed;
                packed.x = cell[0] | (cell[1] << 16);
                packed.y = cell[2] | (cell[3] << 16);
                *reinterpret_cast<uint2*>(&s_cells[row][4 * lane]) = packed;
            }
            __syncthreads();
            // ---- every thread down its column, in read order
            if (valid) {
                // NB rows at a time: the cells, then the three table values of each (the zero row unless the cell is the
                // reference allele), are loaded before any of them is used: the LDS latency is paid once per NB cells
                constexpr int NB = 2;      // (four at a time costs more in spills than the extra LDS round trips save)
                for (int i0 = 0; i0 < nb; i0 += NB) {
                    uint32_t v4[NB];
                    double th[NB], tt[NB], te[NB];
                    bool use4[NB], ref4[NB];
#pragma unroll
                    for (int k = 0; k < NB; k++) v4[k] = i0 + k < nb ? (uint32_t)s_cells[i0 + k][tid] : (uint32_t)CELL_EMPTY;
#pragma unroll
                    for (int k = 0; k < NB; k++) {
                        const uint32_t v = v4[k];
                        const int ri = min(i0 + k, nb - 1);
                        // an EMPTY cell, or a read this chunk did not fetch (normcounts.py:289), adds nothing
                        use4[k] = (v & 15u) != CELL_EMPTY && !(any_edge && edge && !(s_tend[ri] > cs_));
                        ref4[k] = use4[k] && (int)(v & 7u) == ref;
                        const uint32_t qe = ref4[k] ? (v >> 8) : 256u;      // the zero row for everything but the reference allele
                        th[k] = s_lut[qe]; tt[k] = s_lut[257 + qe]; te[k] = s_lut[514 + qe];
                    }
#pragma unroll
                    for (int k = 0; k < NB; k++)
                        NORM_CELL_Z(v4[k], use4[k], ref4[k], s_hap[min(i0 + k, nb - 1)], th[k], tt[k], te[k], acc1, acc2)
                }
            }
            __syncthreads();
            if (((r0 - lo) / NT_ROWS & 511) == 511) {             // a pile tens of thousands of reads deep: empty the 16-bit fields
                cnt[4] += acc1 & 0xffffu; cnt[5] += acc1 >> 16; nref += acc2 & 0xffffu; tri_sum += acc2 >> 16;
                acc1 = 0; acc2 = 0;
            }
        }
        if (!valid) continue;
        cnt[4] += acc1 & 0xffffu; cnt[5] += acc1 >> 16; nref += acc2 & 0xffffu; tri_sum += acc2 >> 16;
        NORM_CLASSIFY()
    }
    __syncthreads();
    if (tid < 14 && s_log[tid]) atomicAdd(&A.log[tid], (unsigned long long)s_log[tid]);
    if (tid < 32 && (s_ccs[tid] || s_ref[tid])) {
        const int cl[4] = {A.cA, A.cC, A.cG, A.cT};
        const int64_t k = ((int64_t)cl[tid >> 3] * A.K + ((tid & 4) ? A.cT : A.cC)) * A.K + cl[tid & 3];
        atomicAdd(&A.ccs_tri[k], (unsigned long long)s_ccs[tid]);
        atomicAdd(&A.ref_tri[k], (unsigned long long)s_ref[tid]);
    }
    if (bad) atomicOr(A.err, bad);
}

// ---------------------------------------------------------------------------------------
// A position k_norm_quad does not classify itself -- its column holds a base of another allele than the reference's, or
// hom-ref is not the smallest of its genotype sums by itself: everything the column walk knows about it.  One position in
// thirty; k_norm_dirty takes them a lane each, where inside k_norm_quad the general classification would run for a lane or
// two of a wave, every other wave, and set the kernel's registers.
struct NormDirty {
    int64_t rpos;
    uint32_t nref, tri_sum, n_ins, n_del, h0, h1;
    uint32_t cnt[4];          // bases of A, T, G, C that are not the reference allele's
    double R[3];              // the reference allele's three sums
    double S[9];              // [table * 3 + slot]: the other alleles', allele c in slot c - (c > ref)
};


__global__ void __launch_bounds__(256) k_norm_dirty(NormArgs A, const NormDirty* recs, const uint32_t* dcount, int64_t cap, int64_t nregions) {
    __shared__ double s_prior[4];
    __shared__ unsigned int s_log[16];
    __shared__ unsigned int s_ccs[32], s_ref[32];
    const int tid = threadIdx.x;
    if (tid < 4) s_prior[tid] = A.lut->prior[tid];
    if (tid < 16) s_log[tid] = 0;
    if (tid < 32) { s_ccs[tid] = 0; s_ref[tid] = 0; }
    __syncthreads();
    const bool phase = A.P.p.phase != 0;
    int bad = 0;
    // the list is in `nregions` parts of `cap` entries (one part per wave of the sweep, filled from its start, a third full or
    // less): a wave per part
    const int lane = tid & 63;
    for (int64_t region = ((int64_t)blockIdx.x * 256 + tid) >> 6; region < nregions; region += (int64_t)gridDim.x * 4)
    for (int64_t i = lane; i < (int64_t)min((int64_t)dcount[region], cap); i += 64) {
        const int64_t t = region * cap + i;
        const NormDirty d = recs[t];
        const int64_t rpos = d.rpos;
        const int refc = (int)A.refseq[rpos];
        const int ref = char2allele(refc);
        uint32_t cnt[6] = {d.cnt[0], d.cnt[1], d.cnt[2], d.cnt[3], d.n_ins, d.n_del};
        double S[3][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int a = max(min(b - (b > ref ? 1 : 0), 2), 0);     // (the reference allele's own entries are replaced in the classification)
            double v0 = d.S[0], v1 = d.S[3], v2 = d.S[6];
            if (a == 1) { v0 = d.S[1]; v1 = d.S[4]; v2 = d.S[7]; }
            if (a == 2) { v0 = d.S[2]; v1 = d.S[5]; v2 = d.S[8]; }
            S[0][b] = v0; S[1][b] = v1; S[2][b] = v2;
        }
        double R0 = d.R[0], R1 = d.R[1], R2 = d.R[2];
        uint32_t nref = d.nref, tri_sum = d.tri_sum, h0 = d.h0, h1 = d.h1;
        const bool bq0 = false;                                       // (a zero quality ended the column in k_norm_quad)
        NORM_CLASSIFY()
    }
    __syncthreads();
    if (tid < 14 && s_log[tid]) atomicAdd(&A.log[tid], (unsigned long long)s_log[tid]);
    if (tid < 32 && (s_ccs[tid] || s_ref[tid])) {
        const int cl[4] = {A.cA, A.cC, A.cG, A.cT};
        const int64_t k = ((int64_t)cl[tid >> 3] * A.K + ((tid & 4) ? A.cT : A.cC)) * A.K + cl[tid & 3];
        atomicAdd(&A.ccs_tri[k], (unsigned long long)s_ccs[tid]);
        atomicAdd(&A.ref_tri[k], (unsigned long long)s_ref[tid]);
    }
    if (bad) atomicOr(A.err, bad);
}

// ---------------------------------------------------------------------------------------
// k_ref_tricounts: reflib.get_chrom_tricount (reflib.py:11-33) over the resident reference string: every
// triplet whose three letters are upper-case A/C/G/T, purine centres read on the other strand; 64 bins
// indexed first * 16 + centre * 4 + last with A0 C1 G2 T3 (only the 32 pyrimidine-centred ones fill).
__global__ void __launch_bounds__(256) k_ref_tricounts(const uint8_t* seq, int64_t len, unsigned long long* out) {
    __shared__ unsigned int s_h[64];
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    auto code = [](int c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 2 < len; i += (int64_t)gridDim.x * blockDim.x) {
        const int a = code(seq[i]), b = code(seq[i + 1]), d = code(seq[i + 2]);
        if (a > 3 || b > 3 || d > 3) continue;
        const bool pur = b == 0 || b == 2;
        const int f = pur ? 3 - d : a, m = pur ? 3 - b : b, l = pur ? 3 - a : d;
        atomicAdd(&s_h[f * 16 + m * 4 + l], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_h[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------
// k_sbs96: mutlib.get_sbs96 + the counting of load_sbs96_counts (mutlib.py:1998-2018, 2058-2102) over the resident
// reference string: one thread per called single-base substitution (0-based position, ASCII ref / alt as the VCF
// holds them).  A purine reference base is reported on the other strand, its neighbours through the
// purine2pyrimidine table (anything outside ACGTN becomes N); a pyrimidine one takes its neighbours as the string
// holds them.  out[0 .. 95]: class (substitution C>A C>G C>T T>A T>C T>G) * 16 + upstream * 4 + downstream with
// A0 C1 G2 T3; out[96]: classes that contain an N (the reference drops them); out[97]: classes outside the 96 with no N
// (KeyError in the reference: a lower-case neighbour of a pyrimidine, an alt outside ACGT); out[98]: position + 1
// behind the string (IndexError).  Position 0 takes its upstream base from the END of the string, as python's seq[-1] does.
__global__ void __launch_bounds__(256) k_sbs96(const uint8_t* seq, int64_t len, const int32_t* pos, const uint8_t* ref,
                                               const uint8_t* alt, int64_t n, unsigned long long* out) {
    __shared__ unsigned int s_h[99];
    if (threadIdx.x < 99) s_h[threadIdx.x] = 0;
    __syncthreads();
    auto code = [](int c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : 5; };   // 5: any other byte
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = pos[i];
        if (p < 0 || p + 1 >= len) { atomicAdd(&s_h[98], 1u); continue; }
        const int r = ref[i], a = alt[i];
        const int before = seq[p > 0 ? p - 1 : len - 1], after = seq[p + 1];
        int up, rf, al, dn;         // codes 0..3, 4 = N, 5 = a letter the class list does not have
        if (r == 'A' || r == 'G') {
            auto comp = [&](int c) { const int k = code(c); return k < 4 ? 3 - k : 4; };     // purine2pyrimidine.get(c, "N")
            up = comp(after); dn = comp(before); rf = comp(r); al = comp(a);
        } else {
            up = code(before); dn = code(after); rf = code(r); al = code(a);
        }
        if (up == 4 || dn == 4 || rf == 4 || al == 4) { atomicAdd(&s_h[96], 1u); continue; }
        if (up > 3 || dn > 3 || al > 3 || (rf != 1 && rf != 3) || al == rf) { atomicAdd(&s_h[97], 1u); continue; }
        const int sub = (rf == 1 ? 0 : 3) + (al > rf ? al - 1 : al);       // C>A C>G C>T | T>A T>C T>G
        atomicAdd(&s_h[sub * 16 + up * 4 + dn], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 99 && s_h[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------
// k_edges: phaselib.get_edges (phaselib.py:16-67).  One wave per read, lanes = the heterozygous SNPs the read
// spans (tstart < pos <= tend).  Every lane finds its SNP's segment by binary search over the read's segment
// list and reads the base and its quality (a deleted position has quality 0, cslib.py:153-170); every ordered
// pair of lanes whose qualities reach min_bq adds one to cis1 / cis2 / trans1 / trans2 of its edge.  The edge
// table is banded: counts[(i * band + (j - i - 1)) * 4 + k].
__global__ void __launch_bounds__(256) k_edges(Reads R, Derived D, const int32_t* hpos, const uint8_t* href, int64_t nhet,
                                               int min_bq, int min_mapq, int64_t band, uint32_t* counts, int* err) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + uni((int)(threadIdx.x >> 6));
    if (r >= R.n) return;
    const ReadMeta Mv = D.meta[r];
    const int mapq = uni((int)R.mapq[r]);
    if ((uni(Mv.flags) & RF_SECONDARY) || mapq < min_mapq) return;
    const int32_t tstart = uni(Mv.tstart), tend = uni(Mv.tend);
    const int ns = uni(Mv.nseg);
    const Seg* segs = D.segs + uni(Mv.segbase);
    const int64_t qo = uni(Mv.qoff);
    const int64_t idx = upper_bound(hpos, (int64_t)0, nhet, tstart), jdx = upper_bound(hpos, (int64_t)0, nhet, tend);
    const int64_t k = jdx - idx;
    if (k < 2) return;
    for (int64_t a0 = 0; a0 < k; a0 += 64) {                   // lanes = SNPs a0 .. a0 + 63 as the first of a pair
        const int64_t a = a0 + lane;
        int st_a = 0;
        bool ok_a = false;
        auto look = [&](int64_t g, int& st, bool& ok) {          // state and usability of hetSNP g for this read
            const int32_t rpos = hpos[g] - 1;
            int lo = 0, hi = ns;                                // last segment that starts at or before rpos
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (segs[mid].t0 <= rpos) lo = mid; else hi = mid; }
            const Seg sg = segs[lo];
            int qb = 0, bq = 0;
            if (rpos >= sg.t0 && rpos < sg.t0 + sg.len) {
                if (sg.flags & SEG_DEL) qb = '-';
                else {
                    const int32_t q = sg.q0 + (rpos - sg.t0);
                    qb = nib2char(nib_at(R.seq, qo + q));
                    bq = R.bq[qo + q];
                }
            } else set_err(err, HIMUT_ERR_COVER);                // KeyError in tpos2qbase
            ok = bq >= min_bq;
            st = (qb == (int)href[g]) ? 0 : 1;
        };
        if (a < k) look(idx + a, st_a, ok_a);
        // second of the pair: the SNPs behind a, a block of 64 at a time (one look-up per lane, then broadcast)
        for (int64_t b0 = a0; b0 < k; b0 += 64) {
            int st_bb = st_a;
            bool ok_bb = ok_a;
            if (b0 != a0) { st_bb = 0; ok_bb = false; if (b0 + lane < k) look(idx + b0 + lane, st_bb, ok_bb); }
            const int nb = (int)min((int64_t)64, k - b0);
            for (int t = 0; t < nb; t++) {
                const int64_t b = b0 + t;
                const int st_b = lane_val(st_bb, t);
                if (!lane_val((int)ok_bb, t)) continue;
                if (a < k && a < b && ok_a) {
                    if (b - a - 1 >= band) { set_err(err, HIMUT_ERR_ARG); continue; }
                    const int kk = (!st_a && !st_b) ? 0 : (st_a && st_b) ? 1 : (!st_a && st_b) ? 2 : 3;
                    atomicAdd(&counts[((idx + a) * band + (b - a - 1)) * 4 + kk], 1u);
                }
            }
        }
    }
}

}  // namespace himut
