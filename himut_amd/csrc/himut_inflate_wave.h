// k_bgzf_inflate_wave: DEFLATE (RFC 1951) with one WAVE per BGZF block -- the sequential part on the scalar unit.
//
// The lane-per-block decoder (himut_inflate.h) is correct and slow: a lane is a sequential machine that gets an
// instruction through every eight cycles or so when its wave has a SIMD to itself, and 45 k blocks are all the
// parallelism a chr20-sized file has.  Here a block belongs to a wave.  What is sequential in DEFLATE -- the bit
// buffer, the symbol-by-symbol walk -- is the same in every lane of the wave; what is parallel uses the lanes: the
// tables of a dynamic block are built by 64 lanes at once (a symbol each: its canonical code from a ballot per code
// length, its entries of the look-up table), a match is copied by as many lanes as it has bytes, the output goes
// through a window in LDS and leaves for HBM in coalesced 256-byte rows.  Measured: 14 GB/s of
// inflated bytes on our BAM blocks (the lane decoder: 6; the host's pool: 10), 300 GB/s on stored blocks and long
// matches.  The walk is ~40 wave-uniform operations per symbol, and a CU gets about one such operation through per
// cycle whether it sits on the four vector units (this build: the compiler keeps the state in vector registers) or is
// pinned to the one scalar unit (tried: slower at seven waves per SIMD than this at three): 45 k blocks x 28 k
// symbols x 40 operations are 80 ms of a whole MI355X.
//
//   input     a 256-byte row of the block's compressed bytes sits in one VGPR (lane i = dword i); the bit buffer
//             (scalar, 64 bits) takes its next dword with a readlane; the next row is loaded a row ahead
//   tables    in LDS, per wave: a 512-entry look-up for literal/length codes of up to 9 bits, a 256-entry one for
//             distance codes of up to 8 bits, and the canonical description (counts per length, symbols sorted by
//             length) for the few longer codes; read with a uniform address (one LDS broadcast) into a scalar
//   window    the last 2 KB of output per wave in LDS; matches that reach further back read the bytes from HBM,
//             where they have been flushed to (everything more than 512 bytes back has)
//
// Device only (readlane / ballot); checked against zlib by tests/test_gpu_inflate.py on streams of every kind and on
// every block of BAM files, as the lane decoder is.
#pragma once
#include "himut_ingest.h"

namespace himut {

constexpr int IW_WIN = 2048;                 // bytes of output kept in LDS per wave (a power of two)
constexpr int IW_ROW = 256;                  // bytes per flush row / input row
constexpr int IW_LBITS = 9, IW_DBITS = 8;    // look-up table widths

struct IwTables {                            // per wave, in LDS
    uint16_t llut[1 << IW_LBITS];            // literal/length: symbol | length << 9 (0: a longer code)
    uint16_t dlut[1 << IW_DBITS];            // distance: symbol | length << 5
    uint16_t lsym[288];                      // literal/length symbols sorted by length then symbol (for the long codes)
    uint16_t dsym[32];
    uint32_t lcnt[16], dcnt[16];             // codes per length
    uint32_t lnext[16];                      // scratch of the build: next code / next place per length
    uint32_t lplace[16];
    uint8_t lens[320];                       // the code lengths of the block being set up
    uint8_t win[IW_WIN];
};

// a wave-uniform value in a scalar register
__device__ __forceinline__ uint32_t iw_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t iw_u64(uint64_t v) { return ((uint64_t)iw_u((uint32_t)(v >> 32)) << 32) | iw_u((uint32_t)v); }

struct IwBits {                              // all uniform
    const uint8_t* in;                       // the block's compressed bytes
    uint32_t row;                            // index of the 256-byte row held in `cur`
    uint32_t dw;                             // next dword of the input to take (absolute index)
    uint64_t buf;
    int cnt;
};

// the bit buffer gets its next 32 bits; cur / nxt: this lane's dword of the current and the next input row
__device__ __forceinline__ void iw_fill(IwBits& b, uint32_t& cur, uint32_t& nxt, int lane) {
    if (b.cnt <= 32) {
        if ((b.dw >> 6) != b.row) {          // the row is used up: the next one is here already, the one behind it sets out
            cur = nxt;
            b.row++;
            __builtin_memcpy(&nxt, b.in + (size_t)(b.row + 1) * IW_ROW + 4 * lane, 4);
        }
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(b.dw & 63u));
        b.buf |= (uint64_t)w << b.cnt;
        b.cnt += 32;
        b.dw++;
    }
}
__device__ __forceinline__ uint32_t iw_bits(IwBits& b, uint32_t& cur, uint32_t& nxt, int lane, int n) {   // n <= 16
    iw_fill(b, cur, nxt, lane);
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// canonical decode, length by length (RFC 1951 3.2.2), for the codes the look-up tables do not hold; -1: no such code
__device__ __forceinline__ int iw_decode_slow(IwBits& b, const uint32_t* cnt, const uint16_t* sym) {
    uint32_t bits = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)iw_u(cnt[len]);
        if (code - count < first) {
            b.buf >>= len;
            b.cnt -= len;
            return (int)iw_u(sym[index + (code - first)]);
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// The tables of one code from n code lengths (lens[0 .. n)), by the whole wave: counts per length (a ballot per length
// and row of 64 symbols), the first code of every length, every symbol's own code = first code of its length + its rank
// among the symbols of that length, its place in the sorted list, and -- for codes of at most `bits` bits -- its
// entries of the look-up table (the code arrives bit-reversed, followed by any bits).  Returns 0 for a complete code,
// < 0 for an over-subscribed one, > 0 for an incomplete one.
template <int SHIFT>
__device__ __forceinline__ int iw_build(const uint8_t* lens, int n, uint32_t* cnt, uint16_t* sym, uint32_t* next, uint32_t* place,
                                        uint16_t* lut, int bits, int lane) {
    if (lane < 16) cnt[lane] = 0;
    for (int i = lane; i < (1 << bits); i += 64) lut[i] = 0;
    __builtin_amdgcn_wave_barrier();
    // counts
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int l = i0 + lane < n ? (int)lens[i0 + lane] : 0;
        for (int len = 1; len <= 15; len++) {
            const unsigned long long m = __ballot(l == len);
            if (m && lane == 0) cnt[len] += (uint32_t)__popcll(m);
        }
    }
    __builtin_amdgcn_wave_barrier();
    int left = 1;
    uint32_t code = 0, pos = 0;
    for (int len = 1; len <= 15; len++) {
        const uint32_t c = iw_u(cnt[len]);
        left <<= 1;
        left -= (int)c;
        if (left < 0) return left;
        if (lane == 0) { next[len] = code; place[len] = pos; }
        code = (code + c) << 1;
        pos += c;
    }
    __builtin_amdgcn_wave_barrier();
    // every symbol: its code and place
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int l = i < n ? (int)lens[i] : 0;
        uint32_t my_code = 0, my_place = 0;
        for (int len = 1; len <= 15; len++) {
            const unsigned long long m = __ballot(l == len);
            if (!m) continue;
            const uint32_t base_code = iw_u(next[len]), base_place = iw_u(place[len]);
            const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (l == len) { my_code = base_code + below; my_place = base_place + below; }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { next[len] = base_code + (uint32_t)__popcll(m); place[len] = base_place + (uint32_t)__popcll(m); }
            __builtin_amdgcn_wave_barrier();
        }
        if (l) {
            sym[my_place] = (uint16_t)i;
            if (l <= bits) {
                const uint32_t rev = __brev(my_code) >> (32 - l);
                const uint16_t e = (uint16_t)(i | (l << SHIFT));
                for (uint32_t j = rev; j < (1u << bits); j += 1u << l) lut[j] = e;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return left;
}

// output rows that are complete leave for memory: 256 bytes, a dword per lane
__device__ __forceinline__ void iw_flush_rows(const uint8_t* win, uint8_t* out, uint32_t pos, uint32_t& flushed, int lane) {
    while (pos - flushed >= (uint32_t)IW_ROW) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(win + ((flushed + 4u * lane) & (IW_WIN - 1)));
        __builtin_memcpy(out + flushed + 4u * lane, &w, 4);          // (out + flushed need not be aligned)
        flushed += IW_ROW;
    }
}

__global__ void __launch_bounds__(256) k_bgzf_inflate_wave(const uint8_t* comp, const BgzfBlock* blocks, int64_t nblocks, uint8_t* out_all,
                                                           int* status) {
    __shared__ IwTables s_tab[4];
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int lane = threadIdx.x & 63, wv = (int)iw_u(threadIdx.x >> 6);
    const int64_t blk = (int64_t)blockIdx.x * 4 + wv;
    if (blk >= nblocks) return;
    IwTables& T = s_tab[wv];
    const BgzfBlock B = blocks[blk];
    const uint32_t isize = iw_u(B.isize);
    if (!isize) return;
    uint8_t* out = out_all + (((uint64_t)iw_u((uint32_t)(B.uoff >> 32)) << 32) | iw_u((uint32_t)B.uoff));
    IwBits b;
    b.in = comp + iw_u(B.coff);
    b.row = 0; b.dw = 0; b.buf = 0; b.cnt = 0;
    const uint32_t clen = iw_u(B.clen);
    uint32_t cur, nxt;
    __builtin_memcpy(&cur, b.in + 4 * lane, 4);                    // (comp carries 1 KB of slack behind its end)
    __builtin_memcpy(&nxt, b.in + IW_ROW + 4 * lane, 4);
    uint32_t pos = 0, flushed = 0;                                 // bytes produced / bytes that have left for memory
    int err = 0;
    bool last = false;
    while (!last && !err) {
        const uint32_t hdr = iw_bits(b, cur, nxt, lane, 3);
        last = (hdr & 1u) != 0;
        const int type = (int)(hdr >> 1);
        if (type == 0) {
            // stored: to the byte boundary, LEN, NLEN, then LEN bytes straight from the input
            const int drop = b.cnt & 7;
            b.buf >>= drop; b.cnt -= drop;
            const uint32_t len = iw_bits(b, cur, nxt, lane, 16), nlen = iw_bits(b, cur, nxt, lane, 16);
            if (len != (~nlen & 0xffffu)) { err = INF_ERR_STORED; break; }
            if (pos + len > isize) { err = INF_ERR_OUTPUT; break; }
            // the bytes behind what the bit buffer has taken (it holds whole bytes now)
            const uint32_t src0 = b.dw * 4u - (uint32_t)(b.cnt >> 3);
            if (src0 + len > clen) { err = INF_ERR_INPUT; break; }
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
                if (i0 + lane < len) T.win[(pos + i0 + lane) & (IW_WIN - 1)] = b.in[src0 + i0 + lane];
                __builtin_amdgcn_wave_barrier();
                const uint32_t done = min(len, i0 + 64);
                iw_flush_rows(T.win, out, pos + done, flushed, lane);
            }
            pos += len;
            // the bit buffer starts again behind the stored bytes
            const uint32_t nb = src0 + len;
            b.dw = nb >> 2; b.buf = 0; b.cnt = 0;
            b.row = b.dw >> 6;
            __builtin_memcpy(&cur, b.in + (size_t)b.row * IW_ROW + 4 * lane, 4);
            __builtin_memcpy(&nxt, b.in + (size_t)(b.row + 1) * IW_ROW + 4 * lane, 4);
            if (nb & 3u) (void)iw_bits(b, cur, nxt, lane, 8 * (int)(nb & 3u));
            continue;
        }
        if (type == 3) { err = INF_ERR_BTYPE; break; }
        int nl, nd;
        if (type == 1) {
            nl = 288; nd = 30;
            for (int i = lane; i < 288; i += 64) T.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
            if (lane < 30) T.lens[288 + lane] = 5;
            __builtin_amdgcn_wave_barrier();
        } else {
            nl = (int)iw_bits(b, cur, nxt, lane, 5) + 257;
            nd = (int)iw_bits(b, cur, nxt, lane, 5) + 1;
            const int ncode = (int)iw_bits(b, cur, nxt, lane, 4) + 4;
            if (nl > 286 || nd > 30) { err = INF_ERR_LENGTHS; break; }
            if (lane < 19) T.lens[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            for (int i = 0; i < ncode; i++) {
                const uint32_t v = iw_bits(b, cur, nxt, lane, 3);
                if (lane == 0) T.lens[iw_u(clorder[i])] = (uint8_t)v;
            }
            __builtin_amdgcn_wave_barrier();
            // the code-length code: its tables sit where the distance code's will be
            if (iw_build<5>(T.lens, 19, T.dcnt, T.dsym, T.lnext, T.lplace, T.dlut, 7, lane) != 0) { err = INF_ERR_LENGTHS; break; }
            int idx = 0;
            const int total = nl + nd;
            while (idx < total && !err) {
                iw_fill(b, cur, nxt, lane);
                int sym;
                const uint32_t e = iw_u(T.dlut[(uint32_t)b.buf & 127u]);
                if (e) { const int l = (int)(e >> 5); b.buf >>= l; b.cnt -= l; sym = (int)(e & 31u); }
                else sym = -1;                                   // (a code-length code has at most 7 bits: every code is in the table)
                if (sym < 0) { err = INF_ERR_CODE; break; }
                if (sym < 16) { if (lane == 0) T.lens[idx] = (uint8_t)sym; idx++; }
                else {
                    int prev = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) { err = INF_ERR_LENGTHS; break; }
                        __builtin_amdgcn_wave_barrier();
                        prev = (int)iw_u(T.lens[idx - 1]);
                        rep = 3 + (int)iw_bits(b, cur, nxt, lane, 2);
                    } else if (sym == 17) rep = 3 + (int)iw_bits(b, cur, nxt, lane, 3);
                    else rep = 11 + (int)iw_bits(b, cur, nxt, lane, 7);
                    if (idx + rep > total) { err = INF_ERR_LENGTHS; break; }
                    for (int i = lane; i < rep; i += 64) T.lens[idx + i] = (uint8_t)prev;
                    idx += rep;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (err) break;
            if (iw_u(T.lens[256]) == 0) { err = INF_ERR_LENGTHS; break; }          // no end-of-block code
        }
        {
            // incomplete codes are allowed only as a single one-bit code (zlib's rule)
            int r = iw_build<5>(T.lens + nl, nd, T.dcnt, T.dsym, T.lnext, T.lplace, T.dlut, IW_DBITS, lane);
            if (r < 0 || (r > 0 && type == 2 && nd != (int)iw_u(T.dcnt[0]) + (int)iw_u(T.dcnt[1]))) { err = INF_ERR_LENGTHS; break; }
            r = iw_build<9>(T.lens, nl, T.lcnt, T.lsym, T.lnext, T.lplace, T.llut, IW_LBITS, lane);
            if (r < 0 || (r > 0 && type == 2 && nl != (int)iw_u(T.lcnt[0]) + (int)iw_u(T.lcnt[1]))) { err = INF_ERR_LENGTHS; break; }
        }
        // ---- the symbols of the block
        for (;;) {
            iw_fill(b, cur, nxt, lane);
            int sym;
            {
                const uint32_t e = iw_u(T.llut[(uint32_t)b.buf & ((1u << IW_LBITS) - 1u)]);
                if (e) { const int l = (int)(e >> 9); b.buf >>= l; b.cnt -= l; sym = (int)(e & 511u); }
                else sym = iw_decode_slow(b, T.lcnt, T.lsym);
            }
            if (sym < 0) { err = INF_ERR_CODE; break; }
            if (sym < 256) {
                if (pos >= isize) { err = INF_ERR_OUTPUT; break; }
                if (lane == 0) T.win[pos & (IW_WIN - 1)] = (uint8_t)sym;
                pos++;
                if ((pos & (IW_ROW - 1)) == 0) { __builtin_amdgcn_wave_barrier(); iw_flush_rows(T.win, out, pos, flushed, lane); }
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = INF_ERR_CODE; break; }
            const uint32_t len = iw_u((uint32_t)lbase[sym]) + iw_bits(b, cur, nxt, lane, (int)iw_u(lext[sym]));
            iw_fill(b, cur, nxt, lane);
            int ds;
            {
                const uint32_t e = iw_u(T.dlut[(uint32_t)b.buf & ((1u << IW_DBITS) - 1u)]);
                if (e) { const int l = (int)(e >> 5); b.buf >>= l; b.cnt -= l; ds = (int)(e & 31u); }
                else ds = iw_decode_slow(b, T.dcnt, T.dsym);
            }
            if (ds < 0) { err = INF_ERR_CODE; break; }
            if (ds >= 30) { err = INF_ERR_DIST; break; }
            const uint32_t dist = iw_u((uint32_t)dbase[ds]) + iw_bits(b, cur, nxt, lane, (int)iw_u(dext[ds]));
            if (dist > pos) { err = INF_ERR_DIST; break; }
            if (pos + len > isize) { err = INF_ERR_OUTPUT; break; }
            // ---- the match: as many lanes as bytes.  Byte i repeats byte i mod dist of the dist bytes in front of it, so
            // every lane reads bytes that exist already; they are in the window unless the match reaches further back
            // than the window keeps, and those bytes have long left for memory.
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i0 = 0; i0 < len; i0 += 64) {
                const uint32_t i = i0 + lane;
                if (i < len) {
                    const uint32_t src = pos - dist + (dist >= len ? i : i % dist);
                    uint8_t v;
                    if (pos + i0 - src <= (uint32_t)(IW_WIN - 64 - 64)) v = T.win[src & (IW_WIN - 1)];
                    else v = out[src];
                    T.win[(pos + i) & (IW_WIN - 1)] = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
            pos += len;
            iw_flush_rows(T.win, out, pos, flushed, lane);
        }
    }
    if (!err) {
        // the last bytes (less than a row)
        __builtin_amdgcn_wave_barrier();
        for (uint32_t p = flushed + lane; p < pos; p += 64) out[p] = T.win[p & (IW_WIN - 1)];
        if (pos != isize) err = INF_ERR_OUTPUT;
        else if (b.dw * 4u - (uint32_t)(b.cnt >> 3) > clen) err = INF_ERR_INPUT;
    }
    if (err && lane == 0) atomicOr(status, 1 << err);
}

}  // namespace himut
