// DEFLATE (RFC 1951) decoder for one BGZF block per LANE: the inflate of a BAM file on the GPU.
//
// A BAM is a series of BGZF blocks (RFC 1952 members of at most 64 KB of payload each, SAM specification 4.1), each an
// independent raw DEFLATE stream: a chr20-sized 30x CCS file has 45 k of them.  DEFLATE is sequential inside a stream
// -- every code's position depends on the lengths of all codes before it -- but the streams do not depend on each
// other, and 45 k lanes are about what the GPU holds at once.  So: one lane per block, the lane's tables in LDS
// (interleaved by lane), a 64-bit bit buffer per lane refilled four bytes at a time, output bytes stored straight to
// HBM, matches copied from the lane's own output.  No lane talks to another.
//
// What costs on a GPU is divergence: the 64 lanes of a wave sit in 64 different places of 64 different streams, and a
// wave runs every branch any of its lanes takes.  The decoder is therefore a state machine that advances every lane by
// one bounded MICRO-STEP per turn (inf_step): a few literal / length symbols, sixteen bytes of a match, one code-length
// symbol of a dynamic block's header ...  Literal / length codes of up to eight bits -- nearly all of them in a BAM,
// whose bytes are mostly base qualities -- are one table look-up (a 256-entry table per lane, filled when the block's
// code is built); longer codes and the distance codes are decoded length by length from the canonical description
// (counts per length + symbols sorted by length, RFC 1951 3.2.2).
//
// Written once for host and device (INF_HD): tests/test_inflate.py runs it on the CPU against zlib on streams of every
// kind (stored, fixed, dynamic; every compression level and strategy), the -m gpu tests run the kernel against zlib on
// the same streams and on whole BAM files.  There is no reference code for this row: the reference reads BAM through
// pysam / htslib (caller.py:267).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define INF_HD __host__ __device__ __forceinline__
#else
#define INF_HD inline
#endif
// A load inside a rarely taken branch is waited for INSIDE the branch (the empty asm uses its result there): otherwise
// the wait sits at the join behind the branch, where every lane's turn passes, and there it waits for every store in
// flight as well (loads and stores retire through one counter): a quarter of a microsecond per literal.
#if defined(__HIP_DEVICE_COMPILE__)
#define INF_LANDED(x) asm volatile("" : "+v"(x))
#else
#define INF_LANDED(x) (void)0
#endif
// host builds count what a stream is made of (tools/inflate_stats.py): [0] symbols by table, [1] by the long-code path,
// [2] matches, [3] match bytes, [4] dynamic headers, [5] code-length symbols, [6] / [7] matches at distances <= 64 / <= 1024
#if !defined(__HIPCC__)
#define INF_STAT(k, n) (inf_stats()[k] += (n))
inline long long* inf_stats() { static long long s[8]; return s; }
#else
#define INF_STAT(k, n) (void)0
#endif

namespace himut {

enum InfErr {
    INF_OK = 0,
    INF_ERR_BTYPE = 1,      // block type 3
    INF_ERR_STORED = 2,     // stored block: LEN != ~NLEN
    INF_ERR_LENGTHS = 3,    // code lengths: too many, repeat without a first length, over-subscribed or incomplete set
    INF_ERR_CODE = 4,       // a code that is in no table
    INF_ERR_DIST = 5,       // distance too far back, or symbol 30 / 31
    INF_ERR_OUTPUT = 6,     // more (or less) output than the block's ISIZE
    INF_ERR_INPUT = 7,      // the stream runs past the block's compressed bytes
};

// Per-decoder tables; element i of an array lives at [i * stride] (stride 64 on the device: the arrays of a wave's 64
// lanes are interleaved, so that lanes reading the same entry hit different banks).
struct InfTables {
    uint16_t* lut;    // [256] literal/length codes of <= 8 bits by their (bit-reversed, as they arrive) low bits: symbol | length << 9; 0 = a longer code
    uint16_t* lcnt;   // [16]  number of literal/length codes of each length
    uint16_t *lnext, *lpos;   // [16] each: scratch while the code is built
    uint16_t* lsym;   // [288] the literal/length symbols with codes LONGER than 8 bits, sorted by length then symbol (may be slow memory)
    uint8_t* lens;    // [320] code lengths while a dynamic block's tables are read
    uint8_t* dsym;    // [32]  distance symbols sorted by length then symbol (the code-length code's while a header is read)
    uint8_t* dcnt;    // [16]
    uint8_t* off;     // [16]  scratch while a table is built
    uint32_t* ring;   // [32]  the next 128 bytes of the lane's input (inf_fill)
    int stride;       // of lut, lcnt, lens, dsym, dcnt, off, ring
    int lstride;      // of lsym
};
constexpr int INF_LDS_U16 = 256 + 16 + 16 + 16;    // uint16 entries per decoder that want fast memory
constexpr int INF_LDS_U8 = 320 + 32 + 16 + 16;     // uint8 entries
constexpr int INF_LDS_U32 = 32;                    // uint32 entries (the input ring)

enum InfMode { INF_M_HEADER = 0, INF_M_LENS, INF_M_BUILD, INF_M_SYMS, INF_M_COPY, INF_M_STORED, INF_M_DONE };

struct InfChunk { uint32_t w[8]; };      // 32 bytes of input on their way from memory
struct InfLane {
    const uint8_t* in;
    int64_t pos, end;      // next input byte to take from the ring, end of the compressed bytes
    uint64_t buf;
    int cnt;               // valid bits in buf
    InfChunk pend;         // the chunk behind the ones in the ring: loaded when the ring last had room, stored when it next has
    uint8_t* out;
    int64_t o, out_len;
    int mode, err, last;
    int nlen, ndist, idx;  // a dynamic header being read
    int nshort;            // literal/length symbols with codes of <= 8 bits (they are not in lsym)
    int rem, dist;         // bytes left of a match (its distance) or of a stored block
};

// At least 32 valid bits.  The input comes through a ring of 128 bytes per lane, 32-byte chunks: a load of the lane's
// input from memory takes a microsecond or two, and a lane that waited for one at every fourth symbol would do little
// else.  The ring holds the three chunks behind the read position; the fourth is in flight (pend) from the moment the
// ring has room for it until the next chunk has been used up -- some forty symbols later -- and only then is waited for.
// (The input buffer carries 160 bytes of slack behind its end: chunks are loaded past it, never used.)
INF_HD void inf_chunk_load(InfChunk& c, const uint8_t* p) { memcpy(c.w, p, 32); }
INF_HD void inf_chunk_store(const InfChunk& c, uint32_t* ring, int slot, int st) {
    for (int k = 0; k < 8; k++) ring[(slot * 8 + k) * st] = c.w[k];
}
INF_HD void inf_ring_begin(InfLane& b, uint32_t* ring, int st) {
    InfChunk c0, c1, c2;
    inf_chunk_load(c0, b.in); inf_chunk_load(c1, b.in + 32); inf_chunk_load(c2, b.in + 64);
    inf_chunk_load(b.pend, b.in + 96);
    inf_chunk_store(c0, ring, 0, st); inf_chunk_store(c1, ring, 1, st); inf_chunk_store(c2, ring, 2, st);
}
INF_HD void inf_fill(InfLane& b, uint32_t* ring, int st) {
    if (b.cnt <= 32) {
        const uint32_t w = ring[((b.pos >> 2) & 31) * st];
        b.buf |= (uint64_t)w << b.cnt;
        b.pos += 4;
        b.cnt += 32;
        if ((b.pos & 31) == 0) {
            // a chunk has been used up: the chunk in flight takes the free slot (three chunks ahead of the read position
            // once more), and the one behind it sets out
            inf_chunk_store(b.pend, ring, (int)(((b.pos >> 5) + 2) & 3), st);
            inf_chunk_load(b.pend, b.in + b.pos + 96);
        }
    }
}
INF_HD uint32_t inf_bits(InfLane& b, const InfTables& T, int n) {     // n <= 16
    inf_fill(b, T.ring, T.stride);
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}
INF_HD void inf_fail(InfLane& b, int e) { b.err = e; b.mode = INF_M_DONE; }

// one symbol of a canonical code (counts per length; sym = the symbols sorted by length, from the skip-th on): length by
// length, RFC 1951 3.2.2.  -1: no such code.
template <class S, class C>
INF_HD int inf_decode(InfLane& b, const InfTables& T, const C* cnt, int cstride, const S* sym, int sstride, int skip) {
    inf_fill(b, T.ring, T.stride);
    uint32_t bits = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)cnt[len * cstride];
        if (code - count < first) {
            b.buf >>= len;
            b.cnt -= len;
            const int k = index + (code - first) - skip;
            if (k < 0) return -1;
            int s = (int)sym[k * sstride];
            INF_LANDED(s);
            return s;
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// counts per length of n code lengths; 0 for a complete set, < 0 over-subscribed, > 0 incomplete
template <class C>
INF_HD int inf_count(const uint8_t* lens, int n, C* cnt, int stride) {
    for (int l = 0; l <= 15; l++) cnt[l * stride] = 0;
    for (int i = 0; i < n; i++) { const int l = lens[i * stride]; cnt[l * stride] = (C)(cnt[l * stride] + 1); }
    int left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= (int)cnt[l * stride];
        if (left < 0) return left;
    }
    return left;
}
// the symbols of a small code (<= 32 symbols) sorted by length then symbol
INF_HD void inf_sort_small(const uint8_t* lens, int n, const uint8_t* cnt, uint8_t* sym, uint8_t* off, int stride) {
    off[1 * stride] = 0;
    for (int l = 1; l < 15; l++) off[(l + 1) * stride] = (uint8_t)(off[l * stride] + cnt[l * stride]);
    for (int i = 0; i < n; i++) {
        const int l = lens[i * stride];
        if (l) { sym[off[l * stride] * stride] = (uint8_t)i; off[l * stride] = (uint8_t)(off[l * stride] + 1); }
    }
}
// The literal/length code from nlen lengths: the look-up table of the codes of <= 8 bits, the sorted list of the symbols
// with longer ones.  Codes are handed out in order of length, then symbol (RFC 1951 3.2.2); a code arrives least
// significant bit first, so the table is indexed by the code's bits reversed, every value of the bits behind it.
INF_HD void inf_build_litlen(InfLane& b, const InfTables& T, int nlen) {
    const int st = T.stride;
    for (int i = 0; i < 256; i++) T.lut[i * st] = 0;
    // first code of every length, and where the symbols of every length > 8 begin in lsym
    uint32_t code = 0;
    int nshort = 0, nlong = 0;
    for (int l = 1; l <= 15; l++) {
        const int c = (int)T.lcnt[l * st];
        T.lnext[l * st] = (uint16_t)code;
        code = (code + (uint32_t)c) << 1;
        if (l <= 8) nshort += c;
        else { T.lpos[l * st] = (uint16_t)nlong; nlong += c; }
    }
    for (int i = 0; i < nlen; i++) {
        const int l = (int)T.lens[i * st];
        if (!l) continue;
        const uint32_t cd = T.lnext[l * st];
        T.lnext[l * st] = (uint16_t)(cd + 1);
        if (l <= 8) {
            uint32_t rev = 0;
            for (int k = 0; k < l; k++) rev |= ((cd >> k) & 1u) << (l - 1 - k);
            const uint16_t e = (uint16_t)(i | (l << 9));
            for (uint32_t j = rev; j < 256; j += 1u << l) T.lut[j * st] = e;
        } else {
            const int p_ = (int)T.lpos[l * st];
            T.lpos[l * st] = (uint16_t)(p_ + 1);
            T.lsym[p_ * T.lstride] = (uint16_t)i;
        }
    }
    b.nshort = nshort;
}

INF_HD void inf_begin(InfLane& b, const InfTables& T, const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_len) {
    b.in = in; b.pos = 0; b.end = in_len; b.buf = 0; b.cnt = 0;
    inf_ring_begin(b, T.ring, T.stride);
    b.out = out; b.o = 0; b.out_len = out_len;
    b.mode = INF_M_HEADER; b.err = 0; b.last = 0;
    b.nlen = b.ndist = b.idx = b.nshort = b.rem = b.dist = 0;
}

INF_HD void inf_end_of_block(InfLane& b) {
    if (!b.last) { b.mode = INF_M_HEADER; return; }
    b.mode = INF_M_DONE;
    if (b.o != b.out_len) b.err = INF_ERR_OUTPUT;
    else if (b.pos - (b.cnt >> 3) > b.end) b.err = INF_ERR_INPUT;
}

constexpr int INF_SYMS_PER_STEP = 4, INF_COPY_PER_STEP = 32;

// One micro-step of a lane that is not done.
INF_HD void inf_step(InfLane& b, const InfTables& T) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int st = T.stride;
    if (b.mode == INF_M_SYMS) {
        for (int k = 0; k < INF_SYMS_PER_STEP; k++) {
            inf_fill(b, T.ring, st);
            int sym;
            const uint32_t e = T.lut[((uint32_t)b.buf & 255u) * st];
            if (e) {
                const int l = (int)(e >> 9);
                b.buf >>= l; b.cnt -= l;
                sym = (int)(e & 511u);
                INF_STAT(0, 1);
            } else {
                INF_STAT(1, 1);
                sym = inf_decode(b, T, T.lcnt, st, T.lsym, T.lstride, b.nshort);
                if (sym < 0) { inf_fail(b, INF_ERR_CODE); return; }
            }
            if (sym < 256) {
                if (b.o >= b.out_len) { inf_fail(b, INF_ERR_OUTPUT); return; }
                b.out[b.o++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256) { inf_end_of_block(b); return; }
            sym -= 257;
            if (sym >= 29) { inf_fail(b, INF_ERR_CODE); return; }
            const int len = (int)lbase[sym] + (int)inf_bits(b, T, lext[sym]);
            const int ds = inf_decode(b, T, T.dcnt, st, T.dsym, st, 0);
            if (ds < 0) { inf_fail(b, INF_ERR_CODE); return; }
            if (ds >= 30) { inf_fail(b, INF_ERR_DIST); return; }
            const int dist = (int)dbase[ds] + (int)inf_bits(b, T, dext[ds]);
            if ((int64_t)dist > b.o) { inf_fail(b, INF_ERR_DIST); return; }
            if (b.o + len > b.out_len) { inf_fail(b, INF_ERR_OUTPUT); return; }
            b.rem = len; b.dist = dist; b.mode = INF_M_COPY;
            INF_STAT(2, 1); INF_STAT(3, len); INF_STAT(6, dist <= 64 ? 1 : 0); INF_STAT(7, dist <= 1024 ? 1 : 0);
            return;
        }
    } else if (b.mode == INF_M_COPY) {
        // A load of the lane's own output waits for every store before it (loads and stores retire in order): the fewer
        // loads the better.  Source and destination at least eight bytes apart: eight bytes per load and store (the
        // buffers carry eight bytes of slack for what is READ past the written output; nothing is written past a match).
        // Closer: the dist bytes in front of the match are the period of everything it writes.
        const int n = b.rem < INF_COPY_PER_STEP ? b.rem : INF_COPY_PER_STEP;
        uint8_t* o = b.out + b.o;
        if (b.dist >= 8) {
            int i = 0;
            for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, o + i - b.dist, 8); memcpy(o + i, &v, 8); }
            if (i < n) {                                  // the last bytes: one load, byte stores (nothing is written past the match)
                uint64_t v;
                memcpy(&v, o + i - b.dist, 8);
                for (int k = 0; i + k < n; k++) o[i + k] = (uint8_t)(v >> (8 * k));
            }
        } else {
            uint64_t pat;
            memcpy(&pat, o - b.dist, 8);
            int j = 0;
            for (int i = 0; i < n; i++) { o[i] = (uint8_t)(pat >> (8 * j)); j = j + 1 == b.dist ? 0 : j + 1; }
        }
        b.o += n; b.rem -= n;
        if (!b.rem) b.mode = INF_M_SYMS;
    } else if (b.mode == INF_M_LENS) {
        // one symbol of the code-length code (its table sits where the distance table will be)
        const int total = b.nlen + b.ndist;
        const int sym = inf_decode(b, T, T.dcnt, st, T.dsym, st, 0);
        INF_STAT(5, 1);
        if (sym < 0) { inf_fail(b, INF_ERR_CODE); return; }
        if (sym < 16) T.lens[(b.idx++) * st] = (uint8_t)sym;
        else {
            int prev = 0, rep;
            if (sym == 16) {
                if (b.idx == 0) { inf_fail(b, INF_ERR_LENGTHS); return; }
                prev = T.lens[(b.idx - 1) * st];
                rep = 3 + (int)inf_bits(b, T, 2);
            } else if (sym == 17) rep = 3 + (int)inf_bits(b, T, 3);
            else rep = 11 + (int)inf_bits(b, T, 7);
            if (b.idx + rep > total) { inf_fail(b, INF_ERR_LENGTHS); return; }
            while (rep--) T.lens[(b.idx++) * st] = (uint8_t)prev;
        }
        if (b.idx == total) b.mode = INF_M_BUILD;
    } else if (b.mode == INF_M_BUILD) {
        if (T.lens[256 * st] == 0) { inf_fail(b, INF_ERR_LENGTHS); return; }               // no end-of-block code
        // incomplete sets are allowed only as a single one-bit code (zlib's rule)
        int r = inf_count(T.lens + (int64_t)b.nlen * st, b.ndist, T.dcnt, st);
        if (r < 0 || (r > 0 && b.ndist != (int)T.dcnt[0] + (int)T.dcnt[1 * st])) { inf_fail(b, INF_ERR_LENGTHS); return; }
        inf_sort_small(T.lens + (int64_t)b.nlen * st, b.ndist, T.dcnt, T.dsym, T.off, st);
        r = inf_count(T.lens, b.nlen, T.lcnt, st);
        if (r < 0 || (r > 0 && b.nlen != (int)T.lcnt[0] + (int)T.lcnt[1 * st])) { inf_fail(b, INF_ERR_LENGTHS); return; }
        inf_build_litlen(b, T, b.nlen);
        b.mode = INF_M_SYMS;
    } else if (b.mode == INF_M_STORED) {
        const int n = b.rem < INF_COPY_PER_STEP ? b.rem : INF_COPY_PER_STEP;
        for (int i = 0; i < n; i++) b.out[b.o++] = (uint8_t)inf_bits(b, T, 8);
        b.rem -= n;
        if (!b.rem) inf_end_of_block(b);
    } else if (b.mode == INF_M_HEADER) {
        const uint32_t hdr = inf_bits(b, T, 3);
        b.last = (int)(hdr & 1u);
        const int type = (int)(hdr >> 1);
        if (type == 0) {
            // stored: to the byte boundary, LEN, NLEN, LEN bytes
            const int drop = b.cnt & 7;
            b.buf >>= drop; b.cnt -= drop;
            const uint32_t len = inf_bits(b, T, 16), nlen = inf_bits(b, T, 16);
            if (len != (~nlen & 0xffffu)) { inf_fail(b, INF_ERR_STORED); return; }
            if (b.o + (int64_t)len > b.out_len) { inf_fail(b, INF_ERR_OUTPUT); return; }
            b.rem = (int)len;
            if (len) b.mode = INF_M_STORED; else inf_end_of_block(b);
        } else if (type == 1) {
            // fixed code: 8 bits for 0..143, 9 for 144..255, 7 for 256..279, 8 for 280..287; distances 5 bits
            for (int i = 0; i < 288; i++) T.lens[i * st] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
            (void)inf_count(T.lens, 288, T.lcnt, st);
            inf_build_litlen(b, T, 288);
            for (int i = 0; i < 30; i++) T.lens[i * st] = 5;
            (void)inf_count(T.lens, 30, T.dcnt, st);
            inf_sort_small(T.lens, 30, T.dcnt, T.dsym, T.off, st);
            b.mode = INF_M_SYMS;
        } else if (type == 2) {
            b.nlen = (int)inf_bits(b, T, 5) + 257; b.ndist = (int)inf_bits(b, T, 5) + 1;
            const int ncode = (int)inf_bits(b, T, 4) + 4;
            if (b.nlen > 286 || b.ndist > 30) { inf_fail(b, INF_ERR_LENGTHS); return; }
            for (int i = 0; i < 19; i++) T.lens[i * st] = 0;
            for (int i = 0; i < ncode; i++) T.lens[clorder[i] * st] = (uint8_t)inf_bits(b, T, 3);
            if (inf_count(T.lens, 19, T.dcnt, st) != 0) { inf_fail(b, INF_ERR_LENGTHS); return; }   // a complete set is required here
            inf_sort_small(T.lens, 19, T.dcnt, T.dsym, T.off, st);
            b.idx = 0;
            b.mode = INF_M_LENS;
            INF_STAT(4, 1);
        } else inf_fail(b, INF_ERR_BTYPE);
    }
}

// Inflates one raw DEFLATE stream of in_len bytes into exactly out_len bytes (host form: a lane of its own).  `in` must
// be readable for 160 bytes past in_len, `out` writable for 8 bytes past out_len (and for 8 bytes in front of it when a
// damaged stream names a short distance at the very start: the caller's buffers carry that slack).  Returns INF_OK or an InfErr.
INF_HD int inf_stream(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_len, const InfTables& T) {
    InfLane b;
    inf_begin(b, T, in, in_len, out, out_len);
    while (b.mode != INF_M_DONE) inf_step(b, T);
    return b.err;
}

}  // namespace himut
