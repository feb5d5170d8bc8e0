// Device side of the BAM ingest (reference call sites caller.py:267,299 and the record wrapper
// bamlib.BAM.__init__, bamlib.py:14-32): inflated BAM records arrive in HBM a window at a time and are turned into the
// read batch of himut_push_reads (include/himut_hip.h, himut_read_batch) without going through host arrays.
//
//   k_bam_decode    one thread per record: the fixed fields, the CIGAR walk (reference length, leading soft clip --
//                   reference_end / query_alignment_start of pysam) and the walk over the auxiliary fields for
//                   cs:Z and tp:A
//   scan            (rocPRIM) running offsets of the padded query lengths and of the cs lengths inside the window
//   k_bam_scatter   one wave per record: SEQ (4-bit packed, as BAM has it), QUAL and the cs text go to their place in
//                   the contig's arrays with 16-byte stores (query offsets are multiples of 32 bases), lane 0 writes
//                   the per-read fields
//   k_bam_advance   the running totals of the contig
//
// The host's part (csrc/bam_ingest.cpp, bam_stream_*): BGZF inflate by a thread pool straight into pinned memory, the
// hop from length field to length field, the read-name table (qid).  Integer / byte work, HBM bound.
#pragma once

#include "himut_kernels.h"

namespace himut {

struct RecDesc {
    uint32_t seq_off, qual_off, cs_off;   // offsets in the window
    uint32_t l_seq, cs_len;
    int32_t pos, ref_len, lead_clip;
    uint32_t flag_mapq_tp;                // flag | mapq << 16 | tp << 24
    uint32_t status;                      // 0 ok, 1 malformed, 2 no cs tag
};

struct IngestState {
    unsigned long long n_reads, bases_padded, cs_n, read_bases;
    unsigned long long n_missing_cs, n_unsorted, n_bad;
    int last_pos, any_longcs, overflow, pad;
};

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

__global__ void __launch_bounds__(256) k_bam_decode(const uint8_t* win, int64_t nbytes, const uint32_t* rec_off, int64_t nrec,
                                                    RecDesc* desc, uint2* sizes) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nrec) return;
    RecDesc d;
    d.seq_off = d.qual_off = d.cs_off = d.l_seq = d.cs_len = 0; d.pos = 0; d.ref_len = 0; d.lead_clip = 0; d.flag_mapq_tp = 0; d.status = 1;
    const uint32_t ro = rec_off[k];
    uint2 sz = make_uint2(0u, 0u);
    if ((int64_t)ro >= 4 && (int64_t)ro + 32 <= nbytes) {
        const uint8_t* rec = win + ro;
        const uint32_t bs = ld_u32(rec - 4);
        if (bs >= 32 && (int64_t)ro + bs <= nbytes) {
            d.pos = (int32_t)ld_u32(rec + 4);
            const uint32_t l_qname = rec[8], mapq = rec[9], n_cigar = ld_u16(rec + 12), flag = ld_u16(rec + 14);
            const uint32_t l_seq = ld_u32(rec + 16);
            uint64_t o = 32ull + l_qname;
            if (o + 4ull * n_cigar + (l_seq + 1ull) / 2 + l_seq <= bs) {
                // CIGAR: reference length (M D N = X) and the leading soft clip
                int64_t ref_len = 0;
                int32_t lead = 0;
                bool seen_query = false;
                for (uint32_t c = 0; c < n_cigar; c++) {
                    const uint32_t v = ld_u32(rec + o + 4ull * c);
                    const uint32_t op = v & 15u, ln = v >> 4;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += ln;
                    if (op == 4 && !seen_query) lead += (int32_t)ln;
                    if (op == 0 || op == 1 || op == 7 || op == 8) seen_query = true;
                }
                o += 4ull * n_cigar;
                d.seq_off = ro + (uint32_t)o;
                o += (l_seq + 1ull) / 2;
                d.qual_off = ro + (uint32_t)o;
                o += l_seq;
                d.l_seq = l_seq; d.ref_len = (int32_t)ref_len; d.lead_clip = lead;
                // auxiliary fields: cs:Z and tp:A
                const uint8_t* p = rec + o;
                const uint8_t* end = rec + bs;
                uint32_t tp = 0;
                bool ok = true, have_cs = false;
                while (p + 3 <= end) {
                    const int t0 = p[0], t1 = p[1], ty = p[2];
                    p += 3;
                    uint64_t sz_ = 0;
                    if (ty == 'A' || ty == 'c' || ty == 'C') sz_ = 1;
                    else if (ty == 's' || ty == 'S') sz_ = 2;
                    else if (ty == 'i' || ty == 'I' || ty == 'f') sz_ = 4;
                    else if (ty == 'Z' || ty == 'H') {
                        const uint8_t* q = p;
                        while (q < end && *q) q++;
                        if (q >= end) { ok = false; break; }
                        if (t0 == 'c' && t1 == 's' && ty == 'Z') { d.cs_off = (uint32_t)(p - win); d.cs_len = (uint32_t)(q - p); have_cs = true; }
                        p = q + 1;
                        continue;
                    } else if (ty == 'B') {
                        if (p + 5 > end) { ok = false; break; }
                        const int sub = p[0];
                        const uint32_t cnt = ld_u32(p + 1);
                        const uint64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                        if ((uint64_t)(end - p) < 5 + es * (uint64_t)cnt) { ok = false; break; }     // the array runs past the record
                        p += 5 + es * cnt;
                        continue;
                    } else { ok = false; break; }
                    if (p + sz_ > end) { ok = false; break; }
                    if (t0 == 't' && t1 == 'p' && ty == 'A') tp = p[0];
                    p += sz_;
                }
                d.flag_mapq_tp = flag | (mapq << 16) | (tp << 24);
                d.status = !ok ? 1u : (have_cs ? 0u : 2u);
                if (ok) sz = make_uint2((l_seq + 31u) & ~31u, d.cs_len);
            }
        }
    }
    desc[k] = d;
    sizes[k] = sz;
}

struct IngestOut {
    int32_t *tstart, *tend, *qstart, *qlen, *qid;
    uint8_t *mapq, *tp;
    uint16_t* flag;
    int64_t *qoff, *cs_off;
    uint8_t *seq, *bq, *cs;
    int64_t cap_reads, cap_bases, cap_cs;
};

__global__ void __launch_bounds__(256) k_bam_scatter(const uint8_t* win, const RecDesc* desc, const uint2* offs, const int32_t* qid_in,
                                                     int64_t nrec, IngestOut O, IngestState* S) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nrec) return;
    const RecDesc d = desc[k];
    if (d.status == 1) { if (lane == 0) atomicAdd(&S->n_bad, 1ull); return; }
    const uint2 o = offs[k];
    const int64_t idx = (int64_t)S->n_reads + k;
    const int64_t qoff = (int64_t)S->bases_padded + o.x, cso = (int64_t)S->cs_n + o.y;
    const int64_t padded = ((int64_t)d.l_seq + 31) & ~(int64_t)31;
    if (idx >= O.cap_reads || qoff + padded > O.cap_bases || cso + d.cs_len > O.cap_cs) { if (lane == 0) S->overflow = 1; return; }
    // SEQ: (l_seq + 1) / 2 bytes, the low nibble of an odd last byte and the padding zeroed
    {
        const uint8_t* src = win + d.seq_off;
        uint8_t* dst = O.seq + qoff / 2;
        const int64_t nsq = ((int64_t)d.l_seq + 1) / 2, tot = padded / 2;
        for (int64_t b = (int64_t)lane * 16; b < tot; b += 1024) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b < nsq) {
                __builtin_memcpy(&v, src + b, 16);                 // may run past the field: masked below (the window has slack)
                const int64_t valid = nsq - b;
                if (valid < 16) {
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int64_t rem = valid - 4 * i;
                        if (rem < 4) w[i] = rem <= 0 ? 0u : (w[i] & (0xffffffffu >> (8 * (4 - rem))));
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if ((d.l_seq & 1u) && nsq - 1 >= b && nsq - 1 < b + 16) {           // odd length: the last byte keeps its high nibble
                    const int64_t j = nsq - 1 - b;
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
                    w[j >> 2] &= ~(0x0fu << (8 * (j & 3)));
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            *reinterpret_cast<uint4*>(dst + b) = v;
        }
    }
    // QUAL
    {
        const uint8_t* src = win + d.qual_off;
        uint8_t* dst = O.bq + qoff;
        const int64_t nq = d.l_seq;
        for (int64_t b = (int64_t)lane * 16; b < padded; b += 1024) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b < nq) {
                __builtin_memcpy(&v, src + b, 16);
                const int64_t valid = nq - b;
                if (valid < 16) {
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int64_t rem = valid - 4 * i;
                        if (rem < 4) w[i] = rem <= 0 ? 0u : (w[i] & (0xffffffffu >> (8 * (4 - rem))));
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            *reinterpret_cast<uint4*>(dst + b) = v;
        }
    }
    // cs text (byte granular: its offsets are not aligned); '=' means long-form tags
    bool longcs = false;
    if (d.status == 0) {
        const uint8_t* src = win + d.cs_off;
        uint8_t* dst = O.cs + cso;
        for (uint32_t b = lane; b < d.cs_len; b += 64) { const uint8_t ch = src[b]; dst[b] = ch; if (ch == '=') longcs = true; }
    }
    if (__ballot(longcs) && lane == 0) S->any_longcs = 1;
    if (lane == 0) {
        O.tstart[idx] = d.pos; O.tend[idx] = d.pos + d.ref_len; O.qstart[idx] = d.lead_clip; O.qlen[idx] = (int32_t)d.l_seq;
        O.flag[idx] = (uint16_t)(d.flag_mapq_tp & 0xffffu); O.mapq[idx] = (uint8_t)((d.flag_mapq_tp >> 16) & 0xffu);
        O.tp[idx] = (uint8_t)(d.flag_mapq_tp >> 24);
        O.qid[idx] = qid_in[k];
        O.qoff[idx] = qoff; O.cs_off[idx] = cso;
        const int prev = k > 0 ? desc[k - 1].pos : S->last_pos;
        if (d.pos < prev) atomicAdd(&S->n_unsorted, 1ull);
        if (d.status == 2) atomicAdd(&S->n_missing_cs, 1ull);
        atomicAdd(&S->read_bases, (unsigned long long)d.l_seq);
    }
}

__global__ void k_bam_advance(const RecDesc* desc, const uint2* sizes, const uint2* offs, int64_t nrec, IngestState* S, int64_t* cs_off,
                              int64_t cap_reads) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || nrec <= 0) return;
    const uint2 lo = offs[nrec - 1], ls = sizes[nrec - 1];
    S->n_reads += (unsigned long long)nrec;
    S->bases_padded += (unsigned long long)lo.x + ls.x;
    S->cs_n += (unsigned long long)lo.y + ls.y;
    S->last_pos = desc[nrec - 1].pos;
    if ((int64_t)S->n_reads <= cap_reads) cs_off[S->n_reads] = (int64_t)S->cs_n;      // n + 1 entries
}

}  // namespace himut
