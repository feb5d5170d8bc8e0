// libhimut_hip.so -- host side of the C ABI declared in include/himut_hip.h.
// Owns the device buffers of one context, builds the small chunk / window tables,
// launches the kernels of himut_kernels.h on the context's stream and reads the
// result back.  No torch types, no C++ exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <chrono>

#include <string.h>  // rocprim's texture iterator needs the host memset declared first
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "himut_hip.h"
#include "himut_kernels.h"
#include "himut_norm.h"
#include "himut_normq.h"
#include "himut_ingest.h"

using namespace himut;

namespace {

struct HipFail {
    hipError_t e;
    const char* what;
    int line;
};

#define HCHECK(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) throw HipFail{_e, #expr, __LINE__}; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    uint64_t gen = 0;      // counts the allocations: a block that was freed and allocated again may come back at the SAME address,
                           // with other contents -- who keeps track of what a buffer holds compares this, not the pointer
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reserve(size_t bytes) {
        if (bytes <= cap && p) return;
        // growing = free + allocate.  Work already queued on the context's (non-blocking) streams may still use the
        // old block, so the device is drained first; this happens when a contig is larger than the ones before it.
        if (p) { HCHECK(hipDeviceSynchronize()); HCHECK(hipFree(p)); p = nullptr; cap = 0; }
        size_t want = std::max<size_t>(bytes, 256);
        HCHECK(hipMalloc(&p, want));
        cap = want; gen++;
    }
    // grows to at least `bytes` keeping the first `used` bytes (the ingest's arrays grow while they are being filled)
    void grow_keep(size_t bytes, size_t used) {
        if (bytes <= cap && p) return;
        const size_t want = std::max<size_t>(std::max(bytes, cap + cap / 2), 256);
        void* q = nullptr;
        HCHECK(hipDeviceSynchronize());
        HCHECK(hipMalloc(&q, want));
        if (p && used) HCHECK(hipMemcpy(q, p, std::min(used, cap), hipMemcpyDeviceToDevice));
        if (p) HCHECK(hipFree(p));
        p = q; cap = want; gen++;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

template <class T>
void upload(DevBuf& b, const T* src, size_t n, hipStream_t st) {
    b.reserve(std::max<size_t>(n, 1) * sizeof(T) + 256);  // slack: kernels read whole 16/32-byte windows
    if (n) HCHECK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, st));
}

template <class T>
void upload(DevBuf& b, const std::vector<T>& v, hipStream_t st) { upload(b, v.data(), v.size(), st); }

enum { EV_START = 0, EV_SIDE, EV_PARSE, EV_HAP, EV_EMIT, EV_INDEX, EV_GATHER, EV_SWEEP, EV_FINAL, EV_COPIED, EV_COUNT };

}  // namespace

struct PendingRun {
    bool active = false, spec = false;
    int64_t ncap = 0, slot_cap = 0, nreserve = 0, positions = 0;
    size_t lead_bytes = 0;
};

struct himut_ctx {
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;   // work that needs nothing from the cs decode runs here, beside it
    hipEvent_t ev[EV_COUNT] = {};
    std::string err;

    Params params{};
    bool have_params = false, have_lut = false, have_reads = false;

    // inputs
    DevBuf d_lut;
    std::vector<int32_t> cstart, cend;
    DevBuf d_cstart, d_cend, d_maskoff, d_tileoff, d_sstart, d_sidx, d_spmax, d_rlo, d_rhi, d_pairoff, d_hint, d_crec, d_mtile;
    std::vector<int32_t> up_cs, up_ce;   // the chunk list the device tables were built for
    bool tables_valid = false, chunks_in_order = false;
    int64_t up_positions = 0, up_tiles = 0, up_pairs = 0, up_maxpairs = 0;
    int64_t nhint = 0;
    std::vector<int64_t> maskoff, tileoff;
    DevBuf d_pon, d_com, d_posbits;
    std::vector<uint64_t> h_pon, h_com;
    int64_t npon = 0, ncom = 0, nposbits = 0;
    bool have_phase = false;
    std::vector<int64_t> h_phoff;
    DevBuf d_phoff, d_hpos, d_href, d_halt, d_hbit, d_hap;

    // reads
    int64_t n = 0, cs_bytes = 0, seq_bytes = 0, bq_bytes = 0, read_bases = 0;
    std::vector<int32_t> h_tstart, h_tend, h_prefmax;
    bool unique_qnames = true, any_longcs = false;
    PendingRun pending;               // a run whose host half is still to come (himut_run_begin / himut_run_end)
    size_t lead_clean_bytes = 0;      // bytes of the position bitmap (and the scalars) a call run left empty for the next one
    DevBuf d_tstart, d_tend, d_qstart, d_qlen, d_mapq, d_flag, d_qid, d_qoff, d_csoff, d_seq, d_bq, d_cs, d_prefmax;
    // derived
    DevBuf d_bqsum, d_nseg, d_nmis, d_nnsub, d_segs, d_mis, d_mq, d_meta, d_rflag, d_ccs, d_order;
    // run state
    DevBuf d_mask, d_recs, d_recs_out, d_keys, d_keys2, d_vals, d_vals2, d_emit, d_pos, d_tmp, d_scalars;
    DevBuf d_tilecnt, d_tileoff2, d_logpart;
    // normcounts
    DevBuf d_nonacgt;                        // per read: SEQ holds a base outside ATGC (k_flag_bases, once per batch)
    DevBuf d_refcode;                        // per reference position: what the sweep wants to know about the letter (k_ref_codes)
    DevBuf d_refseq, d_live, d_callable, d_dirty, d_dcount, d_redo, d_plan, d_plancnt, d_tri;
    int dbg_norm_sweep = 0, dbg_norm_pool = 0;     // himut_debug_normcounts (tests)
    int64_t dbg_norm_dirty_cap = 0;
    int64_t norm_dirty_room = 0;                   // positions per part of k_norm_dirty's list an earlier pass of this context needed
    int64_t reflen = 0;
    uint8_t ref_cls[256] = {};
    int ref_K = 0;
    std::vector<unsigned long long> h_tri;   // ccs[K^3], ref[K^3], log[16]
    bool have_norm = false;
    DevBuf d_dense_counts, d_dense_bqsum, d_tiles, d_cands, d_cands2, d_winlo, d_winhi, d_blkslots, d_blkoff, d_blktab, d_colstore, d_posbits_c, d_posrank, d_poppc, d_tmp2;
    // capacities the candidate / column buffers were last sized for: a run whose counts fit them goes
    // through without a host round trip in the middle (0 = not known yet)
    int64_t cap_cand = 0, cap_slots = 0;
    int timing = 1;                          // himut_set_stage_timing: 0 total only, 1 + the column capture, 2 every stage
    bool mask_clean = false;                 // d_mask and d_tilecnt hold zeros only (k_mask_emit leaves them so)
    // device-side BAM ingest (himut_ingest_*)
    void* ing_pinned[2] = {nullptr, nullptr};
    size_t ing_window = 0, ing_bound = 0;
    bool ing_sized = false;
    DevBuf d_stage[2], d_recoff[2], d_qidin[2], d_desc, d_sizes, d_offs, d_istate, d_tp;
    hipEvent_t ing_copied[2] = {}, ing_parsed[2] = {};
    bool ing_open = false, ing_used[2] = {false, false};
    int64_t ing_reads = 0, ing_bases = 0, ing_cs = 0;   // capacity the windows so far may need (upper bounds)
    bool bases_flagged = false;              // d_nonacgt holds k_flag_bases' answer for the pushed reads
    int64_t win_nblk = 0;                    // d_winlo / d_winhi hold the read windows of the pushed reads for this many
                                             // 256-position blocks (0: not computed yet)
    void* h_scalars = nullptr;               // pinned landing zone of the scalars block
    std::vector<himut_record> h_recs;
    bool h_recs_valid = false;
    int64_t n_out = 0;
    int64_t log[15] = {};
    himut_run_stats stats{};
};

namespace {

// the two pinned ingest windows of the process, and the context whose ingest is open on them (himut_ingest_begin .. _end)
void* g_pinned[2] = {nullptr, nullptr};
size_t g_pinned_bytes = 0;
himut_ctx* g_pinned_owner = nullptr;
std::mutex g_pinned_mx;
void release_pinned(himut_ctx* c) {
    std::lock_guard<std::mutex> lock(g_pinned_mx);
    if (g_pinned_owner == c) g_pinned_owner = nullptr;
}

struct PopcWord {
    __host__ __device__ uint32_t operator()(uint32_t w) const { return (uint32_t)__builtin_popcount(w); }
};

// scalars block in device memory
struct Scalars {
    unsigned long long ncand;
    unsigned long long nrec;
    unsigned long long reserved0;
    unsigned long long nccs;
    unsigned long long log[16];
    int err;
    int qhigh;               // (unused)
    int dirty_over;          // normcounts: k_norm_quad left more positions to k_norm_dirty than a part of the list holds
    unsigned int nredo;      // normcounts: tiles k_norm_quad left to k_norm_tile
    int pad[20];             // 256 bytes: one aligned fill clears it
};
static_assert(sizeof(Scalars) == 256, "Scalars is cleared with one aligned fill");

const char* err_text(int code) {
    switch (code) {
        case HIMUT_ERR_CS: return "cs tag cannot be tokenised or disagrees with SEQ/CIGAR";
        case HIMUT_ERR_BASE: return "KeyError: base outside ATGC (util.py:17)";
        case HIMUT_ERR_BQ0: return "ValueError: math domain error (BQ 0 in a candidate column, gtlib.py:64)";
        case HIMUT_ERR_COVER: return "KeyError: hetSNP position missing from tpos2qbase (haplib.py:51)";
        case HIMUT_ERR_DEPTH: return "pile too deep: the contig's candidate columns need more than 2^32 column-store slots";
    }
    return "device error";
}

int fail(himut_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

template <class F>
int guarded(himut_ctx* c, F f) {
    try {
        (void)hipGetLastError();   // an error some earlier call left behind is not this call's
        return f();
    } catch (const HipFail& h) {
        char buf[512];
        snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s (himut_hip.hip:%d)", (int)h.e, hipGetErrorString(h.e), h.what, h.line);
        return fail(c, HIMUT_ERR_HIP, buf);
    } catch (const std::bad_alloc&) {
        return fail(c, HIMUT_ERR_NOMEM, "host allocation failed");
    } catch (...) {
        return fail(c, HIMUT_ERR_ARG, "unexpected C++ exception");
    }
}

Reads make_reads(himut_ctx* c) {
    Reads R;
    R.n = c->n;
    R.tstart = c->d_tstart.as<int32_t>(); R.tend = c->d_tend.as<int32_t>(); R.qstart = c->d_qstart.as<int32_t>();
    R.qlen = c->d_qlen.as<int32_t>(); R.mapq = c->d_mapq.as<uint8_t>(); R.flag = c->d_flag.as<uint16_t>();
    R.qid = c->d_qid.as<int32_t>(); R.qoff = c->d_qoff.as<int64_t>(); R.cs_off = c->d_csoff.as<int64_t>();
    R.seq = c->d_seq.as<uint8_t>(); R.bq = c->d_bq.as<uint8_t>(); R.cs = c->d_cs.as<uint8_t>();
    R.prefmax_tend = c->d_prefmax.as<int32_t>();
    R.nonacgt = c->d_nonacgt.as<uint8_t>();
    return R;
}

Derived make_derived(himut_ctx* c) {
    Derived D;
    D.bqsum = c->d_bqsum.as<uint32_t>(); D.nseg = c->d_nseg.as<int32_t>(); D.nmis = c->d_nmis.as<int32_t>();
    D.segs = c->d_segs.as<Seg>(); D.mis = c->d_mis.as<int32_t>(); D.mq = c->d_mq.as<uint32_t>();
    D.rflag = c->d_rflag.as<uint8_t>(); D.meta = c->d_meta.as<ReadMeta>(); D.nnsub = c->d_nnsub.as<int32_t>();
    return D;
}

// Uploads the chunk tables for the given chunk list and the current reads.
struct ChunkTables {
    int64_t n = 0, positions = 0, n_tiles = 0, npairs = 0, maxpairs = 0;   // maxpairs: the most reads under one chunk
};

ChunkTables upload_chunks(himut_ctx* c, const std::vector<int32_t>& cs, const std::vector<int32_t>& ce) {
    ChunkTables T;
    const int TP = PD_TP;  // tiles are only used by the dense pile kernel
    const int64_t n = (int64_t)cs.size();
    T.n = n;
    if (c->tables_valid && cs == c->up_cs && ce == c->up_ce) {   // same chunks, same reads: the tables are on the device
        T.positions = c->up_positions; T.n_tiles = c->up_tiles; T.npairs = c->up_pairs; T.maxpairs = c->up_maxpairs;
        return T;
    }
    c->maskoff.assign(n + 1, 0);
    c->tileoff.assign(n + 1, 0);
    for (int64_t k = 0; k < n; k++) {
        int64_t span = (int64_t)ce[k] - cs[k] + 1;
        c->maskoff[k + 1] = c->maskoff[k] + span;
        c->tileoff[k + 1] = c->tileoff[k] + (span + TP - 1) / TP;
    }
    T.positions = c->maskoff[n];
    T.n_tiles = c->tileoff[n];
    std::vector<int32_t> order(n), sstart(n), sidx(n), spmax(n);
    for (int64_t k = 0; k < n; k++) order[k] = (int32_t)k;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cs[a] < cs[b]; });
    int32_t run = INT32_MIN;
    for (int64_t k = 0; k < n; k++) {
        sstart[k] = cs[order[k]];
        sidx[k] = order[k];
        run = std::max(run, ce[order[k]]);
        spmax[k] = run;
    }
    // read windows per chunk: [first read whose running max tend > start, first read with tstart >= end)
    std::vector<int64_t> rlo(n), rhi(n), pairoff(n + 1, 0);
    for (int64_t k = 0; k < n; k++) {
        rlo[k] = std::upper_bound(c->h_prefmax.begin(), c->h_prefmax.end(), cs[k]) - c->h_prefmax.begin();
        rhi[k] = std::lower_bound(c->h_tstart.begin(), c->h_tstart.end(), ce[k]) - c->h_tstart.begin();
        if (rhi[k] < rlo[k]) rhi[k] = rlo[k];
        pairoff[k + 1] = pairoff[k] + (rhi[k] - rlo[k]);
        T.maxpairs = std::max<int64_t>(T.maxpairs, rhi[k] - rlo[k]);
    }
    T.npairs = pairoff[n];
    // look-up hint: for each 16-kb block of positions, the number of sorted starts <= block start
    int32_t maxend = 0;
    for (int64_t k = 0; k < n; k++) maxend = std::max(maxend, ce[k]);
    c->nhint = ((int64_t)maxend >> CHUNK_HINT_SHIFT) + 2;
    std::vector<int32_t> hint((size_t)c->nhint);
    {
        int64_t j = 0;
        for (int64_t b = 0; b < c->nhint; b++) {
            const int64_t p = b << CHUNK_HINT_SHIFT;
            while (j < n && (int64_t)sstart[j] <= p) j++;
            hint[(size_t)b] = (int32_t)j;
        }
    }
    std::vector<ChunkRec> crec((size_t)n);
    for (int64_t k = 0; k < n; k++) {
        const int32_t ci = sidx[k];
        crec[k].start = cs[ci]; crec[k].end = ce[ci]; crec[k].idx = ci; crec[k].pmaxend = spmax[k];
        crec[k].maskoff = c->maskoff[ci]; crec[k].pairbase = pairoff[ci] - rlo[ci];
    }
    // candidates come out of the mask in (chunk, tpos) order; that is the record order when
    // no chunk starts before its predecessor's end (the reference's chunking shares only the edge)
    c->chunks_in_order = true;
    for (int64_t k = 1; k < n; k++) if (cs[k] < ce[k - 1]) c->chunks_in_order = false;
    // mask tiles (MASK_TILE_CELLS cells each): the chunk their first cell belongs to
    const int64_t ntile = std::max<int64_t>(1, ((int64_t)T.positions + MASK_TILE_CELLS - 1) / MASK_TILE_CELLS);
    std::vector<MaskTile> mtile((size_t)ntile);
    {
        int64_t ck = 0;
        for (int64_t b = 0; b < ntile; b++) {
            const int64_t cell = b * MASK_TILE_CELLS;
            while (ck + 1 < n && c->maskoff[ck + 1] <= cell) ck++;
            MaskTile& m = mtile[(size_t)b];
            m.ck0 = (int32_t)ck; m.start0 = n > 0 ? cs[ck] : 0;
            m.off0 = n > 0 ? c->maskoff[ck] : 0; m.off1 = n > 0 ? c->maskoff[ck + 1] : 0; m.pad = 0;
        }
    }
    hipStream_t st = c->stream;
    upload(c->d_hint, hint, st);
    upload(c->d_crec, crec, st);
    upload(c->d_mtile, mtile, st);
    upload(c->d_cstart, cs, st); upload(c->d_cend, ce, st);
    upload(c->d_maskoff, c->maskoff, st); upload(c->d_tileoff, c->tileoff, st);
    upload(c->d_sstart, sstart, st); upload(c->d_sidx, sidx, st); upload(c->d_spmax, spmax, st);
    upload(c->d_rlo, rlo, st); upload(c->d_rhi, rhi, st); upload(c->d_pairoff, pairoff, st);
    HCHECK(hipStreamSynchronize(st));  // the host vectors above go out of scope
    c->up_cs = cs; c->up_ce = ce; c->tables_valid = true;
    c->up_positions = T.positions; c->up_tiles = T.n_tiles; c->up_pairs = T.npairs; c->up_maxpairs = T.maxpairs;
    return T;
}

Chunks make_chunks(himut_ctx* c, int64_t n) {
    Chunks C;
    C.n = n;
    C.rec = c->d_crec.as<ChunkRec>(); C.mtile = c->d_mtile.as<MaskTile>();
    C.start = c->d_cstart.as<int32_t>(); C.end = c->d_cend.as<int32_t>();
    C.maskoff = c->d_maskoff.as<int64_t>();
    C.s_start = c->d_sstart.as<int32_t>(); C.s_idx = c->d_sidx.as<int32_t>(); C.s_pmaxend = c->d_spmax.as<int32_t>();
    C.rlo = c->d_rlo.as<int64_t>(); C.rhi = c->d_rhi.as<int64_t>(); C.pairoff = c->d_pairoff.as<int64_t>();
    C.hint = c->d_hint.as<int32_t>(); C.nhint = c->nhint;
    return C;
}

Phase make_phase(himut_ctx* c) {
    Phase H;
    H.off = c->d_phoff.as<int64_t>(); H.hpos = c->d_hpos.as<int32_t>(); H.href = c->d_href.as<uint8_t>();
    H.halt = c->d_halt.as<uint8_t>(); H.hbit = c->d_hbit.as<uint8_t>(); H.hap = c->d_hap.as<uint8_t>();
    return H;
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }

void launch_pile_dense(himut_ctx* c, const Chunks& C, const Reads& R, const Derived& D, const ChunkTables& T, int* err) {
    hipStream_t st = c->stream;
    c->d_tiles.reserve((size_t)T.n_tiles * sizeof(TileInfo) + 64);
    hipLaunchKernelGGL(k_tile_index<PD_TP>, dim3(blocks_for(T.n_tiles, 256)), dim3(256), 0, st, R, C,
                       c->d_tileoff.as<int64_t>(), T.n_tiles, c->d_tiles.as<TileInfo>());
    DenseArgs A;
    A.R = R; A.D = D; A.C = C; A.tiles = c->d_tiles.as<TileInfo>(); A.n_tiles = T.n_tiles;
    A.counts = c->d_dense_counts.as<uint32_t>(); A.bqsum = c->d_dense_bqsum.as<uint32_t>(); A.err = err;
    hipLaunchKernelGGL((k_pile_dense<PD_TP, PD_RB, PD_NT>), dim3((unsigned)T.n_tiles), dim3(PD_NT), 0, st, A);
}

int check_device_err(himut_ctx* c, int bits) {
    if (!bits) return HIMUT_OK;
    for (int code = 1; code < 31; code++)
        if (bits & (1 << code)) return fail(c, code, err_text(code));
    return fail(c, HIMUT_ERR_ARG, "device error");
}

// stage events cost a barrier packet each (a few microseconds of queue time): only the ones asked for are recorded
inline void stage_event(himut_ctx* c, int ev, int level, hipStream_t st) {
    if (c->timing >= level) HCHECK(hipEventRecord(c->ev[ev], st));
}

template <bool WITH_BQ>
void launch_parse(himut_ctx* c, const Reads& R, const Derived& D, Scalars* sc, uint32_t* posbits, int64_t nposwords,
                  void* fill = nullptr, int64_t fill_slots = 0) {
    // fill: the column store, to be left EMPTY (fill_slots 16-bit slots) by the decode's waves on their way
    const int64_t fill16 = (fill_slots * 2 + 15) / 16;
    const int fill_per = fill ? (int)((fill16 + c->n * 64 - 1) / (c->n * 64)) : 0;
    hipLaunchKernelGGL(k_parse_cs<WITH_BQ>, dim3(blocks_for(c->n, 4)), dim3(256), 0, c->stream, R, D, c->params, &sc->err,
                       c->d_ccs.as<uint8_t>(), posbits, nposwords, (uint4*)fill, fill16, fill_per);
}

// side_work: work for the second stream, done while the quality stream + cs decode run
template <bool WITH_BQ, class F>
void run_parse_stage(himut_ctx* c, const Reads& R, const Derived& D, Scalars* sc, F side_work, uint32_t* posbits = nullptr,
                     int64_t nposwords = 0, void* fill = nullptr, int64_t fill_slots = 0) {
    hipStream_t st = c->stream;
    // the side stream takes the work that needs nothing from the decode (it starts behind EV_START: the
    // previous run on this context is over by then)
    launch_parse<WITH_BQ>(c, R, D, sc, posbits, nposwords, fill, fill_slots);
    HCHECK(hipStreamWaitEvent(c->side, c->ev[EV_START], 0));
    side_work(c->side);
    HCHECK(hipEventRecord(c->ev[EV_SIDE], c->side));
    if (c->any_longcs)
        hipLaunchKernelGGL(k_check_longcs, dim3(blocks_for(c->n, 256)), dim3(256), 0, st, R, D, &sc->err);
    HCHECK(hipStreamWaitEvent(st, c->ev[EV_SIDE], 0));
    stage_event(c, EV_PARSE, 2, st);
}
void run_parse_stage(himut_ctx* c, const Reads& R, const Derived& D, Scalars* sc) {   // the decode alone
    run_parse_stage<false>(c, R, D, sc, [](hipStream_t) {});
}

// once per pushed batch: which reads hold a base outside ATGC somewhere (on `st`, in front of whatever looks at the flags)
void flag_bases_once(himut_ctx* c, hipStream_t st) {
    if (c->bases_flagged) return;             // (d_nonacgt is sized by alloc_derived)
    if (c->n > 0)
        hipLaunchKernelGGL(k_flag_bases, dim3(blocks_for(c->n, 4)), dim3(256), 0, st, c->n, c->d_qoff.as<int64_t>(), c->d_qlen.as<int32_t>(),
                           c->d_seq.as<uint8_t>(), c->d_nonacgt.as<uint8_t>());
    c->bases_flagged = true;
}

void alloc_derived(himut_ctx* c) {
    const int64_t n = c->n;
    const int64_t segcap = (c->cs_bytes >> 1) + n + 2;
    c->d_bqsum.reserve((size_t)n * 4 + 64);
    c->d_nseg.reserve((size_t)n * 4 + 64);
    c->d_nmis.reserve((size_t)n * 4 + 64);
    c->d_nnsub.reserve((size_t)n * 4 + 64);
    c->d_segs.reserve((size_t)segcap * sizeof(Seg));
    c->d_mis.reserve((size_t)segcap * 4);
    c->d_mq.reserve((size_t)segcap * 4);
    c->d_meta.reserve((size_t)(n + 1) * sizeof(ReadMeta));
    c->d_rflag.reserve((size_t)n + 64);
    c->d_ccs.reserve((size_t)n + 64);
    c->d_scalars.reserve(sizeof(Scalars));
    c->d_nonacgt.reserve((size_t)n + 64);
}

// One pass of the scan.  spec: the candidate and column-slot buffers keep the capacities of an earlier run
// (cap_cand, cap_slots) and every kernel behind a count takes the count from device memory, so the host
// launches the whole run without waiting in the middle; *overflow is set if a count did not fit (the caller
// runs again with exact sizes).  Otherwise the host waits for the counts where it needs them and sizes the
// buffers with 25 % of headroom for the runs that follow.
//
// Order: cs decode -> bitmap of the substitution positions of the reads that pass the cheap filters ->
// column windows / offsets -> k_stream_capture (every quality and base byte of the contig exactly once: the
// column store AND the whole-read quality sums) -> k_propose (the read filters now have the quality mean) ->
// candidates out of the mask -> k_eval_columns -> finalisation.
int finish_run(himut_ctx* c, bool* overflow);

int do_run_once(himut_ctx* c, bool allow_spec, bool* overflow, bool defer) {
    *overflow = false;
    c->pending.active = false;
    if (!c->have_params) return fail(c, HIMUT_ERR_ARG, "himut_set_params has not been called");
    if (!c->have_lut) return fail(c, HIMUT_ERR_ARG, "himut_set_gt_lut has not been called");
    if (!c->have_reads) return fail(c, HIMUT_ERR_ARG, "himut_push_reads has not been called");
    const bool phase = c->params.p.phase != 0;
    if (phase && !c->have_phase) return fail(c, HIMUT_ERR_ARG, "phase requested but himut_set_phase has not been called");
    if (phase && (int64_t)c->h_phoff.size() != (int64_t)c->cstart.size() + 1)
        return fail(c, HIMUT_ERR_ARG, "himut_set_phase chunk count differs from himut_set_chunks");
    for (size_t k = 0; k < c->cstart.size(); k++)
        if (c->cstart[k] > c->cend[k]) return fail(c, HIMUT_ERR_CHUNK, "ValueError: invalid coordinates: chunk start > end");
    HCHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->h_recs_valid = false;
    c->n_out = 0;
    memset(c->log, 0, sizeof(c->log));
    memset(&c->stats, 0, sizeof(c->stats));
    c->params.unique_qnames = c->unique_qnames ? 1 : 0;

    ChunkTables T = upload_chunks(c, c->cstart, c->cend);
    // the sorted-chunk path writes the candidates in their final order straight from the mask: only that one
    // has nothing between the count and its consumers that needs the count on the host
    const bool spec = allow_spec && c->chunks_in_order && c->cap_cand > 0 && c->cap_slots > 0 && T.positions > 0 && c->n > 0;
    alloc_derived(c);
    const int64_t n4 = ((int64_t)T.positions * 2 + 15) / 16;     // the mask in 16-byte pieces (8 positions each)
    const size_t mask_bytes = (size_t)n4 * 16;
    const int64_t anyw = ((int64_t)T.positions + 31) / 32;        // the mask sweeps take 32 cells per thread
    const unsigned mtiles = blocks_for(anyw, 256);
    const uint64_t mask_was = c->d_mask.gen;
    c->d_mask.reserve(mask_bytes + 64);
    // the emit sweep zeroes what the propose kernel set: a buffer that went through a whole run is clean
    const bool clear_mask = !c->mask_clean || mask_was != c->d_mask.gen;
    c->mask_clean = false;
    const uint64_t tcnt_was = c->d_tilecnt.gen;
    c->d_tilecnt.reserve((size_t)mtiles * 4 + 64);
    c->d_tileoff2.reserve((size_t)mtiles * 4 + 64);
    const bool clear_all = clear_mask || tcnt_was != c->d_tilecnt.gen;
    if (phase && T.n > 65535) return fail(c, HIMUT_ERR_ARG, "--phase: more than 65,535 chunks in one contig (k_read_hap takes a chunk per grid row)");
    if (phase) c->d_hap.reserve((size_t)T.npairs + 64);
    // bitmap of column positions: probed at every position a read covers, so it spans reads as well as chunks; the
    // read windows and the column offsets are kept per 256 positions of the same span
    int32_t maxpos = c->h_prefmax.empty() ? 0 : c->h_prefmax.back();
    for (int32_t e : c->cend) maxpos = std::max(maxpos, e);
    const int64_t nblk = ((int64_t)maxpos >> WIN_SHIFT) + 2;
    const int64_t nwords = nblk * 8;
    size_t scan_tiles = 0;
    BlockCount BC;
    // the column index: a thread per `idx_per` consecutive blocks, at most 1024 workgroups (k_block_sums / k_block_table3)
    const int idx_per = (int)std::max<int64_t>(1, (nblk + 256 * 1024 - 1) / (256 * 1024));
    const unsigned idx_wgs = blocks_for(nblk, 256 * idx_per);
    {   // buffers and scan scratch whose sizes the host knows now: sized before anything is queued (growing a
        // buffer in the middle of a run would free it under the kernels already queued on it)
        c->d_winlo.reserve((size_t)nblk * 4 + 64);
        c->d_winhi.reserve((size_t)nblk * 4 + 64);
        const uint64_t bits_was = c->d_posbits_c.gen;
        c->d_posbits_c.reserve((size_t)(nwords + 2) * 4 + 256);
        if (bits_was != c->d_posbits_c.gen) c->lead_clean_bytes = 0;
        c->d_posrank.reserve((size_t)idx_wgs * sizeof(uint4) + 256);      // per-workgroup totals of the column index
        c->d_blkslots.reserve((size_t)nblk * 4 + 256); c->d_blkoff.reserve((size_t)nblk * 4 + 256);
        c->d_blktab.reserve((size_t)nblk * sizeof(BlockTab) + 256);
        BC.bits = c->d_posbits_c.as<uint32_t>(); BC.winlo = c->d_winlo.as<int32_t>(); BC.winhi = c->d_winhi.as<int32_t>();
        if (mtiles > 65536u) {                                             // (else one workgroup scans the tile counts: k_scan_small)
            uint32_t* nul = nullptr;
            HCHECK(rocprim::exclusive_scan(nullptr, scan_tiles, nul, nul, 0u, (size_t)mtiles, rocprim::plus<uint32_t>(), st));
        }
        c->d_tmp2.reserve(scan_tiles + 256);
    }

    Reads R = make_reads(c);
    Derived D = make_derived(c);
    Chunks C = make_chunks(c, T.n);
    Phase H = make_phase(c);
    Scalars* sc = c->d_scalars.as<Scalars>();
    Scalars& hs = *reinterpret_cast<Scalars*>(c->h_scalars);

    HCHECK(hipEventRecord(c->ev[EV_START], st));
    flag_bases_once(c, st);
    // The cs decode sets the bits of the column positions and every kernel adds to the scalars: both are empty before
    // the run starts.  A run leaves them so (it clears them behind its last copy, while the host is already reading
    // the results): only a context that has not just been through a run of this kind pays for the fills here.
    const size_t lead_bytes = (size_t)(nwords + 2) * 4;
    if (c->lead_clean_bytes < lead_bytes) {
        HCHECK(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
        HCHECK(hipMemsetAsync(c->d_posbits_c.p, 0, lead_bytes, st));
    }
    c->lead_clean_bytes = 0;
    // The read windows per 256 positions depend on the pushed reads only (like a BAM index they are made once per
    // batch: the first run after himut_push_reads).  That kernel and the fills of a mask that is not known to be
    // empty need nothing from the cs decode: they run beside it on the second stream.  A context that has been
    // through a run has neither to do, and the second stream stays idle.
    const bool need_win = c->n > 0 && c->win_nblk != nblk;
    // (on kept capacities the column store's size is known before the decode has run: the decode's waves fill it)
    const bool fill_early = spec;
    if (fill_early) c->d_colstore.reserve((size_t)c->cap_slots * 2 + 256);
    void* fill_p = fill_early ? c->d_colstore.p : nullptr;
    auto side_work = [&](hipStream_t side) {
        if (clear_all) {
            HCHECK(hipMemsetAsync(c->d_mask.p, 0, c->d_mask.cap, side));
            HCHECK(hipMemsetAsync(c->d_tilecnt.p, 0, c->d_tilecnt.cap, side));
        }
        if (need_win)
            hipLaunchKernelGGL(k_window_index, dim3(blocks_for(nblk, 256)), dim3(256), 0, side, R, nblk,
                               c->d_winlo.as<int32_t>(), c->d_winhi.as<int32_t>());
    };
    if (c->n > 0) {
        if (clear_all || need_win)
            run_parse_stage<false>(c, R, D, sc, side_work, c->d_posbits_c.as<uint32_t>(), nwords, fill_p, c->cap_slots);
        else {
            launch_parse<false>(c, R, D, sc, c->d_posbits_c.as<uint32_t>(), nwords, fill_p, c->cap_slots);
            if (c->any_longcs)
                hipLaunchKernelGGL(k_check_longcs, dim3(blocks_for(c->n, 256)), dim3(256), 0, st, R, D, &sc->err);
            stage_event(c, EV_PARSE, 2, st);
        }
        c->win_nblk = nblk;
    } else {
        side_work(st);
        stage_event(c, EV_PARSE, 2, st);
    }
    if (phase && T.npairs > 0)
        hipLaunchKernelGGL(k_read_hap, dim3((unsigned)blocks_for(T.maxpairs, 16), (unsigned)T.n), dim3(256), 0, st, R, D, C, H, &sc->err);
    stage_event(c, EV_HAP, 2, st);

    // ---- columns: per 256-position block the column positions and the read window -> one scan -> the block table
    size_t slot_cap = 0;
    PosIndex X;
    X.bits = c->d_posbits_c.as<uint32_t>(); X.rank = nullptr; X.nwords = nwords;
    X.bt = c->d_blktab.as<BlockTab>(); X.nblk = nblk;
    if (c->n > 0) {
        hipLaunchKernelGGL(k_block_sums, dim3(idx_wgs), dim3(256), 0, st, BC, nblk, idx_per, c->d_posrank.as<uint4>());
        hipLaunchKernelGGL(k_block_table3, dim3(idx_wgs), dim3(256), 0, st, BC, nblk, idx_per, c->d_posrank.as<uint4>(),
                           c->d_blktab.as<BlockTab>(), c->d_blkoff.as<uint32_t>(), c->d_blkslots.as<uint32_t>(), &sc->err);
        size_t slot_reserve = (size_t)c->cap_slots;
        slot_cap = (size_t)c->cap_slots;
        if (!spec) {
            uint32_t last_off = 0, last_n = 0;
            HCHECK(hipMemcpyAsync(&last_off, c->d_blkoff.as<uint32_t>() + (nblk - 1), 4, hipMemcpyDeviceToHost, st));
            HCHECK(hipMemcpyAsync(&last_n, c->d_blkslots.as<uint32_t>() + (nblk - 1), 4, hipMemcpyDeviceToHost, st));
            HCHECK(hipMemcpyAsync(&hs, sc, sizeof(Scalars), hipMemcpyDeviceToHost, st));
            HCHECK(hipStreamSynchronize(st));
            if (hs.err) return check_device_err(c, hs.err);
            slot_cap = (size_t)last_off + last_n;
            slot_reserve = slot_cap + slot_cap / 4 + 4096;
        }
        if (!fill_early) {
            c->d_colstore.reserve(slot_reserve * 2 + 256);
            if (slot_cap) HCHECK(hipMemsetD16Async(c->d_colstore.p, (unsigned short)CELL_EMPTY, slot_cap, st));
        }
        CaptureArgs G;
        G.R = R; G.D = D; G.X = X; G.colstore = c->d_colstore.as<uint16_t>(); G.nslots = (int64_t)slot_cap;
        G.r_begin = 0; G.r_end = c->n; G.bqsum = c->d_bqsum.as<uint32_t>(); G.err = &sc->err;
        // the proposals of a read (read filters, trim / window filters -> mask) are the tail of its capture wave
        G.C = C; G.H = H; G.P = c->params; G.mask = c->d_mask.as<uint32_t>(); G.tilecnt = c->d_tilecnt.as<uint32_t>();
        G.ccs_flag = c->d_ccs.as<uint8_t>();
        stage_event(c, EV_INDEX, 1, st);
        hipLaunchKernelGGL(k_stream_capture, dim3(blocks_for(c->n, 4)), dim3(256), 0, st, G);
        stage_event(c, EV_GATHER, 1, st);
    } else {
        stage_event(c, EV_INDEX, 1, st);
        stage_event(c, EV_GATHER, 1, st);
    }
    // the candidates = the set bits of the mask; k_propose counted them per tile: the scan places the tiles
    uint32_t last_tcnt = 0, last_toff = 0;
    if (anyw > 0) {
        if (mtiles <= 65536u)
            hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, c->d_tilecnt.as<uint32_t>(), c->d_tileoff2.as<uint32_t>(), (int)mtiles);
        else
            HCHECK(rocprim::exclusive_scan(c->d_tmp2.p, scan_tiles, c->d_tilecnt.as<uint32_t>(), c->d_tileoff2.as<uint32_t>(), 0u,
                                           (size_t)mtiles, rocprim::plus<uint32_t>(), st));
        if (!spec) {
            HCHECK(hipMemcpyAsync(&last_tcnt, c->d_tilecnt.as<uint32_t>() + (mtiles - 1), 4, hipMemcpyDeviceToHost, st));
            HCHECK(hipMemcpyAsync(&last_toff, c->d_tileoff2.as<uint32_t>() + (mtiles - 1), 4, hipMemcpyDeviceToHost, st));
        }
    }

    // number of candidate evaluations -> record capacity
    int64_t ncap = c->cap_cand, nreserve = c->cap_cand;       // grid / scan extent, buffer capacity (records)
    if (!spec) {
        HCHECK(hipMemcpyAsync(&hs, sc, sizeof(Scalars), hipMemcpyDeviceToHost, st));
        HCHECK(hipStreamSynchronize(st));
        if (hs.err) return check_device_err(c, hs.err);
        ncap = (int64_t)last_toff + last_tcnt;
        nreserve = ncap + ncap / 4 + 1024;
    }
    c->d_recs.reserve((size_t)(nreserve + 1) * sizeof(himut_record));
    c->d_recs_out.reserve((size_t)(nreserve + 1) * sizeof(himut_record));
    const unsigned long long* ncand_dev = &sc->ncand;

    size_t sort_tmp = 0;
    if (ncap > 0) {
        // candidates in the order of the final records (tpos, chunk, ref, alt)
        c->d_keys.reserve((size_t)nreserve * 8); c->d_keys2.reserve((size_t)nreserve * 8);
        c->d_cands.reserve((size_t)nreserve * sizeof(Cand) + 256);
        c->d_cands2.reserve((size_t)nreserve * sizeof(Cand) + 256);
        c->d_emit.reserve((size_t)nreserve * 4);
        if (c->chunks_in_order) {
            hipLaunchKernelGGL(k_mask_emit, dim3(mtiles), dim3(256), 0, st, c->d_posbits_c.as<uint32_t>(), (int64_t)T.positions,
                               c->d_mask.as<uint16_t>(), c->d_tileoff2.as<uint32_t>(), C, c->d_cands2.as<Cand>(),
                               c->d_keys2.as<uint64_t>(), ncap, &sc->ncand, c->d_tilecnt.as<uint32_t>());
        } else {
            HCHECK(rocprim::radix_sort_pairs(nullptr, sort_tmp, c->d_keys.as<uint64_t>(), c->d_keys2.as<uint64_t>(),
                                             c->d_cands.as<uint64_t>(), c->d_cands2.as<uint64_t>(), (size_t)ncap, 0, 60, st));
            c->d_tmp.reserve(sort_tmp + 256);
            hipLaunchKernelGGL(k_mask_emit, dim3(mtiles), dim3(256), 0, st, c->d_posbits_c.as<uint32_t>(), (int64_t)T.positions,
                               c->d_mask.as<uint16_t>(), c->d_tileoff2.as<uint32_t>(), C, c->d_cands.as<Cand>(),
                               c->d_keys.as<uint64_t>(), ncap, &sc->ncand, c->d_tilecnt.as<uint32_t>());
            HCHECK(rocprim::radix_sort_pairs(c->d_tmp.p, sort_tmp, c->d_keys.as<uint64_t>(), c->d_keys2.as<uint64_t>(),
                                             c->d_cands.as<uint64_t>(), c->d_cands2.as<uint64_t>(), (size_t)ncap, 0, 60, st));
        }
        stage_event(c, EV_EMIT, 2, st);
        EvalArgs A;
        A.P = c->params;
        A.S.pon = c->d_pon.as<uint64_t>(); A.S.npon = c->npon; A.S.com = c->d_com.as<uint64_t>(); A.S.ncom = c->ncom;
        A.S.posbits = c->d_posbits.as<uint32_t>(); A.S.nposbits = c->nposbits;
        A.lut = c->d_lut.as<GtLut>();
        A.cands = c->d_cands2.as<Cand>(); A.ncand = ncap; A.ncand_dev = ncand_dev;
        A.R = R; A.D = D; A.C = C; A.H = H; A.X = X;
        A.colstore = c->d_colstore.as<uint16_t>(); A.nslots = (int64_t)slot_cap;
        A.recs = c->d_recs.as<himut_record>();
        A.err = &sc->err;
        if (phase) hipLaunchKernelGGL(k_eval_columns<true>, dim3(blocks_for(ncap, 256)), dim3(256), 0, st, A);
        else hipLaunchKernelGGL(k_eval_columns<false>, dim3(blocks_for(ncap, 256)), dim3(256), 0, st, A);
    } else {
        stage_event(c, EV_EMIT, 2, st);
    }
    stage_event(c, EV_SWEEP, 2, st);

    // ---- finalisation: order, cross-chunk som_seen, counters, compaction (every set mask bit is one evaluation)
    if (ncap > 0) {
        const unsigned nb = blocks_for(ncap, 256);
        c->d_logpart.reserve((size_t)nb * 16 * 4 + 64);
        c->d_pos.reserve((size_t)nb * 4 + 64);          // where each workgroup's emitted records begin
        hipLaunchKernelGGL(k_finalize_flags, dim3(nb), dim3(256), 0, st, c->d_recs.as<himut_record>(),
                           c->d_keys2.as<uint64_t>(), (const uint32_t*)nullptr, ncand_dev, ncap, c->d_emit.as<uint32_t>(),
                           c->d_logpart.as<uint32_t>(), c->d_ccs.as<uint8_t>(), c->n);
        hipLaunchKernelGGL(k_run_totals, dim3(1), dim3(1024), 0, st, ncap, c->d_blkoff.as<uint32_t>(), c->d_blkslots.as<uint32_t>(), nblk,
                           &sc->nrec, &sc->reserved0, c->d_logpart.as<uint32_t>(), (int64_t)nb, sc->log, c->d_pos.as<uint32_t>());
        hipLaunchKernelGGL(k_compact, dim3(nb), dim3(256), 0, st, c->d_recs.as<himut_record>(), (const uint32_t*)nullptr,
                           c->d_emit.as<uint32_t>(), c->d_pos.as<uint32_t>(), ncand_dev, ncap, c->d_recs_out.as<himut_record>());
    }
    if (c->n > 0 && ncap <= 0)   // no mask sweep ran: count the flagged reads here
        hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, st, c->d_ccs.as<uint8_t>(), c->n, &sc->nccs);
    HCHECK(hipEventRecord(c->ev[EV_FINAL], st));

    HCHECK(hipMemcpyAsync(&hs, sc, sizeof(Scalars), hipMemcpyDeviceToHost, st));
    HCHECK(hipEventRecord(c->ev[EV_COPIED], st));
    // behind the copy: the scalars and the bitmap empty for the next run (the host does not wait for these)
    HCHECK(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
    HCHECK(hipMemsetAsync(c->d_posbits_c.p, 0, lead_bytes, st));
    // what the second half (finish_run) needs: himut_run_begin returns here, with everything queued
    PendingRun& Q = c->pending;
    Q.active = true; Q.spec = spec; Q.ncap = ncap; Q.slot_cap = (int64_t)slot_cap; Q.nreserve = nreserve; Q.positions = T.positions;
    Q.lead_bytes = lead_bytes;
    if (defer) return HIMUT_OK;
    return finish_run(c, overflow);
}

// The host's half behind a run's last copy: waits for it (not for the stream), checks the device's error word and the
// counts against the capacities, takes the counters and the stage times.
int finish_run(himut_ctx* c, bool* overflow) {
    *overflow = false;
    PendingRun Q = c->pending;
    c->pending.active = false;
    if (!Q.active) return HIMUT_OK;
    HCHECK(hipSetDevice(c->device));
    const Scalars& hs = *reinterpret_cast<const Scalars*>(c->h_scalars);
    HCHECK(hipEventSynchronize(c->ev[EV_COPIED]));
    if (hs.err) return check_device_err(c, hs.err);
    c->lead_clean_bytes = Q.lead_bytes;
    const int64_t ncap = Q.ncap, slot_cap = Q.slot_cap;
    const int64_t ncand = ncap > 0 ? (int64_t)hs.ncand : 0;
    const int64_t nslots = ncap > 0 ? (int64_t)hs.reserved0 : slot_cap;
    if (ncand > ncap || nslots > slot_cap) {             // only a run on kept capacities can get here
        *overflow = true;                                // (mask cells past the capacity may still be set: not clean)
        return HIMUT_OK;
    }
    c->mask_clean = true;      // the emit sweep ran over every cell that was set (or nothing was set)
    if (!Q.spec && c->chunks_in_order) { c->cap_cand = Q.nreserve; c->cap_slots = slot_cap + slot_cap / 4 + 4096; }
    c->stats.column_slots = nslots;
    c->n_out = ncap > 0 ? (int64_t)hs.nrec : 0;
    for (int k = 0; k < 15; k++) c->log[k] = (int64_t)hs.log[k];
    if (ncap <= 0) c->log[0] = (int64_t)hs.nccs;     // (else counter 0 came with the others, k_finalize_flags)

    auto ms = [&](int a, int b) { float f = 0; (void)hipEventElapsedTime(&f, c->ev[a], c->ev[b]); return (double)f; };
    himut_run_stats& S = c->stats;
    S.ms_total = ms(EV_START, EV_FINAL);
    S.ms_bqsum = 0.0;
    if (c->timing >= 1) S.ms_capture = ms(EV_INDEX, EV_GATHER);
    if (c->timing >= 2) {
        S.ms_parse = ms(EV_START, EV_PARSE);
        S.ms_hap = ms(EV_PARSE, EV_HAP);
        S.ms_index = ms(EV_HAP, EV_INDEX);
        S.ms_emit = ms(EV_GATHER, EV_EMIT);
        S.ms_eval = ms(EV_EMIT, EV_SWEEP);
        S.ms_finalize = ms(EV_SWEEP, EV_FINAL);
    }
    S.n_reads = c->n;
    S.read_bases = c->read_bases;
    S.positions = Q.positions;
    S.n_unique_positions = 0;
    S.n_candidates = ncand;
    S.n_records = c->n_out;
    return HIMUT_OK;
}

int do_run(himut_ctx* c) {
    bool overflow = false;
    int rc = do_run_once(c, true, &overflow, false);
    if (rc == HIMUT_OK && overflow) {
        c->cap_cand = c->cap_slots = 0;
        rc = do_run_once(c, false, &overflow, false);
        c->stats.reran = 1;
    }
    return rc;
}

// himut_run in two halves.  begin: everything queued; on kept capacities (any run but a context's first on its reads
// and chunks) without waiting for anything.  end: the wait, the checks, and the second pass with exact sizes if a count
// did not fit.  Between the two the context must not be touched.
int do_run_begin(himut_ctx* c) {
    bool overflow = false;
    int rc = do_run_once(c, true, &overflow, true);
    if (rc == HIMUT_OK && !c->pending.spec && c->pending.active) {     // sized with the host in the loop: nothing left to overlap
        rc = finish_run(c, &overflow);
        c->pending.active = false;
    }
    return rc;
}
int do_run_end(himut_ctx* c) {
    if (!c->pending.active) return HIMUT_OK;
    bool overflow = false;
    int rc = finish_run(c, &overflow);
    if (rc == HIMUT_OK && overflow) {
        c->cap_cand = c->cap_slots = 0;
        rc = do_run_once(c, false, &overflow, false);
        c->stats.reran = 1;
    }
    return rc;
}

}  // namespace

extern "C" {

int himut_abi_version(void) { return HIMUT_ABI_VERSION; }

__global__ void k_warm(int* p) { if (p) *p = 0; }

int himut_create(int device, himut_ctx** out) {
    if (!out) return HIMUT_ERR_ARG;
    *out = nullptr;
    himut_ctx* c = new (std::nothrow) himut_ctx();
    if (!c) return HIMUT_ERR_NOMEM;
    c->device = device;
    int rc = guarded(c, [&]() -> int {
        int ndev = 0;
        HCHECK(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) return fail(c, HIMUT_ERR_ARG, "no such HIP device");
        HCHECK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HCHECK(hipGetDeviceProperties(&prop, device));
        c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HCHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        for (auto& e : c->ev) HCHECK(hipEventCreate(&e));
        HCHECK(hipHostMalloc(&c->h_scalars, sizeof(Scalars), hipHostMallocDefault));
        // the library's code object is loaded with the first launch (tens of milliseconds): here, not in the first
        // contig's ingest or scan
        hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, c->stream, (int*)nullptr);
        HCHECK(hipStreamSynchronize(c->stream));
        return HIMUT_OK;
    });
    if (rc) {
        static thread_local std::string last;
        last = c->err;
        delete c;
        return rc;
    }
    *out = c;
    return HIMUT_OK;
}

void himut_destroy(himut_ctx* c) {
    if (!c) return;
    release_pinned(c);
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    for (int k = 0; k < 2; k++) {
        if (c->ing_copied[k]) (void)hipEventDestroy(c->ing_copied[k]);
        if (c->ing_parsed[k]) (void)hipEventDestroy(c->ing_parsed[k]);
    }
    delete c;
}

int himut_set_stage_timing(himut_ctx* c, int level) {
    if (!c) return HIMUT_ERR_ARG;
    if (level < 0 || level > 2) return fail(c, HIMUT_ERR_ARG, "stage timing level must be 0, 1 or 2");
    c->timing = level;
    return HIMUT_OK;
}

const char* himut_last_error(const himut_ctx* c) { return c ? c->err.c_str() : "null context"; }

int himut_set_params(himut_ctx* c, const himut_params* p) {
    if (!c || !p) return HIMUT_ERR_ARG;
    c->params.p = *p;
    c->have_params = true;
    return HIMUT_OK;
}

int himut_set_gt_lut(himut_ctx* c, const double* hom, const double* het, const double* err_, int n_bq, const double prior[4]) {
    if (!c || !hom || !het || !err_ || !prior || n_bq < 1 || n_bq > 256) return fail(c, HIMUT_ERR_ARG, "bad LUT arguments");
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        GtLut L;
        for (int k = 0; k < 256; k++) {
            const int s = k < n_bq ? k : n_bq - 1;
            L.t[0][k] = hom[s]; L.t[1][k] = het[s]; L.t[2][k] = err_[s];
        }
        for (int k = 0; k < 4; k++) L.prior[k] = prior[k];
        c->d_lut.reserve(sizeof(GtLut));
        HCHECK(hipMemcpyAsync(c->d_lut.p, &L, sizeof(GtLut), hipMemcpyHostToDevice, c->stream));
        HCHECK(hipStreamSynchronize(c->stream));
        c->have_lut = true;
        return HIMUT_OK;
    });
}

int himut_set_chunks(himut_ctx* c, const int32_t* start, const int32_t* end, int64_t n) {
    if (!c || n < 0 || (n > 0 && (!start || !end))) return fail(c, HIMUT_ERR_ARG, "bad chunk arguments");
    if (n >= (1 << 24)) return fail(c, HIMUT_ERR_ARG, "too many chunks");
    c->cstart.assign(start, start + n);
    c->cend.assign(end, end + n);
    return HIMUT_OK;
}

int himut_set_site_set(himut_ctx* c, int which, const uint64_t* keys, int64_t n) {
    if (!c || (which != 0 && which != 1) || n < 0 || (n > 0 && !keys)) return fail(c, HIMUT_ERR_ARG, "bad site-set arguments");
    for (int64_t k = 1; k < n; k++)
        if (keys[k - 1] > keys[k]) return fail(c, HIMUT_ERR_ARG, "site-set keys must be sorted ascending");
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        DevBuf& b = which == 0 ? c->d_pon : c->d_com;
        upload(b, keys, (size_t)n, c->stream);
        (which == 0 ? c->h_pon : c->h_com).assign(keys, keys + n);
        (which == 0 ? c->npon : c->ncom) = n;
        // position bitmap over both sets: lets a candidate skip the two binary searches
        uint64_t maxpos = 0;
        for (const auto* v : {&c->h_pon, &c->h_com})
            if (!v->empty()) maxpos = std::max<uint64_t>(maxpos, v->back() >> 4);
        std::vector<uint32_t> bits((size_t)(maxpos >> 5) + 2, 0u);
        for (const auto* v : {&c->h_pon, &c->h_com})
            for (uint64_t k : *v) bits[(size_t)((k >> 4) >> 5)] |= 1u << ((k >> 4) & 31);
        c->nposbits = (c->h_pon.empty() && c->h_com.empty()) ? 0 : (int64_t)maxpos + 1;
        upload(c->d_posbits, bits, c->stream);
        HCHECK(hipStreamSynchronize(c->stream));
        return HIMUT_OK;
    });
}

int himut_set_phase(himut_ctx* c, const int64_t* off, const int32_t* hpos, const uint8_t* href, const uint8_t* halt,
                    const uint8_t* hbit, int64_t n_chunks) {
    if (!c || !off || n_chunks < 0) return fail(c, HIMUT_ERR_ARG, "bad phase arguments");
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        const int64_t m = off[n_chunks];
        c->h_phoff.assign(off, off + n_chunks + 1);
        upload(c->d_phoff, off, (size_t)n_chunks + 1, c->stream);
        upload(c->d_hpos, hpos, (size_t)m, c->stream);
        upload(c->d_href, href, (size_t)m, c->stream);
        upload(c->d_halt, halt, (size_t)m, c->stream);
        upload(c->d_hbit, hbit, (size_t)m, c->stream);
        HCHECK(hipStreamSynchronize(c->stream));
        c->have_phase = true;
        return HIMUT_OK;
    });
}

int himut_push_reads(himut_ctx* c, const himut_read_batch* b) {
    if (!c || !b || b->n_reads < 0) return fail(c, HIMUT_ERR_ARG, "bad read batch");
    const int64_t n = b->n_reads;
    if (n > 0 && (!b->tstart || !b->tend || !b->qstart || !b->qlen || !b->mapq || !b->flag || !b->qid || !b->qoff ||
                  !b->cs_off || !b->seq || !b->bq || !b->cs))
        return fail(c, HIMUT_ERR_ARG, "read batch has null arrays");
    // host-side shape checks: everything the kernels index with must be in range
    int64_t bases = 0;
    bool unique = true;
    for (int64_t i = 0; i < n; i++) {
        if (i > 0 && b->tstart[i] < b->tstart[i - 1]) return fail(c, HIMUT_ERR_ARG, "reads are not coordinate sorted");
        if (b->qoff[i] < 0 || (b->qoff[i] & 31) || b->qlen[i] < 0 || b->qoff[i] + (((int64_t)b->qlen[i] + 31) & ~(int64_t)31) > b->bq_bytes ||
            (b->qoff[i] + (((int64_t)b->qlen[i] + 31) & ~(int64_t)31)) / 2 > b->seq_bytes)
            return fail(c, HIMUT_ERR_ARG, "read offsets exceed the sequence / quality buffers");
        if (b->cs_off[i] < 0 || b->cs_off[i + 1] < b->cs_off[i] || b->cs_off[i + 1] > b->cs_bytes)
            return fail(c, HIMUT_ERR_ARG, "cs offsets exceed the cs buffer");
        if (b->tend[i] < b->tstart[i] || b->qstart[i] < 0 || b->qstart[i] > b->qlen[i])
            return fail(c, HIMUT_ERR_ARG, "read coordinates are inconsistent");
        if (b->qid[i] < 0 || b->qid[i] >= n) return fail(c, HIMUT_ERR_ARG, "qid out of range");
        if (b->qid[i] != i) unique = false;
        bases += b->qlen[i];
    }
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        hipStream_t st = c->stream;
        c->n = n; c->cs_bytes = b->cs_bytes; c->seq_bytes = b->seq_bytes; c->bq_bytes = b->bq_bytes; c->read_bases = bases;
        c->unique_qnames = unique;
        c->h_tstart.assign(b->tstart, b->tstart + n);
        c->h_tend.assign(b->tend, b->tend + n);
        c->h_prefmax.resize((size_t)n);
        int32_t run = INT32_MIN;
        for (int64_t i = 0; i < n; i++) { run = std::max(run, b->tend[i]); c->h_prefmax[(size_t)i] = run; }
        upload(c->d_tstart, b->tstart, (size_t)n, st); upload(c->d_tend, b->tend, (size_t)n, st);
        upload(c->d_qstart, b->qstart, (size_t)n, st); upload(c->d_qlen, b->qlen, (size_t)n, st);
        upload(c->d_mapq, b->mapq, (size_t)n, st); upload(c->d_flag, b->flag, (size_t)n, st);
        upload(c->d_qid, b->qid, (size_t)n, st); upload(c->d_qoff, b->qoff, (size_t)n, st);
        upload(c->d_csoff, b->cs_off, (size_t)n + 1, st);
        upload(c->d_seq, b->seq, (size_t)b->seq_bytes, st); upload(c->d_bq, b->bq, (size_t)b->bq_bytes, st);
        // k_parse_cs takes the text 1 KB at a time, 16 bytes per lane, whatever is left of the tag: the last read's
        // window runs up to 1 KB past the end of the text
        c->d_cs.reserve((size_t)b->cs_bytes + 2048);
        upload(c->d_cs, b->cs, (size_t)b->cs_bytes, st);
        upload(c->d_prefmax, c->h_prefmax, st);
        // long-form cs ('=' operations) needs one extra checking kernel; find out once, on the host
        c->any_longcs = memchr(b->cs, '=', (size_t)b->cs_bytes) != nullptr;
        HCHECK(hipStreamSynchronize(st));
        c->have_reads = true;
        c->tables_valid = false;   // the chunk tables hold read windows
        c->win_nblk = 0; c->bases_flagged = false;
        c->h_recs_valid = false;
        return HIMUT_OK;
    });
}

// ---- device-side BAM ingest: see include/himut_hip.h and csrc/himut_ingest.h
// First sizes of the contig's arrays from what CCS records look like (two thirds of a record are qualities); the arrays
// grow if a window needs more.
static void ingest_size_arrays(himut_ctx* c, hipStream_t st) {
    const size_t B = c->ing_bound;
    c->d_bq.reserve(B * 7 / 10 + (1 << 20));
    c->d_seq.reserve(B * 7 / 20 + (1 << 20));
    c->d_cs.reserve(B / 16 + (1 << 20));
    const size_t nr0 = B / 4000 + 4096;
    c->d_tstart.reserve(nr0 * 4 + 256); c->d_tend.reserve(nr0 * 4 + 256); c->d_qstart.reserve(nr0 * 4 + 256); c->d_qlen.reserve(nr0 * 4 + 256);
    c->d_qid.reserve(nr0 * 4 + 256); c->d_mapq.reserve(nr0 + 256); c->d_tp.reserve(nr0 + 256); c->d_flag.reserve(nr0 * 2 + 256);
    c->d_qoff.reserve(nr0 * 8 + 256); c->d_csoff.reserve((nr0 + 1) * 8 + 256);
    HCHECK(hipMemsetAsync(c->d_csoff.p, 0, 8, st));
    c->ing_sized = true;
}

int himut_ingest_begin(himut_ctx* c, int64_t inflated_bound, int64_t window_bytes) {
    if (!c || inflated_bound < 0 || window_bytes < (1 << 16)) return fail(c, HIMUT_ERR_ARG, "bad ingest arguments");
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        HCHECK(hipStreamSynchronize(c->stream));
        HCHECK(hipStreamSynchronize(c->side));
        const size_t W = (size_t)window_bytes;
        // The two pinned windows belong to the process, not to the context (pinning 128 MB takes tens of
        // milliseconds; a call makes one context per contig and ingests them one after the other): one ingest at a time
        std::lock_guard<std::mutex> lock(g_pinned_mx);
        if (g_pinned_owner && g_pinned_owner != c)
            return fail(c, HIMUT_ERR_ARG, "another context's ingest is open: the two pinned windows belong to the process");
        g_pinned_owner = c;
        if (g_pinned_bytes < W + 4096) {
            const auto t_pin = std::chrono::steady_clock::now();
            for (int k = 0; k < 2; k++) {
                if (g_pinned[k]) { HCHECK(hipHostFree(g_pinned[k])); g_pinned[k] = nullptr; }
                // cacheable pages: the inflate reads its own output back (LZ77 matches)
                const unsigned fl = getenv("HIMUT_PINNED_COHERENT") ? hipHostMallocDefault : (hipHostMallocNonCoherent | hipHostMallocPortable);
                HCHECK(hipHostMalloc(&g_pinned[k], W + 4096, fl));
            }
            g_pinned_bytes = W + 4096;
            if (getenv("HIMUT_INGEST_PROFILE"))
                fprintf(stderr, "ingest: pinned windows 2 x %zu MB in %.1f ms\n", (W + 4096) >> 20,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pin).count());
        }
        for (int k = 0; k < 2; k++) {
            c->ing_pinned[k] = g_pinned[k];
            if (!c->ing_copied[k]) HCHECK(hipEventCreateWithFlags(&c->ing_copied[k], hipEventDisableTiming));
            if (!c->ing_parsed[k]) HCHECK(hipEventCreateWithFlags(&c->ing_parsed[k], hipEventDisableTiming));
        }
        c->ing_window = W;
        for (int k = 0; k < 2; k++) { c->d_stage[k].reserve(W + 4096); c->ing_used[k] = false; }
        // the contig's arrays are sized when the first window arrives (himut_ingest_window): by then the host's pool
        // is inflating the second window, and gigabytes of hipMalloc are off the path
        c->ing_bound = (size_t)inflated_bound;
        c->ing_sized = false;
        c->d_istate.reserve(sizeof(IngestState));
        IngestState z;
        memset(&z, 0, sizeof(z));
        z.last_pos = -0x7fffffff - 1;
        HCHECK(hipMemcpy(c->d_istate.p, &z, sizeof(z), hipMemcpyHostToDevice));
        c->ing_reads = c->ing_bases = c->ing_cs = 0;
        c->ing_open = true;
        c->have_reads = false; c->tables_valid = false; c->win_nblk = 0; c->bases_flagged = false; c->h_recs_valid = false;
        return HIMUT_OK;
    });
}

void* himut_ingest_buffer(himut_ctx* c, int slot) { return (c && (slot == 0 || slot == 1)) ? c->ing_pinned[slot] : nullptr; }

int himut_ingest_wait(himut_ctx* c, int slot) {
    if (!c || (slot != 0 && slot != 1)) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        if (c->ing_used[slot]) HCHECK(hipEventSynchronize(c->ing_copied[slot]));
        return HIMUT_OK;
    });
}

int himut_ingest_window(himut_ctx* c, int slot, int64_t start, int64_t nbytes, const uint32_t* rec_off, const int32_t* qid, int64_t n_rec,
                        int64_t padded_bases, int64_t tag_bytes) {
    if (!c || (slot != 0 && slot != 1) || n_rec < 0 || nbytes < 0 || start < 0 || (n_rec && (!rec_off || !qid)))
        return fail(c, HIMUT_ERR_ARG, "bad ingest window");
    if (!c->ing_open) return fail(c, HIMUT_ERR_ARG, "himut_ingest_begin has not been called");
    if ((size_t)(start + nbytes) > c->ing_window) return fail(c, HIMUT_ERR_ARG, "ingest window larger than the buffer");
    if (n_rec == 0) return HIMUT_OK;
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        hipStream_t st = c->stream, cp = c->side;
        if (!c->ing_sized) {
            const auto t_sz = std::chrono::steady_clock::now();
            ingest_size_arrays(c, st);
            if (getenv("HIMUT_INGEST_PROFILE"))
                fprintf(stderr, "ingest: contig arrays sized in %.1f ms\n",
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sz).count());
        }
        // room for this window's reads: exact for the per-read arrays and the bases, an upper bound for the cs text
        const int64_t nr = c->ing_reads + n_rec, nb = c->ing_bases + padded_bases, nc = c->ing_cs + tag_bytes;
        const size_t ur = (size_t)c->ing_reads;
        c->d_tstart.grow_keep((size_t)nr * 4 + 256, ur * 4); c->d_tend.grow_keep((size_t)nr * 4 + 256, ur * 4);
        c->d_qstart.grow_keep((size_t)nr * 4 + 256, ur * 4); c->d_qlen.grow_keep((size_t)nr * 4 + 256, ur * 4);
        c->d_qid.grow_keep((size_t)nr * 4 + 256, ur * 4); c->d_mapq.grow_keep((size_t)nr + 256, ur); c->d_tp.grow_keep((size_t)nr + 256, ur);
        c->d_flag.grow_keep((size_t)nr * 2 + 256, ur * 2); c->d_qoff.grow_keep((size_t)nr * 8 + 256, ur * 8);
        c->d_csoff.grow_keep((size_t)(nr + 1) * 8 + 256, (ur + 1) * 8);
        c->d_bq.grow_keep((size_t)nb + 256, (size_t)c->ing_bases); c->d_seq.grow_keep((size_t)nb / 2 + 256, (size_t)c->ing_bases / 2);
        c->d_cs.grow_keep((size_t)nc + 2048 + 256, (size_t)c->ing_cs);
        c->d_recoff[slot].reserve((size_t)n_rec * 4 + 256); c->d_qidin[slot].reserve((size_t)n_rec * 4 + 256);
        c->d_desc.reserve((size_t)n_rec * sizeof(RecDesc) + 256);
        c->d_sizes.reserve((size_t)n_rec * 8 + 256); c->d_offs.reserve((size_t)n_rec * 8 + 256);
        size_t scan_b = 0;
        HCHECK(rocprim::exclusive_scan(nullptr, scan_b, c->d_sizes.as<uint2>(), c->d_offs.as<uint2>(), make_uint2(0u, 0u), (size_t)n_rec, PlusU2(), st));
        c->d_tmp.reserve(scan_b + 256);
        // copy stream: the window's bytes (pinned -> HBM) once the parse of the window that used this staging buffer is over
        if (c->ing_used[slot]) HCHECK(hipStreamWaitEvent(cp, c->ing_parsed[slot], 0));
        HCHECK(hipMemcpyAsync(c->d_stage[slot].p, (const uint8_t*)c->ing_pinned[slot] + start, (size_t)nbytes, hipMemcpyHostToDevice, cp));
        HCHECK(hipEventRecord(c->ing_copied[slot], cp));
        c->ing_used[slot] = true;
        // compute stream: record list, decode, offsets, scatter
        HCHECK(hipMemcpyAsync(c->d_recoff[slot].p, rec_off, (size_t)n_rec * 4, hipMemcpyHostToDevice, st));
        HCHECK(hipMemcpyAsync(c->d_qidin[slot].p, qid, (size_t)n_rec * 4, hipMemcpyHostToDevice, st));
        HCHECK(hipStreamWaitEvent(st, c->ing_copied[slot], 0));
        const uint8_t* win = c->d_stage[slot].as<uint8_t>();
        hipLaunchKernelGGL(k_bam_decode, dim3(blocks_for(n_rec, 256)), dim3(256), 0, st, win, nbytes, c->d_recoff[slot].as<uint32_t>(), n_rec,
                           c->d_desc.as<RecDesc>(), c->d_sizes.as<uint2>());
        HCHECK(rocprim::exclusive_scan(c->d_tmp.p, scan_b, c->d_sizes.as<uint2>(), c->d_offs.as<uint2>(), make_uint2(0u, 0u), (size_t)n_rec, PlusU2(), st));
        IngestOut O;
        O.tstart = c->d_tstart.as<int32_t>(); O.tend = c->d_tend.as<int32_t>(); O.qstart = c->d_qstart.as<int32_t>(); O.qlen = c->d_qlen.as<int32_t>();
        O.qid = c->d_qid.as<int32_t>(); O.mapq = c->d_mapq.as<uint8_t>(); O.tp = c->d_tp.as<uint8_t>(); O.flag = c->d_flag.as<uint16_t>();
        O.qoff = c->d_qoff.as<int64_t>(); O.cs_off = c->d_csoff.as<int64_t>();
        O.seq = c->d_seq.as<uint8_t>(); O.bq = c->d_bq.as<uint8_t>(); O.cs = c->d_cs.as<uint8_t>();
        O.cap_reads = nr; O.cap_bases = nb; O.cap_cs = nc;
        hipLaunchKernelGGL(k_bam_scatter, dim3(blocks_for(n_rec, 4)), dim3(256), 0, st, win, c->d_desc.as<RecDesc>(), c->d_offs.as<uint2>(),
                           c->d_qidin[slot].as<int32_t>(), n_rec, O, c->d_istate.as<IngestState>());
        hipLaunchKernelGGL(k_bam_advance, dim3(1), dim3(64), 0, st, c->d_desc.as<RecDesc>(), c->d_sizes.as<uint2>(), c->d_offs.as<uint2>(), n_rec,
                           c->d_istate.as<IngestState>(), c->d_csoff.as<int64_t>(), nr);
        HCHECK(hipEventRecord(c->ing_parsed[slot], st));
        c->ing_reads = nr; c->ing_bases = nb; c->ing_cs = nc;
        return HIMUT_OK;
    });
}

int himut_ingest_end(himut_ctx* c, int unique_qnames, himut_ingest_result* out) {
    if (!c || !out) return HIMUT_ERR_ARG;
    if (!c->ing_open) return fail(c, HIMUT_ERR_ARG, "himut_ingest_begin has not been called");
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        hipStream_t st = c->stream;
        if (!c->ing_sized) { c->ing_bound = 0; ingest_size_arrays(c, st); }       // a contig without records
        HCHECK(hipStreamSynchronize(c->side));
        HCHECK(hipStreamSynchronize(st));
        c->ing_open = false;
        release_pinned(c);
        IngestState S;
        HCHECK(hipMemcpy(&S, c->d_istate.p, sizeof(S), hipMemcpyDeviceToHost));
        memset(out, 0, sizeof(*out));
        out->n_reads = (int64_t)S.n_reads; out->bases_padded = (int64_t)S.bases_padded; out->cs_bytes = (int64_t)S.cs_n;
        out->read_bases = (int64_t)S.read_bases; out->n_missing_cs = (int64_t)S.n_missing_cs; out->n_unsorted = (int64_t)S.n_unsorted;
        out->n_malformed = (int64_t)S.n_bad;
        if (S.overflow) return fail(c, HIMUT_ERR_NOMEM, "ingest: a window needed more room than the host announced");
        if (S.n_bad) return fail(c, HIMUT_ERR_ARG, "malformed BAM record");
        const int64_t n = (int64_t)S.n_reads;
        c->n = n; c->cs_bytes = (int64_t)S.cs_n; c->bq_bytes = (int64_t)S.bases_padded; c->seq_bytes = (int64_t)S.bases_padded / 2;
        c->read_bases = (int64_t)S.read_bases;
        c->unique_qnames = unique_qnames != 0;
        c->any_longcs = S.any_longcs != 0;
        c->h_tstart.resize((size_t)n); c->h_tend.resize((size_t)n); c->h_prefmax.resize((size_t)n);
        if (n) {
            HCHECK(hipMemcpy(c->h_tstart.data(), c->d_tstart.p, (size_t)n * 4, hipMemcpyDeviceToHost));
            HCHECK(hipMemcpy(c->h_tend.data(), c->d_tend.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        }
        int32_t run = INT32_MIN;
        for (int64_t i = 0; i < n; i++) { run = std::max(run, c->h_tend[(size_t)i]); c->h_prefmax[(size_t)i] = run; }
        upload(c->d_prefmax, c->h_prefmax, st);
        // the kernels read whole 16 / 32-byte windows and 1 KB of cs text past the end: the arrays carry that slack
        c->d_cs.grow_keep((size_t)S.cs_n + 2048 + 256, (size_t)S.cs_n);
        HCHECK(hipStreamSynchronize(st));
        c->have_reads = S.n_unsorted == 0 && S.n_missing_cs == 0;
        c->tables_valid = false; c->win_nblk = 0; c->bases_flagged = false; c->h_recs_valid = false;
        return HIMUT_OK;
    });
}

int himut_ingest_read_meta(himut_ctx* c, int32_t* tstart, int32_t* tend, int32_t* qlen, uint8_t* mapq, uint8_t* tp) {
    if (!c) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        const size_t n = (size_t)c->n;
        if (!n) return HIMUT_OK;
        if (tstart) HCHECK(hipMemcpy(tstart, c->d_tstart.p, n * 4, hipMemcpyDeviceToHost));
        if (tend) HCHECK(hipMemcpy(tend, c->d_tend.p, n * 4, hipMemcpyDeviceToHost));
        if (qlen) HCHECK(hipMemcpy(qlen, c->d_qlen.p, n * 4, hipMemcpyDeviceToHost));
        if (mapq) HCHECK(hipMemcpy(mapq, c->d_mapq.p, n, hipMemcpyDeviceToHost));
        if (tp) HCHECK(hipMemcpy(tp, c->d_tp.p, n, hipMemcpyDeviceToHost));
        return HIMUT_OK;
    });
}

// The read batch as it sits in HBM, back on the host (tests: the device-parsed batch against the host-parsed one).
int himut_download_reads(himut_ctx* c, himut_read_batch* b, uint8_t* tp) {
    if (!c || !b) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        const size_t n = (size_t)c->n;
        if (b->n_reads < c->n || b->seq_bytes < c->seq_bytes || b->bq_bytes < c->bq_bytes || b->cs_bytes < c->cs_bytes)
            return fail(c, HIMUT_ERR_ARG, "destination batch too small");
        auto down = [&](const void* dst, const DevBuf& src, size_t bytes) { if (bytes) HCHECK(hipMemcpy(const_cast<void*>(dst), src.p, bytes, hipMemcpyDeviceToHost)); };
        down(b->tstart, c->d_tstart, n * 4); down(b->tend, c->d_tend, n * 4); down(b->qstart, c->d_qstart, n * 4); down(b->qlen, c->d_qlen, n * 4);
        down(b->mapq, c->d_mapq, n); down(b->flag, c->d_flag, n * 2); down(b->qid, c->d_qid, n * 4); down(b->qoff, c->d_qoff, n * 8);
        down(b->cs_off, c->d_csoff, (n + 1) * 8);
        down(b->seq, c->d_seq, (size_t)c->seq_bytes); down(b->bq, c->d_bq, (size_t)c->bq_bytes); down(b->cs, c->d_cs, (size_t)c->cs_bytes);
        if (tp) down(tp, c->d_tp, n);
        b->n_reads = c->n; b->seq_bytes = c->seq_bytes; b->bq_bytes = c->bq_bytes; b->cs_bytes = c->cs_bytes;
        return HIMUT_OK;
    });
}

int himut_run(himut_ctx* c) {
    if (!c) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int { return do_run(c); });
}
int himut_run_begin(himut_ctx* c) {
    if (!c) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int { return do_run_begin(c); });
}
int himut_run_end(himut_ctx* c) {
    if (!c) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int { return do_run_end(c); });
}

int himut_get_records(himut_ctx* c, const himut_record** records, int64_t* n) {
    if (!c || !records || !n) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        if (!c->h_recs_valid) {
            HCHECK(hipSetDevice(c->device));
            c->h_recs.resize((size_t)c->n_out);
            if (c->n_out)
                HCHECK(hipMemcpyAsync(c->h_recs.data(), c->d_recs_out.p, (size_t)c->n_out * sizeof(himut_record),
                                      hipMemcpyDeviceToHost, c->stream));
            HCHECK(hipStreamSynchronize(c->stream));
            c->h_recs_valid = true;
        }
        *records = c->h_recs.data();
        *n = c->n_out;
        return HIMUT_OK;
    });
}

int himut_get_log(himut_ctx* c, int64_t out[15]) {
    if (!c || !out) return HIMUT_ERR_ARG;
    for (int k = 0; k < 15; k++) out[k] = c->log[k];
    return HIMUT_OK;
}

int himut_get_stats(himut_ctx* c, himut_run_stats* out) {
    if (!c || !out) return HIMUT_ERR_ARG;
    *out = c->stats;
    return HIMUT_OK;
}

int himut_records_device(himut_ctx* c, const void** dev_ptr, int64_t* n) {
    if (!c || !dev_ptr || !n) return HIMUT_ERR_ARG;
    *dev_ptr = c->d_recs_out.p;
    *n = c->n_out;
    return HIMUT_OK;
}

int himut_copy_records_to_device(himut_ctx* c, void* dst, int64_t capacity_records) {
    if (!c || (!dst && c->n_out)) return HIMUT_ERR_ARG;
    if (capacity_records < c->n_out) return fail(c, HIMUT_ERR_ARG, "destination too small");
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        if (c->n_out)
            HCHECK(hipMemcpyAsync(dst, c->d_recs_out.p, (size_t)c->n_out * sizeof(himut_record), hipMemcpyDeviceToDevice, c->stream));
        HCHECK(hipStreamSynchronize(c->stream));
        return HIMUT_OK;
    });
}

namespace {

int do_normcounts(himut_ctx* c, const uint8_t* alt_order, int non_human, bool force_tile = false, int attempt = 0) {
    if (!c->have_params) return fail(c, HIMUT_ERR_ARG, "himut_set_params has not been called");
    if (!c->have_lut) return fail(c, HIMUT_ERR_ARG, "himut_set_gt_lut has not been called");
    if (!c->have_reads) return fail(c, HIMUT_ERR_ARG, "himut_push_reads has not been called");
    if (c->reflen <= 0) return fail(c, HIMUT_ERR_ARG, "himut_set_reference has not been called");
    const bool phase = c->params.p.phase != 0;
    if (phase && !c->have_phase) return fail(c, HIMUT_ERR_ARG, "phase requested but himut_set_phase has not been called");
    if (phase && (int64_t)c->h_phoff.size() != (int64_t)c->cstart.size() + 1)
        return fail(c, HIMUT_ERR_ARG, "himut_set_phase chunk count differs from himut_set_chunks");
    for (size_t k = 0; k < c->cstart.size(); k++)
        if (c->cstart[k] > c->cend[k]) return fail(c, HIMUT_ERR_CHUNK, "ValueError: invalid coordinates: chunk start > end");
    for (int k = 0; k < 12; k++) if (alt_order[k] > 3) return fail(c, HIMUT_ERR_ARG, "alt_order holds alleles 0..3");
    HCHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->have_norm = false;
    memset(&c->stats, 0, sizeof(c->stats));
    c->params.unique_qnames = c->unique_qnames ? 1 : 0;

    if (c->cstart.size() > 65535) return fail(c, HIMUT_ERR_ARG, "more than 65,535 chunks in one contig (the sweep's grids take a chunk per row)");
    ChunkTables T = upload_chunks(c, c->cstart, c->cend);
    alloc_derived(c);
    Reads R = make_reads(c);
    Derived D = make_derived(c);
    Chunks C = make_chunks(c, T.n);
    if (phase) c->d_hap.reserve((size_t)T.npairs + 64);
    Phase H = make_phase(c);
    Scalars* sc = c->d_scalars.as<Scalars>();
    c->lead_clean_bytes = 0;           // (the call path's scalars and bitmap are used here)
    const int K = c->ref_K;
    const size_t ntri = (size_t)K * K * K;
    c->d_tri.reserve((2 * ntri + 16) * 8);
    c->d_live.reserve((size_t)c->n + 64);
    const size_t cwords = (size_t)(c->bq_bytes >> 5) + 64;
    c->d_callable.reserve(cwords * 4);
    // the sweep: k_norm_quad (a wave per 256 columns), k_norm_dirty for the positions it lists, k_norm_tile for the tiles it
    // lists; the whole contig with k_norm_tile when one of the two lists was too short (force_tile), or when a test asks
    const bool sweep_tile = force_tile || c->dbg_norm_sweep == 1;
    const bool sweep_quad = !sweep_tile;
    // the sweep's grid (workgroups of NQ_WAVES waves, a wave per 256 positions; NQ_Q workgroups per XCD class and chunk),
    // and the list of the positions k_norm_quad leaves to k_norm_dirty (a column with another allele: one in thirty): a part
    // per workgroup, room for one of its positions in four
    int32_t maxspan = 1;
    for (size_t k = 0; k < c->cstart.size(); k++) maxspan = std::max(maxspan, c->cend[k] - c->cstart[k]);
    const int64_t q_per = ((int64_t)blocks_for(maxspan, NQ_WG_COLS) + 7) / 8;             // workgroup tiles of a chunk per XCD class
    // (NQ_Q workgroups per class and chunk keep a wave on a dozen tiles of a long contig; a contig of a few chunks gets
    //  more of them, so that the grid still fills the chip: about 4096 workgroups where the tiles allow)
    const int64_t q_want = std::max<int64_t>(NQ_Q, (4096 + 8 * std::max<int64_t>(T.n, 1) - 1) / (8 * std::max<int64_t>(T.n, 1)));
    const unsigned q_gx = 8u * (unsigned)std::min<int64_t>(q_want, q_per);
    const int64_t q_regions = (int64_t)q_gx * (int64_t)std::max<int64_t>(T.n, 1);
    const int64_t q_tiles_per_wg = (q_per + (q_gx / 8) - 1) / (q_gx / 8);
    int64_t dirty_cap = q_tiles_per_wg * NQ_WAVES * NQ_SLOTS + 64;     // (what the waves' pools can hold: a quarter of the positions)
    dirty_cap = std::min<int64_t>(std::max(dirty_cap, c->norm_dirty_room), q_tiles_per_wg * NQ_WG_COLS);
    if (c->dbg_norm_dirty_cap > 0 && attempt == 0) dirty_cap = c->dbg_norm_dirty_cap;   // (tests: the first pass overflows)
    // tiles left to k_norm_tile (more pieces than the plan holds, more columns with another allele than a wave's pool): room for
    // every tile of the contig
    const unsigned redo_cap = (unsigned)std::min<int64_t>((int64_t)std::max<int64_t>(T.n, 1) * blocks_for(maxspan, NQ_COLS) + 64, (int64_t)1 << 28);
    if (sweep_quad) {
        c->d_dirty.reserve((size_t)dirty_cap * (size_t)q_regions * sizeof(NormDirty) + 256);
        c->d_dcount.reserve((size_t)q_regions * 4 + 256);
        c->d_redo.reserve((size_t)redo_cap * sizeof(NormRedo) + 256);
    }

    HCHECK(hipEventRecord(c->ev[EV_START], st));
    flag_bases_once(c, st);
    HCHECK(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
    HCHECK(hipMemsetAsync(c->d_ccs.p, 0, (size_t)c->n + 1, st));
    HCHECK(hipMemsetAsync(c->d_tri.p, 0, (2 * ntri + 16) * 8, st));
    // (d_callable is not cleared: k_callable writes the words of every read)
    if (c->n > 0) run_parse_stage(c, R, D, sc);   // (the quality sums are k_callable's)
    else stage_event(c, EV_PARSE, 2, st);
    int32_t maxend = 0;
    for (int32_t e : c->cend) maxend = std::max(maxend, e);
    const int64_t nblk = ((int64_t)maxend >> WIN_SHIFT) + 2;
    c->d_winlo.reserve((size_t)nblk * 4 + 64);
    c->d_winhi.reserve((size_t)nblk * 4 + 64);
    if (c->n > 0 && T.n > 0) {
        hipLaunchKernelGGL(k_read_live, dim3(blocks_for(c->n, 16)), dim3(256), 0, st, R, D, C, c->params,
                           c->d_live.as<uint8_t>(), c->d_ccs.as<uint8_t>(), &sc->err);
        // (k_callable takes the reads with a low mean quality out of `live`: before the phased runs' count of the reads)
        hipLaunchKernelGGL(k_callable, dim3(blocks_for(c->n, 4)), dim3(256), 0, st, R, D, c->params, c->d_live.as<uint8_t>(),
                           c->d_callable.as<uint32_t>(), c->d_ccs.as<uint8_t>());
        if (phase && T.npairs > 0) {
            hipLaunchKernelGGL(k_read_hap, dim3((unsigned)blocks_for(T.maxpairs, 16), (unsigned)T.n), dim3(256), 0, st, R, D, C, H, &sc->err);
            hipLaunchKernelGGL(k_pair_ccs, dim3(blocks_for(T.npairs, 256)), dim3(256), 0, st, C, H, R, c->d_live.as<uint8_t>(),
                               T.npairs, c->d_ccs.as<uint8_t>());
        }
        hipLaunchKernelGGL(k_window_index, dim3(blocks_for(nblk, 256)), dim3(256), 0, st, R, nblk, c->d_winlo.as<int32_t>(),
                           c->d_winhi.as<int32_t>());
    }
    HCHECK(hipEventRecord(c->ev[EV_EMIT], st));

    NormArgs A;
    A.P = c->params;
    A.S.pon = c->d_pon.as<uint64_t>(); A.S.npon = c->npon; A.S.com = c->d_com.as<uint64_t>(); A.S.ncom = c->ncom;
    A.S.posbits = c->d_posbits.as<uint32_t>(); A.S.nposbits = c->nposbits;
    A.lut = c->d_lut.as<GtLut>();
    A.R = R; A.C = C; A.H = H;
    A.refseq = c->d_refseq.as<uint8_t>(); A.reflen = c->reflen;
    memcpy(A.cls, c->ref_cls, 256);
    A.K = K; A.cA = c->ref_cls['A']; A.cC = c->ref_cls['C']; A.cG = c->ref_cls['G']; A.cT = c->ref_cls['T'];
    memcpy(A.alt_order, alt_order, 12);
    A.non_human = non_human;
    A.ccs_tri = c->d_tri.as<unsigned long long>(); A.ref_tri = A.ccs_tri + ntri; A.log = A.ccs_tri + 2 * ntri;
    A.err = &sc->err;
    if (c->n > 0 && T.n > 0) {
        A.X = PosIndex{}; A.colstore = nullptr; A.p_lo = 0; A.p_hi = 0;
        if (sweep_quad) {
            // the plan: which pieces of which reads lie over each tile of 256 positions (k_norm_plan), then the sweep
            const int64_t tpc = (int64_t)blocks_for(maxspan, NQ_COLS);                  // tiles per chunk (the plan's stride)
            c->d_plan.reserve((size_t)T.n * (size_t)tpc * NQ_ITEMS * sizeof(NqItem) + 256);
            c->d_plancnt.reserve((size_t)T.n * (size_t)tpc * 4 + 256);
            const dim3 pgrid((unsigned)blocks_for(blocks_for(tpc, NQ_PLAN_TILES), 4), (unsigned)T.n);
            if (phase) hipLaunchKernelGGL(k_norm_plan<true>, pgrid, dim3(256), 0, st, A, D, c->d_winlo.as<int32_t>(), c->d_winhi.as<int32_t>(),
                                          nblk, tpc, c->d_plan.as<NqItem>(), c->d_plancnt.as<uint32_t>(), c->d_redo.as<NormRedo>(),
                                          &sc->nredo, redo_cap);
            else hipLaunchKernelGGL(k_norm_plan<false>, pgrid, dim3(256), 0, st, A, D, c->d_winlo.as<int32_t>(), c->d_winhi.as<int32_t>(),
                                    nblk, tpc, c->d_plan.as<NqItem>(), c->d_plancnt.as<uint32_t>(), c->d_redo.as<NormRedo>(),
                                    &sc->nredo, redo_cap);
            const dim3 grid(q_gx, (unsigned)T.n);
            stage_event(c, EV_INDEX, 1, st);                                            // (around k_norm_quad: stats.ms_capture)
            const unsigned pool_limit = c->dbg_norm_pool > 0 ? (unsigned)std::min(c->dbg_norm_pool, NQ_SLOTS) : (unsigned)NQ_SLOTS;
            if (phase) hipLaunchKernelGGL(k_norm_quad<true>, grid, dim3(NQ_WAVES * 64), 0, st, A, c->d_callable.as<uint32_t>(), (int64_t)c->bq_bytes,
                                          c->d_refcode.as<uint16_t>(), c->d_plan.as<NqItem>(), c->d_plancnt.as<uint32_t>(), tpc, q_per, c->d_dirty.as<NormDirty>(),
                                          c->d_dcount.as<uint32_t>(), dirty_cap, &sc->dirty_over, c->d_redo.as<NormRedo>(),
                                          &sc->nredo, redo_cap, pool_limit);
            else hipLaunchKernelGGL(k_norm_quad<false>, grid, dim3(NQ_WAVES * 64), 0, st, A, c->d_callable.as<uint32_t>(), (int64_t)c->bq_bytes,
                                    c->d_refcode.as<uint16_t>(), c->d_plan.as<NqItem>(), c->d_plancnt.as<uint32_t>(), tpc, q_per, c->d_dirty.as<NormDirty>(),
                                    c->d_dcount.as<uint32_t>(), dirty_cap, &sc->dirty_over, c->d_redo.as<NormRedo>(),
                                    &sc->nredo, redo_cap, pool_limit);
            stage_event(c, EV_GATHER, 1, st);
            hipLaunchKernelGGL(k_norm_dirty, dim3((unsigned)std::min<int64_t>(blocks_for(q_regions, 4), 16384)), dim3(256), 0, st, A,
                               c->d_dirty.as<NormDirty>(), c->d_dcount.as<uint32_t>(), dirty_cap, q_regions);
            // (returns at once unless a tile was listed)
            hipLaunchKernelGGL(k_norm_tile, dim3(1024), dim3(256), 0, st, A, D, c->d_callable.as<uint32_t>(), c->d_winlo.as<int32_t>(),
                               c->d_winhi.as<int32_t>(), nblk, (int64_t)0, c->d_redo.as<NormRedo>(), &sc->nredo, redo_cap);
        } else {
            const int64_t per = ((int64_t)blocks_for(maxspan, 256) + 7) / 8;
            const dim3 grid(8u * (unsigned)std::min<int64_t>(NT_Q, per), (unsigned)T.n);
            hipLaunchKernelGGL(k_norm_tile, grid, dim3(256), 0, st, A, D, c->d_callable.as<uint32_t>(), c->d_winlo.as<int32_t>(),
                               c->d_winhi.as<int32_t>(), nblk, per, (const NormRedo*)nullptr, (const unsigned int*)nullptr, 0u);
        }
    }
    if (c->n > 0 && T.n > 0)
        hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, st, c->d_ccs.as<uint8_t>(), c->n, &sc->nccs);
    HCHECK(hipEventRecord(c->ev[EV_FINAL], st));
    c->h_tri.assign(2 * ntri + 16, 0ULL);
    Scalars hs;
    HCHECK(hipMemcpyAsync(c->h_tri.data(), c->d_tri.p, (2 * ntri + 16) * 8, hipMemcpyDeviceToHost, st));
    HCHECK(hipMemcpyAsync(&hs, sc, sizeof(Scalars), hipMemcpyDeviceToHost, st));
    HCHECK(hipStreamSynchronize(st));
    // The list of positions left to k_norm_dirty was too short in some part (a region where more than one position in four
    // holds another allele: deep piles, a sample far from the reference): the same sweep once more with the room the
    // counters say it needs -- the context keeps it for its later passes, as himut_run keeps its capacities.  The list of
    // tiles was too short (or the room still is, which the counters rule out): the whole contig with k_norm_tile.
    if (hs.dirty_over && !force_tile && attempt == 0 && hs.nredo <= redo_cap) {
        std::vector<uint32_t> need((size_t)q_regions);
        HCHECK(hipMemcpy(need.data(), c->d_dcount.p, (size_t)q_regions * 4, hipMemcpyDeviceToHost));
        uint32_t mx = 0;
        for (uint32_t v : need) mx = std::max(mx, v);
        c->norm_dirty_room = (int64_t)mx + (int64_t)mx / 8 + 64;
        return do_normcounts(c, alt_order, non_human, false, 1);
    }
    if ((hs.dirty_over || hs.nredo > redo_cap) && !force_tile) return do_normcounts(c, alt_order, non_human, true, attempt + 1);
    c->stats.reran = (force_tile || attempt > 0) ? 1 : 0;
    if (hs.err) return check_device_err(c, hs.err);
    c->h_tri[2 * ntri + 0] = hs.nccs;
    float f = 0;
    (void)hipEventElapsedTime(&f, c->ev[EV_START], c->ev[EV_FINAL]);
    c->stats.ms_total = (double)f;
    if (c->timing >= 2) {   // recorded by run_parse_stage only then (an unrecorded event leaves a sticky HIP error)
        (void)hipEventElapsedTime(&f, c->ev[EV_START], c->ev[EV_PARSE]);
        c->stats.ms_parse = (double)f;
        (void)hipEventElapsedTime(&f, c->ev[EV_PARSE], c->ev[EV_EMIT]);
        c->stats.ms_index = (double)f;      // the read pass: filters, callable bits, window index
    }
    (void)hipEventElapsedTime(&f, c->ev[EV_EMIT], c->ev[EV_FINAL]);
    c->stats.ms_eval = (double)f;           // the position sweep: plan, k_norm_quad, k_norm_dirty, listed tiles
    if (c->timing >= 1 && sweep_quad && c->n > 0 && T.n > 0) {
        (void)hipEventElapsedTime(&f, c->ev[EV_INDEX], c->ev[EV_GATHER]);
        c->stats.ms_capture = (double)f;    // k_norm_quad by itself, the pass's dominant kernel
    }
    c->stats.n_reads = c->n; c->stats.read_bases = c->read_bases; c->stats.positions = T.positions;
    c->stats.column_slots = hs.nredo;        // (normcounts: tiles k_norm_quad left to k_norm_tile)
    c->have_norm = true;
    return HIMUT_OK;
}

}  // namespace

int himut_set_reference(himut_ctx* c, const uint8_t* seq, int64_t len, const uint8_t* cls, int n_classes) {
    if (!c) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        if (!seq || len <= 0 || !cls || n_classes < 5 || n_classes > 32) return fail(c, HIMUT_ERR_ARG, "bad reference / class table");
        HCHECK(hipSetDevice(c->device));
        upload(c->d_refseq, seq, (size_t)len, c->stream);
        // the letters' codes for the sweep, with room behind the string (a tile's last lanes read past it: codes of 0)
        c->d_refcode.reserve(((size_t)len + 512) * 2);
        hipLaunchKernelGGL(k_ref_codes, dim3(2048), dim3(256), 0, c->stream, c->d_refseq.as<uint8_t>(), len, c->d_refcode.as<uint16_t>(), len + 512);
        HCHECK(hipStreamSynchronize(c->stream));
        c->reflen = len;
        memcpy(c->ref_cls, cls, 256);
        c->ref_K = n_classes;
        for (int k = 0; k < 256; k++) if (cls[k] >= n_classes) return fail(c, HIMUT_ERR_ARG, "class id out of range");
        return HIMUT_OK;
    });
}

int himut_run_edges(himut_ctx* c, const int32_t* hpos, const uint8_t* href, int64_t n_het, int min_bq, int min_mapq,
                    int64_t band, uint32_t* counts) {
    if (!c || !counts || band < 1 || n_het < 0 || (n_het && (!hpos || !href))) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        if (!c->have_reads) return fail(c, HIMUT_ERR_ARG, "himut_push_reads has not been called");
        if (!c->have_params) return fail(c, HIMUT_ERR_ARG, "himut_set_params has not been called");   // the cs decode reads them
        HCHECK(hipSetDevice(c->device));
        hipStream_t st = c->stream;
        alloc_derived(c);
        Reads R = make_reads(c);
        Derived D = make_derived(c);
        Scalars* sc = c->d_scalars.as<Scalars>();
    c->lead_clean_bytes = 0;           // (the call path's scalars and bitmap are used here)
        const size_t nc = (size_t)std::max<int64_t>(n_het, 1) * (size_t)band * 4;
        c->d_tmp.reserve(nc * 4 + 256);
        upload(c->d_hpos, hpos, (size_t)n_het, st);
        upload(c->d_href, href, (size_t)n_het, st);
        HCHECK(hipEventRecord(c->ev[EV_START], st));
        HCHECK(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
        HCHECK(hipMemsetAsync(c->d_tmp.p, 0, nc * 4, st));
        if (c->n > 0) {
            run_parse_stage(c, R, D, sc);
            if (n_het >= 2)
                hipLaunchKernelGGL(k_edges, dim3(blocks_for(c->n, 4)), dim3(256), 0, st, R, D, c->d_hpos.as<int32_t>(),
                                   c->d_href.as<uint8_t>(), n_het, min_bq, min_mapq, band, c->d_tmp.as<uint32_t>(), &sc->err);
        }
        HCHECK(hipEventRecord(c->ev[EV_FINAL], st));
        Scalars hs;
        HCHECK(hipMemcpyAsync(counts, c->d_tmp.p, nc * 4, hipMemcpyDeviceToHost, st));
        HCHECK(hipMemcpyAsync(&hs, sc, sizeof(Scalars), hipMemcpyDeviceToHost, st));
        HCHECK(hipStreamSynchronize(st));
        c->have_phase = false;        // d_hpos / d_href were reused
        float f = 0;
        (void)hipEventElapsedTime(&f, c->ev[EV_START], c->ev[EV_FINAL]);
        memset(&c->stats, 0, sizeof(c->stats));
        c->stats.ms_total = (double)f;
        c->stats.n_reads = c->n; c->stats.read_bases = c->read_bases;
        if (hs.err) return check_device_err(c, hs.err);
        return HIMUT_OK;
    });
}

int himut_ref_tricounts(himut_ctx* c, int64_t out[64]) {
    if (!c || !out) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        if (c->reflen <= 0) return fail(c, HIMUT_ERR_ARG, "himut_set_reference has not been called");
        HCHECK(hipSetDevice(c->device));
        c->d_tmp2.reserve(64 * 8 + 256);
        unsigned long long* d = c->d_tmp2.as<unsigned long long>();
        HCHECK(hipMemsetAsync(d, 0, 64 * 8, c->stream));
        hipLaunchKernelGGL(k_ref_tricounts, dim3(2048), dim3(256), 0, c->stream, c->d_refseq.as<uint8_t>(), c->reflen, d);
        unsigned long long h[64];
        HCHECK(hipMemcpyAsync(h, d, 64 * 8, hipMemcpyDeviceToHost, c->stream));
        HCHECK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 64; k++) out[k] = (int64_t)h[k];
        return HIMUT_OK;
    });
}

int himut_sbs96_counts(himut_ctx* c, const int32_t* pos0, const uint8_t* ref, const uint8_t* alt, int64_t n, int64_t out[99]) {
    if (!c || !out || n < 0 || (n && (!pos0 || !ref || !alt))) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int {
        if (c->reflen <= 0) return fail(c, HIMUT_ERR_ARG, "himut_set_reference has not been called");
        HCHECK(hipSetDevice(c->device));
        hipStream_t st = c->stream;
        const size_t nn = (size_t)std::max<int64_t>(n, 1);
        c->d_tmp.reserve(nn * 4 + 256);
        c->d_tmp2.reserve(nn * 2 + 99 * 8 + 512);
        uint8_t* d_ref = c->d_tmp2.as<uint8_t>();
        uint8_t* d_alt = d_ref + nn;
        unsigned long long* d_out = reinterpret_cast<unsigned long long*>(c->d_tmp2.as<uint8_t>() + ((2 * nn + 255) & ~(size_t)255));
        if (n) {
            HCHECK(hipMemcpyAsync(c->d_tmp.p, pos0, (size_t)n * 4, hipMemcpyHostToDevice, st));
            HCHECK(hipMemcpyAsync(d_ref, ref, (size_t)n, hipMemcpyHostToDevice, st));
            HCHECK(hipMemcpyAsync(d_alt, alt, (size_t)n, hipMemcpyHostToDevice, st));
        }
        HCHECK(hipMemsetAsync(d_out, 0, 99 * 8, st));
        if (n)
            hipLaunchKernelGGL(k_sbs96, dim3(std::min<unsigned>(blocks_for(n, 256), 2048u)), dim3(256), 0, st, c->d_refseq.as<uint8_t>(),
                               c->reflen, c->d_tmp.as<int32_t>(), d_ref, d_alt, n, d_out);
        unsigned long long h[99];
        HCHECK(hipMemcpyAsync(h, d_out, 99 * 8, hipMemcpyDeviceToHost, st));
        HCHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 99; k++) out[k] = (int64_t)h[k];
        return HIMUT_OK;
    });
}

int himut_run_normcounts(himut_ctx* c, const uint8_t* alt_order, int non_human_sample) {
    if (!c || !alt_order) return HIMUT_ERR_ARG;
    return guarded(c, [&]() -> int { return do_normcounts(c, alt_order, non_human_sample); });
}

int himut_debug_normcounts(himut_ctx* c, int sweep, int64_t dirty_cap, int pool_slots) {
    if (!c || sweep < 0 || sweep > 1 || dirty_cap < 0 || pool_slots < 0) return HIMUT_ERR_ARG;
    c->dbg_norm_sweep = sweep; c->dbg_norm_dirty_cap = dirty_cap; c->dbg_norm_pool = pool_slots;
    return HIMUT_OK;
}

int himut_get_normcounts(himut_ctx* c, int64_t* ccs_tri, int64_t* ref_tri, int64_t log[14]) {
    if (!c) return HIMUT_ERR_ARG;
    if (!c->have_norm) return fail(c, HIMUT_ERR_ARG, "himut_run_normcounts has not completed");
    const size_t ntri = (size_t)c->ref_K * c->ref_K * c->ref_K;
    for (size_t k = 0; k < ntri; k++) { ccs_tri[k] = (int64_t)c->h_tri[k]; ref_tri[k] = (int64_t)c->h_tri[ntri + k]; }
    for (int k = 0; k < 14; k++) log[k] = (int64_t)c->h_tri[2 * ntri + k];
    return HIMUT_OK;
}

int himut_pile_counts(himut_ctx* c, int32_t p0, int32_t p1, uint32_t* counts, uint32_t* bqsum) {
    if (!c || !counts || !bqsum || p1 <= p0) return fail(c, HIMUT_ERR_ARG, "bad pile range");
    if (!c->have_reads || !c->have_params || !c->have_lut) return fail(c, HIMUT_ERR_ARG, "context not initialised");
    return guarded(c, [&]() -> int {
        HCHECK(hipSetDevice(c->device));
        hipStream_t st = c->stream;
        // one pseudo chunk (p0, p1): its tiles cover rpos p0-1 .. p1-1, the columns p0 .. p1-1 are complete
        std::vector<int32_t> cs{p0}, ce{p1};
        ChunkTables T = upload_chunks(c, cs, ce);
        alloc_derived(c);
        Reads R = make_reads(c);
        Derived D = make_derived(c);
        Chunks C = make_chunks(c, 1);
        Scalars* sc = c->d_scalars.as<Scalars>();
    c->lead_clean_bytes = 0;           // (the call path's scalars and bitmap are used here)
        HCHECK(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
        if (c->n > 0) run_parse_stage(c, R, D, sc);
        c->d_dense_counts.reserve((size_t)T.positions * 6 * 4 + 64);
        c->d_dense_bqsum.reserve((size_t)T.positions * 4 * 4 + 64);
        HCHECK(hipMemsetAsync(c->d_dense_counts.p, 0, (size_t)T.positions * 24, st));
        HCHECK(hipMemsetAsync(c->d_dense_bqsum.p, 0, (size_t)T.positions * 16, st));
        if (c->n > 0) launch_pile_dense(c, C, R, D, T, &sc->err);
        const int64_t npos = (int64_t)p1 - p0;
        HCHECK(hipMemcpyAsync(counts, c->d_dense_counts.as<uint32_t>() + 6, (size_t)npos * 24, hipMemcpyDeviceToHost, st));
        HCHECK(hipMemcpyAsync(bqsum, c->d_dense_bqsum.as<uint32_t>() + 4, (size_t)npos * 16, hipMemcpyDeviceToHost, st));
        Scalars hs;
        HCHECK(hipMemcpyAsync(&hs, sc, sizeof(Scalars), hipMemcpyDeviceToHost, st));
        HCHECK(hipStreamSynchronize(st));
        if (hs.err) return check_device_err(c, hs.err);
        return HIMUT_OK;
    });
}

}  // extern "C"
