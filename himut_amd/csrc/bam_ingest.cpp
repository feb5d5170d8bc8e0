// Host-side BAM ingest for the himut hot path (own implementation over zlib; no htslib).
//
// Reads a coordinate-sorted BAM (BGZF) sequentially and builds, per reference
// sequence, the structure-of-arrays read batch that himut_push_reads() takes
// (himut_amd/readbatch.py): what `pysam.AlignmentFile.fetch` + `bamlib.BAM` deliver to the
// reference worker (src/himut/bamlib.py:14-32, caller.py:299-300).  Also writes such a
// batch back out as BAM (synthetic inputs for end-to-end runs and round-trip tests).
//
// Record fields kept: reference_start, reference_end (from CIGAR), leading soft clip
// (query_alignment_start), query length, MAPQ, FLAG, packed SEQ and QUAL as stored in the
// BAM, the cs:Z and tp:A tags, and the index of the first read with the same name.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <future>
#include <thread>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// HIMUT_INGEST_PROFILE=1: seconds per stage of bam_load_threads on stderr
struct Prof { double inflate0 = 0, wait = 0, hop = 0, decode = 0, place = 0, grow = 0, copy = 0; };
static Prof g_prof;
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// BGZF reader: the file is mapped, its block headers are walked once (no inflate), and
// the blocks are inflated a window (~64 MB of output) at a time by a pool of threads --
// BGZF blocks are independent deflate streams.  The window after the one being parsed is
// inflated in the background, so record parsing and inflate overlap.
// libdeflate (about twice as fast as zlib's inflate) is used when the shared library is on the
// system; its three entry points are looked up at run time, zlib is the fallback.
struct Deflate {
    void* (*alloc)() = nullptr;
    int (*run)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    Deflate() {
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        run = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !run || !release) { alloc = nullptr; run = nullptr; release = nullptr; }
    }
    bool ok() const { return run != nullptr; }
};
const Deflate& deflate_lib() { static Deflate d; return d; }

struct BlockRef {
    size_t file_off;      // first byte of the block in the file (virtual file offsets of the index point here)
    size_t cdata_off;     // first byte of the deflate stream
    uint32_t cdata_len;
    uint32_t isize;       // inflated size
    size_t uoff;          // offset inside its window
};

struct Bgzf {
    const uint8_t* base = nullptr;   // mapped file
    size_t fsize = 0;
    int fd = -1;
    std::vector<uint8_t> owned;      // fallback when mmap is not possible
    std::vector<BlockRef> blocks;
    std::vector<size_t> win_first;   // first block of every window, + one past the end
    int threads = 1;
    std::string err;
    bool eof = false;

    std::vector<uint8_t> buf[2];     // two windows: one being parsed, one being inflated
    size_t cur = 0;                  // window being parsed
    size_t pos = 0, len = 0;
    std::future<std::string> pending;
    bool started = false;

    size_t scan_parts = 1;           // stretches the block table was made from (threads used)
    size_t WINDOW = (size_t)64 << 20;   // HIMUT_INGEST_WINDOW_KB overrides (tests use small windows)

    bool open(const char* path, int nthreads, const std::vector<size_t>* hints = nullptr) {
        threads = nthreads < 1 ? 1 : nthreads;
        if (const char* e = getenv("HIMUT_INGEST_WINDOW_KB")) { const long kb = atol(e); if (kb > 0) WINDOW = (size_t)kb << 10; }
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0) { err = "fstat failed"; return false; }
        fsize = (size_t)st.st_size;
        if (fsize == 0) { err = "not a BAM file"; return false; }
        void* m = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            base = (const uint8_t*)m;
            (void)madvise(m, fsize, MADV_SEQUENTIAL);
        } else {
            owned.resize(fsize);
            size_t got = 0;
            while (got < fsize) {
                const ssize_t k = ::read(fd, owned.data() + got, fsize - got);
                if (k <= 0) { err = "read failed"; return false; }
                got += (size_t)k;
            }
            base = owned.data();
        }
        return scan(hints);
    }
    void close() {
        if (pending.valid()) (void)pending.get();
        if (base && owned.empty()) munmap((void*)base, fsize);
        if (fd >= 0) ::close(fd);
        base = nullptr; fd = -1;
    }
    // Block headers of the file range [p0, p1) -> (offset, compressed length, inflated length).  One pread per block:
    // the last four bytes of a block (its inflated length) and the header of the next block are neighbours.  (Through
    // the mapping a page fault per block costs several times more, and the pages are faulted in by the inflate threads
    // in parallel anyway.)  Returns "" or an error; "split" when the range does not end on a block boundary.
    std::string scan_range(size_t p0, size_t p1, std::vector<BlockRef>& out) const {
        uint8_t cur[64], nx[68];
        const bool pr = fd >= 0 && owned.empty();
        auto fetch = [&](size_t at, uint8_t* dst, size_t want) {
            const size_t w = std::min(want, fsize - at);
            if (!(pr && ::pread(fd, dst, w, (off_t)at) == (ssize_t)w)) memcpy(dst, base + at, w);
            return w;
        };
        size_t p = p0;
        if (p < p1) (void)fetch(p, cur, sizeof(cur));
        while (p < p1) {
            if (p + 18 > fsize) return "truncated BGZF header";
            const uint8_t* h = cur;
            if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return "not a BGZF block";
            const unsigned xlen = h[10] | (h[11] << 8);
            if (p + 12 + xlen > fsize) return "truncated BGZF header";
            if (12 + xlen > sizeof(cur)) h = base + p;          // unusually long extra field: through the mapping
            int bsize = -1;
            for (size_t k = 0; k + 4 <= xlen;) {      // BC is normally the first subfield; tolerate others
                const uint8_t* e = h + 12 + k;
                const unsigned slen = e[2] | (e[3] << 8);
                if (e[0] == 'B' && e[1] == 'C' && slen == 2 && k + 6 <= xlen) bsize = e[4] | (e[5] << 8);
                k += 4 + slen;
            }
            if (bsize < 0) return "BGZF block without BC field";
            const size_t total = (size_t)bsize + 1;
            if (total < 12 + xlen + 8 || p + total > fsize) return "truncated BGZF block";
            BlockRef b;
            b.file_off = p;
            b.cdata_off = p + 12 + xlen;
            b.cdata_len = (uint32_t)(total - 12 - xlen - 8);
            const size_t got = fetch(p + total - 4, nx, sizeof(nx));
            b.isize = nx[0] | (nx[1] << 8) | (nx[2] << 16) | ((uint32_t)nx[3] << 24);
            b.uoff = 0;
            out.push_back(b);
            p += total;
            memset(cur, 0, sizeof(cur));
            if (got > 4) memcpy(cur, nx + 4, got - 4);
        }
        return p == p1 ? "" : "split";
    }
    // The whole file's block table; windows of ~WINDOW output bytes.  ``hints``: file offsets at which blocks are known
    // to start (an index beside the file lists some): the table is then made by several threads, a stretch of the file
    // each; a hint that turns out wrong only costs the serial scan.
    bool scan(const std::vector<size_t>* hints = nullptr) {
        std::vector<size_t> cut(1, 0);
        const size_t T = (size_t)std::min(threads, 16);
        size_t min_size = (size_t)8 << 20;            // below it the serial scan is as fast (tests lower it)
        if (const char* e = getenv("HIMUT_INGEST_SCAN_MIN_KB")) min_size = (size_t)atol(e) << 10;
        if (hints && !hints->empty() && T > 1 && fsize > min_size)
            for (size_t i = 1; i < T; i++) {
                auto it = std::lower_bound(hints->begin(), hints->end(), fsize / T * i);
                if (it != hints->end() && *it > cut.back() && *it < fsize) cut.push_back(*it);
            }
        cut.push_back(fsize);
        const size_t np = cut.size() - 1;
        scan_parts = np;
        std::vector<std::vector<BlockRef>> part(np);
        std::vector<std::string> perr(np);
        if (np > 1) {
            std::vector<std::thread> pool;
            for (size_t i = 1; i < np; i++) pool.emplace_back([&, i]() { perr[i] = scan_range(cut[i], cut[i + 1], part[i]); });
            perr[0] = scan_range(cut[0], cut[1], part[0]);
            for (auto& th : pool) th.join();
            bool ok = true;
            for (const auto& e : perr) ok = ok && e.empty();
            if (!ok) { part.assign(1, {}); perr.assign(1, ""); cut = {0, fsize}; scan_parts = 1; }
        }
        if (part.size() == 1 && part[0].empty()) perr[0] = scan_range(0, fsize, part[0]);
        if (!perr[0].empty()) { err = perr[0] == "split" ? "truncated BGZF block" : perr[0]; return false; }
        size_t nb = 0;
        for (const auto& v : part) nb += v.size();
        blocks.reserve(nb);
        for (const auto& v : part) blocks.insert(blocks.end(), v.begin(), v.end());
        size_t wbytes = 0;
        win_first.push_back(0);
        for (size_t k = 0; k < blocks.size(); k++) {
            BlockRef& b = blocks[k];
            if (wbytes && wbytes + b.isize > WINDOW) { win_first.push_back(k); wbytes = 0; }
            b.uoff = wbytes;
            wbytes += b.isize;
        }
        win_first.push_back(blocks.size());
        return true;
    }
    size_t n_windows() const { return win_first.size() - 1; }
    // inflates window w into out with the pool; returns an error text or ""
    std::string inflate_window(size_t w, std::vector<uint8_t>& out) const {
        const size_t b0 = win_first[w], b1 = win_first[w + 1];
        size_t total = 0;
        for (size_t k = b0; k < b1; k++) total += blocks[k].isize;
        out.resize(total);
        return inflate_blocks(b0, b1, out.data(), blocks[b0 < b1 ? b0 : 0].uoff);
    }
    // inflates blocks [b0, b1) with the pool: block k lands at dst + (uoff[k] - base_uoff) when the blocks belong to one
    // window, or back to back from dst when ``packed`` (any range)
    std::string inflate_blocks(size_t b0, size_t b1, uint8_t* dst, size_t base_uoff, const std::vector<size_t>* packed_off = nullptr) const {
        std::atomic<size_t> next(b0);
        std::atomic<int> bad(0);
        auto work = [&]() {
            const Deflate& L = deflate_lib();
            void* dec = (L.ok() && !getenv("HIMUT_INGEST_ZLIB")) ? L.alloc() : nullptr;   // the variable forces zlib
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (!dec && inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            // The compressed bytes are taken with pread into a buffer of the thread's own: through the shared mapping every
            // first touch of a page is a fault that takes the address space's lock, and the pool's threads queue up on it.
            std::vector<uint8_t> cbuf;
            const bool use_pread = fd >= 0 && owned.empty() && !getenv("HIMUT_INGEST_MMAP");
            if (use_pread) cbuf.resize(1 << 16);
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= b1) break;
                const BlockRef& b = blocks[k];
                if (!b.isize) continue;
                uint8_t* to = packed_off ? dst + (*packed_off)[k - b0] : dst + (b.uoff - base_uoff);
                const uint8_t* cin = base + b.cdata_off;
                if (use_pread && b.cdata_len <= cbuf.size() && ::pread(fd, cbuf.data(), b.cdata_len, (off_t)b.cdata_off) == (ssize_t)b.cdata_len)
                    cin = cbuf.data();
                if (dec) {
                    size_t got = 0;
                    if (L.run(dec, cin, b.cdata_len, to, b.isize, &got) != 0 || got != b.isize) { bad = 2; break; }
                } else {
                    inflateReset(&zs);
                    zs.next_in = (Bytef*)cin; zs.avail_in = b.cdata_len;
                    zs.next_out = to; zs.avail_out = b.isize;
                    if (inflate(&zs, Z_FINISH) != Z_STREAM_END) { bad = 2; break; }
                }
            }
            if (dec) L.release(dec); else inflateEnd(&zs);
        };
        const int nt = (int)std::min<size_t>((size_t)threads, b1 - b0 ? b1 - b0 : 1);
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
        return bad == 0 ? "" : (bad == 1 ? "inflateInit2 failed" : "inflate failed");
    }
    bool next_window() {
        if (!started) {
            started = true;
            if (n_windows() == 0) { eof = true; return false; }
            const double t0 = now_s();
            const std::string e = inflate_window(0, buf[0]);
            g_prof.inflate0 += now_s() - t0;
            if (!e.empty()) { err = e; return false; }
            cur = 0;
        } else {
            if (cur + 1 >= n_windows()) { eof = true; return false; }
            const double t0 = now_s();
            const std::string e = pending.get();
            g_prof.wait += now_s() - t0;
            if (!e.empty()) { err = e; return false; }
            cur++;
        }
        if (cur + 1 < n_windows()) {
            const size_t w = cur + 1;
            pending = std::async(std::launch::async, [this, w]() { return inflate_window(w, buf[w & 1]); });
        }
        pos = 0; len = buf[cur & 1].size();
        return true;
    }
    // n bytes of the stream without a copy when they lie inside the current window (scratch otherwise)
    const uint8_t* view(size_t n, std::vector<uint8_t>& scratch) {
        if (pos == len && !next_window()) return nullptr;
        if (len - pos >= n) { const uint8_t* p = buf[cur & 1].data() + pos; pos += n; return p; }
        scratch.resize(n);
        return read(scratch.data(), n) ? scratch.data() : nullptr;
    }
    bool read(void* dst, size_t n) {
        uint8_t* d = (uint8_t*)dst;
        while (n) {
            if (pos == len) { if (!next_window()) return false; continue; }
            const size_t k = len - pos < n ? len - pos : n;
            memcpy(d, buf[cur & 1].data() + pos, k);
            d += k; pos += k; n -= k;
        }
        return true;
    }
};

// growable byte buffer that does not zero what it grows by
// A contig's big arrays.  The copy pass that fills them is bound by first-touch page faults, so the arrays live in
// one anonymous mapping reserved up front (address space only: MAP_NORESERVE) and backed by huge pages where the
// kernel hands them out; growing inside the reservation costs nothing.  Without a reservation (or past it) the
// buffer grows by realloc / mremap.
struct RawBuf {
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    bool mapped = false;
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    RawBuf(RawBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap), mapped(o.mapped) { o.p = nullptr; o.n = o.cap = 0; o.mapped = false; }
    ~RawBuf() { if (mapped) munmap(p, cap); else free(p); }
    void reserve(size_t bytes) {
        if (p || bytes < ((size_t)8 << 20)) return;
        bytes = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) return;
        (void)madvise(m, bytes, MADV_HUGEPAGE);
        p = (uint8_t*)m; cap = bytes; mapped = true;
    }
    bool grow(size_t want) {
        if (want > cap) {
            size_t nc = cap ? cap : 4096;
            while (nc < want) nc += nc / 2 + 4096;
            if (mapped) {
                void* q = mremap(p, cap, nc, MREMAP_MAYMOVE);
                if (q == MAP_FAILED) return false;
                p = (uint8_t*)q; cap = nc;
            } else {
                uint8_t* q = (uint8_t*)realloc(p, nc);
                if (!q) return false;
                p = q; cap = nc;
            }
        }
        n = want;
        return true;
    }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
};

struct Contig {
    std::string name;
    int64_t length = 0;
    std::vector<int32_t> tstart, tend, qstart, qlen, qid;
    std::vector<uint8_t> mapq, tp;
    std::vector<uint16_t> flag;
    std::vector<int64_t> qoff, cs_off;
    RawBuf seq, bq, cs;
    std::unordered_map<std::string, int32_t> first_by_name;
    int64_t bases_padded = 0, cs_n = 0;
};

struct Bam {
    std::string header_text, err;
    std::vector<Contig> contigs;
    int64_t n_missing_cs = 0, n_unmapped = 0, n_unsorted = 0;
};

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// walks the auxiliary fields; returns false on a malformed block
bool scan_tags(const uint8_t* p, const uint8_t* end, const uint8_t** cs, size_t* cs_len, uint8_t* tp) {
    *cs = nullptr; *cs_len = 0; *tp = 0;
    while (p + 3 <= end) {
        const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
        p += 3;
        size_t sz = 0;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': {
                const uint8_t* q = p;
                while (q < end && *q) q++;
                if (q >= end) return false;
                if (t0 == 'c' && t1 == 's' && ty == 'Z') { *cs = p; *cs_len = (size_t)(q - p); }
                p = q + 1;
                continue;
            }
            case 'B': {
                if (p + 5 > end) return false;
                const char sub = (char)p[0];
                const uint32_t cnt = le32(p + 1);
                size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                p += 5 + es * cnt;
                continue;
            }
            default: return false;
        }
        if (p + sz > end) return false;
        if (t0 == 't' && t1 == 'p' && ty == 'A') *tp = p[0];
        p += sz;
    }
    return true;
}

void w32(std::vector<uint8_t>& v, uint32_t x) { for (int k = 0; k < 4; k++) v.push_back((uint8_t)(x >> (8 * k))); }

struct BgzfWriter {
    FILE* f;
    std::vector<uint8_t> buf;
    bool ok = true;
    uint64_t foff = 0;      // bytes of finished blocks
    uint64_t voffset() const { return (foff << 16) | (uint64_t)buf.size(); }   // virtual file offset of the next byte
    void flush_block(const uint8_t* data, size_t n) {
        std::vector<uint8_t> comp(n + 1024);
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)data; zs.avail_in = (uInt)n;
        zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n);
        const uint16_t bsize = (uint16_t)(clen + 25);
        uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255), (uint8_t)(bsize >> 8)};
        uint8_t tail[8];
        for (int k = 0; k < 4; k++) { tail[k] = (uint8_t)(crc >> (8 * k)); tail[4 + k] = (uint8_t)((uint32_t)n >> (8 * k)); }
        ok = ok && fwrite(hdr, 1, 18, f) == 18 && fwrite(comp.data(), 1, clen, f) == clen && fwrite(tail, 1, 8, f) == 8;
        foff += 18 + clen + 8;
    }
    void write(const void* p, size_t n) {
        const uint8_t* d = (const uint8_t*)p;
        while (n) {
            size_t k = 0xff00 - buf.size() < n ? 0xff00 - buf.size() : n;
            buf.insert(buf.end(), d, d + k);
            d += k; n -= k;
            if (buf.size() == 0xff00) { flush_block(buf.data(), buf.size()); buf.clear(); }
        }
    }
    void finish() {
        if (!buf.empty()) { flush_block(buf.data(), buf.size()); buf.clear(); }
        flush_block(nullptr, 0);  // EOF marker block
    }
};

}  // namespace

extern "C" {

// Loads the whole file with `threads` inflate threads (0: one per hardware thread, at most 16).
// Returns a handle (never null); check bam_error().
static void* bam_load_threads_impl(Bam* B, const char* path, int threads);

void* bam_load_threads(const char* path, int threads) {
    Bam* B = new Bam();
    try {
        return bam_load_threads_impl(B, path, threads);
    } catch (const std::exception& e) {         // e.g. bad_alloc / length_error on a corrupt header: an error, not an abort
        B->err = std::string("BAM load failed: ") + e.what();
        return B;
    }
}

static void* bam_load_threads_impl(Bam* B, const char* path, int threads) {
    Bgzf z;
    if (threads <= 0) {
        const char* e = getenv("HIMUT_INGEST_THREADS");
        threads = e ? atoi(e) : 0;
        if (threads <= 0) threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    }
    if (!z.open(path, threads)) { B->err = z.err; z.close(); return B; }
    uint8_t magic[4];
    uint8_t b4[4];
    auto fail = [&](const std::string& m) { B->err = m.empty() ? "unexpected end of BAM" : m; z.close(); return (void*)B; };
    if (!z.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) return fail(z.err.empty() ? "not a BAM file" : z.err);
    if (!z.read(b4, 4)) return fail(z.err);
    const uint32_t l_text = le32(b4);
    size_t inflated_all = 0;                // sizes read from the header are checked against what the file can hold
    for (const BlockRef& b : z.blocks) inflated_all += b.isize;
    if ((size_t)l_text > inflated_all) return fail("BAM header text longer than the file");
    B->header_text.resize(l_text);
    if (l_text && !z.read(&B->header_text[0], l_text)) return fail(z.err);
    while (!B->header_text.empty() && B->header_text.back() == '\0') B->header_text.pop_back();
    if (!z.read(b4, 4)) return fail(z.err);
    const uint32_t n_ref = le32(b4);
    if ((size_t)n_ref * 8 > inflated_all) return fail("BAM header lists more contigs than the file can hold");
    B->contigs.resize(n_ref);
    for (uint32_t i = 0; i < n_ref; i++) {
        if (!z.read(b4, 4)) return fail(z.err);
        const uint32_t l_name = le32(b4);
        if ((size_t)l_name > inflated_all) return fail("contig name longer than the file");
        std::string nm(l_name, '\0');
        if (!z.read(&nm[0], l_name)) return fail(z.err);
        while (!nm.empty() && nm.back() == '\0') nm.pop_back();
        if (!z.read(b4, 4)) return fail(z.err);
        B->contigs[i].name = nm;
        B->contigs[i].length = le32(b4);
    }
    // Records are parsed a window at a time: the record boundaries of the window are found by hopping from
    // length field to length field, the records are decoded by the pool (CIGAR walk, tag scan), a short
    // sequential pass assigns every kept record its place in its contig's arrays, and the pool copies the bytes.
    struct RecInfo {
        const uint8_t* rec; uint32_t bs;
        int32_t ref_id, pos, lead_clip; int64_t ref_len;
        uint32_t l_seq, cs_len; uint16_t flag; uint8_t mapq, tp, l_qname, status;   // status: 0 keep, 1 unmapped, 2 no cs
        const uint8_t *seq, *qual, *cs; const char* qname;
        int64_t dst_bases, dst_cs; Contig* C;
    };
    std::string perr;
    size_t inflated_total = 0;              // of the whole file: an upper bound for any contig's bytes
    for (const BlockRef& b : z.blocks) inflated_total += b.isize;
    auto decode = [&](RecInfo& I) -> bool {
        const uint8_t* rec = I.rec;
        const uint32_t bs = I.bs;
        I.ref_id = (int32_t)le32(&rec[0]);
        I.pos = (int32_t)le32(&rec[4]);
        I.l_qname = rec[8];
        I.mapq = rec[9];
        const uint16_t n_cigar = le16(&rec[12]);
        I.flag = le16(&rec[14]);
        I.l_seq = le32(&rec[16]);
        I.status = 0;
        if (I.ref_id < 0 || (uint32_t)I.ref_id >= n_ref || (I.flag & 4)) { I.status = 1; return true; }
        size_t o = 32;
        if (o + I.l_qname + 4ull * n_cigar + (I.l_seq + 1) / 2 + I.l_seq > bs) return false;
        I.qname = (const char*)&rec[o];
        o += I.l_qname;
        int64_t ref_len = 0;
        int32_t lead_clip = 0;
        bool seen_query = false;
        for (uint16_t k = 0; k < n_cigar; k++) {
            const uint32_t c = le32(&rec[o + 4 * k]);
            const uint32_t op = c & 15, ln = c >> 4;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += ln;   // M D N = X
            if (op == 4 && !seen_query) lead_clip += (int32_t)ln;                   // leading S
            if (op == 0 || op == 1 || op == 7 || op == 8) seen_query = true;        // M I = X
        }
        I.ref_len = ref_len; I.lead_clip = lead_clip;
        o += 4ull * n_cigar;
        I.seq = &rec[o];
        o += (I.l_seq + 1) / 2;
        I.qual = &rec[o];
        o += I.l_seq;
        size_t cs_len = 0;
        if (!scan_tags(&rec[o], rec + bs, &I.cs, &cs_len, &I.tp)) return false;
        I.cs_len = (uint32_t)cs_len;
        if (!I.cs) I.status = 2;
        return true;
    };
    auto place = [&](RecInfo& I) {          // sequential: order matters
        if (I.status == 1) { B->n_unmapped++; return; }
        if (I.status == 2) { B->n_missing_cs++; return; }
        Contig& C = B->contigs[(size_t)I.ref_id];
        if (!C.tstart.empty() && I.pos < C.tstart.back()) B->n_unsorted++;
        if (C.tstart.empty()) {             // first record of the contig: reserve address space for its arrays
            const size_t bound = std::min<size_t>(2 * inflated_total, (size_t)48 << 30);
            C.bq.reserve(bound); C.seq.reserve(bound / 2); C.cs.reserve(std::min<size_t>(inflated_total, (size_t)16 << 30));
        }
        const int32_t idx = (int32_t)C.tstart.size();
        C.tstart.push_back(I.pos);
        C.tend.push_back((int32_t)(I.pos + I.ref_len));
        C.qstart.push_back(I.lead_clip);
        C.qlen.push_back((int32_t)I.l_seq);
        C.mapq.push_back(I.mapq);
        C.flag.push_back(I.flag);
        C.tp.push_back(I.tp);
        auto it = C.first_by_name.emplace(std::string(I.qname, strnlen(I.qname, I.l_qname)), idx);
        C.qid.push_back(it.first->second);
        C.qoff.push_back(C.bases_padded);
        I.C = &C; I.dst_bases = C.bases_padded; I.dst_cs = C.cs_n;
        C.bases_padded += ((int64_t)I.l_seq + 31) & ~(int64_t)31;
        C.cs_off.push_back(C.cs_n);
        C.cs_n += I.cs_len;
    };
    auto copy_bytes = [&](const RecInfo& I) {
        if (I.status) return;
        Contig& C = *I.C;
        const int64_t padded = ((int64_t)I.l_seq + 31) & ~(int64_t)31;
        uint8_t* sq = C.seq.p + I.dst_bases / 2;
        const size_t nsq = (I.l_seq + 1) / 2;
        memcpy(sq, I.seq, nsq);
        if (I.l_seq & 1) sq[nsq - 1] &= 0xf0;
        memset(sq + nsq, 0, (size_t)(padded / 2) - nsq);               // pad to the 32-base boundary
        uint8_t* bq = C.bq.p + I.dst_bases;
        memcpy(bq, I.qual, I.l_seq);
        memset(bq + I.l_seq, 0, (size_t)padded - I.l_seq);
        if (I.cs_len) memcpy(C.cs.p + I.dst_cs, I.cs, I.cs_len);
    };
    auto run_pool = [&](size_t count, const std::function<void(size_t)>& f) {
        const int nt = (int)std::min<size_t>((size_t)z.threads, count / 64 + 1);
        if (nt <= 1) { for (size_t k = 0; k < count; k++) f(k); return; }
        std::atomic<size_t> next(0);
        auto work = [&]() { for (;;) { const size_t k0 = next.fetch_add(64); if (k0 >= count) break; for (size_t k = k0; k < std::min(count, k0 + 64); k++) f(k); } };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    };
    auto finish_batch = [&](std::vector<RecInfo>& recs) -> bool {
        std::atomic<int> bad(0);
        double t0 = now_s();
        run_pool(recs.size(), [&](size_t k) { if (!decode(recs[k])) bad = 1; });
        g_prof.decode += now_s() - t0;
        if (bad) { perr = "malformed BAM record"; return false; }
        t0 = now_s();
        for (auto& I : recs) place(I);
        g_prof.place += now_s() - t0;
        t0 = now_s();
        for (auto& C : B->contigs)
            if (!C.seq.grow((size_t)(C.bases_padded / 2)) || !C.bq.grow((size_t)C.bases_padded) || !C.cs.grow((size_t)C.cs_n)) {
                perr = "out of memory"; return false;
            }
        g_prof.grow += now_s() - t0;
        t0 = now_s();
        run_pool(recs.size(), [&](size_t k) { copy_bytes(recs[k]); });
        g_prof.copy += now_s() - t0;
        recs.clear();
        return true;
    };
    std::vector<uint8_t> scratch;
    std::vector<RecInfo> recs;
    for (;;) {
        // whole records inside the current window
        if (z.pos == z.len && !z.next_window()) { if (z.eof && z.err.empty()) break; return fail(z.err); }
        const uint8_t* wbuf = z.buf[z.cur & 1].data();
        const double t_hop = now_s();
        while (z.pos + 4 <= z.len) {
            const uint32_t bs = le32(wbuf + z.pos);
            if (bs < 32) return fail("BAM record too short");
            if (z.pos + 4 + (size_t)bs > z.len) break;
            RecInfo I;
            memset(&I, 0, sizeof(I));
            I.rec = wbuf + z.pos + 4; I.bs = bs;
            recs.push_back(I);
            z.pos += 4 + (size_t)bs;
        }
        g_prof.hop += now_s() - t_hop;
        if (!finish_batch(recs)) return fail(perr);
        if (z.pos == z.len) continue;
        // a record that runs into the next window: assembled in scratch, handled on its own
        if (!z.read(b4, 4)) { if (z.eof && z.err.empty()) break; return fail(z.err); }
        const uint32_t bs = le32(b4);
        if (bs < 32) return fail("BAM record too short");
        const uint8_t* rec = z.view(bs, scratch);
        if (!rec) return fail(z.err.empty() ? "truncated BAM record" : z.err);
        RecInfo I;
        memset(&I, 0, sizeof(I));
        I.rec = rec; I.bs = bs;
        recs.push_back(I);
        if (!finish_batch(recs)) return fail(perr);
    }
    for (auto& C : B->contigs) { C.cs_off.push_back(C.cs_n); C.first_by_name.clear(); }
    z.close();
    if (getenv("HIMUT_INGEST_PROFILE")) {
        fprintf(stderr, "ingest profile (s): first window %.3f, waiting for inflate %.3f, hop %.3f, decode %.3f, place %.3f, "
                        "grow %.3f, copy %.3f (threads %d)\n", g_prof.inflate0, g_prof.wait, g_prof.hop, g_prof.decode,
                g_prof.place, g_prof.grow, g_prof.copy, threads);
        g_prof = Prof();
    }
    return B;
}

void* bam_load(const char* path) { return bam_load_threads(path, 0); }

const char* bam_error(void* h) { return ((Bam*)h)->err.c_str(); }
const char* bam_header_text(void* h) { return ((Bam*)h)->header_text.c_str(); }
int64_t bam_n_ref(void* h) { return (int64_t)((Bam*)h)->contigs.size(); }
const char* bam_ref_name(void* h, int64_t i) { return ((Bam*)h)->contigs[(size_t)i].name.c_str(); }
int64_t bam_ref_len(void* h, int64_t i) { return ((Bam*)h)->contigs[(size_t)i].length; }
int64_t bam_ref_nreads(void* h, int64_t i) { return (int64_t)((Bam*)h)->contigs[(size_t)i].tstart.size(); }
int64_t bam_ref_bases_padded(void* h, int64_t i) { return ((Bam*)h)->contigs[(size_t)i].bases_padded; }
int64_t bam_ref_cs_bytes(void* h, int64_t i) { return (int64_t)((Bam*)h)->contigs[(size_t)i].cs.size(); }
int64_t bam_count(void* h, int what) {
    Bam* B = (Bam*)h;
    return what == 0 ? B->n_missing_cs : what == 1 ? B->n_unmapped : B->n_unsorted;
}

void bam_ref_copy(void* h, int64_t i, int32_t* tstart, int32_t* tend, int32_t* qstart, int32_t* qlen, uint8_t* mapq,
                  uint16_t* flag, int32_t* qid, int64_t* qoff, int64_t* cs_off, uint8_t* seq, uint8_t* bq, uint8_t* cs,
                  uint8_t* tp) {
    const Contig& C = ((Bam*)h)->contigs[(size_t)i];
    const size_t n = C.tstart.size();
    auto cp = [](void* d, const void* s, size_t b) { if (b) memcpy(d, s, b); };
    cp(tstart, C.tstart.data(), n * 4); cp(tend, C.tend.data(), n * 4); cp(qstart, C.qstart.data(), n * 4);
    cp(qlen, C.qlen.data(), n * 4); cp(mapq, C.mapq.data(), n); cp(flag, C.flag.data(), n * 2);
    cp(qid, C.qid.data(), n * 4); cp(qoff, C.qoff.data(), n * 8); cp(cs_off, C.cs_off.data(), (n + 1) * 8);
    if (seq) cp(seq, C.seq.data(), C.seq.size());
    if (bq) cp(bq, C.bq.data(), C.bq.size());
    if (cs) cp(cs, C.cs.data(), C.cs.size());
    cp(tp, C.tp.data(), n);
}

// the three big arrays of a contig in place (valid until bam_free): 0 seq, 1 bq, 2 cs
const uint8_t* bam_ref_bytes(void* h, int64_t i, int which) {
    const Contig& C = ((Bam*)h)->contigs[(size_t)i];
    return which == 0 ? C.seq.data() : which == 1 ? C.bq.data() : C.cs.data();
}

void bam_free(void* h) { delete (Bam*)h; }

// ---- VCF body text from the device's integer records ---------------------------------------
// What caller.records_to_tuples + vcflib._body_line print (reference: bamlib.py:181-219, caller.py:174-192,
// vcflib.py:820-1021), in C: the divisions are the same IEEE doubles and "%.1f" / "%.0f" / "%.2f" round
// like python's format (both print the correctly rounded decimal).  One 64-byte himut_record per entry
// (include/himut_hip.h).  sm_file = 1 writes only the rows of the single_molecule_mutations file.
// Returns the number of bytes written, or -1 if `cap` is too small.
struct VcfRec {
    int32_t tpos, chunk, phase_set, gq;
    uint8_t ref, alt, gt0, gt1, status, gt_state, flags, pad;
    uint32_t counts[6], bqsum[4];
};

static bool vcf_format_slice(const void* records, int64_t k0, int64_t k1, const char* chrom, int phased, int sm_file, std::string& out) {
    static const char* STATUS[] = {"PASS", "LowBQ", "LowGQ", "IndelSite", "HetSite", "HetAltSite", "HomAltSite", "ComSnp",
                                   "PanelOfNormal", "LowDepth", "HighDepth", "Unphased"};
    auto idx = [](int ch) { return ch == 'A' ? 0 : ch == 'T' ? 1 : ch == 'G' ? 2 : 3; };
    const VcfRec* R = (const VcfRec*)records;
    for (int64_t k = k0; k < k1; k++) {
        const VcfRec& r = R[k];
        const uint32_t* c = r.counts;
        const double depth = (double)(c[0] + c[1] + c[2] + c[3] + c[5]);
        const double ref_count = (double)c[idx(r.ref)];
        const bool hetalt = r.status == 5;
        char ps[16];
        if (r.phase_set >= 0) snprintf(ps, sizeof(ps), "%d", r.phase_set); else snprintf(ps, sizeof(ps), ".");
        char line[512];
        int m;
        if (hetalt) {
            const int pi = idx(r.gt0), qi = idx(r.gt1);
            const double pc = (double)c[pi], qc = (double)c[qi];
            if ((int64_t)ref_count != 1 && sm_file) continue;
            const char* fmt = (phased && sm_file) ? "GT:GQ:BQ:DP:AD:VAF:PS" : "GT:GQ:BQ:DP:AD:VAF";
            m = snprintf(line, sizeof(line), "%s\t%d\t.\t%c\t%c,%c\t.\t%s\t.\t%s\t./.:%d:%.1f,%.1f:%.0f:%.0f,%.0f,%.0f:%.2f,%.2f",
                         chrom, r.tpos, r.ref, r.gt0, r.gt1, STATUS[r.status], fmt, r.gq, (double)r.bqsum[pi] / pc,
                         (double)r.bqsum[qi] / qc, depth, ref_count, pc, qc, pc / depth, qc / depth);
        } else {
            const int ai = idx(r.alt);
            const double alt_count = (double)c[ai];
            if ((int64_t)alt_count != 1 && sm_file) continue;
            const double alt_bq = alt_count != 0 ? (double)r.bqsum[ai] / alt_count : 0.0;
            const char* fmt = phased ? "GT:GQ:BQ:DP:AD:VAF:PS" : "GT:GQ:BQ:DP:AD:VAF";
            m = snprintf(line, sizeof(line), "%s\t%d\t.\t%c\t%c\t.\t%s\t.\t%s\t./.:%d:%.1f:%.0f:%.0f,%.0f:%.2f", chrom, r.tpos,
                         r.ref, r.alt, r.status < 12 ? STATUS[r.status] : "?", fmt, r.gq, alt_bq, depth, ref_count, alt_count,
                         alt_count / depth);
        }
        if (m < 0 || m >= (int)sizeof(line) - 24) return false;
        out.append(line, (size_t)m);
        if (phased) { out.push_back(':'); out.append(ps); }
        out.push_back('\n');
    }
    return true;
}

// VCF body lines of n records into out (cap bytes); returns the length, or -1 when cap is too small.  Large inputs
// are formatted by a few threads, a slice of the records each, and the slices are laid end to end.
int64_t vcf_format_records(const void* records, int64_t n, const char* chrom, int phased, int sm_file, char* out, int64_t cap) {
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)32, n / 8000, (int64_t)std::thread::hardware_concurrency()}));
    std::vector<std::string> part((size_t)nt);
    std::vector<char> ok((size_t)nt, 1);
    auto work = [&](int t) {
        part[(size_t)t].reserve((size_t)((n / nt + 1) * 64));
        ok[(size_t)t] = vcf_format_slice(records, n * t / nt, n * (t + 1) / nt, chrom, phased, sm_file, part[(size_t)t]) ? 1 : 0;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    int64_t w = 0;
    for (int t = 0; t < nt; t++) {
        if (!ok[(size_t)t] || w + (int64_t)part[(size_t)t].size() > cap) return -1;
        memcpy(out + w, part[(size_t)t].data(), part[(size_t)t].size());
        w += (int64_t)part[(size_t)t].size();
    }
    return w;
}

// ---- writer: one call per file; contigs given as parallel arrays of batches -------------
struct BamWriteContig {
    const char* name;
    int64_t length;
    int64_t n;
    const int32_t *tstart, *qstart, *qlen;
    const uint8_t* mapq;
    const uint16_t* flag;
    const int32_t* qid;
    const int64_t *qoff, *cs_off;
    const uint8_t *seq, *bq, *cs, *tp;
};

int bam_write(const char* path, const char* sample, const BamWriteContig* contigs, int64_t n_contigs) {
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    BgzfWriter w{f, {}};
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int64_t i = 0; i < n_contigs; i++)
        text += "@SQ\tSN:" + std::string(contigs[i].name) + "\tLN:" + std::to_string(contigs[i].length) + "\n";
    text += "@RG\tID:1\tSM:" + std::string(sample) + "\n";
    std::vector<uint8_t> hdr;
    hdr.insert(hdr.end(), {'B', 'A', 'M', 1});
    w32(hdr, (uint32_t)text.size());
    hdr.insert(hdr.end(), text.begin(), text.end());
    w32(hdr, (uint32_t)n_contigs);
    for (int64_t i = 0; i < n_contigs; i++) {
        const std::string nm = contigs[i].name;
        w32(hdr, (uint32_t)nm.size() + 1);
        hdr.insert(hdr.end(), nm.begin(), nm.end());
        hdr.push_back(0);
        w32(hdr, (uint32_t)contigs[i].length);
    }
    w.write(hdr.data(), hdr.size());
    std::vector<uint8_t> rec;
    std::vector<uint32_t> cigar;
    // index (.bai, SAM spec section 5.2): per contig the bins with their chunks and the 16-kb linear index
    struct RefIndex { std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lin; uint64_t beg = 0, end = 0, n = 0; };
    std::vector<RefIndex> index((size_t)n_contigs);
    auto reg2bin = [](int64_t beg, int64_t end) -> uint32_t {
        --end;
        if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
        if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
        if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
        if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
        if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
        return 0;
    };
    for (int64_t ci = 0; ci < n_contigs; ci++) {
        const BamWriteContig& C = contigs[ci];
        for (int64_t r = 0; r < C.n; r++) {
            // CIGAR from the cs tag: S (clip) then M / I / D
            cigar.clear();
            const uint8_t* cs = C.cs + C.cs_off[r];
            const int64_t cn = C.cs_off[r + 1] - C.cs_off[r];
            int64_t qcons = C.qstart[r];
            if (C.qstart[r] > 0) cigar.push_back(((uint32_t)C.qstart[r] << 4) | 4);
            auto push = [&](uint32_t op, uint32_t ln) {
                if (!ln) return;
                if (!cigar.empty() && (cigar.back() & 15) == op) cigar.back() += ln << 4;
                else cigar.push_back((ln << 4) | op);
            };
            for (int64_t i = 0; i < cn;) {
                const char c = (char)cs[i];
                int64_t j = i + 1;
                if (c == ':') { uint32_t v = 0; while (j < cn && cs[j] >= '0' && cs[j] <= '9') v = v * 10 + (cs[j++] - '0'); push(0, v); qcons += v; }
                else if (c == '*') { j = i + 3; push(0, 1); qcons += 1; }
                else { while (j < cn && ((cs[j] | 32) >= 'a' && (cs[j] | 32) <= 'z')) j++; const uint32_t ln = (uint32_t)(j - i - 1);
                       if (c == '=') { push(0, ln); qcons += ln; } else if (c == '+') { push(1, ln); qcons += ln; } else push(2, ln); }
                i = j;
            }
            if (C.qlen[r] > qcons) cigar.push_back(((uint32_t)(C.qlen[r] - qcons) << 4) | 4);
            const std::string qname = "ccs/" + std::to_string((long long)C.qid[r]);
            const uint32_t l_seq = (uint32_t)C.qlen[r];
            rec.clear();
            w32(rec, 0);  // block_size, patched below
            w32(rec, (uint32_t)ci);
            w32(rec, (uint32_t)C.tstart[r]);
            rec.push_back((uint8_t)(qname.size() + 1));
            rec.push_back(C.mapq[r]);
            rec.push_back(0x48); rec.push_back(0x12);  // bin (unused by this reader)
            rec.push_back((uint8_t)(cigar.size() & 255)); rec.push_back((uint8_t)(cigar.size() >> 8));
            rec.push_back((uint8_t)(C.flag[r] & 255)); rec.push_back((uint8_t)(C.flag[r] >> 8));
            w32(rec, l_seq);
            w32(rec, 0xffffffffu); w32(rec, 0xffffffffu); w32(rec, 0);
            rec.insert(rec.end(), qname.begin(), qname.end());
            rec.push_back(0);
            for (uint32_t c : cigar) w32(rec, c);
            const uint8_t* sq = C.seq + C.qoff[r] / 2;
            rec.insert(rec.end(), sq, sq + (l_seq + 1) / 2);
            const uint8_t* bq = C.bq + C.qoff[r];
            rec.insert(rec.end(), bq, bq + l_seq);
            rec.insert(rec.end(), {'c', 's', 'Z'});
            rec.insert(rec.end(), cs, cs + cn);
            rec.push_back(0);
            if (C.tp[r]) { rec.insert(rec.end(), {'t', 'p', 'A'}); rec.push_back(C.tp[r]); }
            const uint32_t bs = (uint32_t)rec.size() - 4;
            for (int k = 0; k < 4; k++) rec[k] = (uint8_t)(bs >> (8 * k));
            int64_t ref_len = 0;
            for (uint32_t c : cigar) if ((c & 15) == 0 || (c & 15) == 2) ref_len += c >> 4;
            const uint64_t v0 = w.voffset();
            w.write(rec.data(), rec.size());
            const uint64_t v1 = w.voffset();
            RefIndex& X = index[(size_t)ci];
            const int64_t beg = C.tstart[r], end = beg + (ref_len > 0 ? ref_len : 1);
            auto& ch = X.bins[reg2bin(beg, end)];
            if (!ch.empty() && ch.back().second == v0) ch.back().second = v1; else ch.emplace_back(v0, v1);
            for (int64_t wdw = beg >> 14; wdw <= (end - 1) >> 14; wdw++) {
                if ((size_t)wdw >= X.lin.size()) X.lin.resize((size_t)wdw + 1, 0);
                if (!X.lin[(size_t)wdw]) X.lin[(size_t)wdw] = v0;
            }
            if (!X.n) X.beg = v0;
            X.end = v1; X.n++;
        }
    }
    w.finish();
    bool ok = w.ok;
    fclose(f);
    if (ok) {
        std::vector<uint8_t> bai = {'B', 'A', 'I', 1};
        auto w64 = [&](uint64_t x) { for (int k = 0; k < 8; k++) bai.push_back((uint8_t)(x >> (8 * k))); };
        w32(bai, (uint32_t)n_contigs);
        for (auto& X : index) {
            w32(bai, (uint32_t)X.bins.size() + (X.n ? 1u : 0u));
            for (auto& kv : X.bins) {
                w32(bai, kv.first); w32(bai, (uint32_t)kv.second.size());
                for (auto& c : kv.second) { w64(c.first); w64(c.second); }
            }
            if (X.n) { w32(bai, 37450u); w32(bai, 2u); w64(X.beg); w64(X.end); w64(X.n); w64(0); }   // samtools' metadata pseudo-bin
            for (size_t k = 1; k < X.lin.size(); k++) if (!X.lin[k]) X.lin[k] = X.lin[k - 1];
            w32(bai, (uint32_t)X.lin.size());
            for (uint64_t v : X.lin) w64(v);
        }
        FILE* g = fopen((std::string(path) + ".bai").c_str(), "wb");
        ok = g && fwrite(bai.data(), 1, bai.size(), g) == bai.size();
        if (g) fclose(g);
    }
    return ok ? 0 : 2;
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming ingest of ONE contig for the device-side record parser (libhimut_hip.so: himut_ingest_*): the host's part
// is what has to be sequential or is cheap -- block inflate (thread pool) straight into a caller's (pinned) buffer,
// the hop from length field to length field, the read-name table -- and the device does the rest (CIGAR walk, tag
// scan, placement, byte copies).  With an index beside the file (x.bam.bai) only the contig's own BGZF blocks are
// inflated; without one the blocks in front of it are inflated and hopped over.
struct BamStream {
    Bgzf z;
    Bam hdr;
    size_t first_block = 0, first_skip = 0;            // where the first record of the file starts
    bool have_bai = false;
    std::vector<std::pair<uint64_t, uint64_t>> ref_range;   // per contig: virtual offsets of its first record / end of its last
    int32_t target = -1;
    size_t blk = 0, blk_end = 0, skip = 0;
    std::vector<uint8_t> carry;
    std::unordered_map<std::string, int32_t> names;
    int64_t nkept = 0;
    bool done = false, unique = true;
    std::string err;
    double t_inflate = 0, t_hop = 0;      // HIMUT_INGEST_PROFILE
    int64_t n_windows = 0, inflated_bytes = 0;   // inflated_bytes: what this stream has inflated since it was opened (header blocks excluded)
    std::future<std::string> inflating;   // the window being inflated in the background (bam_stream_prefetch)
    std::vector<size_t> inf_off;
    size_t inf_tot = 0;
    uint8_t* inf_buf = nullptr;
    std::vector<uint32_t> pump_off[2];    // bam_stream_pump's record lists: handed to the device asynchronously, so they
    std::vector<int32_t> pump_qid[2];     // live as long as the stream, not as long as the call
    bool ready = false;                   // a window is inflated and waits for its hop (bam_stream_wait)
    size_t ready_tot = 0;
    uint8_t* ready_buf = nullptr;
};
constexpr size_t BAM_STREAM_HEAD = (size_t)4 << 20;

// File offsets at which the index says BGZF blocks start (chunk begins and linear-index entries), sorted.  Only hints
// for the parallel block scan: a wrong or stale index costs nothing but the serial scan.
static std::vector<size_t> bai_block_hints(const std::string& path) {
    std::vector<size_t> out;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return out;
    std::vector<uint8_t> d;
    uint8_t tmp[1 << 16];
    size_t k;
    while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) d.insert(d.end(), tmp, tmp + k);
    fclose(f);
    size_t p = 0;
    auto need = [&](uint64_t n) { return p + n <= d.size(); };
    auto r32 = [&]() { const uint32_t v = le32(&d[p]); p += 4; return v; };
    auto r64 = [&]() { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)d[p + i] << (8 * i); p += 8; return v; };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0) return out;
    p = 4;
    const uint32_t n = r32();
    for (uint32_t i = 0; i < n; i++) {
        if (!need(4)) return out;
        const uint32_t nbin = r32();
        for (uint32_t b = 0; b < nbin; b++) {
            if (!need(8)) return out;
            const uint32_t bin = r32(), nch = r32();
            if (!need(16ull * nch)) return out;
            for (uint32_t c = 0; c < nch; c++) {
                const uint64_t beg = r64();
                (void)r64();
                if (bin != 37450) out.push_back((size_t)(beg >> 16));
            }
        }
        if (!need(4)) return out;
        const uint32_t nint = r32();
        if (!need(8ull * nint)) return out;
        for (uint32_t c = 0; c < nint; c++) { const uint64_t v = r64(); if (v) out.push_back((size_t)(v >> 16)); }
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

static bool read_bai(const std::string& path, size_t n_ref, std::vector<std::pair<uint64_t, uint64_t>>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> d;
    uint8_t tmp[1 << 16];
    size_t k;
    while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) d.insert(d.end(), tmp, tmp + k);
    fclose(f);
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= d.size(); };
    auto r32 = [&]() { const uint32_t v = le32(&d[p]); p += 4; return v; };
    auto r64 = [&]() { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)d[p + i] << (8 * i); p += 8; return v; };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0) return false;
    p = 4;
    const uint32_t n = r32();
    if (n != n_ref) return false;
    out.assign(n, {~0ull, 0ull});
    for (uint32_t i = 0; i < n; i++) {
        if (!need(4)) return false;
        const uint32_t nbin = r32();
        for (uint32_t b = 0; b < nbin; b++) {
            if (!need(8)) return false;
            const uint32_t bin = r32(), nch = r32();
            if (!need(16ull * nch)) return false;
            for (uint32_t c = 0; c < nch; c++) {
                const uint64_t beg = r64(), end = r64();
                if (bin == 37450) continue;
                out[i].first = std::min(out[i].first, beg);
                out[i].second = std::max(out[i].second, end);
            }
        }
        if (!need(4)) return false;
        const uint32_t nint = r32();
        if (!need(8ull * nint)) return false;
        p += 8ull * nint;
    }
    return true;
}

void* bam_stream_open(const char* path, int threads) {
    BamStream* S = new BamStream();
    try {
        if (threads <= 0) {
            const char* e = getenv("HIMUT_INGEST_THREADS");
            threads = e ? atoi(e) : 0;
            if (threads <= 0) threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
        }
        Bgzf& z = S->z;
        std::vector<size_t> hints;
        if (!getenv("HIMUT_INGEST_NO_INDEX")) hints = bai_block_hints(std::string(path) + ".bai");
        if (!z.open(path, threads, &hints)) { S->err = z.err; return S; }
        // the header sits in the first block or two: they are inflated one at a time, not a window at a time
        size_t consumed = 0, hb_next = 0;
        std::vector<uint8_t> hbuf;
        auto rd = [&](void* dst, size_t n) {
            while (hbuf.size() < consumed + n) {
                if (hb_next >= z.blocks.size()) return false;
                const size_t at = hbuf.size();
                hbuf.resize(at + z.blocks[hb_next].isize);
                std::vector<size_t> off(1, 0);
                if (!z.inflate_blocks(hb_next, hb_next + 1, hbuf.data() + at, 0, &off).empty()) return false;
                hb_next++;
            }
            memcpy(dst, hbuf.data() + consumed, n);
            consumed += n;
            return true;
        };
        uint8_t b4[4];
        if (!rd(b4, 4) || memcmp(b4, "BAM\1", 4) != 0) { S->err = z.err.empty() ? "not a BAM file" : z.err; return S; }
        size_t inflated_total = 0;
        for (const BlockRef& b : z.blocks) inflated_total += b.isize;
        if (!rd(b4, 4)) { S->err = "truncated BAM header"; return S; }
        const uint32_t l_text = le32(b4);
        if ((size_t)l_text > inflated_total) { S->err = "BAM header text longer than the file"; return S; }
        S->hdr.header_text.resize(l_text);
        if (l_text && !rd(&S->hdr.header_text[0], l_text)) { S->err = "truncated BAM header"; return S; }
        while (!S->hdr.header_text.empty() && S->hdr.header_text.back() == '\0') S->hdr.header_text.pop_back();
        if (!rd(b4, 4)) { S->err = "truncated BAM header"; return S; }
        const uint32_t n_ref = le32(b4);
        if ((size_t)n_ref * 8 > inflated_total) { S->err = "BAM header lists more contigs than the file can hold"; return S; }
        S->hdr.contigs.resize(n_ref);
        for (uint32_t i = 0; i < n_ref; i++) {
            if (!rd(b4, 4)) { S->err = "truncated BAM header"; return S; }
            const uint32_t l_name = le32(b4);
            if ((size_t)l_name > inflated_total) { S->err = "contig name longer than the file"; return S; }
            std::string nm(l_name, '\0');
            if (l_name && !rd(&nm[0], l_name)) { S->err = "truncated BAM header"; return S; }
            while (!nm.empty() && nm.back() == '\0') nm.pop_back();
            if (!rd(b4, 4)) { S->err = "truncated BAM header"; return S; }
            S->hdr.contigs[i].name = nm;
            S->hdr.contigs[i].length = le32(b4);
        }
        size_t acc = 0;
        for (size_t k = 0; k < z.blocks.size(); k++) {
            if (consumed < acc + z.blocks[k].isize) { S->first_block = k; S->first_skip = consumed - acc; break; }
            acc += z.blocks[k].isize;
            S->first_block = k + 1; S->first_skip = 0;
        }
        if (!getenv("HIMUT_INGEST_NO_INDEX"))
            S->have_bai = read_bai(std::string(path) + ".bai", n_ref, S->ref_range);
    } catch (const std::exception& e) { S->err = std::string("BAM header: ") + e.what(); }
    return S;
}

const char* bam_stream_error(void* h) { return ((BamStream*)h)->err.c_str(); }
const char* bam_stream_header_text(void* h) { return ((BamStream*)h)->hdr.header_text.c_str(); }
int64_t bam_stream_n_ref(void* h) { return (int64_t)((BamStream*)h)->hdr.contigs.size(); }
const char* bam_stream_ref_name(void* h, int64_t i) { return ((BamStream*)h)->hdr.contigs[(size_t)i].name.c_str(); }
int64_t bam_stream_ref_len(void* h, int64_t i) { return ((BamStream*)h)->hdr.contigs[(size_t)i].length; }
int bam_stream_indexed(void* h) { return ((BamStream*)h)->have_bai ? 1 : 0; }
int64_t bam_stream_scan_parts(void* h) { return (int64_t)((BamStream*)h)->z.scan_parts; }
int64_t bam_stream_inflated_bytes(void* h) { return ((BamStream*)h)->inflated_bytes; }
int bam_stream_unique_names(void* h) { return ((BamStream*)h)->unique ? 1 : 0; }
void bam_stream_close(void* h) {
    BamStream* S = (BamStream*)h;
    if (S->inflating.valid()) (void)S->inflating.get();
    if (getenv("HIMUT_INGEST_PROFILE"))
        fprintf(stderr, "stream profile (s): inflate %.3f in %lld windows, hop + names %.3f (threads %d)\n", S->t_inflate,
                (long long)S->n_windows, S->t_hop, S->z.threads);
    S->z.close();
    delete S;
}

// Restricts the stream to one contig.  *inflated_bound = inflated bytes of the blocks that will be read (an upper
// bound of the contig's record bytes when the file is indexed, of everything from the first record on otherwise).
int bam_stream_select(void* h, int32_t ref_id, int64_t* inflated_bound) {
    BamStream* S = (BamStream*)h;
    if (ref_id < 0 || (size_t)ref_id >= S->hdr.contigs.size()) { S->err = "no such contig"; return 1; }
    if (S->inflating.valid()) (void)S->inflating.get();
    S->ready = false;
    S->target = ref_id; S->carry.clear(); S->names.clear(); S->nkept = 0; S->done = false; S->unique = true;
    const auto& B = S->z.blocks;
    S->blk = S->first_block; S->skip = S->first_skip; S->blk_end = B.size();
    if (S->have_bai) {
        const auto rg = S->ref_range[(size_t)ref_id];
        if (rg.first == ~0ull) { S->blk = S->blk_end = 0; S->done = true; }       // no records on this contig
        else {
            auto find = [&](uint64_t foff) { size_t lo = 0, hi = B.size(); while (lo < hi) { const size_t m = (lo + hi) / 2; if (B[m].file_off < foff) lo = m + 1; else hi = m; } return lo; };
            const size_t b0 = find(rg.first >> 16), b1 = find(rg.second >> 16);
            if (b0 >= B.size() || B[b0].file_off != (rg.first >> 16)) { S->err = "index does not match the BAM file"; return 1; }
            S->blk = b0; S->skip = (size_t)(rg.first & 0xffff);
            S->blk_end = std::min(B.size(), b1 + 1);
        }
    }
    int64_t tot = 0;
    for (size_t k = S->blk; k < S->blk_end; k++) tot += B[k].isize;
    if (inflated_bound) *inflated_bound = tot;
    return 0;
}

// A window buffer = BAM_STREAM_HEAD bytes of head room + the inflated blocks.  The inflate of window k + 1 (thread pool,
// started by bam_stream_prefetch, running in the background) overlaps the hop over window k and its hand-over to the
// GPU; the partial record that window k ends with is then put in FRONT of window k + 1's bytes, in the head room, so
// the inflate never has to wait for it.
int64_t bam_stream_head(void) { return (int64_t)BAM_STREAM_HEAD; }

// Starts inflating the next blocks of the contig into buf + HEAD (as many as fit cap - HEAD).  Returns 1 when an
// inflate is in flight, 0 when the contig has no more blocks, -2 on error.  One prefetch may be in flight.
int bam_stream_prefetch(void* h, uint8_t* buf, int64_t cap) {
    BamStream* S = (BamStream*)h;
    try {
        if (S->inflating.valid()) { S->err = "a prefetch is in flight already"; return -2; }
        if (S->done || S->blk >= S->blk_end) return 0;
        const auto& B = S->z.blocks;
        S->inf_off.clear();
        size_t tot = 0, b0 = S->blk, b1 = S->blk;
        while (b1 < S->blk_end && (int64_t)(BAM_STREAM_HEAD + tot + B[b1].isize) <= cap) { S->inf_off.push_back(tot); tot += B[b1].isize; b1++; }
        if (b1 == b0) { S->err = "ingest window smaller than a BGZF block"; return -2; }
        S->blk = b1;
        S->inf_tot = tot; S->inf_buf = buf; S->inflated_bytes += (int64_t)tot;
        uint8_t* dst = buf + BAM_STREAM_HEAD;
        S->inflating = std::async(std::launch::async, [S, b0, b1, dst]() {
            const double t0 = now_s();
            std::string e = S->z.inflate_blocks(b0, b1, dst, 0, &S->inf_off);
            S->t_inflate += now_s() - t0; S->n_windows++;
            return e;
        });
        return 1;
    } catch (const std::exception& e) { S->err = std::string("BAM stream: ") + e.what(); return -2; }
}

// Waits for the inflate in flight.  Returns 1 when a window is now ready for bam_stream_next, 0 when none was in flight,
// -2 on error.  After it the next prefetch may be started BEFORE the hop over this window (bam_stream_next), so the
// pool never idles while one thread walks the records.
int bam_stream_wait(void* h) {
    BamStream* S = (BamStream*)h;
    try {
        if (S->ready) return 1;
        if (!S->inflating.valid()) return 0;
        const std::string e = S->inflating.get();
        if (!e.empty()) { S->err = e; return -2; }
        S->ready = true; S->ready_tot = S->inf_tot; S->ready_buf = S->inf_buf;
        return 1;
    } catch (const std::exception& e) { S->err = std::string("BAM stream: ") + e.what(); return -2; }
}

// Finishes the window whose inflate bam_stream_prefetch(buf) started (or, with none in flight, inflates one now): hops
// over the records and lists the kept ones (this contig, mapped): rec_off[k] = offset of record k's body (behind its
// length field) from buf + *start, qid[k] = index of the first kept record with the same read name.  The records occupy
// *nbytes bytes from buf + *start.  sums: of the kept records, query lengths rounded up to 32 and bytes of the auxiliary
// fields.  Returns the number of kept records (0: a window of other contigs' records, go on), -1 at the end of the
// contig, -2 on error.
int64_t bam_stream_next(void* h, uint8_t* buf, int64_t cap, uint32_t* rec_off, int32_t* qid, int64_t rec_cap, int64_t* start,
                        int64_t* nbytes, int64_t* sums) {
    BamStream* S = (BamStream*)h;
    *nbytes = 0; *start = 0;
    sums[0] = sums[1] = 0;
    try {
        if (!S->ready && !S->inflating.valid()) {
            if (S->done) return -1;
            const int r = bam_stream_prefetch(h, buf, cap);
            if (r < 0) return -2;
            if (r == 0 && S->carry.empty()) return -1;
        }
        if (!S->ready && bam_stream_wait(h) < 0) return -2;
        size_t tot = 0;
        if (S->ready) {
            S->ready = false;
            if (S->ready_buf != buf) { S->err = "bam_stream_next on a buffer other than the prefetched one"; return -2; }
            tot = S->ready_tot;
        }
        const double t_h0 = now_s();
        const size_t c = S->carry.size();
        if (c > BAM_STREAM_HEAD) { S->err = "a BAM record is larger than the head room of the ingest window"; return -2; }
        uint8_t* dst = buf + BAM_STREAM_HEAD - c;
        if (c) memcpy(dst, S->carry.data(), c);
        S->carry.clear();
        const size_t nb = c + tot;
        size_t pos = S->skip;
        S->skip = 0;
        const size_t first = pos;
        int64_t n = 0;
        while (pos + 4 <= nb && n < rec_cap) {
            const uint32_t bs = le32(dst + pos);
            if (bs < 32) { S->err = "BAM record too short"; return -2; }
            if (pos + 4 + (size_t)bs > nb) break;
            const uint8_t* rec = dst + pos + 4;
            const int32_t ref_id = (int32_t)le32(rec);
            const uint16_t flag = le16(rec + 14);
            if (ref_id > S->target || ref_id < 0) { S->done = true; break; }       // coordinate sorted: the contig is over
            if (ref_id == S->target && !(flag & 4)) {
                const size_t l_qname = rec[8];
                if (32 + l_qname > bs) { S->err = "malformed BAM record"; return -2; }
                const char* qn = (const char*)rec + 32;
                auto it = S->names.emplace(std::string(qn, strnlen(qn, l_qname)), (int32_t)S->nkept);
                if (!it.second) S->unique = false;
                const uint64_t n_cigar = le16(rec + 12), l_seq = le32(rec + 16);
                const uint64_t fixed = 32 + l_qname + 4 * n_cigar + (l_seq + 1) / 2 + l_seq;
                if (fixed > bs) { S->err = "malformed BAM record"; return -2; }
                sums[0] += (int64_t)((l_seq + 31) & ~(uint64_t)31);
                sums[1] += (int64_t)(bs - fixed);
                rec_off[n] = (uint32_t)(pos + 4 - first);
                qid[n] = it.first->second;
                S->nkept++; n++;
            }
            pos += 4 + (size_t)bs;
        }
        *start = (int64_t)(BAM_STREAM_HEAD - c + first);
        *nbytes = (int64_t)(pos - first);
        S->t_hop += now_s() - t_h0;
        // nothing more comes once every block is taken AND no window is being inflated or waits for its hop (the next
        // prefetch may have been started before this hop)
        const bool last = S->blk >= S->blk_end && !S->inflating.valid() && !S->ready;
        if (!S->done) {
            if (pos < nb) S->carry.assign(dst + pos, dst + nb);
            if (last && !S->carry.empty() && n < rec_cap) {
                // the last block ended inside a record: an indexed range ends with the contig's last record, so what is
                // left belongs to the next contig; without an index the file is truncated
                if (!S->have_bai && S->carry.size() >= 4) { S->err = "truncated BAM record"; return -2; }
                S->carry.clear();
            }
        }
        if (n > 0) return n;
        if (S->done || (last && S->carry.empty())) return -1;
        return 0;
    } catch (const std::exception& e) { S->err = std::string("BAM stream: ") + e.what(); return -2; }
}

// The whole loop of one contig's ingest in one call (no interpreter between the steps: a Python thread that parses the
// side VCFs meanwhile would otherwise hold the GIL against every one of them).  ``wait_fn`` / ``window_fn`` are
// libhimut_hip.so's himut_ingest_wait / himut_ingest_window, ``bufs`` its two pinned windows of ``cap`` bytes.
// Returns 0, -2 on a stream error (bam_stream_error), or the positive error code of the device library.
typedef int (*ingest_wait_fn)(void*, int);
typedef int (*ingest_window_fn)(void*, int, int64_t, int64_t, const uint32_t*, const int32_t*, int64_t, int64_t, int64_t);
int bam_stream_pump(void* h, void* ctx, void* wait_fn, void* window_fn, uint8_t* buf0, uint8_t* buf1, int64_t cap, int64_t rec_cap) {
    BamStream* S = (BamStream*)h;
    try {
        ingest_wait_fn wait = (ingest_wait_fn)wait_fn;
        ingest_window_fn window = (ingest_window_fn)window_fn;
        uint8_t* bufs[2] = {buf0, buf1};
        std::vector<uint32_t>* rec_off = S->pump_off;
        std::vector<int32_t>* qid = S->pump_qid;
        for (int k = 0; k < 2; k++) { rec_off[k].resize((size_t)rec_cap); qid[k].resize((size_t)rec_cap); }
        int slot = 0;
        const bool prof = getenv("HIMUT_INGEST_PROFILE") != nullptr;
        double t_wait_inf = 0, t_wait_dev = 0, t_window = 0, t_first_window = 0, t_hop0 = S->t_hop;
        int64_t nwin = 0;
        if (bam_stream_prefetch(h, bufs[0], cap) < 0) return -2;
        for (;;) {
            // window `slot` is inflated: the pool goes on with the next one (into the other buffer, once its bytes of two
            // windows ago have left the host) while this thread hops over the records and hands the window to the GPU
            double t0 = prof ? now_s() : 0;
            if (bam_stream_wait(h) < 0) return -2;
            double t1 = prof ? now_s() : 0;
            int rc = wait(ctx, slot ^ 1);
            if (rc) return rc;
            if (prof) { t_wait_inf += t1 - t0; t_wait_dev += now_s() - t1; }
            if (bam_stream_prefetch(h, bufs[slot ^ 1], cap) < 0) return -2;
            int64_t start = 0, nbytes = 0, sums[2] = {0, 0};
            const int64_t n = bam_stream_next(h, bufs[slot], cap, rec_off[slot].data(), qid[slot].data(), rec_cap, &start, &nbytes, sums);
            if (n == -1) break;
            if (n < 0) return -2;
            t0 = prof ? now_s() : 0;
            if (n > 0 && (rc = window(ctx, slot, start, nbytes, rec_off[slot].data(), qid[slot].data(), n, sums[0], sums[1])) != 0) return rc;
            if (prof) { const double dt = now_s() - t0; t_window += dt; if (!nwin) t_first_window = dt; nwin++; }
            slot ^= 1;
        }
        if (prof)
            fprintf(stderr, "pump (s): waiting for inflate %.3f, for the device %.3f, hop %.3f, handing over %.3f (first window %.3f) in %lld windows\n",
                    t_wait_inf, t_wait_dev, S->t_hop - t_hop0, t_window, t_first_window, (long long)nwin);
        return 0;
    } catch (const std::exception& e) { S->err = std::string("BAM stream: ") + e.what(); return -2; }
}

}  // extern "C"
