// Synthetic CCS read-batch generator (test / bench infrastructure, own design).
//
// Produces the neutral read-batch layout consumed by himut_push_reads()
// (include/himut_hip.h) for a synthetic diploid sample, following the recipe
// in SURVEY.md §8(d): iid ACGT contig, germline SNPs, sorted CCS reads copied
// from one haplotype with substitution / insertion / deletion errors, cs:short
// tags, a two-level base-quality model and a few "bad" read classes so the
// read filters of the caller fire.  Deterministic for a given parameter block:
// every read draws from its own counter-based RNG stream, so generation is
// parallel over reads and independent of the thread count.
//
// Nothing in here is derived from the reference implementation; it only has to
// emit inputs the reference would accept (no query N, no BQ 0).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

struct SynthParams {
    uint64_t seed;
    int32_t contig_len;
    int32_t read_len_min;
    int32_t read_len_max;
    int32_t softclip_max;
    int32_t cs_long;   // 1: emit '=' long-form matches
    int32_t threads;
    double depth;
    double read_len_mean;
    double read_len_sd;
    double snp_rate;   // germline SNPs per bp
    double het_frac;   // fraction of SNPs that are heterozygous
    double hetalt_frac; // fraction of SNPs where the two haplotypes carry different alts
    double sub_rate;   // per-base sequencing substitution rate
    double ins_rate;
    double del_rate;
    double som_rate;   // extra single-read substitutions per base
    double frac_noisy; // reads with noisy_mult x error rates
    double noisy_mult;
    double frac_lowmapq;
    double frac_lowbq;
    double bq93_prob;
    double frac_softclip;
    double pile_frac;  // fraction of the contig covered by pile-up regions
    double pile_mult;  // coverage multiplier inside them
};

}  // extern "C"

namespace {

struct Rng {
    uint64_t s;
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t a, uint64_t b) {
        s = mix(seed + 0x9e3779b97f4a7c15ULL * (a + 1)) ^ mix(b * 0xd1342543de82ef95ULL + 0x632be59bd9b4e019ULL);
    }
    uint64_t next() {
        s += 0x9e3779b97f4a7c15ULL;
        return mix(s);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    int64_t geometric(double p) {
        if (p <= 0.0) return (int64_t)1 << 40;
        double u = uniform();
        if (u < 1e-300) u = 1e-300;
        return (int64_t)std::floor(std::log(u) / std::log1p(-p));
    }
};

const char kBase[4] = {'A', 'C', 'G', 'T'};
// BAM nibble codes for A C G T
const uint8_t kNib[4] = {1, 2, 4, 8};

struct ReadPlan {
    int32_t start;
    int32_t reflen;
    int32_t clipl, clipr;
    uint8_t hap, noisy, lowbq, mapq;
};

struct Synth {
    SynthParams p;
    std::vector<uint8_t> ref;       // 0..3
    std::vector<int32_t> snp_pos;   // sorted
    std::vector<uint8_t> snp_alt;   // 0..3
    std::vector<uint8_t> snp_alt2;  // second alt (gt 4 only)
    std::vector<uint8_t> snp_gt;    // 1: hap0 only, 2: hap1 only, 3: both, 4: hap0 alt / hap1 alt2
    std::vector<ReadPlan> plan;
    std::vector<int32_t> qlen, tend, cslen;
    std::vector<int64_t> qoff, csoff;
    int64_t total_bases_padded = 0;
    int64_t cs_total = 0;
};

struct Out {
    uint8_t* seq;  // packed nibbles for this read (may be null)
    uint8_t* cs;   // may be null
    int64_t q = 0;
    int64_t c = 0;
    void base(int b) {
        if (seq) {
            uint8_t n = kNib[b];
            if (q & 1) seq[q >> 1] |= n; else seq[q >> 1] = (uint8_t)(n << 4);
        }
        q++;
    }
    void ch(char x) {
        if (cs) cs[c] = (uint8_t)x;
        c++;
    }
    void num(int64_t v) {
        char tmp[24];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) ch(tmp[--n]);
    }
};

inline char lower(int b) { return (char)(kBase[b] | 0x20); }

// Generates read i; returns tend.  Writes are skipped where Out has nulls.
int32_t gen_read(const Synth& S, int64_t i, Out& o) {
    const SynthParams& P = S.p;
    const ReadPlan& pl = S.plan[(size_t)i];
    Rng rng(P.seed, 0x52454144ULL, (uint64_t)i);
    const double mult = pl.noisy ? P.noisy_mult : 1.0;
    const double psub = (P.sub_rate + P.som_rate) * mult;
    const double pins = P.ins_rate * mult, pdel = P.del_rate * mult;
    const double perr = psub + pins + pdel;
    const int32_t start = pl.start, end = pl.start + pl.reflen;
    const int hapbit = pl.hap ? 2 : 1;
    auto carried = [&](size_t k) { return S.snp_gt[k] >= 3 || (S.snp_gt[k] & hapbit); };

    for (int k = 0; k < pl.clipl; k++) o.base((int)rng.below(4));

    int64_t run = 0;
    int64_t run_q0 = o.q;
    auto flush = [&]() {
        if (run == 0) return;
        if (P.cs_long) {
            o.ch('=');
            // bases of the run equal the reference bases
            for (int64_t k = 0; k < run; k++) o.ch(kBase[S.ref[(size_t)(run_q0 + k)]]);
        } else {
            o.ch(':');
            o.num(run);
        }
        run = 0;
    };
    // For the long form we need the reference coordinates of the run, so track
    // them in run_q0 as a reference position instead of a query position.
    size_t j = (size_t)(std::lower_bound(S.snp_pos.begin(), S.snp_pos.end(), start) - S.snp_pos.begin());
    auto skip_snps = [&](int32_t t) {
        while (j < S.snp_pos.size() && (S.snp_pos[j] < t || !carried(j))) j++;
    };
    int32_t t = start;
    skip_snps(t);
    int64_t next_err = (int64_t)t + 1 + rng.geometric(perr);
    while (t < end) {
        int64_t te = end;
        if (next_err < te) te = next_err;
        if (j < S.snp_pos.size() && S.snp_pos[j] < te) te = S.snp_pos[j];
        if (te > t) {
            if (run == 0) run_q0 = t;
            for (int32_t x = t; x < (int32_t)te; x++) o.base(S.ref[(size_t)x]);
            run += te - t;
            t = (int32_t)te;
        }
        if (t >= end) break;
        const bool is_snp = (j < S.snp_pos.size() && S.snp_pos[j] == t);
        int hb = is_snp ? ((S.snp_gt[j] == 4 && pl.hap) ? S.snp_alt2[j] : S.snp_alt[j]) : S.ref[(size_t)t];
        int b = hb;
        if ((int64_t)t == next_err) {
            double u = rng.uniform() * perr;
            if (u < pins && t > start) {
                flush();
                int k = 1 + (int)rng.below(3);
                o.ch('+');
                for (int x = 0; x < k; x++) {
                    int ib = (int)rng.below(4);
                    o.base(ib);
                    o.ch(lower(ib));
                }
            } else if (u < pins + pdel && t > start) {
                int k = 1 + (int)rng.below(3);
                if (t + k < end) {
                    flush();
                    o.ch('-');
                    for (int x = 0; x < k; x++) o.ch(lower(S.ref[(size_t)(t + x)]));
                    t += k;
                    skip_snps(t);
                    next_err = (int64_t)t + 1 + rng.geometric(perr);
                    continue;
                }
            } else {
                b = (hb + 1 + (int)rng.below(3)) & 3;
            }
            next_err = (int64_t)t + 1 + rng.geometric(perr);
        }
        if (b == S.ref[(size_t)t]) {
            if (run == 0) run_q0 = t;
            run++;
            o.base(b);
        } else {
            flush();
            o.ch('*');
            o.ch(lower(S.ref[(size_t)t]));
            o.ch(lower(b));
            o.base(b);
        }
        t++;
        if (is_snp) { j++; skip_snps(t); }
    }
    flush();
    for (int k = 0; k < pl.clipr; k++) o.base((int)rng.below(4));
    return t;
}

void fill_bq(const Synth& S, int64_t i, uint8_t* bq, int32_t qlen) {
    const SynthParams& P = S.p;
    Rng rng(P.seed, 0x42515F5FULL, (uint64_t)i);
    const bool low = S.plan[(size_t)i].lowbq;
    const uint32_t thr = (uint32_t)(P.bq93_prob * 65536.0);
    int32_t k = 0;
    while (k < qlen) {
        uint64_t r = rng.next();
        uint64_t r2 = rng.next();
        for (int x = 0; x < 4 && k < qlen; x++, k++) {
            uint32_t a = (uint32_t)(r >> (16 * x)) & 0xffff;
            uint32_t v = (uint32_t)(r2 >> (16 * x)) & 0xffff;
            uint8_t q;
            if (low) q = (uint8_t)(1 + (v * 40u >> 16));
            else if (a < thr) q = 93;
            else q = (uint8_t)(1 + (v * 92u >> 16));
            bq[k] = q;
        }
    }
}

template <class F>
void parallel_for(int64_t n, int threads, F f) {
    if (threads < 1) threads = 1;
    if (threads == 1 || n < 64) { for (int64_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    int64_t chunk = (n + threads - 1) / threads;
    for (int w = 0; w < threads; w++) {
        int64_t a = w * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        th.emplace_back([=]() { for (int64_t i = a; i < b; i++) f(i); });
    }
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

void* synth_create(const SynthParams* pp) {
    Synth* S = new Synth();
    S->p = *pp;
    const SynthParams& P = S->p;
    const int32_t L = P.contig_len;
    S->ref.resize((size_t)L);
    {
        // reference bases: parallel in blocks of 64k, each its own stream
        const int64_t nblk = (L + 65535) / 65536;
        parallel_for(nblk, P.threads, [&](int64_t b) {
            Rng r(P.seed, 0x524546ULL, (uint64_t)b);
            int64_t a = b * 65536, e = std::min<int64_t>(L, a + 65536);
            for (int64_t x = a; x < e;) {
                uint64_t v = r.next();
                for (int k = 0; k < 32 && x < e; k++, x++) S->ref[(size_t)x] = (uint8_t)((v >> (2 * k)) & 3);
            }
        });
    }
    {
        Rng r(P.seed, 0x534e50ULL, 0);
        int64_t pos = r.geometric(P.snp_rate);
        while (pos < L) {
            S->snp_pos.push_back((int32_t)pos);
            int rb = S->ref[(size_t)pos];
            S->snp_alt.push_back((uint8_t)((rb + 1 + r.below(3)) & 3));
            double u = r.uniform();
            uint8_t gt = 3;
            if (u < P.het_frac) gt = (r.next() & 1) ? 1 : 2;
            else if (u < P.het_frac + P.hetalt_frac) gt = 4;
            uint8_t a1 = S->snp_alt.back();
            uint8_t a2 = (uint8_t)((a1 + 1 + r.below(3)) & 3);
            if (a2 == rb) a2 = (uint8_t)((a2 + 1) & 3);
            if (a2 == a1) a2 = (uint8_t)((a2 + 1) & 3);
            if (a2 == rb) a2 = (uint8_t)((a2 + 1) & 3);
            S->snp_alt2.push_back(a2);
            S->snp_gt.push_back(gt);
            pos += 1 + r.geometric(P.snp_rate);
        }
    }
    // read plan
    const double mean_len = P.read_len_mean;
    int64_t n_base = (int64_t)std::llround(P.depth * (double)L / mean_len);
    if (n_base < 1) n_base = 1;
    std::vector<int32_t> starts;
    {
        Rng r(P.seed, 0x5354ULL, 0);
        const int32_t span = std::max(1, L - P.read_len_min);
        for (int64_t k = 0; k < n_base; k++) starts.push_back((int32_t)r.below((uint32_t)span));
        if (P.pile_frac > 0.0 && P.pile_mult > 1.0) {
            // pile-up regions of 2 kb each
            const int32_t w = 2000;
            int64_t nreg = (int64_t)std::ceil(P.pile_frac * L / w);
            for (int64_t g = 0; g < nreg; g++) {
                int32_t a = (int32_t)r.below((uint32_t)std::max(1, L - w));
                int64_t extra = (int64_t)std::llround((P.pile_mult - 1.0) * P.depth * (w + mean_len) / mean_len);
                for (int64_t k = 0; k < extra; k++) {
                    int64_t s0 = (int64_t)a - (int64_t)mean_len + (int64_t)r.below((uint32_t)(w + (int32_t)mean_len));
                    if (s0 < 0) s0 = 0;
                    if (s0 >= span) s0 = span - 1;
                    starts.push_back((int32_t)s0);
                }
            }
        }
    }
    std::sort(starts.begin(), starts.end());
    const int64_t n = (int64_t)starts.size();
    S->plan.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        Rng r(P.seed, 0x504c414eULL, (uint64_t)i);
        ReadPlan& pl = S->plan[(size_t)i];
        pl.start = starts[(size_t)i];
        double len = mean_len + P.read_len_sd * r.normal();
        int32_t rl = (int32_t)std::llround(len);
        rl = std::max(P.read_len_min, std::min(P.read_len_max, rl));
        if (pl.start + rl > L) rl = L - pl.start;
        if (rl < 1) rl = 1;
        pl.reflen = rl;
        pl.hap = (uint8_t)(r.next() & 1);
        pl.noisy = r.uniform() < P.frac_noisy;
        pl.mapq = 60;
        if (r.uniform() < P.frac_lowmapq) pl.mapq = (uint8_t)r.below(60);
        pl.lowbq = r.uniform() < P.frac_lowbq;
        pl.clipl = pl.clipr = 0;
        if (P.softclip_max > 0 && r.uniform() < P.frac_softclip) {
            pl.clipl = (int32_t)r.below((uint32_t)P.softclip_max + 1);
            pl.clipr = (int32_t)r.below((uint32_t)P.softclip_max + 1);
        }
    }
    // pass 1: sizes
    S->qlen.resize((size_t)n);
    S->tend.resize((size_t)n);
    S->cslen.resize((size_t)n);
    parallel_for(n, P.threads, [&](int64_t i) {
        Out o{nullptr, nullptr};
        S->tend[(size_t)i] = gen_read(*S, i, o);
        S->qlen[(size_t)i] = (int32_t)o.q;
        S->cslen[(size_t)i] = (int32_t)o.c;
    });
    S->qoff.resize((size_t)n);
    S->csoff.resize((size_t)n + 1);
    int64_t qo = 0, co = 0;
    for (int64_t i = 0; i < n; i++) {
        S->qoff[(size_t)i] = qo;
        S->csoff[(size_t)i] = co;
        qo += ((int64_t)S->qlen[(size_t)i] + 31) & ~(int64_t)31;
        co += S->cslen[(size_t)i];
    }
    S->csoff[(size_t)n] = co;
    S->total_bases_padded = qo;
    S->cs_total = co;
    return S;
}

int64_t synth_n_reads(void* h) { return (int64_t)((Synth*)h)->plan.size(); }
int64_t synth_total_bases_padded(void* h) { return ((Synth*)h)->total_bases_padded; }
int64_t synth_cs_total(void* h) { return ((Synth*)h)->cs_total; }
int64_t synth_n_snps(void* h) { return (int64_t)((Synth*)h)->snp_pos.size(); }

// seq: total_bases_padded/2 bytes (zero-initialised by the caller), bq:
// total_bases_padded bytes, cs: cs_total bytes.
void synth_fill(void* h, int32_t* tstart, int32_t* tend, int32_t* qstart, int32_t* qlen, uint8_t* mapq,
                uint16_t* flag, int32_t* qid, int64_t* qoff, int64_t* cs_off, uint8_t* seq, uint8_t* bq,
                uint8_t* cs, uint8_t* tp) {
    Synth* S = (Synth*)h;
    const int64_t n = (int64_t)S->plan.size();
    parallel_for(n, S->p.threads, [&](int64_t i) {
        const size_t k = (size_t)i;
        Out o{seq + (S->qoff[k] >> 1), cs + S->csoff[k]};
        gen_read(*S, i, o);
        fill_bq(*S, i, bq + S->qoff[k], S->qlen[k]);
        tstart[k] = S->plan[k].start;
        tend[k] = S->tend[k];
        qstart[k] = S->plan[k].clipl;
        qlen[k] = S->qlen[k];
        mapq[k] = S->plan[k].mapq;
        flag[k] = 0;
        qid[k] = (int32_t)i;
        qoff[k] = S->qoff[k];
        cs_off[k] = S->csoff[k];
        tp[k] = 'P';
    });
    cs_off[n] = S->csoff[(size_t)n];
}

void synth_get_snps(void* h, int32_t* pos, uint8_t* ref, uint8_t* alt, uint8_t* gt) {
    Synth* S = (Synth*)h;
    for (size_t k = 0; k < S->snp_pos.size(); k++) {
        pos[k] = S->snp_pos[k];
        ref[k] = (uint8_t)kBase[S->ref[(size_t)S->snp_pos[k]]];
        alt[k] = (uint8_t)kBase[S->snp_alt[k]];
        gt[k] = S->snp_gt[k];
    }
}

void synth_get_ref(void* h, uint8_t* out) {
    Synth* S = (Synth*)h;
    for (size_t k = 0; k < S->ref.size(); k++) out[k] = (uint8_t)kBase[S->ref[k]];
}

void synth_destroy(void* h) { delete (Synth*)h; }

}  // extern "C"
