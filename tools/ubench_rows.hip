// Micro-benchmark of the normcounts sweep's access pattern (tools only; no product code): a wave per 256 reference positions
// loads, for each of the ~30 reads over them, a row of 256 qualities (a dword per lane, any alignment), the 128 bytes of packed
// bases under it and 32 bytes of bits, NB rows in flight, and does next to nothing with them.  What rate does the memory
// system give this shape, by rows in flight, waves per CU, alignment and which of the three arrays are read?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_rows tools/ubench_rows.hip && /tmp/ubench_rows
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int L = 15008;        // bases per read (multiple of 32)
constexpr int STEP = 500;       // reference positions between read starts (30x)

template <int NB, int MODE, bool ALIGNED>
__global__ void __launch_bounds__(256) k_rows(const uint8_t* bq, const uint8_t* seq, const uint8_t* bits, int64_t ntiles, int64_t nreads,
                                              uint32_t* out, int lds_pad) {
    extern __shared__ uint32_t s_pad[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t acc = 0;
    if (lds_pad && threadIdx.x == 0) s_pad[0] = 1;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t base = tile * 256;
        int64_t r_lo = (base + 256 - L + STEP - 1) / STEP; if (r_lo < 0) r_lo = 0;
        int64_t r_hi = base / STEP + 1; if (r_hi > nreads) r_hi = nreads;
        for (int64_t r0 = r_lo; r0 < r_hi; r0 += NB) {
            uint32_t q[NB], s[NB], b[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                int64_t r = r0 + k < r_hi ? r0 + k : r_hi - 1;
                int64_t K = r * L + (base - r * STEP);
                if (K < r * L) K = r * L;
                if (K > r * L + L - 256) K = r * L + L - 256;
                if (ALIGNED) K &= ~(int64_t)7;
                q[k] = 0; s[k] = 0; b[k] = 0;
                if (MODE & 1) __builtin_memcpy(&q[k], bq + K + 4 * lane, 4);
                if (MODE & 2) __builtin_memcpy(&s[k], seq + (K >> 1) + 2 * lane, 4);
                if (MODE & 4) { uint16_t t; __builtin_memcpy(&t, bits + (K >> 3) + (lane >> 1), 2); b[k] = t; }
            }
#pragma unroll
            for (int k = 0; k < NB; k++) acc += q[k] ^ (s[k] >> 3) ^ b[k];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int NB, int MODE, bool ALIGNED>
int run(const char* name, const uint8_t* bq, const uint8_t* seq, const uint8_t* bits, int64_t ntiles, int64_t nreads, uint32_t* out,
        int blocks_per_cu) {
    // occupancy is set by dynamic LDS: 160 KB / blocks_per_cu per block
    const int lds = blocks_per_cu >= 8 ? 0 : (160 * 1024 / blocks_per_cu) - 1024;
    CK(hipFuncSetAttribute((const void*)k_rows<NB, MODE, ALIGNED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu * 4;
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_rows<NB, MODE, ALIGNED>), dim3(grid), dim3(256), lds, 0, bq, seq, bits, ntiles, nreads, out, lds);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; it++) hipLaunchKernelGGL((k_rows<NB, MODE, ALIGNED>), dim3(grid), dim3(256), lds, 0, bq, seq, bits, ntiles, nreads, out, lds);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double rows = (double)ntiles * 30.0;
    const double bytes = rows * ((MODE & 1 ? 256.0 : 0) + (MODE & 2 ? 128.0 : 0) + (MODE & 4 ? 32.0 : 0));
    printf("%-34s waves/CU %2d  %7.3f ms  %6.2f TB/s useful  %5.1f Mrows/ms\n", name, blocks_per_cu * 4, ms, bytes / ms / 1e9, rows / ms / 1e6);
    return 0;
}

int main() {
    const int64_t npos = 64444167, nreads = npos / STEP, ntiles = npos / 256;
    const size_t nb = (size_t)nreads * L + 4096;
    uint8_t *bq, *seq, *bits; uint32_t* out;
    CK(hipMalloc(&bq, nb)); CK(hipMalloc(&seq, nb / 2 + 4096)); CK(hipMalloc(&bits, nb / 8 + 4096)); CK(hipMalloc(&out, 256));
    CK(hipMemset(bq, 93, nb)); CK(hipMemset(seq, 0x12, nb / 2 + 4096)); CK(hipMemset(bits, 0xff, nb / 8 + 4096));
    printf("rows of 256 positions x %lld tiles, %lld reads of %d bases (%.2f GB of qualities)\n", (long long)ntiles, (long long)nreads, L, nb / 1e9);
#define R(NB, MODE, AL, BPC) if (run<NB, MODE, AL>("NB=" #NB " mode=" #MODE " aligned=" #AL, bq, seq, bits, ntiles, nreads, out, BPC)) return 1;
    R(4, 7, false, 4) R(4, 7, false, 8) R(4, 7, false, 2)
    R(2, 7, false, 4) R(8, 7, false, 4) R(8, 7, false, 8) R(16, 7, false, 4)
    R(4, 7, true, 4) R(8, 7, true, 8)
    R(4, 1, false, 4) R(4, 3, false, 4) R(4, 1, true, 4) R(8, 1, false, 8) R(8, 1, true, 8)
    return 0;
}
