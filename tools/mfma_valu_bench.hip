// Stand-alone micro-benchmark (measurement aid, not part of libsvx): does the matrix pipe run beside the vector ALU?
// Times (a) a stream of v_mfma_i32_32x32x32_i8, (b) a stream of v_xor + v_bcnt pairs, (c) both interleaved in the same
// wavefronts, at 1/2/4 wavefronts per SIMD, and prints cycles per SIMD for each so that (c) can be compared with
// max(a, b) and a + b.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_valu_bench tools/mfma_valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MFMA_PER_TRIP = 8, PAIRS_PER_MFMA = 8;     // 8 xor+bcnt pairs (16 instructions) beside each MFMA

template <bool DO_MFMA, bool DO_VALU>
__global__ __launch_bounds__(256) void k(uint32_t *sink, int iters) {
    v4i a, b;
    for (int i = 0; i < 4; ++i) { a[i] = (int)(threadIdx.x * 2654435761u + i); b[i] = (int)(threadIdx.x * 40503u + 7 * i); }
    v16i acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = i + j;
    uint32_t s[8], x[8];
    for (int i = 0; i < 8; ++i) { s[i] = threadIdx.x + i; x[i] = threadIdx.x * 747796405u + i * 2891336453u; }
    const uint32_t key = blockIdx.x * 0x9e3779b9u + 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < MFMA_PER_TRIP; ++u) {
            if (DO_MFMA) acc[u & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u & 3], 0, 0, 0);
            if (DO_VALU) {
#pragma unroll
                for (int i = 0; i < PAIRS_PER_MFMA; ++i)
                    asm volatile("v_xor_b32_e32 %1, %2, %1\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(s[i]), "+v"(x[i]) : "v"(key));
            }
        }
    }
    uint32_t r = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) r += (uint32_t)acc[j][i];
    for (int i = 0; i < 8; ++i) r += s[i] + x[i];
    if (r == 0x12345678u) sink[0] = r;                     // keeps everything alive
}

template <bool M, bool V>
static int run(const char *name, int waves_per_simd, uint32_t *sink, double ghz) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 4000;
    const int blocks = cus * waves_per_simd;                // 256 threads = 4 wavefronts = one per SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<M, V><<<blocks, 256>>>(sink, 100);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<M, V><<<blocks, 256>>>(sink, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double trips = (double)iters * MFMA_PER_TRIP * waves_per_simd;    // per SIMD
    const double cyc = ms * 1e-3 * ghz * 1e9 / trips;
    printf("%-22s waves/SIMD %d  %8.3f ms  %7.1f cycles per (MFMA%s) slot per SIMD at %.2f GHz", name, waves_per_simd, ms, cyc,
           V ? " + 8 xor/bcnt pairs" : "", ghz);
    if (M) printf("  -> %.0f T i8-MAC/s chip-wide", (double)cus * 4 * trips * 32768.0 / (ms * 1e-3) / 1e12);
    if (V) printf("  -> %.1f T lane-op/s", (double)cus * 4 * trips * PAIRS_PER_MFMA * 2 * 64 / (ms * 1e-3) / 1e12);
    printf("\n");
    return 0;
}

int main() {
    uint32_t *sink;
    CHK(hipMalloc(&sink, 256));
    const double ghz = 2.4;
    for (int w : {1, 2, 4}) {
        if (run<true, false>("mfma only", w, sink, ghz)) return 1;
        if (run<false, true>("xor+bcnt only", w, sink, ghz)) return 1;
        if (run<true, true>("interleaved", w, sink, ghz)) return 1;
    }
    CHK(hipFree(sink));
    return 0;
}
