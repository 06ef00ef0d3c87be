// Issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands: one dependent accumulator chain against four
// independent ones, one and two wavefronts per SIMD.  Measurement aid (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc) {
    v16f acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    v8i x = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    v8i y = {0x2222aaaa, 0x2a2a2a2a, 0x22222222, (int)0xaa22aa22, 0, 0, 0, 0};
    x[0] ^= threadIdx.x << 3 & 0x88888888;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc[u % NACC] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, acc[u % NACC], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int blocks, const char *name) {
    float *out; long long *cyc, h;
    hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 64>>>(out, 100, cyc);
    hipEventRecord(e0);
    k<NACC><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 16.0 * iters;
    printf("%-34s blocks=%5d  %.3f ms  %.1f ns per instruction per wavefront  counter ticks per instruction %.2f\n", name, blocks, ms, ms * 1e6 / n, h / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>(1024, "one accumulator, 1 wave/SIMD");
    run<4>(1024, "four accumulators, 1 wave/SIMD");
    run<1>(2048, "one accumulator, 2 waves/SIMD");
    run<4>(2048, "four accumulators, 2 waves/SIMD");
    return 0;
}
