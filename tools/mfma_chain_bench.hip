// Does it matter in which order a wavefront issues the fp4 products of its five accumulator tiles?  Per trip 50 products
// (v_mfma_f32_32x32x64_f8f6f4, fp4 operands): ten in a row into the same accumulator (what the search kernel issues), the
// products of two tiles alternating, or round-robin over the five tiles; alone and with three independent v_max3_f32 behind
// every product; one and two wavefronts per SIMD.  Measurement aid (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int ORDER, int VALU>      // ORDER 0 chain of ten, 1 pairs of tiles alternating, 2 round-robin over five
__global__ __launch_bounds__(64, 2) void k(float *out, int iters) {
    const int lane = threadIdx.x;
    v8i a[5], b[5];
    for (int s = 0; s < 5; ++s) {
        a[s] = v8i{0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
        b[s] = a[s];
        a[s][0] ^= ((lane + s) << 3) & 0x88888888; b[s][1] ^= ((lane + s) << 7) & 0x88888888;
    }
    v16f acc[5];
    for (int q = 0; q < 5; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    float f[6];
    for (int i = 0; i < 6; ++i) f[i] = (float)(lane + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 50; ++s) {
            int t;
            if (ORDER == 0) t = s / 10;
            else if (ORDER == 1) t = s < 40 ? 2 * (s / 20) + (s & 1) : 4;
            else t = s % 5;
            acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s % 5], b[(s / 5) % 5], acc[t], 4, 4, 0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU; ++v)
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[(s * VALU + v) % 6]) : "v"(f[(s * VALU + v + 2) % 6]), "v"(f[(s * VALU + v + 4) % 6]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int q = 0; q < 5; ++q) for (int i = 0; i < 16; ++i) sum += acc[q][i];
    for (int i = 0; i < 6; ++i) sum += f[i];
    out[blockIdx.x * 64 + lane] = sum;
}

template <int ORDER, int VALU>
void run(const char *name, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 2000;
    float *out;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ORDER, VALU><<<blocks, 64>>>(out, 100);
    hipEventRecord(e0);
    k<ORDER, VALU><<<blocks, 64>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %d VALU per product, %d wavefront(s) per SIMD: %8.3f ms  %6.2f ns per product per SIMD\n", name, VALU, waves_per_simd, ms,
           ms * 1e6 / (iters * 50.0 * waves_per_simd));
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("ten in a row per tile", w); run<1, 0>("two tiles alternating", w); run<2, 0>("round-robin over five", w);
        run<0, 3>("ten in a row per tile", w); run<1, 3>("two tiles alternating", w); run<2, 3>("round-robin over five", w);
    }
    return 0;
}
