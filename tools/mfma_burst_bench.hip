// Does the order of fp4 matrix instructions and vector instructions inside a wavefront change what a SIMD sustains?
// Every variant issues the same work per trip (G groups of: M products of v_mfma_f32_32x32x64_f8f6f4 on four accumulators,
// and 4 M independent v_max3_f32): finely interleaved (one product, four vector instructions), in bursts (M products, then
// 4 M vector instructions), with the two wavefronts of a SIMD running the bursts in opposite phase, and the two parts alone.
// Measurement aid (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int M, int MODE>      // MODE 0 interleaved, 1 bursts, 2 bursts in opposite phase in odd blocks, 3 products only, 4 vector only
__global__ __launch_bounds__(64, 2) void k(float *out, int iters) {
    const int lane = threadIdx.x;
    v8i a = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0}, b = a;
    a[0] ^= (lane << 3) & 0x88888888; b[1] ^= (lane << 7) & 0x88888888;
    v16f acc[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)(lane + i);
    auto prod = [&](int u) __attribute__((always_inline)) {
        acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[u & 3], 4, 4, 0, 0, 0, 0);
        asm volatile("" : "+v"(acc[u & 3]));
    };
    auto vec = [&](int u) __attribute__((always_inline)) {
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[u & 7]) : "v"(f[(u + 3) & 7]), "v"(f[(u + 5) & 7]));
    };
    const bool flip = MODE == 2 && (blockIdx.x & 1);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 2 * M; ++u) { prod(u); vec(4 * u); vec(4 * u + 1); vec(4 * u + 2); vec(4 * u + 3); __builtin_amdgcn_sched_barrier(0); }
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 2 * M; ++u) { prod(u); __builtin_amdgcn_sched_barrier(0); }
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 8 * M; ++u) { vec(u); }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (!flip) {
#pragma unroll
                    for (int u = 0; u < M; ++u) { prod(u); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                    for (int u = 0; u < 4 * M; ++u) vec(u);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int u = 0; u < 4 * M; ++u) vec(u);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < M; ++u) { prod(u); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 64 + lane] = s;
}

template <int M, int MODE>
void run(int wps, const char *what) {
    const int blocks = 1024 * wps, iters = 4000;
    float *out;
    (void)hipMalloc(&out, blocks * 64 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<M, MODE><<<blocks, 64>>>(out, 400);
    (void)hipEventRecord(e0);
    k<M, MODE><<<blocks, 64>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("M=%2d %-34s waves/SIMD %d: %8.3f ms  %6.1f ns per product per SIMD\n", M, what, wps, ms, ms * 1e6 / (2.0 * M * iters) / wps);
    (void)hipFree(out);
}

template <int M>
void all() {
    for (int w = 1; w <= 2; ++w) {
        run<M, 3>(w, "products only");
        run<M, 4>(w, "vector instructions only");
        run<M, 0>(w, "interleaved 1 : 4");
        run<M, 1>(w, "bursts");
        if (w == 2) run<M, 2>(w, "bursts, opposite phase");
    }
}

int main() {
    all<5>();
    all<25>();
    return 0;
}
