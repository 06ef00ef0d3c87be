// The fp4 matrix instruction with unit block scales (v_mfma_scale_f32_32x32x64_f8f6f4, scale bytes 0x7f) against the form
// without scale operands (v_mfma_f32_32x32x64_f8f6f4, what the compiler selects for zero scale arguments): are the results
// the same, and does the shorter encoding issue faster?  Measurement aid (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <bool SCALED, int NV>
__global__ __launch_bounds__(64, 2) void k(float *out, int iters) {
    __shared__ uint4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) {
        unsigned r = i * 2654435761u + blockIdx.x * 40503u;
        lds[i] = make_uint4((r & 0x88888888u) | 0x22222222u, ((r >> 1) & 0x88888888u) | 0x22222222u, ((r >> 2) & 0x88888888u) | 0x22222222u, ((r >> 3) & 0x88888888u) | 0x22222222u);
    }
    __syncthreads();
    const int lane = threadIdx.x;
    v8i x = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    x[0] ^= (lane << 3) & 0x88888888;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)(lane + i);
    uint4 fb[4];
    for (int s = 0; s < 4; ++s) fb[s] = lds[lane + s];
    v16f acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v8i b = {(int)fb[u % 4].x, (int)fb[u % 4].y, (int)fb[u % 4].z, (int)fb[u % 4].w, 0, 0, 0, 0};
            fb[u % 4] = lds[lane + ((u * 37 + it) & 511)];
            if (SCALED) acc[u % 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, b, acc[u % 4], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else acc[u % 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, b, acc[u % 4], 4, 4, 0, 0, 0, 0);
            asm volatile("" : "+v"(acc[u % 4]));
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[(u * NV + v) % 8]) : "v"(f[(u + v + 3) % 8]), "v"(f[(u + v + 5) % 8]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) sum += acc[a][i] * (float)(1 + i + 16 * a);
    for (int i = 0; i < 8; ++i) sum += f[i];
    out[blockIdx.x * 64 + lane] = sum;
}

template <bool SCALED, int NV>
double run(int wps, std::vector<float> &keep) {
    const int blocks = 1024 * wps;
    float *out;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SCALED, NV><<<blocks, 64>>>(out, 7);
    keep.resize(blocks * 64);
    hipMemcpy(keep.data(), out, keep.size() * sizeof(float), hipMemcpyDeviceToHost);
    k<SCALED, NV><<<blocks, 64>>>(out, 2000);
    hipEventRecord(e0);
    k<SCALED, NV><<<blocks, 64>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 64 * 16.0 * iters * blocks;
    printf("%s  VOP3 per instruction %d  waves/SIMD %d:  %.2f ms  %.0f TFLOP/s\n", SCALED ? "unit scales" : "no scales  ", NV, wps, ms, flop / (ms * 1e-3) / 1e12);
    hipFree(out);
    return ms;
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        std::vector<float> a, b;
        run<true, 0>(w, a); run<false, 0>(w, b);
        printf("   results equal: %s (first values %g %g)\n", a.size() == b.size() && !memcmp(a.data(), b.data(), a.size() * 4) ? "yes" : "NO", a[5], b[5]);
        run<true, 4>(w, a); run<false, 4>(w, b);
        printf("   results equal: %s\n", a.size() == b.size() && !memcmp(a.data(), b.data(), a.size() * 4) ? "yes" : "NO");
    }
    return 0;
}
