// fp4 matrix instruction shapes under sustained load: v_mfma_scale_f32_32x32x64_f8f6f4 against v_mfma_scale_f32_16x16x128_f8f6f4,
// the same multiply-adds per wavefront, operands re-read from LDS (one 16-byte read per instruction), random data,
// one and two wavefronts per SIMD.  Measurement aid (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int SHAPE, int NV>
__global__ __launch_bounds__(64, 2) void k(float *out, int iters, long long *cyc) {
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
    float sum = 0;
    long long t0, t1;
    if (SHAPE == 32) {
        v16f acc[4];
        for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v8i b = {(int)fb[u % 4].x, (int)fb[u % 4].y, (int)fb[u % 4].z, (int)fb[u % 4].w, 0, 0, 0, 0};
                fb[u % 4] = lds[lane + ((u * 37 + it) & 511)];
                acc[u % 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, b, acc[u % 4], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                asm volatile("" : "+v"(acc[u % 4]));
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[(u * NV + v) % 8]) : "v"(f[(u + v + 3) % 8]), "v"(f[(u + v + 5) % 8]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_readcyclecounter();
        for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) sum += acc[a][i];
    } else {
        v4f acc[16];
        for (int a = 0; a < 16; ++a) for (int i = 0; i < 4; ++i) acc[a][i] = 0.0f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {           // twice the instructions for the same multiply-adds
                v8i b = {(int)fb[u % 4].x, (int)fb[u % 4].y, (int)fb[u % 4].z, (int)fb[u % 4].w, 0, 0, 0, 0};
                fb[u % 4] = lds[lane + ((u * 37 + it) & 511)];
                acc[u % 16] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, b, acc[u % 16], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                asm volatile("" : "+v"(acc[u % 16]));
                if (u % 2 == 1) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[(u * NV + v) % 8]) : "v"(f[(u + v + 3) % 8]), "v"(f[(u + v + 5) % 8]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_readcyclecounter();
        for (int a = 0; a < 16; ++a) for (int i = 0; i < 4; ++i) sum += acc[a][i];
    }
    for (int i = 0; i < 8; ++i) sum += f[i];
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int SHAPE, int NV>
void run(int wps) {
    const int blocks = 1024 * wps;
    float *out; long long *cyc, h;
    hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, NV><<<blocks, 64>>>(out, 2000, cyc);
    hipEventRecord(e0);
    k<SHAPE, NV><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 2.0 * 32 * 32 * 64 * 16.0 * iters * blocks;
    printf("shape %2dx%2d  VOP3 per 32x32x64 worth of work %d  waves/SIMD %d:  %.2f ms  %.0f TFLOP/s  wave ticks per 32x32x64 worth %.1f  (clock est. %.2f GHz)\n",
           SHAPE, SHAPE, NV, wps, ms, flop / (ms * 1e-3) / 1e12, h / (16.0 * iters), h / (ms * 1e-3) / 1e9 / 1.0);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<32, 0>(w); run<16, 0>(w);
        run<32, 4>(w); run<16, 4>(w);
    }
    return 0;
}
