// What one SIMD sustains when every fp4 matrix instruction comes with NV independent VOP3 instructions (v_max3_f32) and
// NL 16-byte LDS reads whose data is the instruction's B operand, one and two wavefronts per SIMD.  Measurement aid.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NV, int NL, int NW>
__global__ __launch_bounds__(64, 2) void k(float *out, int iters, long long *cyc) {
    __shared__ uint4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = make_uint4(0x22222222u, 0x2a2a2a2au, 0xa2a2a2a2u ^ i, 0x22aa22aau);
    __syncthreads();
    v16f acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    v8i x = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)(threadIdx.x + i);
    const int lane = threadIdx.x;
    uint4 fb[4];
    for (int s = 0; s < 4; ++s) fb[s] = lds[lane + s];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v8i b = {(int)fb[u % 4].x, (int)fb[u % 4].y, (int)fb[u % 4].z, (int)fb[u % 4].w, 0, 0, 0, 0};
            if (NL >= 1) fb[u % 4] = lds[lane + ((u * 37 + it) & 511)];
            if (NL >= 2) { uint4 e = lds[lane + 512 + ((u * 11 + it) & 255)]; asm volatile("" :: "v"(e.x), "v"(e.y), "v"(e.z), "v"(e.w)); }
            acc[u % 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, b, acc[u % 4], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            asm volatile("" : "+v"(acc[u % 4]));
#pragma unroll
            for (int v = 0; v < NV; ++v)
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[(u * NV + v) % 8]) : "v"(f[(u + v + 3) % 8]), "v"(f[(u + v + 5) % 8]));
            if (NW) {   // a 16-byte LDS write every fourth instruction, as the staging does
                if (u % 4 == 3) lds[lane + 768 + (u & 12)] = fb[(u + 1) % 4];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, int NL, int NW>
void run(int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd;
    float *out; long long *cyc, h;
    hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, NL, NW><<<blocks, 64>>>(out, 100, cyc);
    hipEventRecord(e0);
    k<NV, NL, NW><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 16.0 * iters;
    printf("VOP3 per MFMA %d  LDS reads per MFMA %d  LDS writes %d  waves/SIMD %d:  %.3f ms  %.1f ns per MFMA per SIMD  wave ticks per MFMA %.1f\n",
           NV, NL, NW, waves_per_simd, ms, ms * 1e6 / n / waves_per_simd, h / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, 0>(w); run<0, 1, 0>(w); run<0, 2, 0>(w);
        run<2, 1, 0>(w); run<4, 1, 0>(w); run<5, 1, 0>(w); run<6, 1, 0>(w); run<8, 1, 0>(w);
        run<4, 1, 1>(w); run<5, 2, 1>(w);
    }
    return 0;
}
