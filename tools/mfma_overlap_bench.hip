// What kind of instruction beside fp4 matrix instructions fails to hide behind them?  Two wavefronts per SIMD, each trip issues ten
// products (v_mfma_f32_32x32x64_f8f6f4 on five accumulator tiles, two per tile) with, between consecutive products, N
// instructions of one kind:
//   priv   v_max3_f32 on private registers
//   acc    v_max3_f32 whose three sources are accumulator registers of OTHER tiles (as the winner selection reads them)
//   dpp    v_max_f32_dpp row_ror on private registers (with the s_nop the DPP rule needs)
//   andor  v_lshlrev + v_and_or_b32 pairs (the expansion)
//   ldsr   ds_read_b128 (conflict-free, result unused until the end of the trip)
//   ldsw   ds_write_b128
// Measurement aid (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
enum { NONE, PRIV, ACC, DPP, ANDOR, LDSR, LDSW, MIX };

template <int KIND, int N>
__global__ __launch_bounds__(64, 2) void k(float *out, int iters) {
    __shared__ uint4 lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    v8i a = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0}, b = a;
    a[0] ^= (lane << 3) & 0x88888888; b[1] ^= (lane << 7) & 0x88888888;
    v16f acc[5];
    for (int q = 0; q < 5; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = (float)(i - lane);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)(lane + i);
    unsigned w = lane * 2654435761u, c22;
    asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
    uint4 r[4] = {lds[lane], lds[lane + 64], lds[lane + 128], lds[lane + 192]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int t = u / 2, o = (t + 2) % 5, o2 = (t + 3) % 5;      // the tiles read are not the one being written
            acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 4, 4, 0, 0, 0, 0);
            asm volatile("" : "+v"(acc[t]));
#pragma unroll
            for (int v = 0; v < N; ++v) {
                const int e = (u * N + v) % 16;
                if (KIND == PRIV || (KIND == MIX && v % 4 == 0)) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[e % 8]) : "v"(f[(e + 3) % 8]), "v"(f[(e + 5) % 8]));
                if (KIND == ACC || (KIND == MIX && v % 4 == 1)) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[e % 8]) : "v"(acc[o][e]), "v"(acc[o2][e]));
                if (KIND == DPP || (KIND == MIX && v % 4 == 2)) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(f[e % 8]) : "v"(f[(e + 3) % 8]));
                if (KIND == ANDOR || (KIND == MIX && v % 4 == 3)) { unsigned s; asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(s) : "v"(w)); asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(w) : "v"(s), "s"(0x88888888u), "v"(c22)); }
                if (KIND == LDSR) r[v % 4] = lds[lane + 64 * ((u + v + it) & 15)];
                if (KIND == LDSW) lds[1024 + lane + 64 * ((u + v) & 7)] = r[v % 4];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == LDSR) asm volatile("" :: "v"(r[0].x), "v"(r[1].y), "v"(r[2].z), "v"(r[3].w));
    }
    float s = (float)w;
    for (int q = 0; q < 5; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    s += (float)(r[0].x + r[1].y + r[2].z + r[3].w);
    out[blockIdx.x * 64 + lane] = s;
}

template <int KIND, int N>
void run(const char *what) {
    const int blocks = 2048, iters = 4000;
    float *out;
    (void)hipMalloc(&out, blocks * 64 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND, N><<<blocks, 64>>>(out, 400);
    (void)hipEventRecord(e0);
    k<KIND, N><<<blocks, 64>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s x %d per product: %8.3f ms  %6.2f ns per product per SIMD\n", what, N, ms, ms * 1e6 / (10.0 * iters) / 2);
    (void)hipFree(out);
}

int main() {
    run<NONE, 0>("none");
    run<PRIV, 2>("priv"); run<PRIV, 4>("priv"); run<PRIV, 6>("priv");
    run<ACC, 2>("acc"); run<ACC, 4>("acc"); run<ACC, 6>("acc");
    run<DPP, 1>("dpp"); run<DPP, 2>("dpp"); run<DPP, 4>("dpp");
    run<ANDOR, 1>("andor"); run<ANDOR, 2>("andor");
    run<LDSR, 1>("ldsr"); run<LDSR, 2>("ldsr");
    run<LDSW, 1>("ldsw");
    run<MIX, 4>("mix"); run<MIX, 8>("mix");
    run<NONE, 0>("none");
    return 0;
}
