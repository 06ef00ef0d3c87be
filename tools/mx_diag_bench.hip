// Instruction mix of one wavefront-row of the matrix-core search under three formulations of the 5-column window, two
// wavefronts per SIMD, every SIMD of the chip busy.  Measurement aid (run on the GPU box); results are not meaningful,
// only the time per row is.
//
//   K5    what dense_wta_mx_kernel issues today: the five window columns in K, i.e. per 32 x 32 tile and census row ten fp4
//         products (five for the row entering the 5-row window, five for the row leaving it), each with a 16-byte operand
//         read from LDS.
//   DIAG  single-pixel cost tiles: two products per tile and row (entering, leaving), the accumulators hold the 5-ROW sum of
//         single-pixel costs, and the 5-COLUMN sum is taken along the tile's diagonal: out[i] = acc[i-2] >> 4 lanes + ... +
//         acc[i+2], a chain of four v_add_f32_dpp wave_shr:1 per accumulator register (64 per tile).
//   PFX   the same with a prefix sum along the diagonal (one v_add_f32_dpp wave_shr:1 per register), a five-lane shift of
//         the prefix through the LDS crossbar (ds_bpermute_b32) and one plain subtraction per register.
// Common to all three, per tile and row: the winner selection (13 v_max3_f32 + 6 v_max_f32_dpp), the tile's share of the
// staging (5 shifts, 6 v_and_or_b32, 2 16-byte LDS writes), one ds_max_f32, five plain additions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

enum { K5 = 0, DIAG = 1, PFX = 2, K5_NOSEL = 3, DIAG_ONLY = 4 };

template <int MODE>
__global__ __launch_bounds__(64, 2) void k(float *out, int rows) {
    __shared__ uint4 lds[1024];
    __shared__ float keys[256];
    for (int i = threadIdx.x; i < 1024; i += 64) {
        unsigned r = i * 2654435761u + blockIdx.x * 40503u;
        lds[i] = make_uint4((r & 0x88888888u) | 0x22222222u, ((r >> 1) & 0x88888888u) | 0x22222222u, ((r >> 2) & 0x88888888u) | 0x22222222u, ((r >> 3) & 0x88888888u) | 0x22222222u);
    }
    for (int i = threadIdx.x; i < 256; i += 64) keys[i] = -1e9f;
    __syncthreads();
    const int lane = threadIdx.x;
    v16f acc[5];
    for (int a = 0; a < 5; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = (float)(lane - i);
    float best[16], tmp[16];
    for (int i = 0; i < 16; ++i) { best[i] = -1e9f; tmp[i] = 0.f; }
    uint4 fb[5];
    for (int s = 0; s < 5; ++s) fb[s] = lds[lane + s];
    v8i an[5];
    for (int s = 0; s < 5; ++s) an[s] = v8i{(int)lds[lane + 64 + s].x, (int)lds[lane + 64 + s].y, (int)lds[lane + 64 + s].z, (int)lds[lane + 64 + s].w, 0, 0, 0, 0};
    unsigned w0 = lane * 2654435761u, w1 = ~w0, c22;
    asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
    const unsigned m88 = 0x88888888u;
    int perm_addr = ((lane + 59) & 63) * 4;
    float misc = (float)lane;
    for (int r = 0; r < rows; ++r) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            constexpr int dummy = 0; (void)dummy;
            const int rd = (t + 1) % 5;                  // the tile whose accumulators are read: finished at least four tiles ago
            // --- products
            constexpr int NP = (MODE == K5 || MODE == K5_NOSEL) ? 10 : 2;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                v8i b = {(int)fb[j % 5].x, (int)fb[j % 5].y, (int)fb[j % 5].z, (int)fb[j % 5].w, 0, 0, 0, 0};
                fb[j % 5] = lds[lane + ((t * 37 + j * 11 + r) & 511)];
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(an[j % 5], b, acc[t], 4, 4, 0, 0, 0, 0);
                asm volatile("" : "+v"(acc[t]));
                // the common vector work of the tile, spread over its products (K5) or behind them
                if (MODE == K5) {
                    // 13 max3 + 6 dpp max + 5 shifts + 6 and_or + 5 adds over ten products
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(best[j]) : "v"(acc[rd][j]), "v"(acc[(rd + 1) % 5][j]));
                    if (j < 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(best[10 + j]) : "v"(acc[rd][10 + j]), "v"(acc[(rd + 1) % 5][10 + j]));
                    if (j < 6) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(tmp[j]) : "v"(best[j + 8]));
                    if (j < 5) { asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(w1) : "v"(w0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(misc) : "v"(tmp[j])); }
                    if (j < 6) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(w0) : "v"(w1), "s"(m88), "v"(c22));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // left operands: ten 16-byte reads per row with the five columns in K (entering and leaving row), two with single-pixel tiles
            auto left = [&](int o) { const uint4 v = lds[lane + 64 + ((o + r) & 255)]; return v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0}; };
            if (MODE == K5 || MODE == K5_NOSEL) {
                an[t] = left(t);
                an[(t + 1) % 5] = left(t + 100);
            } else if (t == 0) {
                an[0] = left(0);
                an[1] = left(100);
            }
            // --- the 5-column sum along the diagonal of tile rd
            float c[16];
            if (MODE == DIAG || MODE == DIAG_ONLY) {
                // four interleaved chains at a time: c[i] = a[i+2] + shr1(a[i+1] + shr1(a[i] + shr1(a[i-1] + shr1(a[i-2]))))
#pragma unroll
                for (int g = 0; g < 16; g += 4) {
#pragma unroll
                    for (int s = 1; s <= 4; ++s) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int i = g + q;
                            if (s == 1) asm volatile("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=&v"(c[i]) : "v"(acc[rd][(i + 14) % 16]), "v"(acc[rd][(i + 15) % 16]));
                            else asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c[i]) : "v"(acc[rd][(i + 14 + s) % 16]));
                        }
                    }
                }
            } else if (MODE == PFX) {
                float p[16];
                asm volatile("v_mov_b32 %0, %1" : "=v"(p[0]) : "v"(acc[rd][0]));
#pragma unroll
                for (int i = 1; i < 16; ++i) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=&v"(p[i]) : "v"(p[i - 1]), "v"(acc[rd][i]));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float q = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_addr, __builtin_bit_cast(int, p[(i + 11) % 16])));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(c[i]) : "v"(p[i]), "v"(q));
                }
            }
            if (MODE == DIAG || MODE == PFX) {
                // the common vector work on the summed tile
#pragma unroll
                for (int j = 0; j < 10; ++j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(best[j]) : "v"(c[j]), "v"(c[(j + 5) % 16]));
#pragma unroll
                for (int j = 0; j < 3; ++j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(best[10 + j]) : "v"(c[10 + j]), "v"(c[13 + j]));
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(tmp[j]) : "v"(best[j + 8]));
#pragma unroll
                for (int j = 0; j < 5; ++j) { asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(w1) : "v"(w0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(misc) : "v"(tmp[j])); }
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(w0) : "v"(w1), "s"(m88), "v"(c22));
            }
            if (MODE == DIAG_ONLY) {
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(c[j]));
            }
            if (MODE != K5_NOSEL && MODE != DIAG_ONLY) {
                lds[512 + lane + 64 * (t & 1)] = make_uint4(w0, w1, w0 ^ c22, w1 ^ c22);
                lds[640 + lane + 64 * (t & 1)] = make_uint4(w1, w0, w0 ^ c22, w1 ^ c22);
                __hip_atomic_fetch_max(&keys[(lane & 31) + 32 * t], best[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = misc + __builtin_bit_cast(float, w0);
    for (int a = 0; a < 5; ++a) for (int i = 0; i < 16; ++i) sum += acc[a][i];
    for (int i = 0; i < 16; ++i) sum += best[i] + tmp[i];
    out[blockIdx.x * 64 + lane] = sum + keys[lane];
}

template <int MODE>
void run(const char *name) {
    const int blocks = 2048;          // two wavefronts on each of the 1024 SIMDs
    float *out;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    const int rows = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, 400);
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, rows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one launch of the search at KITTI size walks 64 frames x 40 x-blocks x (364 + 4 x 4 halo) rows = 973 k wavefront-rows on 2048 slots
    printf("%-10s %8.3f ms for %d rows  %7.1f ns per wavefront-row of five tiles (two wavefronts per SIMD)  -> %.3f ms per KITTI launch of 973 k wavefront-rows\n",
           name, ms, rows, ms * 1e6 / rows, ms / rows * 973000.0 / 2048.0);
    hipFree(out);
}

int main() {
    run<K5>("K5");
    run<DIAG>("DIAG");
    run<PFX>("PFX");
    run<K5_NOSEL>("K5 bare");
    run<DIAG_ONLY>("DIAG bare");
    run<K5>("K5");
    run<DIAG>("DIAG");
    return 0;
}
