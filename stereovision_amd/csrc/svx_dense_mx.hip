// Dense 5x5-aggregated Hamming winner-takes-all on the matrix cores (two-word descriptors, left view).
//
// Same result as the vector sweep of svx_dense.hip (search rule of denseCensusStereo/denseCensusStereo.cpp:87-131
// on 64-bit descriptors); the cost values are formed as dot products.  With every descriptor bit b written as the
// number 1 - 2b, the dot product of two 64-element descriptors is 64 - 2 * Hamming distance, and the cost of a
// 5-column window is the dot product of the two windows' concatenated descriptors (K = 5 x 64).  One wavefront
// owns 32 left columns and the 32*T right columns they can match (T tiles of 32 x 32 left/right pairs): per census
// row it adds the row that enters the 5-row window (5 k-steps per tile) and subtracts the row that leaves it (the
// same products with the left operand negated), so the accumulators always hold
//     acc[tile][left column m][right column n] = 1600 - 2 * window cost.
// Operands are fp4 (E2M1: +1.0 = 0x2, -1.0 = 0xA) with unit block scales: every product and sum is an exact small
// integer in the f32 accumulators.  The disparity tag and the admissibility mask are one fused multiply-add per
// accumulator element (key = acc * 1024 + bias, bias = 1023 - d or a large negative number), the winner of a left
// column is the maximum key over the tiles and over the 32 lanes that hold its right columns.
#include "svx_common.hpp"

typedef int mx_v8i __attribute__((ext_vector_type(8)));
typedef float mx_v16f __attribute__((ext_vector_type(16)));

constexpr int MX_M = 32;                 // left columns per wavefront
constexpr int MX_MAX_STRIP = 96;

template <int T>
struct MxGeom {
    static constexpr int RPX = 32 * T + 4;          // right pixels staged per row (2-column window halo each side)
    static constexpr int LPX = MX_M + 4;
    static constexpr int RW = 2 * RPX, LW = 2 * LPX, WORDS = RW + LW;   // census words per staged row
    static constexpr int ROUNDS = (WORDS + 63) / 64;
};

// 32 descriptor bits -> 32 fp4 values: dword j holds bits j, j+4, .. j+28 (the same permutation on both operands).
// c22 = 0x22222222 in a vector register (an instruction takes one scalar constant only).
__device__ __forceinline__ uint4 mx_expand(uint32_t w, uint32_t c22) {
    uint4 r;
    const uint32_t m88 = 0x88888888u;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.x) : "v"(w << 3), "s"(m88), "v"(c22));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.y) : "v"(w << 2), "s"(m88), "v"(c22));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.z) : "v"(w << 1), "s"(m88), "v"(c22));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.w) : "v"(w), "s"(m88), "v"(c22));
    return r;
}

__device__ __forceinline__ mx_v8i mx_frag(const uint4 v) {
    mx_v8i f = {(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    return f;
}

// Maximum over the 16 lanes of every row for 16 registers at once: lane l ends up with register (l & 15).
__device__ __forceinline__ float mx_reduce16(const float (&k)[16], int lane) {
    float r[8], q[4], u[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // lanes 0..7 of a row (banks 0,1) keep register i, lanes 8..15 register i + 8
        asm volatile("s_nop 1\n\t"
                     "v_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_max_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
                     : "=&v"(r[i]) : "v"(k[i]), "v"(k[i + 8]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // banks 0,2 keep register i (partner lane + 4), banks 1,3 register i + 4 (partner lane - 4)
        asm volatile("s_nop 1\n\t"
                     "v_max_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_max_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(q[i]) : "v"(r[i]), "v"(r[i + 4]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm volatile("s_nop 1\n\t"
                     "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                     : "=&v"(u[i]) : "v"(q[i]));
    }
    const float w0 = (lane & 1) ? u[1] : u[0];
    const float w1 = (lane & 1) ? u[3] : u[2];
    float x0, x1;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(x0), "=&v"(x1) : "v"(w0), "v"(w1));
    return (lane & 2) ? x1 : x0;
}

// One census row: stage the entering row (and the leaving one) as fp4 operands, run the products, pick the winners.
// The staging buffers are single: LDS operations of one wavefront execute in order, so the writes of a row follow the
// reads of the previous one.
template <int T, bool LEAVING, bool OUTPUT>
__device__ __forceinline__ void mx_row(mx_v16f (&acc)[T], uint4 *__restrict__ bn, uint4 *__restrict__ bo,
                                       const uint32_t (&wn)[MxGeom<T>::ROUNDS], const uint32_t (&wo)[MxGeom<T>::ROUNDS],
                                       uint32_t left_mask_hi, uint32_t left_mask_lo, int lane, float *__restrict__ out) {
    using G = MxGeom<T>;
    const int n = lane & 31, h = lane >> 5;
    uint32_t c22;
    asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
#pragma unroll
    for (int i = 0; i < G::ROUNDS; ++i) bn[i * 64 + lane] = mx_expand(wn[i], c22);
    if (LEAVING) {
        // the leaving row's left words are complemented: its products are subtracted
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i) {
            const uint32_t flip = i == G::ROUNDS - 1 ? left_mask_hi : (i == G::ROUNDS - 2 ? left_mask_lo : 0u);
            bo[i * 64 + lane] = mx_expand(wo[i] ^ flip, c22);
        }
    }
    __builtin_amdgcn_wave_barrier();
    mx_v8i an[5], ao[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        an[k] = mx_frag(bn[G::RW + (n + k) * 2 + h]);
        if (LEAVING) ao[k] = mx_frag(bo[G::RW + (n + k) * 2 + h]);
    }
    // T * PER products; the right operand of product s + P is fetched before product s is issued
    constexpr int PER = LEAVING ? 10 : 5, S = T * PER, P = 4;
    auto right = [&](int s) -> uint4 {
        const int t = s / PER, j = s % PER, k = j % 5;
        return (j >= 5 ? bo : bn)[(32 * t + n + k) * 2 + h];
    };
    uint4 fb[P];
#pragma unroll
    for (int s = 0; s < P; ++s) fb[s] = right(s);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int t = s / PER, j = s % PER, k = j % 5;
        const mx_v8i b = mx_frag(fb[s % P]);
        if (s + P < S) fb[s % P] = right(s + P);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(j >= 5 ? ao[k] : an[k], b, acc[t], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
    if (OUTPUT) {
        float best[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float b = acc[0][i];
#pragma unroll
            for (int t = 1; t + 1 < T; t += 2) b = __builtin_fmaxf(__builtin_fmaxf(b, acc[t][i]), acc[t + 1][i]);
            if ((T & 1) == 0) b = __builtin_fmaxf(b, acc[T - 1][i]);
            best[i] = b;
        }
        float z = mx_reduce16(best, lane);                       // lane l: register (l & 15) over its 16-lane row
        const float zo = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, z), 0x401f));   // lane ^ 16
        z = __builtin_fmaxf(z, zo);
        // key = window score + (1023 - d) / 1024; an inadmissible pair sits near -2^20
        const float frac = z - __builtin_floorf(z);
        const float d = z < -524288.0f ? 0.0f : 1023.0f - frac * 1024.0f;
        if ((lane & 16) == 0) *out = d;
    }
}

// grid: (x-blocks, strips, frames), one wavefront per block.  x-block b owns output columns [xl0, xl0 + 32),
// xl0 = min(2 + 32 b, cols - 34) (the last block is shifted left to end at column cols-3).
// Requires cols >= 36, rows >= 5, max_disp <= 32 * (T - 1).
template <int T>
__global__ __launch_bounds__(64) void dense_wta_mx_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                          int rows, int cols, int max_disp, int strip_rows,
                                                          float *__restrict__ disp_all, int write_border) {
    using G = MxGeom<T>;
    __shared__ uint4 lds[2][G::ROUNDS * 64];            // [entering row, leaving row][census word, padded to whole rounds]
    const int lane = threadIdx.x;
    const int n = lane & 31, h = lane >> 5;
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    unsigned wi = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((total & 7u) == 0u) wi = (wi & 7u) * (total >> 3) + (wi >> 3);      // contiguous run of work items per XCD
    const unsigned bx = wi % gridDim.x, by = (wi / gridDim.x) % gridDim.y, bz = wi / (gridDim.x * gridDim.y);
    const long npx = (long)rows * cols;
    const uint32_t *cl = cl_all + bz * npx * 2;
    const uint32_t *cr = cr_all + bz * npx * 2;
    float *disp = disp_all + bz * npx;
    const int n_xblocks = gridDim.x;
    const int xl0 = min(2 + (int)bx * MX_M, cols - 2 - MX_M);
    const int r0 = xl0 - 32 * (T - 1);               // right column of tile 0, lane 0
    const int y_lo = 2 + (int)by * strip_rows, y_hi = min(y_lo + strip_rows, rows - 2);
    const int n_rows = y_hi - y_lo + 4;              // census rows y_lo-2 .. y_hi+1

    // Accumulator element i of tile t is (left column xl0 + m, right column r0 + 32 t + n).  It starts at
    // (1023 - d) / 1024, so that the running window score carries its disparity tag below the integer part
    // (21 significant bits: exact), or near -2^20 when the pair is inadmissible.
    mx_v16f acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int xr = r0 + 32 * t + n, d = xl0 + m - xr;
            const bool ok = d >= 0 && d < max_disp && xr >= 5;
            acc[t][i] = ok ? (float)(1023 - d) * (1.0f / 1024.0f) : -1048576.0f;
        }

    // staging: word q of a row is right word (r0 - 2) * 2 + q for q < RW, left word (xl0 - 2) * 2 + q - RW after it;
    // words left of the frame and the padding of the last round are clamped: they only feed inadmissible pairs
    const uint32_t *src[G::ROUNDS];
    uint32_t left_mask[G::ROUNDS];
#pragma unroll
    for (int i = 0; i < G::ROUNDS; ++i) {
        const int q = min(i * 64 + lane, G::WORDS - 1);
        const bool is_left = q >= G::RW;
        src[i] = is_left ? cl + (xl0 - 2) * 2 + (q - G::RW) : cr + max((r0 - 2) * 2 + q, 0);
        left_mask[i] = is_left ? 0xffffffffu : 0u;
    }
    static_assert(G::RW >= (G::ROUNDS - 2) * 64, "left words must lie in the last two rounds");
    auto fetch = [&](int r, uint32_t (&w)[G::ROUNDS]) {
        const long rowoff = (long)(y_lo - 2 + r) * cols * 2;
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i) w[i] = src[i][rowoff];
    };
    uint32_t wn[G::ROUNDS], wo[G::ROUNDS];
    const int i_out = lane & 15;
    float *out = disp + (long)y_lo * cols + xl0 + (i_out & 3) + 8 * (i_out >> 2) + 4 * h;
    const int last = n_rows - 1;

    fetch(0, wn);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        uint32_t wc[G::ROUNDS];
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i) wc[i] = wn[i];
        fetch(r + 1, wn);
        mx_row<T, false, false>(acc, lds[0], lds[1], wc, wc, 0u, 0u, lane, out);
    }
    {
        uint32_t wc[G::ROUNDS];
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i) wc[i] = wn[i];
        fetch(min(5, last), wn);
        fetch(0, wo);
        mx_row<T, false, true>(acc, lds[0], lds[1], wc, wc, 0u, 0u, lane, out);
        out += cols;
    }
#pragma unroll 1
    for (int r = 5; r < n_rows; ++r) {
        uint32_t wc[G::ROUNDS], wd[G::ROUNDS];
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i) { wc[i] = wn[i]; wd[i] = wo[i]; }
        fetch(min(r + 1, last), wn);
        fetch(r - 4, wo);
        mx_row<T, true, true>(acc, lds[0], lds[1], wc, wd, left_mask[G::ROUNDS - 1], left_mask[G::ROUNDS - 2], lane, out);
        out += cols;
    }

    if (write_border) {
        // rows outside [2, rows-2) over this block's columns, and the two border columns on either side
        const int ys = by == 0 ? 0 : y_lo, y_end = by == gridDim.y - 1 ? rows : y_hi;
        for (int y = ys + h; y < y_end; y += 2)
            if (y < 2 || y >= rows - 2) disp[(long)y * cols + xl0 + n] = 0.0f;
        if ((int)bx == 0)
            for (int y = ys + (lane >> 1); y < y_end; y += 32) disp[(long)y * cols + (lane & 1)] = 0.0f;
        if ((int)bx == n_xblocks - 1)
            for (int y = ys + (lane >> 1); y < y_end; y += 32) disp[(long)y * cols + cols - 2 + (lane & 1)] = 0.0f;
    }
}

bool svx_mx_sweep_eligible(int rows, int cols, int max_disp) { return cols >= MX_M + 4 && rows >= 5 && max_disp >= 1 && max_disp <= 128; }

// Left-view search on two-word descriptors; see svx_dense.hip for the meaning of the arguments.
int svx_launch_mx_sweep(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int max_disp,
                        float *disparity, int write_border) {
    const int strip_rows = MX_MAX_STRIP;
    const int n_strips = (rows - 4 + strip_rows - 1) / strip_rows;
    const int n_xblocks = (cols - 4 + MX_M - 1) / MX_M;
    dim3 grid(n_xblocks, n_strips, n_frames);
    if (max_disp <= 64)
        dense_wta_mx_kernel<3><<<grid, 64, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, strip_rows, disparity, write_border);
    else
        dense_wta_mx_kernel<5><<<grid, 64, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, strip_rows, disparity, write_border);
    SVX_LAUNCH_CHECK(ctx, "dense_wta_mx_kernel");
    return SVX_OK;
}
