// a1 (unsigned -> signed pixels) and a2 (5x5 census transform) for gfx950.
//
// Both kernels are HBM-bound: a1 moves 1 B in + 1 B out per pixel, a2 1 B in +
// 4 B out per pixel.  The census kernel stages a 256 x 16 pixel tile plus its
// 2-pixel halo in LDS with 16-byte-coalesced row loads (the halo re-reads hit
// L2) and every lane emits four consecutive descriptors as one 16-byte store.
#include "svx_common.hpp"

// ---------------------------------------------------------------- a1 -----
__global__ __launch_bounds__(256) void u8_to_s8_vec_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long nvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
        dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void u8_to_s8_byte_kernel(const uint8_t *__restrict__ src, int8_t *__restrict__ dst, long first, long n) {
    for (long i = first + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = (int8_t)(src[i] ^ 0x80u);
}

SVX_API int svx_u8_to_s8(svx_ctx *ctx, const uint8_t *src, int8_t *dst, int rows, int cols, int n_frames) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, src && dst, SVX_ERR_INVALID_ARG, "svx_u8_to_s8: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_u8_to_s8: empty image");
    const long n = (long)rows * cols * n_frames;
    long nvec = 0;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) nvec = n / 16;
    if (nvec > 0) {
        unsigned grid = svx_cdiv(nvec, 256);
        if (grid > 2048) grid = 2048;
        u8_to_s8_vec_kernel<<<grid, 256, 0, ctx->stream>>>((const uint4 *)src, (uint4 *)dst, nvec);
        SVX_LAUNCH_CHECK(ctx, "u8_to_s8_vec_kernel");
    }
    if (nvec * 16 < n) {
        unsigned grid = svx_cdiv(n - nvec * 16, 256);
        if (grid > 2048) grid = 2048;
        u8_to_s8_byte_kernel<<<grid, 256, 0, ctx->stream>>>(src, dst, nvec * 16, n);
        SVX_LAUNCH_CHECK(ctx, "u8_to_s8_byte_kernel");
    }
    return SVX_OK;
}

// ---------------------------------------------------------------- a2 -----
// Bit position of neighbour (dy,dx) in the 24-bit word: idx=(dy+2)*5+(dx+2),
// g=24-idx, bit = g<12 ? g : g-1  (the centre, idx 12, has no bit).
__host__ __device__ constexpr int census_bit24(int dy, int dx) {
    return (24 - ((dy + 2) * 5 + (dx + 2))) < 12 ? (24 - ((dy + 2) * 5 + (dx + 2))) : (24 - ((dy + 2) * 5 + (dx + 2))) - 1;
}
// 24-bit word -> 25-bit word of the generic template (bit = 24-idx).
__device__ __forceinline__ uint32_t census24_to_25(uint32_t w) { return (w & 0xFFFu) | ((w >> 12) << 13); }

constexpr int CT_W = 256;            // tile width in pixels (one wave spans it: 64 lanes x 4 px)
constexpr int CT_H = 16;             // tile height: 4 waves x 4 rows
constexpr int CT_LROW = CT_W / 4 + 2;  // dwords per staged row: columns x0-4 .. x0+CT_W+3

// The 24 bits of one pixel, most significant first (idx = 0..24 without the centre): word = 2*word +
// (centre > neighbour) per step, both operands picked as unsigned bytes straight out of packed registers by
// SDWA (no extraction instructions), the compare result carried into the word by an add-with-carry.  Pixels
// are kept in the UNSIGNED domain (signed input is biased by 0x80 at staging), where the reference's signed
// compare is an unsigned one.  Row windows: L[r] = columns x-2..x+1, H[r] = columns x+2..x+5 of row r of the
// 5-row window; neighbour (dy,dx) of pixel I is window byte 2+I+dx of row dy+2, the centre byte 2+I of row 2.
// One asm block per pixel (48 instructions).
#define CENSUS_PX0(WORD, L, H)                                                             \
    asm(                                                                                       \
        "v_cmp_gt_u32_sdwa vcc, %3, %1 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %1 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %1 src0_sel:BYTE_2 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %1 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %6 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:BYTE_2 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %7 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %3 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %3 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %3 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %8 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %4 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %4 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %4 src0_sel:BYTE_2 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %4 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %9 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %5 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %5 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %5 src0_sel:BYTE_2 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %5 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %10 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        : "+v"(WORD)                                                                       \
        : "v"(L[0]), "v"(L[1]), "v"(L[2]), "v"(L[3]), "v"(L[4]), "v"(H[0]), "v"(H[1]), "v"(H[2]), "v"(H[3]), "v"(H[4]) \
        : "vcc")

#define CENSUS_PX1(WORD, L, H)                                                             \
    asm(                                                                                       \
        "v_cmp_gt_u32_sdwa vcc, %3, %1 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %1 src0_sel:BYTE_3 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %1 src0_sel:BYTE_3 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %6 src0_sel:BYTE_3 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %6 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:BYTE_3 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %2 src0_sel:BYTE_3 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %7 src0_sel:BYTE_3 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %7 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %3 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %3 src0_sel:BYTE_3 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %8 src0_sel:BYTE_3 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %8 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %4 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %4 src0_sel:BYTE_3 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %4 src0_sel:BYTE_3 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %9 src0_sel:BYTE_3 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %9 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %5 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %5 src0_sel:BYTE_3 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %5 src0_sel:BYTE_3 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %10 src0_sel:BYTE_3 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %3, %10 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        : "+v"(WORD)                                                                       \
        : "v"(L[0]), "v"(L[1]), "v"(L[2]), "v"(L[3]), "v"(L[4]), "v"(H[0]), "v"(H[1]), "v"(H[2]), "v"(H[3]), "v"(H[4]) \
        : "vcc")

#define CENSUS_PX2(WORD, L, H)                                                             \
    asm(                                                                                       \
        "v_cmp_gt_u32_sdwa vcc, %8, %1 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %1 src0_sel:BYTE_0 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %6 src0_sel:BYTE_0 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %6 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %6 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %2 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %2 src0_sel:BYTE_0 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %7 src0_sel:BYTE_0 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %7 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %7 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %3 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %3 src0_sel:BYTE_0 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %8 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %8 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %4 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %4 src0_sel:BYTE_0 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %9 src0_sel:BYTE_0 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %9 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %9 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %5 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %5 src0_sel:BYTE_0 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %10 src0_sel:BYTE_0 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %10 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %10 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        : "+v"(WORD)                                                                       \
        : "v"(L[0]), "v"(L[1]), "v"(L[2]), "v"(L[3]), "v"(L[4]), "v"(H[0]), "v"(H[1]), "v"(H[2]), "v"(H[3]), "v"(H[4]) \
        : "vcc")

#define CENSUS_PX3(WORD, L, H)                                                             \
    asm(                                                                                       \
        "v_cmp_gt_u32_sdwa vcc, %8, %1 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %6 src0_sel:BYTE_1 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %6 src0_sel:BYTE_1 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %6 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %6 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %2 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %7 src0_sel:BYTE_1 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %7 src0_sel:BYTE_1 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %7 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %7 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %3 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %8 src0_sel:BYTE_1 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %8 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %8 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %4 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %9 src0_sel:BYTE_1 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %9 src0_sel:BYTE_1 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %9 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %9 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %5 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %10 src0_sel:BYTE_1 src1_sel:BYTE_0\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %10 src0_sel:BYTE_1 src1_sel:BYTE_1\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %10 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        "v_cmp_gt_u32_sdwa vcc, %8, %10 src0_sel:BYTE_1 src1_sel:BYTE_3\n\t" \
        "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t" \
        : "+v"(WORD)                                                                       \
        : "v"(L[0]), "v"(L[1]), "v"(L[2]), "v"(L[3]), "v"(L[4]), "v"(H[0]), "v"(H[1]), "v"(H[2]), "v"(H[3]), "v"(H[4]) \
        : "vcc")

// Requires cols % 4 == 0 and 4-byte aligned frame bases, so that every staged
// dword lies entirely inside or entirely outside the frame.
template <bool IS_U8>
__global__ __launch_bounds__(256) void census5x5_tile_kernel(const uint8_t *__restrict__ in_all, uint32_t *__restrict__ out_all,
                                                             int rows, int cols, int flags) {
    __shared__ uint32_t tile[(CT_H + 4) * CT_LROW];
    const long n = (long)rows * cols;
    const uint8_t *in = in_all + (long)blockIdx.z * n;
    uint32_t *out = out_all + (long)blockIdx.z * n;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    const int tid = threadIdx.x;

    // staging in the unsigned domain; outside the frame reads the signed value 0, i.e. 0x80 unsigned
    for (int i = tid; i < (CT_H + 4) * CT_LROW; i += 256) {
        const int r = i / CT_LROW, k = i - r * CT_LROW;
        const long flat = (long)(y0 - 2 + r) * cols + (x0 - 4 + 4 * k);
        uint32_t v = 0x80808080u;
        if (flat >= 0 && flat < n) {
            v = *reinterpret_cast<const uint32_t *>(in + flat);
            if (!IS_U8) v ^= 0x80808080u;
        }
        tile[i] = v;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int x = x0 + lane * 4;
    if (x >= cols) return;
    const bool generic = (flags & 15) == SVX_CENSUS_GENERIC25;
    const bool zero_rest = (flags & SVX_CENSUS_ZERO_REST) != 0;

    // lo/hi windows of the 8 staged rows this lane needs: columns x-2..x+1 and x+2..x+5
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint32_t *p = tile + (wave * 4 + r) * CT_LROW + lane;
        const uint32_t a = p[0], b = p[1], c = p[2];
        lo[r] = __builtin_amdgcn_alignbyte(b, a, 2);
        hi[r] = __builtin_amdgcn_alignbyte(c, b, 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + wave * 4 + j;
        if (y >= rows) break;
        uint32_t res[4] = {0u, 0u, 0u, 0u};
        if (y >= 2 && y < rows - 2) {
            const uint32_t *L = lo + j, *H = hi + j;
            CENSUS_PX0(res[0], L, H);
            CENSUS_PX1(res[1], L, H);
            CENSUS_PX2(res[2], L, H);
            CENSUS_PX3(res[3], L, H);
            if (generic) {
#pragma unroll
                for (int i = 0; i < 4; ++i) res[i] = census24_to_25(res[i]);
            }
            uint32_t *o = out + (long)y * cols + x;
            if (!generic || (x >= 2 && x + 3 < cols - 2)) {
                *reinterpret_cast<uint4 *>(o) = make_uint4(res[0], res[1], res[2], res[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool interior = (x + i >= 2) && (x + i < cols - 2);
                    if (interior) o[i] = res[i];
                    else if (zero_rest) o[i] = 0u;
                }
            }
        } else if (zero_rest) {
            *reinterpret_cast<uint4 *>(out + (long)y * cols + x) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

// Any width / alignment: one thread per pixel straight from global memory.
template <bool IS_U8>
__global__ __launch_bounds__(256) void census5x5_pixel_kernel(const uint8_t *__restrict__ in_all, uint32_t *__restrict__ out_all,
                                                              int rows, int cols, int flags) {
    const long n = (long)rows * cols;
    const uint8_t *in = in_all + (long)blockIdx.z * n;
    uint32_t *out = out_all + (long)blockIdx.z * n;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const bool generic = (flags & 15) == SVX_CENSUS_GENERIC25;
    const bool zero_rest = (flags & SVX_CENSUS_ZERO_REST) != 0;
    const bool written = (y >= 2 && y < rows - 2) && (!generic || (x >= 2 && x < cols - 2));
    if (!written) {
        if (zero_rest) out[(long)y * cols + x] = 0u;
        return;
    }
    auto px = [&](long flat) -> int {
        if (flat < 0 || flat >= n) return 0;
        return IS_U8 ? (int)(int8_t)(in[flat] ^ 0x80u) : (int)(int8_t)in[flat];
    };
    const int c = px((long)y * cols + x);
    uint32_t word = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            if (dy == 0 && dx == 0) continue;
            word |= (uint32_t)(c > px((long)(y + dy) * cols + (x + dx))) << census_bit24(dy, dx);
        }
    out[(long)y * cols + x] = generic ? census24_to_25(word) : word;
}

template <bool IS_U8>
static int census_dispatch(svx_ctx *ctx, const void *in, uint32_t *out, int rows, int cols, int n_frames, int flags, const char *who) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, in && out, SVX_ERR_INVALID_ARG, "census5x5: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "census5x5: empty image");
    const int layout = flags & 15;
    SVX_REQUIRE(ctx, layout == SVX_CENSUS_SSE24 || layout == SVX_CENSUS_GENERIC25, SVX_ERR_INVALID_ARG, "census5x5: unknown layout flag");
    SVX_REQUIRE(ctx, (flags & ~(15 | SVX_CENSUS_ZERO_REST)) == 0, SVX_ERR_INVALID_ARG, "census5x5: unknown flag bits");
    SVX_REQUIRE(ctx, n_frames <= 65535, SVX_ERR_UNSUPPORTED, "census5x5: more than 65535 frames per call");
    const bool fast = (cols % 4 == 0) && (((uintptr_t)in & 3u) == 0) && (((uintptr_t)out & 15u) == 0);
    if (fast) {
        dim3 grid(svx_cdiv(cols, CT_W), svx_cdiv(rows, CT_H), n_frames);
        census5x5_tile_kernel<IS_U8><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)in, out, rows, cols, flags);
    } else {
        dim3 grid(svx_cdiv(cols, 64), svx_cdiv(rows, 4), n_frames);
        census5x5_pixel_kernel<IS_U8><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)in, out, rows, cols, flags);
    }
    SVX_LAUNCH_CHECK(ctx, who);
    return SVX_OK;
}

SVX_API int svx_census5x5_s8(svx_ctx *ctx, const int8_t *in, uint32_t *out, int rows, int cols, int n_frames, int flags) {
    return census_dispatch<false>(ctx, in, out, rows, cols, n_frames, flags, "census5x5 (s8)");
}
SVX_API int svx_census5x5_u8(svx_ctx *ctx, const uint8_t *in, uint32_t *out, int rows, int cols, int n_frames, int flags) {
    return census_dispatch<true>(ctx, in, out, rows, cols, n_frames, flags, "census5x5 (u8)");
}

// ---------------------------------------------------------------- a14 ----
// Generic N x N census of denseCensusStereo/CensusTransform.cpp:33-67: bit j*m+i set when neighbour >=
// centre, the window-index-versus-image-coordinate skip test included.  One thread per pixel; this
// variant is not on the fast path of the reference either.
template <bool FROM_U8>
__global__ __launch_bounds__(256) void census_nxn_kernel(const uint8_t *__restrict__ in_all, int rows, int cols, int m, int nw,
                                                         uint32_t *__restrict__ out_all) {
    const long n = (long)rows * cols;
    const uint8_t *in = in_all + (long)blockIdx.z * n;
    uint32_t *out = out_all + (long)blockIdx.z * n * nw;
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= cols || v >= rows) return;
    auto px = [&](long o) -> int { return FROM_U8 ? min((int)in[o], 127) : (int)(int8_t)in[o]; };   // u8 -> s8 saturates
    uint32_t *w = out + ((long)v * cols + u) * nw;
    const int h = m / 2;
    if (u < h || v < h || u >= cols - h || v >= rows - h) {
        for (int k = 0; k < nw; ++k) w[k] = 0u;
        return;
    }
    const int ref = px((long)v * cols + u);
    uint32_t acc = 0;
    int word = 0, pass = 0;
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) {
            if (j == v && i == u) { pass = -1; continue; }
            const int bit = j * m + i + pass;
            if ((bit >> 5) != word) { w[word] = acc; acc = 0u; ++word; }
            if (px((long)(v - h + j) * cols + (u - h + i)) >= ref) acc |= 1u << (bit & 31);
        }
    w[word] = acc;
    for (int k = word + 1; k < nw; ++k) w[k] = 0u;
}

template <bool FROM_U8>
static int census_nxn_dispatch(svx_ctx *ctx, const void *in, int rows, int cols, int n_frames, int size, uint32_t *out) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, in && out, SVX_ERR_INVALID_ARG, "svx_census_nxn: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_census_nxn: empty image");
    SVX_REQUIRE(ctx, size >= 1 && size <= 31, SVX_ERR_UNSUPPORTED, "svx_census_nxn: size must be in [1, 31]");
    SVX_REQUIRE(ctx, n_frames <= 65535, SVX_ERR_UNSUPPORTED, "svx_census_nxn: more than 65535 frames per call");
    const int m = (size % 2 == 0) ? size + 1 : size;
    dim3 grid(svx_cdiv(cols, 64), svx_cdiv(rows, 4), n_frames);
    census_nxn_kernel<FROM_U8><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)in, rows, cols, m, (m * m) / 32 + 1, out);
    SVX_LAUNCH_CHECK(ctx, "census_nxn_kernel");
    return SVX_OK;
}

SVX_API int svx_census_nxn_words(int size) {
    if (size < 1) return 0;
    const int m = (size % 2 == 0) ? size + 1 : size;
    return (m * m) / 32 + 1;
}
SVX_API int svx_census_nxn_s8(svx_ctx *ctx, const int8_t *in, int rows, int cols, int n_frames, int size, uint32_t *out) {
    return census_nxn_dispatch<false>(ctx, in, rows, cols, n_frames, size, out);
}
SVX_API int svx_census_nxn_u8(svx_ctx *ctx, const uint8_t *in, int rows, int cols, int n_frames, int size, uint32_t *out) {
    return census_nxn_dispatch<true>(ctx, in, rows, cols, n_frames, size, out);
}
