// a8: dense winner-takes-all over 5x5-aggregated census Hamming costs (gfx950).
#include "svx_common.hpp"

// ---------------------------------------------------------------------------
// Plain formulation: one thread per pixel, every disparity, 25 popcounts each.
// Kept as the simplest correct statement of the search (any size, any
// max_disp); the tiled kernel below is the fast path.
__global__ __launch_bounds__(256) void dense_wta5x5_plain_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                                 int rows, int cols, int max_disp, float *__restrict__ disp_all,
                                                                 int write_border) {
    const long n = (long)rows * cols;
    const uint32_t *cl = cl_all + (long)blockIdx.z * n;
    const uint32_t *cr = cr_all + (long)blockIdx.z * n;
    float *disp = disp_all + (long)blockIdx.z * n;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    if (x < 2 || x >= cols - 2 || y < 2 || y >= rows - 2) {
        if (write_border) disp[(long)y * cols + x] = 0.0f;
        return;
    }
    uint32_t l[25];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int k = 0; k < 5; ++k) l[m * 5 + k] = cl[(long)(y + m - 2) * cols + (x + k - 2)];
    int best = 0x7fffffff, bestx = x;
    for (int i = x; i >= 5 && i > x - max_disp; --i) {
        int cost = 0;
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int k = 0; k < 5; ++k) cost += popc32(l[m * 5 + k] ^ cr[(long)(y + m - 2) * cols + (i + k - 2)]);
        if (cost < best) { best = cost; bestx = i; }
    }
    disp[(long)y * cols + x] = (float)(x - bestx);
}

static int dense_args_ok(svx_ctx *ctx, const void *a, const void *b, const void *c, int rows, int cols, int n_frames, int max_disp) {
    SVX_REQUIRE(ctx, a && b && c, SVX_ERR_INVALID_ARG, "dense_wta: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "dense_wta: empty image");
    SVX_REQUIRE(ctx, max_disp >= 1, SVX_ERR_INVALID_ARG, "dense_wta: max_disp must be >= 1");
    SVX_REQUIRE(ctx, n_frames <= 65535, SVX_ERR_UNSUPPORTED, "dense_wta: more than 65535 frames per call");
    return SVX_OK;
}

SVX_API int svx_dense_wta5x5(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right,
                             int rows, int cols, int n_frames, int max_disp, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, census_left, census_right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    dim3 grid(svx_cdiv(cols, 64), svx_cdiv(rows, 4), n_frames);
    dense_wta5x5_plain_kernel<<<grid, 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, max_disp, disparity, 0);
    SVX_LAUNCH_CHECK(ctx, "dense_wta5x5_plain_kernel");
    return SVX_OK;
}

SVX_API int svx_dense_census_wta_u8(svx_ctx *ctx, const uint8_t *left, const uint8_t *right,
                                    int rows, int cols, int n_frames, int max_disp, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, left, right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    const size_t n = (size_t)rows * cols * n_frames;
    void *ws;
    rc = svx_ws_reserve(ctx, 2 * n * sizeof(uint32_t), &ws);
    if (rc) return rc;
    uint32_t *cl = (uint32_t *)ws, *cr = cl + n;
    rc = svx_census5x5_u8(ctx, left, cl, rows, cols, n_frames, SVX_CENSUS_SSE24 | SVX_CENSUS_ZERO_REST);
    if (rc) return rc;
    rc = svx_census5x5_u8(ctx, right, cr, rows, cols, n_frames, SVX_CENSUS_SSE24 | SVX_CENSUS_ZERO_REST);
    if (rc) return rc;
    dim3 grid(svx_cdiv(cols, 64), svx_cdiv(rows, 4), n_frames);
    dense_wta5x5_plain_kernel<<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity, 1);
    SVX_LAUNCH_CHECK(ctx, "dense_wta5x5_plain_kernel");
    return SVX_OK;
}
