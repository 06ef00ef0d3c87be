// Plane-prior path for gfx950: a9 (batched full-range epipolar search),
// a10 (cost of the prior), a11 (accept / refine + occupancy grid) and
// a12 (plane fit + prior rasterisation).  All of it is gather-bound integer
// work: one census word per side per evaluated pixel.
#include "svx_common.hpp"

// ---------------------------------------------------------------- a9 -----
// One wavefront per query point; lanes take disparities d = lane, lane+64, ...
// and evaluate the whole A x A aggregate; the winner is the minimum of
// (cost << 32 | d), i.e. the smallest disparity among equal costs, which is
// what a strict '<' scan from d = 0 upwards keeps.
// Full-range search for one point by one wavefront; every lane returns the winning disparity.
__device__ __forceinline__ int epipolar_search(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                               int half, int margin, int x, int y, int max_disp, int lane) {
    const long n = (long)rows * cols;
    unsigned long long best = ~0ull;
    for (int d = lane; d < max_disp; d += 64) {
        const int i = x - d;
        if (i < half + margin) break;
        unsigned cost = 0;
        for (int m = -half; m <= half && y + m < rows; ++m) {
            if (y + m < 0) continue;
            const long ro = (long)(y + m) * cols;
            for (int k = -half; k <= half && i + k < cols; ++k)
                cost += popc32(ld_guard_u32(cl, n, ro + x + k) ^ ld_guard_u32(cr, n, ro + i + k));
        }
        const unsigned long long key = ((unsigned long long)cost << 32) | (unsigned)d;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(best, s, 64);
        best = o < best ? o : best;
    }
    return best == ~0ull ? 0 : (int)(unsigned)(best & 0xffffffffu);
}

// The same search with the window's loops unrolled (A = 2 HALF + 1 known at compile time): the 2 A words of a window row are
// independent reads in flight together, where the general form waits for every pair of words in turn (121 round trips a
// disparity at A = 11), and the left words come from addresses that are the same in every lane.  INSIDE: the point's window
// lies inside the frame at every disparity (all points but those next to the frame's borders): no guard, no clipping.
// Otherwise the clipping rules of the general form as predicates: rows outside the frame are skipped, a right column at or
// beyond cols ends its row, every read is a flat index with zeros outside the frame.  x, y: the same in every lane.
template <int HALF, bool INSIDE>
__device__ __forceinline__ int epipolar_search_unrolled(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                        int margin, int x, int y, int max_disp, int lane) {
    const long n = (long)rows * cols;
    unsigned long long best = ~0ull;
    for (int d0 = 0; d0 < max_disp && x - d0 >= HALF + margin; d0 += 64) {
        const int d = d0 + lane, i = x - d;
        const bool live = d < max_disp && i >= HALF + margin;
        const int ic = live ? i : x - d0;                 // (a lane without a disparity reads the window of the round's first one)
        unsigned cost = 0;
#pragma unroll
        for (int m = -HALF; m <= HALF; ++m) {
            const long ro = (long)(y + m) * cols;
            if (INSIDE) {
                const uint32_t *lrow = cl + ro + (x - HALF), *rrow = cr + ro + (ic - HALF);
#pragma unroll
                for (int k = 0; k <= 2 * HALF; ++k) cost += popc32(lrow[k] ^ rrow[k]);
            } else if (y + m >= 0 && y + m < rows) {
#pragma unroll
                for (int k = -HALF; k <= HALF; ++k) {
                    const unsigned c = popc32(ld_guard_u32(cl, n, ro + x + k) ^ ld_guard_u32(cr, n, ro + ic + k));
                    cost += ic + k < cols ? c : 0u;
                }
            }
        }
        const unsigned long long key = live ? ((unsigned long long)cost << 32) | (unsigned)d : ~0ull;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(best, s, 64);
        best = o < best ? o : best;
    }
    return best == ~0ull ? 0 : (int)(unsigned)(best & 0xffffffffu);
}

// Picks the form: x, y are made wavefront-uniform (one point per wavefront).
__device__ __forceinline__ int epipolar_search_point(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                     int half, int margin, int x, int y, int max_disp, int lane) {
    x = __builtin_amdgcn_readfirstlane(x);
    y = __builtin_amdgcn_readfirstlane(y);
    if (half == 5 && margin >= 0) {
        if (y - 5 >= 0 && y + 5 < rows && x + 5 < cols && x >= 5) return epipolar_search_unrolled<5, true>(cl, cr, rows, cols, margin, x, y, max_disp, lane);
        return epipolar_search_unrolled<5, false>(cl, cr, rows, cols, margin, x, y, max_disp, lane);
    }
    return epipolar_search(cl, cr, rows, cols, half, margin, x, y, max_disp, lane);
}

__device__ __forceinline__ int single_pixel_cost(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                 int x, int y, int d) {
    const long n = (long)rows * cols;
    return popc32(ld_guard_u32(cl, n, (long)y * cols + x) ^ ld_guard_u32(cr, n, (long)y * cols + (x - d)));
}

__global__ __launch_bounds__(256) void epipolar_match_kernel(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                             int half, int margin, const int32_t *__restrict__ pts_xy, int n_pts, int max_disp,
                                                             float *__restrict__ out_disp, int8_t *__restrict__ out_cost) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= n_pts) return;
    const int x = pts_xy[2 * p], y = pts_xy[2 * p + 1];
    const int d = epipolar_search_point(cl, cr, rows, cols, half, margin, x, y, max_disp, lane);
    if (lane == 0) {
        out_disp[p] = (float)d;
        out_cost[p] = (int8_t)single_pixel_cost(cl, cr, rows, cols, x, y, d);
    }
}

// a13: one wavefront per occupancy-grid cell.  Slot 2c holds the re-matched invalid point of cell c,
// slot 2c+1 its confident support; out_valid marks the slots that became support points.
__global__ __launch_bounds__(256) void support_resampling_kernel(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                                 int half, int margin, int n_cells,
                                                                 const int32_t *__restrict__ conf_x, const int32_t *__restrict__ conf_y,
                                                                 const float *__restrict__ conf_d, const int8_t *__restrict__ conf_c,
                                                                 const int32_t *__restrict__ inv_x, const int32_t *__restrict__ inv_y,
                                                                 const int8_t *__restrict__ inv_c, int t_low, int t_high, int max_disp,
                                                                 float *__restrict__ disparities, int32_t *__restrict__ out_xy,
                                                                 float *__restrict__ out_d, uint8_t *__restrict__ out_valid) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= n_cells) return;
    bool inv_ok = false;
    int ix = inv_x[c], iy = inv_y[c], d = 0;
    if ((int)inv_c[c] > t_high) {          // wavefront-uniform
        d = epipolar_search_point(cl, cr, rows, cols, half, margin, ix, iy, max_disp, lane);
        inv_ok = (int)(int8_t)single_pixel_cost(cl, cr, rows, cols, ix, iy, d) < t_low;
    }
    if (lane == 0) {
        out_valid[2 * c] = inv_ok ? 1 : 0;
        out_xy[4 * c] = ix; out_xy[4 * c + 1] = iy; out_d[2 * c] = (float)d;
        if (inv_ok && ix >= 0 && ix < cols && iy >= 0 && iy < rows) disparities[(long)iy * cols + ix] = (float)d;
        const bool conf_ok = (int)conf_c[c] < t_low;
        const int cx = conf_x[c], cy = conf_y[c];
        out_valid[2 * c + 1] = conf_ok ? 1 : 0;
        out_xy[4 * c + 2] = cx; out_xy[4 * c + 3] = cy; out_d[2 * c + 1] = conf_d[c];
        if (conf_ok && cx >= 0 && cx < cols && cy >= 0 && cy < rows) disparities[(long)cy * cols + cx] = conf_d[c];
    }
}

SVX_API int svx_epipolar_match_batch(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right,
                                     int rows, int cols, int census_size, int cost_aggr_window_size,
                                     const int32_t *pts_xy, int n_pts, int max_disp, float *out_disp, int8_t *out_cost) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, census_left && census_right, SVX_ERR_INVALID_ARG, "svx_epipolar_match_batch: null census pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_epipolar_match_batch: empty image");
    SVX_REQUIRE(ctx, census_size >= 1 && cost_aggr_window_size >= 1, SVX_ERR_INVALID_ARG, "svx_epipolar_match_batch: window sizes must be >= 1");
    SVX_REQUIRE(ctx, n_pts >= 0, SVX_ERR_INVALID_ARG, "svx_epipolar_match_batch: negative point count");
    if (n_pts == 0) return SVX_OK;
    SVX_REQUIRE(ctx, pts_xy && out_disp && out_cost, SVX_ERR_INVALID_ARG, "svx_epipolar_match_batch: null point or output pointer");
    epipolar_match_kernel<<<svx_cdiv(n_pts, 4), 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, cost_aggr_window_size / 2,
                                                                      census_size / 2, pts_xy, n_pts, max_disp, out_disp, out_cost);
    SVX_LAUNCH_CHECK(ctx, "epipolar_match_kernel");
    return SVX_OK;
}

SVX_API int svx_support_resampling(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                                   int census_size, int cost_aggr_window_size, int grid_w, int grid_h,
                                   const int32_t *conf_x, const int32_t *conf_y, const float *conf_d, const int8_t *conf_c,
                                   const int32_t *inv_x, const int32_t *inv_y, const int8_t *inv_c,
                                   int t_low, int t_high, int max_disp, float *disparities,
                                   int32_t *out_xy, float *out_d, uint8_t *out_valid) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, census_left && census_right && disparities, SVX_ERR_INVALID_ARG, "svx_support_resampling: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_support_resampling: empty image");
    SVX_REQUIRE(ctx, grid_w > 0 && grid_h > 0 && (long)grid_w * grid_h <= 0x3fffffff, SVX_ERR_INVALID_ARG, "svx_support_resampling: bad grid size");
    SVX_REQUIRE(ctx, conf_x && conf_y && conf_d && conf_c && inv_x && inv_y && inv_c && out_xy && out_d && out_valid,
                SVX_ERR_INVALID_ARG, "svx_support_resampling: null grid or output pointer");
    SVX_REQUIRE(ctx, census_size >= 1 && cost_aggr_window_size >= 1, SVX_ERR_INVALID_ARG, "svx_support_resampling: window sizes must be >= 1");
    const int n_cells = grid_w * grid_h;
    support_resampling_kernel<<<svx_cdiv(n_cells, 4), 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, cost_aggr_window_size / 2,
                                                                            census_size / 2, n_cells, conf_x, conf_y, conf_d, conf_c, inv_x, inv_y,
                                                                            inv_c, t_low, t_high, max_disp, disparities, out_xy, out_d, out_valid);
    SVX_LAUNCH_CHECK(ctx, "support_resampling_kernel");
    return SVX_OK;
}

// --------------------------------------------------------------- a10 -----
__global__ __launch_bounds__(256) void cost_eval_kernel(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                        const int32_t *__restrict__ pts_xy, int n_pts, const float *__restrict__ disp,
                                                        int8_t *__restrict__ costs) {
    const long n = (long)rows * cols;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pts; p += gridDim.x * blockDim.x) {
        const int2 pt = reinterpret_cast<const int2 *>(pts_xy)[p];
        const int x = pt.x, y = pt.y;
        if (x < 0 || x >= cols || y < 0 || y >= rows) continue;
        const long o = (long)y * cols + x;
        const int xr = x - (int)floor((double)disp[o] + 0.5);
        if (xr < 2) continue;
        costs[o] = (int8_t)popc32(cl[o] ^ ld_guard_u32(cr, n, (long)y * cols + xr));
    }
}

SVX_API int svx_cost_eval(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                          const int32_t *pts_xy, int n_pts, const float *disparities, int8_t *costs) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, census_left && census_right && disparities && costs, SVX_ERR_INVALID_ARG, "svx_cost_eval: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_cost_eval: empty image");
    SVX_REQUIRE(ctx, n_pts >= 0, SVX_ERR_INVALID_ARG, "svx_cost_eval: negative point count");
    if (n_pts == 0) return SVX_OK;
    SVX_REQUIRE(ctx, pts_xy != nullptr, SVX_ERR_INVALID_ARG, "svx_cost_eval: null point list");
    unsigned grid = svx_cdiv(n_pts, 256);
    if (grid > 4096) grid = 4096;
    cost_eval_kernel<<<grid, 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, pts_xy, n_pts, disparities, costs);
    SVX_LAUNCH_CHECK(ctx, "cost_eval_kernel");
    return SVX_OK;
}

// Banded search around the plane prior (no counterpart in the reference, which evaluates the prior's own disparity
// only; the band follows libelas' plane-prior matcher, elas/elas.cpp:683-780): candidates d0 - radius .. d0 + radius in
// ascending order, strict '<'; d0 is admitted as costEvaluation admits it, the others when they are disparities of the
// search range whose right pixel is at column 2 or beyond.  The prior takes the winner when that is not d0.
__global__ __launch_bounds__(256) void cost_eval_band_kernel(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                             const int32_t *__restrict__ pts_xy, int n_pts, float *__restrict__ disp,
                                                             int8_t *__restrict__ costs, int radius, int max_disp) {
    const long n = (long)rows * cols;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pts; p += gridDim.x * blockDim.x) {
        const int2 pt = reinterpret_cast<const int2 *>(pts_xy)[p];
        const int x = pt.x, y = pt.y;
        if (x < 0 || x >= cols || y < 0 || y >= rows) continue;
        const long o = (long)y * cols + x;
        const int d0 = (int)floor((double)disp[o] + 0.5);
        const uint32_t l = cl[o];
        int best = 1 << 30, best_d = 0;
        for (int k = -radius; k <= radius; ++k) {
            const long d = (long)d0 + k, xr = (long)x - d;
            if (xr < 2 || (k != 0 && (d < 0 || (max_disp > 0 && d >= max_disp)))) continue;
            const int c = popc32(l ^ ld_guard_u32(cr, n, (long)y * cols + xr));
            if (c < best) { best = c; best_d = (int)d; }
        }
        if (best == 1 << 30) continue;
        costs[o] = (int8_t)best;
        if (best_d != d0) disp[o] = (float)best_d;
    }
}

SVX_API int svx_cost_eval_band(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                               const int32_t *pts_xy, int n_pts, float *disparities, int8_t *costs, int radius, int max_disp) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, census_left && census_right && disparities && costs, SVX_ERR_INVALID_ARG, "svx_cost_eval_band: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_cost_eval_band: empty image");
    SVX_REQUIRE(ctx, n_pts >= 0 && radius >= 0 && radius <= 1024, SVX_ERR_INVALID_ARG, "svx_cost_eval_band: bad point count or radius");
    if (n_pts == 0) return SVX_OK;
    SVX_REQUIRE(ctx, pts_xy != nullptr, SVX_ERR_INVALID_ARG, "svx_cost_eval_band: null point list");
    unsigned grid = svx_cdiv(n_pts, 256);
    if (grid > 4096) grid = 4096;
    cost_eval_band_kernel<<<grid, 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, pts_xy, n_pts, disparities, costs, radius, max_disp);
    SVX_LAUNCH_CHECK(ctx, "cost_eval_band_kernel");
    return SVX_OK;
}

// --------------------------------------------------------------- a11 -----
// Per-point accept step plus per-cell winners.  A sequential scan with strict
// compares keeps, per cell, the lowest cost below tLow (confident support) and
// the highest cost above tHigh (invalid match), earliest point first; that is
// the minimum of the keys below, found with a segmented wavefront scan over
// runs of equal cells and one 64-bit atomic per run.
// EVAL: the point's cost is worked out here, as cost_eval_kernel works it out (a10: the prior's own disparity, the right pixel
// at column 2 or beyond, t_high otherwise), instead of being read from a cost map: svx_high_perf_stereo's form, which needs
// neither the map nor the launch that fills it.
template <bool EVAL>
__global__ __launch_bounds__(256) void refine_points_kernel(const int32_t *__restrict__ pts_xy, int n_pts, const float *__restrict__ disp,
                                                            const int8_t *__restrict__ costs, int rows, int cols, int t_low, int t_high,
                                                            int grid_size, int grid_w, float *__restrict__ final_disp, int8_t *__restrict__ final_costs,
                                                            unsigned long long *__restrict__ conf_key, unsigned long long *__restrict__ inv_key,
                                                            const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long ck = ~0ull, ik = ~0ull;
    int cell = -1;
    if (p < n_pts) {
        const int2 pt = reinterpret_cast<const int2 *>(pts_xy)[p];
        const int u = pt.x, v = pt.y;
        if (u >= 0 && u < cols && v >= 0 && v < rows) {
            const long o = (long)v * cols + u;
            const float d = disp[o];
            int mc;
            if (EVAL) {
                const int xr = u - (int)floor((double)d + 0.5);
                mc = xr < 2 ? t_high : (int)(int8_t)popc32(cl[o] ^ ld_guard_u32(cr, (long)rows * cols, (long)v * cols + xr));
            } else mc = costs[o];
            if (mc < (int)(uint8_t)final_costs[o]) { final_disp[o] = d; final_costs[o] = (int8_t)mc; }
            cell = (v / grid_size) * grid_w + (u / grid_size);
            if (mc < t_low) ck = ((unsigned long long)(unsigned)(mc + 128) << 32) | (unsigned)p;
            if (mc > t_high) ik = ((unsigned long long)(unsigned)(127 - mc) << 32) | (unsigned)p;
        }
    }
    // Points come in raster order, so the lanes of one cell form contiguous runs.  A segmented min-scan
    // (lane l takes lane l-s while it is in the same cell) leaves each run's minimum in the run's last
    // lane, which issues the only atomics of that run.
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int oc = __shfl_up(cell, s, 64);
        const unsigned long long c2 = __shfl_up(ck, s, 64), i2 = __shfl_up(ik, s, 64);
        if (lane >= s && oc == cell) {
            ck = c2 < ck ? c2 : ck;
            ik = i2 < ik ? i2 : ik;
        }
    }
    const int next = __shfl_down(cell, 1, 64);
    const bool tail = lane == 63 || next != cell;
    if (tail && cell >= 0) {
        if (ck != ~0ull) atomicMin(conf_key + cell, ck);
        if (ik != ~0ull) atomicMin(inv_key + cell, ik);
    }
}

// (a winner's cost is in its key: no cost map is read)
__global__ __launch_bounds__(256) void refine_cells_kernel(const int32_t *__restrict__ pts_xy, const float *__restrict__ disp,
                                                           int cols, int t_low, int t_high, int n_cells,
                                                           const unsigned long long *__restrict__ conf_key, const unsigned long long *__restrict__ inv_key,
                                                           int32_t *conf_x, int32_t *conf_y, float *conf_d, int8_t *conf_c,
                                                           int32_t *inv_x, int32_t *inv_y, int8_t *inv_c) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const unsigned long long ck = conf_key[c], ik = inv_key[c];
    if (ck == ~0ull) { conf_x[c] = 0; conf_y[c] = 0; conf_d[c] = 0.0f; conf_c[c] = (int8_t)t_low; }
    else {
        const unsigned p = (unsigned)(ck & 0xffffffffu);
        const int u = pts_xy[2 * p], v = pts_xy[2 * p + 1];
        conf_x[c] = u; conf_y[c] = v; conf_d[c] = disp[(long)v * cols + u]; conf_c[c] = (int8_t)((int)(ck >> 32) - 128);
    }
    if (ik == ~0ull) { inv_x[c] = 0; inv_y[c] = 0; inv_c[c] = (int8_t)t_high; }
    else {
        const unsigned p = (unsigned)(ik & 0xffffffffu);
        const int u = pts_xy[2 * p], v = pts_xy[2 * p + 1];
        inv_x[c] = u; inv_y[c] = v; inv_c[c] = (int8_t)(127 - (int)(ik >> 32));
    }
}

// a11 on a cost map (costs != nullptr), or a10 + a11 in one pass over the points (costs == nullptr: census images given)
static int refine_impl(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, const int32_t *pts_xy, int n_pts,
                       const float *disparities, const int8_t *costs, int rows, int cols, int t_low, int t_high, unsigned occ_grid_size,
                       float *final_disparities, int8_t *final_costs, int32_t *conf_x, int32_t *conf_y, float *conf_d, int8_t *conf_c,
                       int32_t *inv_x, int32_t *inv_y, int8_t *inv_c, const char *who) {
    SVX_REQUIRE(ctx, disparities && final_disparities && final_costs, SVX_ERR_INVALID_ARG, "svx_refine: null image pointer");
    SVX_REQUIRE(ctx, conf_x && conf_y && conf_d && conf_c && inv_x && inv_y && inv_c, SVX_ERR_INVALID_ARG, "svx_refine: null grid pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_refine: empty image");
    SVX_REQUIRE(ctx, occ_grid_size >= 1 && occ_grid_size <= 0x7fffffffu, SVX_ERR_INVALID_ARG, "svx_refine: occ_grid_size must be >= 1");
    SVX_REQUIRE(ctx, n_pts >= 0, SVX_ERR_INVALID_ARG, "svx_refine: negative point count");
    SVX_REQUIRE(ctx, n_pts == 0 || pts_xy, SVX_ERR_INVALID_ARG, "svx_refine: null point list");
    SVX_REQUIRE(ctx, t_low >= -128 && t_low <= 127 && t_high >= -128 && t_high <= 127, SVX_ERR_INVALID_ARG, "svx_refine: thresholds must fit a signed char");
    (void)who;
    const int g = (int)occ_grid_size;
    const int gw = cols / g + 1, gh = rows / g + 1;
    const long n_cells = (long)gw * gh;
    SVX_REQUIRE(ctx, n_cells <= 0x7fffffff, SVX_ERR_UNSUPPORTED, "svx_refine: grid too large");
    void *ws;
    int rc = svx_ws_reserve(ctx, 2 * sizeof(unsigned long long) * (size_t)n_cells, &ws);
    if (rc) return rc;
    unsigned long long *ck = (unsigned long long *)ws, *ik = ck + n_cells;
    SVX_HIP(ctx, hipMemsetAsync(ws, 0xff, 2 * sizeof(unsigned long long) * (size_t)n_cells, ctx->stream));
    if (n_pts > 0) {
        if (costs)
            refine_points_kernel<false><<<svx_cdiv(n_pts, 256), 256, 0, ctx->stream>>>(pts_xy, n_pts, disparities, costs, rows, cols, t_low, t_high, g, gw,
                                                                                      final_disparities, final_costs, ck, ik, nullptr, nullptr);
        else
            refine_points_kernel<true><<<svx_cdiv(n_pts, 256), 256, 0, ctx->stream>>>(pts_xy, n_pts, disparities, nullptr, rows, cols, t_low, t_high, g, gw,
                                                                                     final_disparities, final_costs, ck, ik, census_left, census_right);
        SVX_LAUNCH_CHECK(ctx, "refine_points_kernel");
    }
    refine_cells_kernel<<<svx_cdiv(n_cells, 256), 256, 0, ctx->stream>>>(pts_xy, disparities, cols, t_low, t_high, (int)n_cells, ck, ik,
                                                                        conf_x, conf_y, conf_d, conf_c, inv_x, inv_y, inv_c);
    SVX_LAUNCH_CHECK(ctx, "refine_cells_kernel");
    return SVX_OK;
}

SVX_API int svx_refine(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, const float *disparities, const int8_t *costs,
                       int rows, int cols, int t_low, int t_high, unsigned occ_grid_size,
                       float *final_disparities, int8_t *final_costs,
                       int32_t *conf_x, int32_t *conf_y, float *conf_d, int8_t *conf_c,
                       int32_t *inv_x, int32_t *inv_y, int8_t *inv_c) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, costs != nullptr, SVX_ERR_INVALID_ARG, "svx_refine: null image pointer");
    return refine_impl(ctx, nullptr, nullptr, pts_xy, n_pts, disparities, costs, rows, cols, t_low, t_high, occ_grid_size, final_disparities, final_costs,
                       conf_x, conf_y, conf_d, conf_c, inv_x, inv_y, inv_c, "svx_refine");
}

// svx_cost_eval followed by svx_refine without the cost map between them (what svx_high_perf_stereo runs per iteration).
int svx_cost_eval_refine(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, const int32_t *pts_xy, int n_pts,
                         const float *disparities, int rows, int cols, int t_low, int t_high, unsigned occ_grid_size,
                         float *final_disparities, int8_t *final_costs, int32_t *conf_x, int32_t *conf_y, float *conf_d, int8_t *conf_c,
                         int32_t *inv_x, int32_t *inv_y, int8_t *inv_c) {
    SVX_REQUIRE(ctx, census_left && census_right, SVX_ERR_INVALID_ARG, "svx_cost_eval: null image pointer");
    return refine_impl(ctx, census_left, census_right, pts_xy, n_pts, disparities, nullptr, rows, cols, t_low, t_high, occ_grid_size, final_disparities,
                       final_costs, conf_x, conf_y, conf_d, conf_c, inv_x, inv_y, inv_c, "svx_cost_eval_refine");
}

// --------------------------------------------------------------- a12 -----
#pragma clang fp contract(off)
__device__ __forceinline__ double det3_first_row(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1])
         - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
         + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

__global__ __launch_bounds__(256) void plane_fit_kernel(const int32_t *__restrict__ tri_xy, int n_tri, const float *__restrict__ disp,
                                                        int rows, int cols, float *__restrict__ tri_abc) {
#pragma clang fp contract(off)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tri) return;
    double A[3][3], rhs[3];
    for (int i = 0; i < 3; ++i) {
        const int x = tri_xy[6 * t + 2 * i], y = tri_xy[6 * t + 2 * i + 1];
        A[i][0] = (double)x; A[i][1] = (double)y; A[i][2] = 1.0;
        rhs[i] = (x >= 0 && x < cols && y >= 0 && y < rows) ? (double)disp[(long)y * cols + x] : 0.0;
    }
    float abc[3] = {0.0f, 0.0f, 0.0f};
    const double dA = det3_first_row(A);
    if (dA != 0.0) {
        for (int k = 0; k < 3; ++k) {
            double M[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) M[i][j] = (j == k) ? rhs[i] : A[i][j];
            abc[k] = (float)(det3_first_row(M) / dA);
        }
    }
    tri_abc[3 * t] = abc[0]; tri_abc[3 * t + 1] = abc[1]; tri_abc[3 * t + 2] = abc[2];
}

SVX_API int svx_plane_fit(svx_ctx *ctx, const int32_t *tri_xy, int n_tri, const float *disparities, int rows, int cols, float *tri_abc) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n_tri >= 0, SVX_ERR_INVALID_ARG, "svx_plane_fit: negative triangle count");
    if (n_tri == 0) return SVX_OK;
    SVX_REQUIRE(ctx, tri_xy && disparities && tri_abc, SVX_ERR_INVALID_ARG, "svx_plane_fit: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_plane_fit: empty image");
    plane_fit_kernel<<<svx_cdiv(n_tri, 256), 256, 0, ctx->stream>>>(tri_xy, n_tri, disparities, rows, cols, tri_abc);
    SVX_LAUNCH_CHECK(ctx, "plane_fit_kernel");
    return SVX_OK;
}

// Rasterisation: one wavefront per triangle walks the triangle's bounding box;
// a pixel is inside when the three integer edge functions agree in sign (zero
// counts as inside).  Overlap on shared edges is resolved towards the lowest
// triangle index with an atomicMin on a per-pixel owner map, then a second
// pass evaluates the owner's plane.
__device__ __forceinline__ bool tri_contains(const int32_t *v, int x, int y) {
    const long e0 = (long)(v[2] - v[0]) * (y - v[1]) - (long)(v[3] - v[1]) * (x - v[0]);
    const long e1 = (long)(v[4] - v[2]) * (y - v[3]) - (long)(v[5] - v[3]) * (x - v[2]);
    const long e2 = (long)(v[0] - v[4]) * (y - v[5]) - (long)(v[1] - v[5]) * (x - v[4]);
    return (e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0);
}

__global__ __launch_bounds__(256) void raster_owner_kernel(const int32_t *__restrict__ tri_xy, int n_tri, int rows, int cols, unsigned *__restrict__ owner) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tri) return;
    int32_t v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = tri_xy[6 * t + i];
    const long area2 = (long)(v[2] - v[0]) * (v[5] - v[1]) - (long)(v[3] - v[1]) * (v[4] - v[0]);
    if (area2 == 0) return;   // degenerate triangles own nothing
    const int xmin = max(0, min(v[0], min(v[2], v[4]))), xmax = min(cols - 1, max(v[0], max(v[2], v[4])));
    const int ymin = max(0, min(v[1], min(v[3], v[5]))), ymax = min(rows - 1, max(v[1], max(v[3], v[5])));
    if (xmin > xmax || ymin > ymax) return;
    const int bw = xmax - xmin + 1;
    const long total = (long)bw * (ymax - ymin + 1);
    for (long i = lane; i < total; i += 64) {
        const int y = ymin + (int)(i / bw), x = xmin + (int)(i % bw);
        if (tri_contains(v, x, y)) atomicMin(owner + (long)y * cols + x, (unsigned)t);
    }
}

__global__ __launch_bounds__(256) void raster_eval_kernel(const unsigned *__restrict__ owner, const float *__restrict__ tri_abc, int rows, int cols,
                                                          float *__restrict__ prior) {
#pragma clang fp contract(off)
    const long n = (long)rows * cols;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
        const unsigned t = owner[o];
        if (t == 0xffffffffu) continue;
        const int y = (int)(o / cols), x = (int)(o - (long)y * cols);
        const float a = tri_abc[3 * t], b = tri_abc[3 * t + 1], c = tri_abc[3 * t + 2];
        prior[o] = (float)((double)a * (double)x + (double)b * (double)y + (double)c);
    }
}

SVX_API int svx_plane_prior_raster(svx_ctx *ctx, const int32_t *tri_xy, const float *tri_abc, int n_tri, int rows, int cols, float *prior) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n_tri >= 0, SVX_ERR_INVALID_ARG, "svx_plane_prior_raster: negative triangle count");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_plane_prior_raster: empty image");
    SVX_REQUIRE(ctx, prior != nullptr, SVX_ERR_INVALID_ARG, "svx_plane_prior_raster: null prior map");
    if (n_tri == 0) return SVX_OK;
    SVX_REQUIRE(ctx, tri_xy && tri_abc, SVX_ERR_INVALID_ARG, "svx_plane_prior_raster: null triangle pointer");
    const size_t n = (size_t)rows * cols;
    void *ws;
    rc = svx_ws_reserve(ctx, n * sizeof(unsigned), &ws);
    if (rc) return rc;
    SVX_HIP(ctx, hipMemsetAsync(ws, 0xff, n * sizeof(unsigned), ctx->stream));
    raster_owner_kernel<<<svx_cdiv(n_tri, 4), 256, 0, ctx->stream>>>(tri_xy, n_tri, rows, cols, (unsigned *)ws);
    SVX_LAUNCH_CHECK(ctx, "raster_owner_kernel");
    unsigned grid = svx_cdiv((long)n, 256);
    if (grid > 4096) grid = 4096;
    raster_eval_kernel<<<grid, 256, 0, ctx->stream>>>((const unsigned *)ws, tri_abc, rows, cols, prior);
    SVX_LAUNCH_CHECK(ctx, "raster_eval_kernel");
    return SVX_OK;
}
