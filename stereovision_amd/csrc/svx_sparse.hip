// a4 (census window cost) and a5/a6/a7 (sparse feature matching with the
// combined uniqueness + left-right consistency check) for gfx950.
//
// The matcher runs one 64-lane wavefront per left feature: lanes stride over
// the candidate right features (a5) or over the positions of the epipolar
// scan (a6), each lane evaluates whole 5x6 windows with v_xor + v_bcnt, and
// the winner / the rejection is found with wavefront shuffles.  Sorting and
// the row-offset table are O(N) host work on the caller's keypoint arrays.
#include "svx_common.hpp"
#include <algorithm>
#include <cmath>
#include <vector>

// 5 rows x NCOLS columns starting at column x-2, flat-guard addressing.
template <int NCOLS>
__device__ __forceinline__ int window_cost(const uint32_t *__restrict__ ref, const uint32_t *__restrict__ cmp, long n, int cols,
                                           int rx, int ry, int px, int py) {
    int cost = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        const long ro = (long)(ry + dy) * cols + (rx - 2);
        const long po = (long)(py + dy) * cols + (px - 2);
#pragma unroll
        for (int k = 0; k < NCOLS; ++k) cost += popc32(ld_guard_u32(ref, n, ro + k) ^ ld_guard_u32(cmp, n, po + k));
    }
    return cost;
}

// The matcher's form of the same cost: the reference window's 30 words are read once per left feature (they are the same in
// every lane) and kept in registers, and a window that lies wholly inside the frame's flat index range, which is every
// window but those of the last rows, is read without the guard, six consecutive words a row.
struct Window6 { uint32_t w[30]; };
__device__ __forceinline__ bool window6_inside(long n, int cols, int x, int y) {
    return (long)(y - 2) * cols + (x - 2) >= 0 && (long)(y + 2) * cols + (x + 3) < n;
}
__device__ __forceinline__ Window6 window6_load(const uint32_t *__restrict__ img, long n, int cols, int x, int y) {
    Window6 r;
    if (window6_inside(n, cols, x, y)) {
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int k = 0; k < 6; ++k) r.w[6 * dy + k] = img[(long)(y + dy - 2) * cols + (x - 2) + k];
    } else {
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int k = 0; k < 6; ++k) r.w[6 * dy + k] = ld_guard_u32(img, n, (long)(y + dy - 2) * cols + (x - 2) + k);
    }
    return r;
}
__device__ __forceinline__ int window6_cost(const Window6 &ref, const uint32_t *__restrict__ cmp, long n, int cols, int px, int py) {
    int cost = 0;
    if (window6_inside(n, cols, px, py)) {
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const uint32_t *row = cmp + (long)(py + dy - 2) * cols + (px - 2);
#pragma unroll
            for (int k = 0; k < 6; ++k) cost += popc32(ref.w[6 * dy + k] ^ row[k]);
        }
    } else {
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int k = 0; k < 6; ++k) cost += popc32(ref.w[6 * dy + k] ^ ld_guard_u32(cmp, n, (long)(py + dy - 2) * cols + (px - 2) + k));
    }
    return cost;
}

// ---------------------------------------------------------------- a4 -----
__global__ __launch_bounds__(256) void window_cost_kernel(const uint32_t *__restrict__ ref, const uint32_t *__restrict__ cmp, int rows, int cols,
                                                          const int32_t *__restrict__ ref_xy, const int32_t *__restrict__ cmp_xy, int n_pairs,
                                                          int16_t *__restrict__ cost, int mode) {
    const long n = (long)rows * cols;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += gridDim.x * blockDim.x) {
        const int rx = ref_xy[2 * p], ry = ref_xy[2 * p + 1], px = cmp_xy[2 * p], py = cmp_xy[2 * p + 1];
        const int c = mode == SVX_WINDOW_5X6 ? window_cost<6>(ref, cmp, n, cols, rx, ry, px, py)
                                             : window_cost<5>(ref, cmp, n, cols, rx, ry, px, py);
        cost[p] = (int16_t)c;
    }
}

SVX_API int svx_census_window_cost(svx_ctx *ctx, const uint32_t *ref, const uint32_t *cmp, int rows, int cols,
                                   const int32_t *ref_xy, const int32_t *cmp_xy, int n_pairs, int16_t *cost, int mode) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, ref && cmp, SVX_ERR_INVALID_ARG, "svx_census_window_cost: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_census_window_cost: empty image");
    SVX_REQUIRE(ctx, mode == SVX_WINDOW_5X6 || mode == SVX_WINDOW_5X5, SVX_ERR_INVALID_ARG, "svx_census_window_cost: unknown window mode");
    SVX_REQUIRE(ctx, n_pairs >= 0, SVX_ERR_INVALID_ARG, "svx_census_window_cost: negative pair count");
    if (n_pairs == 0) return SVX_OK;
    SVX_REQUIRE(ctx, ref_xy && cmp_xy && cost, SVX_ERR_INVALID_ARG, "svx_census_window_cost: null point or cost pointer");
    unsigned grid = svx_cdiv(n_pairs, 256);
    if (grid > 4096) grid = 4096;
    window_cost_kernel<<<grid, 256, 0, ctx->stream>>>(ref, cmp, rows, cols, ref_xy, cmp_xy, n_pairs, cost, mode);
    SVX_LAUNCH_CHECK(ctx, "window_cost_kernel");
    return SVX_OK;
}

// ----------------------------------------------------------- a5 / a6 -----
struct SparseLeft {     // one per left feature, (y,x)-sorted
    float x, y;         // rectified == image coordinates (rect == NULL)
    int xi, yi;         // int(x+0.5), int(y+0.5) evaluated in double on the host
    int r_start, r_end; // candidate range in the sorted right list
    int state;          // 0 = match normally, 1 = skipped by the top rule, 2 = invalidated by the bottom rule
};
struct SparseRight {
    float x, y;
    int xi, yi;
};

__global__ __launch_bounds__(256) void sparse_calc_costs_kernel(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                                const SparseLeft *__restrict__ L, int nl, const SparseRight *__restrict__ R,
                                                                int max_disparity, float y_tol, int *__restrict__ m_first, int *__restrict__ m_cost) {
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (l >= nl) return;
    const SparseLeft f = L[l];
    if (f.state != 0) {   // top rows keep the value-initialised pair (0,0); bottom rows are invalidated
        if (lane == 0) { m_first[l] = f.state == 1 ? 0 : -1; m_cost[l] = 0; }
        return;
    }
    const long n = (long)rows * cols;
    const Window6 ref = window6_load(cl, n, cols, __builtin_amdgcn_readfirstlane(f.xi), __builtin_amdgcn_readfirstlane(f.yi));
    unsigned long long best = ~0ull;   // (cost << 32) | r ; strict '<' in ascending r == minimum of this key
    for (int r = f.r_start + lane; r < f.r_end; r += 64) {
        const SparseRight g = R[r];
        const bool ok = g.x <= f.x && g.x >= f.x - (float)max_disparity && fabsf(f.y - g.y) <= y_tol &&
                        g.x >= 2.0f && g.x < (float)(cols - 2);
        if (ok) {
            const int c = (int)(short)window6_cost(ref, cr, n, cols, g.xi, g.yi);
            const unsigned long long key = ((unsigned long long)(unsigned)(c + 32768) << 32) | (unsigned)r;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(best, s, 64);
        best = o < best ? o : best;
    }
    if (lane == 0) {
        if (best == ~0ull) { m_first[l] = -1; m_cost[l] = 32767; }
        else { m_first[l] = (int)(unsigned)(best & 0xffffffffu); m_cost[l] = (int)(best >> 32) - 32768; }
    }
}

__global__ __launch_bounds__(256) void sparse_consistency_kernel(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                                 const SparseLeft *__restrict__ L, int nl, const SparseRight *__restrict__ R,
                                                                 int max_disparity, float uniqueness, int lr_step, int *__restrict__ m_first,
                                                                 const int *__restrict__ m_cost) {
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (l >= nl) return;
    const SparseLeft f = L[l];
    if (f.state != 0) return;
    const int r = m_first[l];
    if (r == -1) return;
    const SparseRight g = R[r];
    const long n = (long)rows * cols;
    // identity epiline-start table stored as short; column == cols reads the next row's entry 0
    int col = min(cols, max(0, g.xi));
    const float left_start = (float)(short)(col == cols ? 0 : col);
    const int start_x = max(2, (int)left_start);
    const int end_x = min(cols - 2, (int)(left_start + (float)max_disparity));
    const int y = f.yi;
    if (y < 2 || y >= rows - 2) return;
    const float limit = __fdiv_rn((float)(short)m_cost[l], uniqueness);
    const Window6 ref = window6_load(cr, n, cols, __builtin_amdgcn_readfirstlane(g.xi), __builtin_amdgcn_readfirstlane(g.yi));
    bool reject = false;
    for (int x = start_x + lane * lr_step; x < end_x; x += 64 * lr_step) {
        const int c = (int)(short)window6_cost(ref, cl, n, cols, x, y);
        if ((float)c < limit && abs(x - f.xi) > lr_step) reject = true;
    }
    if (__any(reject) && lane == 0) m_first[l] = -1;
}

// The records of svx_sparse_match's host loop built on the device, for key points that are already there in raster
// order (what svx_exfast_detect writes: sorted by (y, x), every position once; integer coordinates, so int(x + 0.5) = x).
// Row offsets of the right list as sparsestereo-inl.h:145-169 fills them: offsets(j) = first right feature whose row is
// >= j for j up to the last row below rows - 1 that holds a feature, and n_right beyond it.  The state rule (top rows
// skipped, everything from the first feature in the bottom rows on invalidated) depends on the order only through the
// row, because the list is sorted.
// row_first (may be null): row_first[j * row_pitch] = index of the first right feature in row j or below, from the bucket index.
__global__ void sparse_prepare_kernel(const int32_t *__restrict__ kl, int nl, const int32_t *__restrict__ kr, int nr, int rows, float y_tol,
                                      const int *__restrict__ row_first, int row_pitch, SparseLeft *__restrict__ L, SparseRight *__restrict__ R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nr) {
        SparseRight g;
        g.xi = kr[2 * i]; g.yi = kr[2 * i + 1]; g.x = (float)g.xi; g.y = (float)g.yi;
        R[i] = g;
    }
    if (i >= nl) return;
    auto first_row_at_least = [&](int j) {            // lower bound over the right rows
        if (row_first) return row_first[j * row_pitch];
        int lo = 0, hi = nr;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (kr[2 * mid + 1] < j) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int lim = first_row_at_least(rows - 1);
    const int last_row = lim > 0 ? kr[2 * (lim - 1) + 1] : -1;
    auto offsets = [&](int j) { return j <= last_row ? first_row_at_least(j) : nr; };
    SparseLeft f;
    f.xi = kl[2 * i]; f.yi = kl[2 * i + 1]; f.x = (float)f.xi; f.y = (float)f.yi;
    const int ly = f.yi;
    auto clampi = [rows](int v) { return min(rows - 1, max(0, v)); };
    f.r_start = offsets(clampi((int)((float)ly - y_tol)));
    f.r_end = offsets(clampi((int)((float)ly + y_tol) + 2));
    f.state = ly >= max(2, rows - 3) ? 2 : (ly < 2 ? 1 : 0);
    L[i] = f;
}

// ---- candidate index for raster-ordered integer key points ----
// The right list of svx_sparse_match_lists is sorted by (y, x) and holds integer positions, so the candidates of a left feature
// that pass the tests of sparse_calc_costs_kernel (|dy| <= yTolerance, x_l - maxDisparity <= x_r <= x_l, 2 <= x_r < cols - 2)
// are one contiguous run per row.  bucket_first[y * (nb + 1) + b] = index of the first right feature at or behind (row y,
// column 64 b): a left feature then visits, in each of its rows, only the buckets its column range touches (about a tenth of
// the row at 70 disparities on a KITTI frame) instead of every feature of the rows [r_start, r_end).  Any superset of the
// passing candidates gives the same minimum, so the result is that of the plain kernel.
constexpr int SP_BUCKET_SHIFT = 6;
constexpr int SP_MAX_TOL_ROWS = 3;          // rows either side of the feature's own the indexed kernel walks (yTolerance < 4)

__global__ void sparse_bucket_index_kernel(const int32_t *__restrict__ kr, int nr, int rows, int nb, int *__restrict__ bucket_first) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * (nb + 1)) return;
    const int y = e / (nb + 1), x = (e % (nb + 1)) << SP_BUCKET_SHIFT;
    int lo = 0, hi = nr;                       // first feature with (row, column) >= (y, x)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int my = kr[2 * mid + 1], mx = kr[2 * mid];
        if (my < y || (my == y && mx < x)) lo = mid + 1; else hi = mid;
    }
    bucket_first[e] = lo;
}

// G lanes serve one left feature: a run of candidates holds a handful of features (about four a row at 70 disparities on a KITTI
// frame), so a whole wavefront per feature leaves most lanes idle; with 16 lanes a wavefront takes four features at once (their
// reference windows then sit in vector registers instead of scalar ones).
template <int G>
__global__ __launch_bounds__(256) void sparse_calc_costs_indexed_kernel(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int rows, int cols,
                                                                        const SparseLeft *__restrict__ L, int nl, const SparseRight *__restrict__ R,
                                                                        const int *__restrict__ bucket_first, int nb, int tol_rows,
                                                                        int max_disparity, float y_tol, int *__restrict__ m_first, int *__restrict__ m_cost) {
    const int l = blockIdx.x * (256 / G) + (threadIdx.x / G);
    const int lane = threadIdx.x & (G - 1);
    if (l >= nl) return;                                   // (a whole group of lanes: the exchanges below stay inside a group)
    const SparseLeft f = L[l];
    if (f.state != 0) {
        if (lane == 0) { m_first[l] = f.state == 1 ? 0 : -1; m_cost[l] = 0; }
        return;
    }
    const long n = (long)rows * cols;
    const Window6 ref = G == 64 ? window6_load(cl, n, cols, __builtin_amdgcn_readfirstlane(f.xi), __builtin_amdgcn_readfirstlane(f.yi))
                                : window6_load(cl, n, cols, f.xi, f.yi);
    const int x_lo = max(f.xi - max_disparity, 2), x_hi = min(f.xi, cols - 3);
    unsigned long long best = ~0ull;
    if (x_lo <= x_hi) {
        const int b0 = x_lo >> SP_BUCKET_SHIFT, b1 = x_hi >> SP_BUCKET_SHIFT;
        for (int y = max(f.yi - tol_rows, 0); y <= min(f.yi + tol_rows, rows - 1); ++y) {
            const int s = max(bucket_first[y * (nb + 1) + b0], f.r_start), e = min(bucket_first[y * (nb + 1) + b1 + 1], f.r_end);
            for (int r = s + lane; r < e; r += G) {
                const SparseRight g = R[r];
                const bool ok = g.x <= f.x && g.x >= f.x - (float)max_disparity && fabsf(f.y - g.y) <= y_tol &&
                                g.x >= 2.0f && g.x < (float)(cols - 2);
                if (ok) {
                    const int c = (int)(short)window6_cost(ref, cr, n, cols, g.xi, g.yi);
                    const unsigned long long key = ((unsigned long long)(unsigned)(c + 32768) << 32) | (unsigned)r;
                    best = key < best ? key : best;
                }
            }
        }
    }
#pragma unroll
    for (int s = G / 2; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(best, s, 64);
        best = o < best ? o : best;
    }
    if (lane == 0) {
        if (best == ~0ull) { m_first[l] = -1; m_cost[l] = 32767; }
        else { m_first[l] = (int)(unsigned)(best & 0xffffffffu); m_cost[l] = (int)(best >> 32) - 32768; }
    }
}

// svx_sparse_match for device-resident key-point lists in raster order: first_dev[i] = index of left feature i's match
// in the right list or -1, cost_dev[i] its cost (as svx_sparse_match's records, in the order of the left list).
int svx_sparse_match_lists(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, const int32_t *left_xy,
                           int n_left, const int32_t *right_xy, int n_right, int max_disparity, float y_tolerance, float uniqueness,
                           int left_right_step, int *first_dev, int *cost_dev) {
    SVX_REQUIRE(ctx, cols <= 32767, SVX_ERR_UNSUPPORTED, "svx_sparse_match: cols above 32767 (epiline table is 16-bit)");
    SVX_REQUIRE(ctx, left_right_step >= 1 && uniqueness > 0.0f, SVX_ERR_INVALID_ARG, "svx_sparse_match: bad step or uniqueness");
    if (n_left == 0 || n_right == 0) return SVX_OK;
    const size_t lbytes = sizeof(SparseLeft) * (size_t)n_left, rbytes = sizeof(SparseRight) * (size_t)n_right;
    const int nb = ((cols - 1) >> SP_BUCKET_SHIFT) + 1;
    const size_t ro_off = (lbytes + 255) & ~(size_t)255, bf_off = (ro_off + rbytes + 255) & ~(size_t)255;
    const size_t total = bf_off + sizeof(int) * (size_t)rows * (nb + 1);
    void *ws;
    int rc = svx_ws_reserve(ctx, total, &ws);
    if (rc) return rc;
    SparseLeft *dl = (SparseLeft *)ws;
    SparseRight *dr = (SparseRight *)((char *)ws + ro_off);
    int *bucket_first = (int *)((char *)ws + bf_off);
    const unsigned grid = svx_cdiv(n_left, 4), prep_grid = svx_cdiv(std::max(n_left, n_right), 256);
    // (a tolerance of four rows or more, or a disparity range wider than the frame, keeps the kernel that walks whole rows)
    if (y_tolerance >= 0.0f && y_tolerance < (float)(SP_MAX_TOL_ROWS + 1) && max_disparity >= 0 && max_disparity < cols) {
        sparse_bucket_index_kernel<<<svx_cdiv((long)rows * (nb + 1), 256), 256, 0, ctx->stream>>>(right_xy, n_right, rows, nb, bucket_first);
        SVX_LAUNCH_CHECK(ctx, "sparse_bucket_index_kernel");
        // (the index's first column is the row-offset table the records are built from)
        sparse_prepare_kernel<<<prep_grid, 256, 0, ctx->stream>>>(left_xy, n_left, right_xy, n_right, rows, y_tolerance, bucket_first, nb + 1, dl, dr);
        SVX_LAUNCH_CHECK(ctx, "sparse_prepare_kernel");
        sparse_calc_costs_indexed_kernel<16><<<svx_cdiv(n_left, 16), 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, dl, n_left, dr, bucket_first, nb,
                                                                                           (int)y_tolerance, max_disparity, y_tolerance, first_dev, cost_dev);
        SVX_LAUNCH_CHECK(ctx, "sparse_calc_costs_indexed_kernel");
    } else {
        sparse_prepare_kernel<<<prep_grid, 256, 0, ctx->stream>>>(left_xy, n_left, right_xy, n_right, rows, y_tolerance, nullptr, 0, dl, dr);
        SVX_LAUNCH_CHECK(ctx, "sparse_prepare_kernel");
        sparse_calc_costs_kernel<<<grid, 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, dl, n_left, dr, max_disparity, y_tolerance, first_dev, cost_dev);
        SVX_LAUNCH_CHECK(ctx, "sparse_calc_costs_kernel");
    }
    sparse_consistency_kernel<<<grid, 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, dl, n_left, dr, max_disparity, uniqueness,
                                                            left_right_step, first_dev, cost_dev);
    SVX_LAUNCH_CHECK(ctx, "sparse_consistency_kernel");
    return SVX_OK;
}

SVX_API int svx_sparse_match(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                             const float *left_xy_host, int n_left, const float *right_xy_host, int n_right,
                             int max_disparity, float y_tolerance, float uniqueness, int left_right_step,
                             int32_t *out_left_host, int32_t *out_right_host, int16_t *out_cost_host, int *n_out) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n_out != nullptr, SVX_ERR_INVALID_ARG, "svx_sparse_match: null n_out");
    *n_out = 0;
    SVX_REQUIRE(ctx, census_left && census_right, SVX_ERR_INVALID_ARG, "svx_sparse_match: null census pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_sparse_match: empty image");
    SVX_REQUIRE(ctx, n_left >= 0 && n_right >= 0, SVX_ERR_INVALID_ARG, "svx_sparse_match: negative feature count");
    SVX_REQUIRE(ctx, cols <= 32767, SVX_ERR_UNSUPPORTED, "svx_sparse_match: cols above 32767 (epiline table is 16-bit)");
    SVX_REQUIRE(ctx, left_right_step >= 1, SVX_ERR_INVALID_ARG, "svx_sparse_match: left_right_step must be >= 1");
    SVX_REQUIRE(ctx, uniqueness > 0.0f, SVX_ERR_INVALID_ARG, "svx_sparse_match: uniqueness must be > 0");
    if (n_left == 0 || n_right == 0) return SVX_OK;   // nothing to match (sparsestereo-inl.h:44-45)
    SVX_REQUIRE(ctx, left_xy_host && right_xy_host && out_left_host && out_right_host && out_cost_host,
                SVX_ERR_INVALID_ARG, "svx_sparse_match: null keypoint or output pointer");

    // Sort both lists by (y, x, original index).
    std::vector<int> lo(n_left), ro(n_right);
    for (int i = 0; i < n_left; ++i) lo[i] = i;
    for (int i = 0; i < n_right; ++i) ro[i] = i;
    auto by_yx = [](const float *xy) {
        return [xy](int a, int b) {
            const float ay = xy[2 * a + 1], by = xy[2 * b + 1];
            if (ay != by) return ay < by;
            const float ax = xy[2 * a], bx = xy[2 * b];
            if (ax != bx) return ax < bx;
            return a < b;
        };
    };
    // detector output is already in raster order: one linear pass then replaces the sort
    if (!std::is_sorted(lo.begin(), lo.end(), by_yx(left_xy_host))) std::sort(lo.begin(), lo.end(), by_yx(left_xy_host));
    if (!std::is_sorted(ro.begin(), ro.end(), by_yx(right_xy_host))) std::sort(ro.begin(), ro.end(), by_yx(right_xy_host));

    // Row offsets into the sorted right list.
    std::vector<unsigned> offsets(rows);
    {
        int last_row = -1;
        for (int i = 0; i < n_right; ++i) {
            const int y = (int)(right_xy_host[2 * ro[i] + 1] + 0.5);
            if (y >= rows - 1) break;
            if (y < 0) continue;
            if (y != last_row) {
                for (int j = last_row + 1; j <= y; ++j) offsets[j] = (unsigned)i;
                last_row = y;
            }
        }
        for (int j = last_row + 1; j < rows; ++j) offsets[j] = (unsigned)n_right;
    }

    const size_t lbytes = sizeof(SparseLeft) * (size_t)n_left, rbytes = sizeof(SparseRight) * (size_t)n_right;
    const size_t mbytes = sizeof(int) * (size_t)n_left;
    const size_t lo_off = 0, ro_off = (lbytes + 255) & ~(size_t)255, mf_off = (ro_off + rbytes + 255) & ~(size_t)255;
    const size_t mc_off = (mf_off + mbytes + 255) & ~(size_t)255, total = mc_off + mbytes;
    void *ws, *pin;
    rc = svx_ws_reserve(ctx, total, &ws);
    if (rc) return rc;
    rc = svx_pinned_reserve(ctx, total, &pin);
    if (rc) return rc;
    SparseLeft *hl = (SparseLeft *)((char *)pin + lo_off);
    SparseRight *hr = (SparseRight *)((char *)pin + ro_off);
    bool bottom = false;
    for (int i = 0; i < n_left; ++i) {
        SparseLeft &f = hl[i];
        f.x = left_xy_host[2 * lo[i]];
        f.y = left_xy_host[2 * lo[i] + 1];
        f.xi = (int)(f.x + 0.5);
        f.yi = (int)(f.y + 0.5);
        const int ly = f.yi;
        auto clampi = [rows](int v) { return std::min(rows - 1, std::max(0, v)); };
        f.r_start = (int)offsets[clampi((int)(ly - y_tolerance))];
        f.r_end = (int)offsets[clampi((int)(ly + y_tolerance) + 2)];
        if (bottom) f.state = 2;
        else if (ly < 2) f.state = 1;
        else if (ly >= rows - 3) { f.state = 2; bottom = true; }
        else f.state = 0;
    }
    for (int i = 0; i < n_right; ++i) {
        SparseRight &g = hr[i];
        g.x = right_xy_host[2 * ro[i]];
        g.y = right_xy_host[2 * ro[i] + 1];
        g.xi = (int)(g.x + 0.5);
        g.yi = (int)(g.y + 0.5);
    }
    char *dws = (char *)ws;
    SVX_HIP(ctx, hipMemcpyAsync(dws + lo_off, hl, lbytes, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(dws + ro_off, hr, rbytes, hipMemcpyHostToDevice, ctx->stream));
    const SparseLeft *dl = (const SparseLeft *)(dws + lo_off);
    const SparseRight *dr = (const SparseRight *)(dws + ro_off);
    int *d_first = (int *)(dws + mf_off), *d_cost = (int *)(dws + mc_off);
    const unsigned grid = svx_cdiv(n_left, 4);
    sparse_calc_costs_kernel<<<grid, 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, dl, n_left, dr, max_disparity,
                                                           y_tolerance, d_first, d_cost);
    SVX_LAUNCH_CHECK(ctx, "sparse_calc_costs_kernel");
    sparse_consistency_kernel<<<grid, 256, 0, ctx->stream>>>(census_left, census_right, rows, cols, dl, n_left, dr, max_disparity,
                                                            uniqueness, left_right_step, d_first, d_cost);
    SVX_LAUNCH_CHECK(ctx, "sparse_consistency_kernel");
    int *h_first = (int *)((char *)pin + mf_off), *h_cost = (int *)((char *)pin + mc_off);
    SVX_HIP(ctx, hipMemcpyAsync(h_first, d_first, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipMemcpyAsync(h_cost, d_cost, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_left; ++i) {
        out_left_host[i] = lo[i];
        out_right_host[i] = h_first[i] == -1 ? -1 : ro[h_first[i]];
        out_cost_host[i] = h_first[i] == -1 ? (int16_t)0 : (int16_t)h_cost[i];
    }
    *n_out = n_left;
    return SVX_OK;
}
