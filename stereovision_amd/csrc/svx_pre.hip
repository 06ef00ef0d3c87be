// Steps either side of the census path (gfx950): the pre-processing that feeds it (histogram
// equalisation, Laplacian magnitude, the raster-ordered high-gradient point list) and a raster-order
// stream compaction shared with the feature detector.  All HBM-bound byte work.
#include "svx_common.hpp"
#include "svx_compact.hpp"

// ------------------------------------------------------------ equalizeHist
__global__ __launch_bounds__(256) void hist256_kernel(const uint8_t *__restrict__ src_all, long n, unsigned *__restrict__ hist_all) {
    __shared__ unsigned h[256];
    const uint8_t *src = src_all + (long)blockIdx.y * n;
    unsigned *hist = hist_all + blockIdx.y * 256;
    h[threadIdx.x] = 0;
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) atomicAdd(&h[src[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// One block per frame builds the look-up table exactly as the sequential definition does.
__global__ __launch_bounds__(256) void hist_lut_kernel(const unsigned *__restrict__ hist_all, long n, uint8_t *__restrict__ lut_all) {
    __shared__ unsigned cdf[256];
    __shared__ int first;
    const unsigned *hist = hist_all + blockIdx.x * 256;
    uint8_t *lut = lut_all + blockIdx.x * 256;
    const int t = threadIdx.x;
    cdf[t] = hist[t];
    if (t == 0) first = 256;
    __syncthreads();
    if (cdf[t]) atomicMin(&first, t);
    __syncthreads();
    const int i0 = first;
    // inclusive sum of hist[i0+1 .. t]
    unsigned v = t > i0 ? cdf[t] : 0u;
    __syncthreads();
    cdf[t] = v;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        const unsigned add = t >= s ? cdf[t - s] : 0u;
        __syncthreads();
        cdf[t] += add;
        __syncthreads();
    }
    if (i0 >= 256) { lut[t] = (uint8_t)t; return; }
    if ((long)hist[i0] == n) { lut[t] = (uint8_t)i0; return; }      // a constant image stays constant
    const float scale = 255.f / (float)(n - (long)hist[i0]);
    int r = t > i0 ? (int)rintf((float)cdf[t] * scale) : 0;
    lut[t] = (uint8_t)min(max(r, 0), 255);
}

__global__ __launch_bounds__(256) void apply_lut_kernel(const uint8_t *__restrict__ src_all, uint8_t *__restrict__ dst_all, long n,
                                                        const uint8_t *__restrict__ lut_all) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = lut_all[blockIdx.y * 256 + threadIdx.x];
    __syncthreads();
    const uint8_t *src = src_all + (long)blockIdx.y * n;
    uint8_t *dst = dst_all + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = lut[src[i]];
}

SVX_API int svx_equalize_hist(svx_ctx *ctx, const uint8_t *src, uint8_t *dst, int rows, int cols, int n_frames) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, src && dst, SVX_ERR_INVALID_ARG, "svx_equalize_hist: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_equalize_hist: empty image");
    SVX_REQUIRE(ctx, n_frames <= 65535, SVX_ERR_UNSUPPORTED, "svx_equalize_hist: more than 65535 frames per call");
    const long n = (long)rows * cols;
    void *ws;
    rc = svx_ws_reserve(ctx, (size_t)n_frames * (256 * sizeof(unsigned) + 256), &ws);
    if (rc) return rc;
    unsigned *hist = (unsigned *)ws;
    uint8_t *lut = (uint8_t *)(hist + (size_t)n_frames * 256);
    SVX_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)n_frames * 256 * sizeof(unsigned), ctx->stream));
    unsigned gx = svx_cdiv(n, 256 * 16);
    if (gx > 256) gx = 256;
    hist256_kernel<<<dim3(gx, n_frames), 256, 0, ctx->stream>>>(src, n, hist);
    SVX_LAUNCH_CHECK(ctx, "hist256_kernel");
    hist_lut_kernel<<<n_frames, 256, 0, ctx->stream>>>(hist, n, lut);
    SVX_LAUNCH_CHECK(ctx, "hist_lut_kernel");
    apply_lut_kernel<<<dim3(gx, n_frames), 256, 0, ctx->stream>>>(src, dst, n, lut);
    SVX_LAUNCH_CHECK(ctx, "apply_lut_kernel");
    return SVX_OK;
}

// ------------------------------------------------- Laplacian + convertScaleAbs
__device__ __forceinline__ int refl101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__global__ __launch_bounds__(256) void laplacian_abs_kernel(const uint8_t *__restrict__ src_all, uint8_t *__restrict__ dst_all, int rows, int cols,
                                                            int ksize, double scale, double delta) {
    const long n = (long)rows * cols;
    const uint8_t *src = src_all + (long)blockIdx.z * n;
    uint8_t *dst = dst_all + (long)blockIdx.z * n;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const int xm = refl101(x - 1, cols), xp = refl101(x + 1, cols), ym = refl101(y - 1, rows), yp = refl101(y + 1, rows);
    auto p = [&](int yy, int xx) -> int { return (int)src[(long)yy * cols + xx]; };
    int sum;
    if (ksize == 1) sum = p(ym, x) + p(y, xm) - 4 * p(y, x) + p(y, xp) + p(yp, x);
    else sum = 2 * (p(ym, xm) + p(ym, xp) + p(yp, xm) + p(yp, xp)) - 8 * p(y, x);
    double v = rint((double)sum * scale + delta);
    v = fmin(fmax(v, -32768.0), 32767.0);
    const int a = (int)fabs(v);
    dst[(long)y * cols + x] = (uint8_t)min(a, 255);
}

SVX_API int svx_laplacian_abs(svx_ctx *ctx, const uint8_t *src, uint8_t *abs_grad, int rows, int cols, int n_frames,
                              int ksize, double scale, double delta) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, src && abs_grad, SVX_ERR_INVALID_ARG, "svx_laplacian_abs: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_laplacian_abs: empty image");
    SVX_REQUIRE(ctx, ksize == 1 || ksize == 3, SVX_ERR_UNSUPPORTED, "svx_laplacian_abs: ksize must be 1 or 3");
    SVX_REQUIRE(ctx, n_frames <= 65535, SVX_ERR_UNSUPPORTED, "svx_laplacian_abs: more than 65535 frames per call");
    laplacian_abs_kernel<<<dim3(svx_cdiv(cols, 64), svx_cdiv(rows, 4), n_frames), 256, 0, ctx->stream>>>(src, abs_grad, rows, cols, ksize, scale, delta);
    SVX_LAUNCH_CHECK(ctx, "laplacian_abs_kernel");
    return SVX_OK;
}

// ------------------------------------------------- high-gradient point list
struct HighGradPred {
    const uint8_t *img;
    int rows, cols, thr;
    __device__ bool operator()(int x, int y, int frame) const {
        return y >= 2 && y < rows - 2 && x >= 2 && x < cols - 2 && (int)img[((long)frame * rows + y) * cols + x] > thr;
    }
};
struct HighGradEmit {
    int32_t *pts;
    __device__ void operator()(unsigned slot, int x, int y, int) const {
        *reinterpret_cast<int2 *>(pts + 2 * (size_t)slot) = make_int2(x, y);
    }
};

SVX_API int svx_high_gradient_points(svx_ctx *ctx, const uint8_t *abs_grad, int rows, int cols, int n_frames, int threshold,
                                     int32_t *pts_xy, int capacity, int *frame_offsets_host) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, abs_grad && frame_offsets_host, SVX_ERR_INVALID_ARG, "svx_high_gradient_points: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_high_gradient_points: empty image");
    SVX_REQUIRE(ctx, capacity >= 0 && (capacity == 0 || pts_xy), SVX_ERR_INVALID_ARG, "svx_high_gradient_points: bad output buffer");
    HighGradPred pred{abs_grad, rows, cols, threshold};
    HighGradEmit emit{pts_xy};
    void *ws;
    rc = svx_ws_reserve(ctx, svx_compact_bytes(rows, n_frames), &ws);
    if (rc) return rc;
    return svx_compact_raster(ctx, ws, rows, cols, n_frames, pred, emit, capacity, frame_offsets_host, "high_gradient_points");
}

// ------------------------------------------------- crop, blur, row alteration
__global__ __launch_bounds__(256) void crop_kernel(const uint8_t *__restrict__ src, int rows, int cols, int r2, int c2, uint8_t *__restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < c2) dst[((long)blockIdx.z * r2 + y) * c2 + x] = src[((long)blockIdx.z * rows + y) * cols + x];
}

SVX_API int svx_crop16(svx_ctx *ctx, const uint8_t *src, int rows, int cols, int n_frames, uint8_t *dst) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, src && dst, SVX_ERR_INVALID_ARG, "svx_crop16: null image pointer");
    const int r2 = 16 * (rows / 16), c2 = 16 * (cols / 16);
    SVX_REQUIRE(ctx, r2 > 0 && c2 > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_crop16: nothing left after cropping");
    SVX_REQUIRE(ctx, n_frames <= 65535 && r2 <= 65535, SVX_ERR_UNSUPPORTED, "svx_crop16: more than 65535 frames or rows per call");
    crop_kernel<<<dim3(svx_cdiv(c2, 256), r2, n_frames), 256, 0, ctx->stream>>>(src, rows, cols, r2, c2, dst);
    SVX_LAUNCH_CHECK(ctx, "crop_kernel");
    return SVX_OK;
}

template <int K>
__global__ __launch_bounds__(256) void gaussian_blur_kernel(const uint8_t *__restrict__ src_all, uint8_t *__restrict__ dst_all, int rows, int cols) {
    constexpr int H = K / 2, NORM = K == 1 ? 1 : (K == 3 ? 16 : 256);
    constexpr int w[5] = {1, K == 3 ? 2 : 4, K == 3 ? 1 : 6, 4, 1};     // binomial row of order K-1
    const long n = (long)rows * cols;
    const uint8_t *src = src_all + (long)blockIdx.z * n;
    uint8_t *dst = dst_all + (long)blockIdx.z * n;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    int sum = 0;
#pragma unroll
    for (int dy = -H; dy <= H; ++dy) {
        const uint8_t *row = src + (long)refl101(y + dy, rows) * cols;
        int rs = 0;
#pragma unroll
        for (int dx = -H; dx <= H; ++dx) rs += w[dx + H] * (int)row[refl101(x + dx, cols)];
        sum += w[dy + H] * rs;
    }
    dst[(long)y * cols + x] = (uint8_t)((sum + NORM / 2) / NORM);
}

SVX_API int svx_gaussian_blur(svx_ctx *ctx, const uint8_t *src, uint8_t *dst, int rows, int cols, int n_frames, int ksize) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, src && dst, SVX_ERR_INVALID_ARG, "svx_gaussian_blur: null image pointer");
    SVX_REQUIRE(ctx, src != dst, SVX_ERR_INVALID_ARG, "svx_gaussian_blur: in-place operation is not supported");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_gaussian_blur: empty image");
    SVX_REQUIRE(ctx, ksize == 1 || ksize == 3 || ksize == 5, SVX_ERR_UNSUPPORTED, "svx_gaussian_blur: ksize must be 1, 3 or 5");
    SVX_REQUIRE(ctx, n_frames <= 65535, SVX_ERR_UNSUPPORTED, "svx_gaussian_blur: more than 65535 frames per call");
    const dim3 grid(svx_cdiv(cols, 64), svx_cdiv(rows, 4), n_frames);
    if (ksize == 1) gaussian_blur_kernel<1><<<grid, 256, 0, ctx->stream>>>(src, dst, rows, cols);
    else if (ksize == 3) gaussian_blur_kernel<3><<<grid, 256, 0, ctx->stream>>>(src, dst, rows, cols);
    else gaussian_blur_kernel<5><<<grid, 256, 0, ctx->stream>>>(src, dst, rows, cols);
    SVX_LAUNCH_CHECK(ctx, "gaussian_blur_kernel");
    return SVX_OK;
}

__global__ __launch_bounds__(256) void alter_rows_kernel(uint8_t *__restrict__ img, int rows, int cols, int row_size, int nb, int step, int line_size) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= cols) return;
    bool invert = false, zero = false;
    if (row_size > 0 && nb >= 1) {
        int band = y / row_size;
        if (band > nb - 1) band = nb - 1;                 // the last band runs to the bottom row
        invert = (band & 1) == 0;
    }
    if (step > 0) {
        const int t = y - step / 2;
        zero = t >= 0 && (t % step) < line_size;
    }
    uint8_t *p = img + ((long)blockIdx.z * rows + y) * cols + x;
    if (zero) *p = 0;
    else if (invert) *p = (uint8_t)(255 - *p);
}

SVX_API int svx_alter_rows(svx_ctx *ctx, uint8_t *img, int rows, int cols, int n_frames, int invert_rows, int nb_rows,
                           int trace_lines, int nb_lines, int line_size) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, img, SVX_ERR_INVALID_ARG, "svx_alter_rows: null image pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_alter_rows: empty image");
    SVX_REQUIRE(ctx, n_frames <= 65535 && rows <= 65535, SVX_ERR_UNSUPPORTED, "svx_alter_rows: more than 65535 frames or rows per call");
    int row_size = 0, nb = 0, step = 0;
    if (invert_rows) {
        SVX_REQUIRE(ctx, nb_rows > 0, SVX_ERR_INVALID_ARG, "svx_alter_rows: nb_rows must be positive");
        row_size = rows / nb_rows + 1;
        nb = rows / row_size;
    }
    if (trace_lines) {
        SVX_REQUIRE(ctx, nb_lines > 0 && rows / nb_lines > 0, SVX_ERR_INVALID_ARG, "svx_alter_rows: nb_lines must be in 1..rows");
        step = rows / nb_lines;
    }
    if (!row_size && !step) return SVX_OK;
    alter_rows_kernel<<<dim3(svx_cdiv(cols, 256), rows, n_frames), 256, 0, ctx->stream>>>(img, rows, cols, row_size, nb, step, line_size);
    SVX_LAUNCH_CHECK(ctx, "alter_rows_kernel");
    return SVX_OK;
}
