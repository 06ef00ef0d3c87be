// Reprojection of the final disparities to a 3-D point list (gfx950): the consumer of the disparity
// map, highPerfStereo/highPerfStereo.cpp:689-723.  cv::perspectiveTransform with the 4x4 matrix Q
// is restated from OpenCV's documented formula in double precision; points are kept when z > 0 and
// their range is below the limit, in input order.
#include "svx_common.hpp"
#include "svx_compact.hpp"
#include <cfloat>

namespace {

struct QMat { double m[16]; };

constexpr int CLOUD_LINE = 1024;      // the point list is cut into lines of this many points for the compaction

#pragma clang fp contract(off)
__device__ __forceinline__ bool reproject(const QMat &q, double x, double y, double d, double max_range, double &X, double &Y, double &Z) {
    double w = x * q.m[12] + y * q.m[13] + d * q.m[14] + q.m[15];
    if (fabs(w) > (double)FLT_EPSILON) {
        w = 1. / w;
        X = (x * q.m[0] + y * q.m[1] + d * q.m[2] + q.m[3]) * w;
        Y = (x * q.m[4] + y * q.m[5] + d * q.m[6] + q.m[7]) * w;
        Z = (x * q.m[8] + y * q.m[9] + d * q.m[10] + q.m[11]) * w;
    } else {
        X = Y = Z = 0.;
    }
    return Z > 0 && sqrt(X * X + Y * Y + Z * Z) < max_range;
}

struct CloudPred {
    QMat q;
    const int32_t *pts;
    const float *disp;
    int n, cols, roi_x, roi_y;
    double max_range;
    __device__ bool operator()(int x, int y, int) const {
        const long i = (long)y * CLOUD_LINE + x;
        if (i >= n) return false;
        const int px = pts[2 * i], py = pts[2 * i + 1];
        double X, Y, Z;
        return reproject(q, (double)(px + roi_x), (double)(py + roi_y), (double)disp[(long)py * cols + px], max_range, X, Y, Z);
    }
};

struct CloudEmit {
    QMat q;
    const int32_t *pts;
    const float *disp;
    const uint8_t *bgr;
    int cols, img_cols, roi_x, roi_y;
    double max_range;
    double *xyz;
    uint8_t *rgb;
    __device__ void operator()(unsigned slot, int x, int y, int) const {
        const long i = (long)y * CLOUD_LINE + x;
        const int px = pts[2 * i], py = pts[2 * i + 1];
        double X, Y, Z;
        reproject(q, (double)(px + roi_x), (double)(py + roi_y), (double)disp[(long)py * cols + px], max_range, X, Y, Z);
        xyz[3 * (size_t)slot] = X; xyz[3 * (size_t)slot + 1] = Y; xyz[3 * (size_t)slot + 2] = Z;
        if (rgb) {
            const uint8_t *c = bgr + 3 * ((long)(py + roi_y) * img_cols + px + roi_x);
            rgb[3 * (size_t)slot] = c[2]; rgb[3 * (size_t)slot + 1] = c[1]; rgb[3 * (size_t)slot + 2] = c[0];
        }
    }
};

// Dense variant: every pixel whose disparity exceeds a floor, in raster order.
struct DensePred {
    QMat q;
    const float *disp;
    int rows, cols, roi_x, roi_y;
    float min_disp;
    double max_range;
    __device__ bool operator()(int x, int y, int frame) const {
        const float d = disp[((long)frame * rows + y) * cols + x];
        if (!(d > min_disp)) return false;
        double X, Y, Z;
        return reproject(q, (double)(x + roi_x), (double)(y + roi_y), (double)d, max_range, X, Y, Z);
    }
};

struct DenseEmit {
    QMat q;
    const float *disp;
    int rows, cols, roi_x, roi_y;
    double max_range;
    double *xyz;
    int32_t *pix;
    __device__ void operator()(unsigned slot, int x, int y, int frame) const {
        double X, Y, Z;
        reproject(q, (double)(x + roi_x), (double)(y + roi_y), (double)disp[((long)frame * rows + y) * cols + x], max_range, X, Y, Z);
        xyz[3 * (size_t)slot] = X; xyz[3 * (size_t)slot + 1] = Y; xyz[3 * (size_t)slot + 2] = Z;
        if (pix) { pix[2 * (size_t)slot] = x; pix[2 * (size_t)slot + 1] = y; }
    }
};

__global__ void cloud_check_points_kernel(const int32_t *__restrict__ pts, int n, int rows, int cols, int roi_x, int roi_y,
                                          int img_rows, int img_cols, int check_img, unsigned *__restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = pts[2 * i], y = pts[2 * i + 1];
    bool ok = x >= 0 && x < cols && y >= 0 && y < rows;
    if (check_img) ok = ok && x + roi_x >= 0 && x + roi_x < img_cols && y + roi_y >= 0 && y + roi_y < img_rows;
    if (!ok) atomicAdd(bad, 1u);
}

}  // namespace

SVX_API int svx_point_cloud(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, const float *disparity_map, int rows, int cols,
                            int roi_x, int roi_y, const double *q_host, const uint8_t *bgr, int img_rows, int img_cols,
                            double max_range, double *xyz_out, uint8_t *rgb_out, int capacity, int *n_out_host) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n_pts >= 0 && capacity >= 0 && n_out_host, SVX_ERR_INVALID_ARG, "svx_point_cloud: bad count or null result");
    *n_out_host = 0;
    if (n_pts == 0) return SVX_OK;
    SVX_REQUIRE(ctx, pts_xy && disparity_map && q_host, SVX_ERR_INVALID_ARG, "svx_point_cloud: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_point_cloud: empty disparity map");
    SVX_REQUIRE(ctx, capacity == 0 || xyz_out, SVX_ERR_INVALID_ARG, "svx_point_cloud: null output");
    SVX_REQUIRE(ctx, !rgb_out || (bgr && img_rows > 0 && img_cols > 0), SVX_ERR_INVALID_ARG, "svx_point_cloud: colours requested without an image");
    const int n_lines = (int)svx_cdiv(n_pts, CLOUD_LINE);
    void *ws;
    rc = svx_ws_reserve(ctx, 256 + svx_compact_bytes(n_lines, 1), &ws);
    if (rc) return rc;
    // the points index the disparity map (and the colour image): reject a list that would read outside
    unsigned *bad = (unsigned *)ws;
    SVX_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(unsigned), ctx->stream));
    cloud_check_points_kernel<<<svx_cdiv(n_pts, 256), 256, 0, ctx->stream>>>(pts_xy, n_pts, rows, cols, roi_x, roi_y, img_rows, img_cols,
                                                                              rgb_out ? 1 : 0, bad);
    SVX_LAUNCH_CHECK(ctx, "cloud_check_points_kernel");
    unsigned n_bad = 0;
    SVX_HIP(ctx, hipMemcpyAsync(&n_bad, bad, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_bad) return svx_fail(ctx, SVX_ERR_INVALID_ARG, "svx_point_cloud: %u points lie outside the disparity map or the colour image", n_bad);
    CloudPred pred;
    CloudEmit emit;
    for (int i = 0; i < 16; ++i) pred.q.m[i] = emit.q.m[i] = q_host[i];
    pred.pts = emit.pts = pts_xy; pred.disp = emit.disp = disparity_map; pred.n = n_pts; pred.cols = emit.cols = cols;
    pred.roi_x = emit.roi_x = roi_x; pred.roi_y = emit.roi_y = roi_y; pred.max_range = emit.max_range = max_range;
    emit.bgr = bgr; emit.img_cols = img_cols; emit.xyz = xyz_out; emit.rgb = rgb_out;
    int off[2] = {0, 0};
    rc = svx_compact_raster(ctx, (char *)ws + 256, n_lines, CLOUD_LINE, 1, pred, emit, capacity, off, "point_cloud");
    if (rc) return rc;
    *n_out_host = off[1];
    return SVX_OK;
}

SVX_API int svx_dense_point_cloud(svx_ctx *ctx, const float *disparity_map, int rows, int cols, int n_frames, int roi_x, int roi_y,
                                  const double *q_host, float min_disparity, double max_range, double *xyz_out, int32_t *pix_xy_out,
                                  int capacity, int *frame_offsets_host) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, disparity_map && q_host && frame_offsets_host, SVX_ERR_INVALID_ARG, "svx_dense_point_cloud: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_dense_point_cloud: empty disparity map");
    SVX_REQUIRE(ctx, capacity >= 0 && (capacity == 0 || xyz_out), SVX_ERR_INVALID_ARG, "svx_dense_point_cloud: bad output buffer");
    DensePred pred;
    DenseEmit emit;
    for (int i = 0; i < 16; ++i) pred.q.m[i] = emit.q.m[i] = q_host[i];
    pred.disp = emit.disp = disparity_map; pred.rows = emit.rows = rows; pred.cols = emit.cols = cols;
    pred.roi_x = emit.roi_x = roi_x; pred.roi_y = emit.roi_y = roi_y; pred.max_range = emit.max_range = max_range;
    pred.min_disp = min_disparity;
    emit.xyz = xyz_out; emit.pix = pix_xy_out;
    void *ws;
    rc = svx_ws_reserve(ctx, svx_compact_bytes(rows, n_frames), &ws);
    if (rc) return rc;
    return svx_compact_raster(ctx, ws, rows, cols, n_frames, pred, emit, capacity, frame_offsets_host, "dense_point_cloud");
}
