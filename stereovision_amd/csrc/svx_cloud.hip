// Reprojection of the final disparities to a 3-D point list (gfx950): the consumer of the disparity
// map, highPerfStereo/highPerfStereo.cpp:689-723.  cv::perspectiveTransform with the 4x4 matrix Q
// is restated from OpenCV's documented formula in double precision; points are kept when z > 0 and
// their range is below the limit, in input order.
#include "svx_common.hpp"
#include "svx_compact.hpp"
#include <cfloat>
#include <cmath>
#include <fstream>
#include <vector>

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

// Stereo against Kinect depth, highPerfStereo/PerformanceEvaluator.cpp:132-245: the per-point part.  A considered pixel
// is reprojected with Q (double arithmetic, float result), moved to the Kinect frame with the float matrix [R|T] and
// projected into the Kinect image with the double R, T and camera matrix; it is a correspondence when the rounded pixel
// lies in 640x480, 0 < z <= 5000 and the Kinect point of the truncated pixel has z > 0.
struct KinectGeom {
    QMat q;
    double r[9], t[3], fx, fy, cx, cy;
    float m[12];
    const int32_t *pts, *raw;
    const float *disp;
    int n, cols, depth_cols, x_off, y_off;
};

__device__ __forceinline__ float kinect_millimetres(int raw) {                  // PerformanceEvaluator.cpp:176-181
    return raw < 2047 ? (float)(1000 * (1.0 / ((double)raw * -0.0030711016 + 3.3309495161))) : 0.0f;
}

__device__ __forceinline__ bool kinect_correspondence(const KinectGeom &g, long i, int &bin, float err[4]) {
    const int px = g.pts[2 * i], py = g.pts[2 * i + 1];
    const double x = (double)(float)(px + g.x_off), y = (double)(float)(py + g.y_off), z = (double)g.disp[(long)py * g.cols + px];
    double w = x * g.q.m[12] + y * g.q.m[13] + z * g.q.m[14] + g.q.m[15];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (fabs(w) > (double)FLT_EPSILON) {
        w = 1. / w;
        sx = (float)((x * g.q.m[0] + y * g.q.m[1] + z * g.q.m[2] + g.q.m[3]) * w);
        sy = (float)((x * g.q.m[4] + y * g.q.m[5] + z * g.q.m[6] + g.q.m[7]) * w);
        sz = (float)((x * g.q.m[8] + y * g.q.m[9] + z * g.q.m[10] + g.q.m[11]) * w);
    }
    const float kx = g.m[0] * sx + g.m[1] * sy + g.m[2] * sz + g.m[3] * 1.0f;
    const float ky = g.m[4] * sx + g.m[5] * sy + g.m[6] * sz + g.m[7] * 1.0f;
    const float kz = g.m[8] * sx + g.m[9] * sy + g.m[10] * sz + g.m[11] * 1.0f;
    const double X = g.r[0] * sx + g.r[1] * sy + g.r[2] * sz + g.t[0];
    const double Y = g.r[3] * sx + g.r[4] * sy + g.r[5] * sz + g.t[1];
    const double Z = g.r[6] * sx + g.r[7] * sy + g.r[8] * sz + g.t[2];
    const double iz = Z != 0 ? 1. / Z : 1.;
    const float u = (float)(X * iz * g.fx + g.cx), v = (float)(Y * iz * g.fy + g.cy);
    if (!(u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f)) return false;
    const int ur = __float2int_rn(u), vr = __float2int_rn(v);                   // Point2f -> Point rounds to nearest even
    if (!(ur >= 0 && ur < 640 && vr >= 0 && vr < 480 && kz > 0 && kz <= 5000)) return false;
    const int ut = (int)u, vt = (int)v;                                         // getKinect3DPoint(int, int) truncates
    const float mm = kinect_millimetres(g.raw[(long)vt * g.depth_cols + ut]);
    if (!(mm > 0)) return false;
    const double fx_d = 1.0 / 5.9421434211923247e+02, fy_d = 1.0 / 5.9104053696870778e+02;      // :163-166
    const double cx_d = 3.3930780975300314e+02, cy_d = 2.4273913761751615e+02;
    const float qx = (float)((ut - cx_d) * (double)mm * fx_d), qy = (float)((vt - cy_d) * (double)mm * fy_d);
    err[0] = kx - qx; err[1] = ky - qy; err[2] = kz - mm;
    err[3] = (float)sqrt((double)err[0] * err[0] + (double)err[1] * err[1] + (double)err[2] * err[2]);
    bin = g.raw[(long)vr * g.depth_cols + ur];
    return true;
}

struct KinectPred {
    KinectGeom g;
    __device__ bool operator()(int x, int y, int) const {
        const long i = (long)y * CLOUD_LINE + x;
        if (i >= g.n) return false;
        int bin;
        float err[4];
        return kinect_correspondence(g, i, bin, err);
    }
};

struct KinectEmit {
    KinectGeom g;
    int32_t *bin_out;
    float *err_out;
    __device__ void operator()(unsigned slot, int x, int y, int) const {
        int bin = 0;
        float err[4] = {0.f, 0.f, 0.f, 0.f};
        kinect_correspondence(g, (long)y * CLOUD_LINE + x, bin, err);
        bin_out[slot] = bin;
        *(float4 *)(err_out + 4 * (size_t)slot) = make_float4(err[0], err[1], err[2], err[3]);
    }
};

__global__ void kinect_check_depth_kernel(const int32_t *__restrict__ raw, long n, unsigned *__restrict__ bad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (raw[i] < 0 || raw[i] >= 2048)) atomicAdd(bad, 1u);
}

// Boost.Accumulators' lazy variance on float samples, as PerformanceEvaluator.cpp:258-310 uses it: sum and sum of squares in
// float, mean = sum / n, variance = sumsq / n - mean * mean; NaN for an empty set.
struct LazyVariance {
    float sum = 0.f, sumsq = 0.f;
    size_t n = 0;
    void add(float v) { sum += v; sumsq += v * v; ++n; }
    float mean() const { return sum / (float)n; }
    float variance() const { const float m = mean(); return sumsq / (float)n - m * m; }
};

// SparseMatch::projectMatches, sparsestereo/sparsematch.h:35-50: float Q times (x, y, disparity, 1), divided by w.
struct QMatF { float m[16]; };
__global__ void project_matches_kernel(QMatF q, const float *__restrict__ xyd, int n, float *__restrict__ xyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyd[3 * i], y = xyd[3 * i + 1], d = xyd[3 * i + 2];
    float p[4];
    for (int r = 0; r < 4; ++r) p[r] = q.m[4 * r] * x + q.m[4 * r + 1] * y + q.m[4 * r + 2] * d + q.m[4 * r + 3] * 1.0f;
    xyz[3 * i] = p[0] / p[3]; xyz[3 * i + 1] = p[1] / p[3]; xyz[3 * i + 2] = p[2] / p[3];
}

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

SVX_API int svx_kinect_evaluate(svx_ctx *ctx, const int32_t *raw_depth, int depth_rows, int depth_cols, const float *disparity_map,
                                int rows, int cols, const int32_t *pts_xy, int n_pts, int x_offset, int y_offset, const double *q_host,
                                const double *cam_host, const double *r_host, const double *t_host, float *bin_mean_host,
                                float *bin_sigma_host, int32_t *bin_count_host, float *totals_host, int *n_corr_host) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, raw_depth && q_host && cam_host && r_host && t_host && bin_mean_host && bin_sigma_host && bin_count_host &&
                         totals_host && n_corr_host, SVX_ERR_INVALID_ARG, "svx_kinect_evaluate: null pointer");
    SVX_REQUIRE(ctx, n_pts >= 0, SVX_ERR_INVALID_ARG, "svx_kinect_evaluate: negative point count");
    SVX_REQUIRE(ctx, n_pts == 0 || (pts_xy && disparity_map && rows > 0 && cols > 0), SVX_ERR_INVALID_ARG,
                "svx_kinect_evaluate: points without a disparity map");
    // the reference indexes the Kinect cloud with pixels of Rect(0,0,640,480) (PerformanceEvaluator.cpp:202-208)
    SVX_REQUIRE(ctx, depth_rows >= 480 && depth_cols >= 640, SVX_ERR_INVALID_ARG, "svx_kinect_evaluate: the raw depth image must be at least 640x480");
    const int n_lines = (int)svx_cdiv(n_pts > 0 ? n_pts : 1, CLOUD_LINE);
    const size_t corr_bytes = ((size_t)(n_pts > 0 ? n_pts : 1) * 20 + 255) / 256 * 256;
    void *ws;
    rc = svx_ws_reserve(ctx, 256 + corr_bytes + svx_compact_bytes(n_lines, 1), &ws);
    if (rc) return rc;
    unsigned *bad = (unsigned *)ws;
    float *err_dev = (float *)((char *)ws + 256);
    int32_t *bin_dev = (int32_t *)(err_dev + 4 * (size_t)(n_pts > 0 ? n_pts : 1));
    void *area = (char *)ws + 256 + corr_bytes;
    SVX_HIP(ctx, hipMemsetAsync(bad, 0, 2 * sizeof(unsigned), ctx->stream));
    const long n_depth = (long)depth_rows * depth_cols;
    kinect_check_depth_kernel<<<svx_cdiv(n_depth, 256), 256, 0, ctx->stream>>>(raw_depth, n_depth, bad);
    SVX_LAUNCH_CHECK(ctx, "kinect_check_depth_kernel");
    if (n_pts > 0) {
        cloud_check_points_kernel<<<svx_cdiv(n_pts, 256), 256, 0, ctx->stream>>>(pts_xy, n_pts, rows, cols, 0, 0, 0, 0, 0, bad + 1);
        SVX_LAUNCH_CHECK(ctx, "cloud_check_points_kernel");
    }
    unsigned n_bad[2] = {0, 0};
    SVX_HIP(ctx, hipMemcpyAsync(n_bad, bad, sizeof(n_bad), hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_bad[0]) return svx_fail(ctx, SVX_ERR_INVALID_ARG, "svx_kinect_evaluate: %u raw depth values lie outside [0, 2048)", n_bad[0]);
    if (n_bad[1]) return svx_fail(ctx, SVX_ERR_INVALID_ARG, "svx_kinect_evaluate: %u points lie outside the disparity map", n_bad[1]);

    int n_corr = 0;
    std::vector<float> err;
    std::vector<int32_t> bin;
    if (n_pts > 0) {
        KinectPred pred;
        for (int i = 0; i < 16; ++i) pred.g.q.m[i] = q_host[i];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) { pred.g.r[3 * i + j] = r_host[3 * i + j]; pred.g.m[4 * i + j] = (float)r_host[3 * i + j]; }
            pred.g.t[i] = t_host[i];
            pred.g.m[4 * i + 3] = (float)t_host[i];
        }
        pred.g.fx = cam_host[0]; pred.g.cx = cam_host[2]; pred.g.fy = cam_host[4]; pred.g.cy = cam_host[5];
        pred.g.pts = pts_xy; pred.g.raw = raw_depth; pred.g.disp = disparity_map;
        pred.g.n = n_pts; pred.g.cols = cols; pred.g.depth_cols = depth_cols; pred.g.x_off = x_offset; pred.g.y_off = y_offset;
        KinectEmit emit;
        emit.g = pred.g; emit.bin_out = bin_dev; emit.err_out = err_dev;
        int off[2] = {0, 0};
        rc = svx_compact_raster(ctx, area, n_lines, CLOUD_LINE, 1, pred, emit, n_pts, off, "kinect_evaluate");
        if (rc) return rc;
        n_corr = off[1];
        err.resize(4 * (size_t)n_corr);
        bin.resize((size_t)n_corr);
        if (n_corr > 0) {
            SVX_HIP(ctx, hipMemcpyAsync(err.data(), err_dev, sizeof(float) * 4 * (size_t)n_corr, hipMemcpyDeviceToHost, ctx->stream));
            SVX_HIP(ctx, hipMemcpyAsync(bin.data(), bin_dev, sizeof(int32_t) * (size_t)n_corr, hipMemcpyDeviceToHost, ctx->stream));
            SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    *n_corr_host = n_corr;

    // The statistics are sequential float sums whose value depends on the order: bins in order, every bin in the order
    // the correspondences were found (PerformanceEvaluator.cpp:263-296).  Stable counting sort, then one pass.
    std::vector<int> start(2049, 0), order((size_t)n_corr);
    for (int i = 0; i < n_corr; ++i) ++start[bin[i] + 1];
    for (int b = 0; b < 2048; ++b) { bin_count_host[b] = start[b + 1]; start[b + 1] += start[b]; }
    {
        std::vector<int> fill(start.begin(), start.begin() + 2048);
        for (int i = 0; i < n_corr; ++i) order[fill[bin[i]]++] = i;
    }
    LazyVariance total[4], rel;
    for (int b = 0; b < 2048; ++b) {
        for (int k = 0; k < 4; ++k) {
            LazyVariance acc;
            for (int j = start[b]; j < start[b + 1]; ++j) {
                const float e = err[4 * (size_t)order[j] + k];
                acc.add(e);
                total[k].add(e);
            }
            bin_mean_host[k * 2048 + b] = acc.mean();
            bin_sigma_host[k * 2048 + b] = std::sqrt(acc.variance());
        }
        if (b != 2047 && bin_count_host[b] > 0) {
            const float mm = (float)(1000 * (1.0 / ((double)b * -0.0030711016 + 3.3309495161)));
            rel.add(bin_mean_host[3 * 2048 + b] / mm);                           // :292-294
        }
    }
    for (int k = 0; k < 4; ++k) { totals_host[k] = total[k].mean(); totals_host[4 + k] = std::sqrt(total[k].variance()); }
    totals_host[8] = rel.mean();
    totals_host[9] = std::sqrt(rel.variance());
    return SVX_OK;
}

SVX_API int svx_write_point_cloud_txt(svx_ctx *ctx, const char *path, const double *xyz_host, const uint8_t *rgb_host, int n_pts,
                                      const uint8_t *const_rgb) {
    SVX_REQUIRE(ctx, path && n_pts >= 0 && (n_pts == 0 || xyz_host) && (rgb_host || const_rgb), SVX_ERR_INVALID_ARG,
                "svx_write_point_cloud_txt: bad argument");
    std::ofstream out(path);
    if (!out) return svx_fail(ctx, SVX_ERR_INVALID_ARG, "svx_write_point_cloud_txt: cannot open %s", path);
    for (long i = 0; i < n_pts; ++i) {
        const uint8_t *c = rgb_host ? rgb_host + 3 * i : const_rgb;
        // the reference streams the colours as doubles (highPerfStereo.cpp:716-721) or int literals: the same digits
        out << xyz_host[3 * i] << " " << xyz_host[3 * i + 1] << " " << xyz_host[3 * i + 2] << " " << (int)c[0] << " " << (int)c[1] << " "
            << (int)c[2] << '\n';
    }
    out.close();
    if (!out) return svx_fail(ctx, SVX_ERR_INVALID_ARG, "svx_write_point_cloud_txt: write to %s failed", path);
    return SVX_OK;
}

SVX_API int svx_project_matches(svx_ctx *ctx, const float *xyd, int n, const double *q_host, float *xyz_out) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n >= 0 && q_host, SVX_ERR_INVALID_ARG, "svx_project_matches: bad count or null matrix");
    if (n == 0) return SVX_OK;
    SVX_REQUIRE(ctx, xyd && xyz_out, SVX_ERR_INVALID_ARG, "svx_project_matches: null pointer");
    QMatF q;
    for (int i = 0; i < 16; ++i) q.m[i] = (float)q_host[i];
    project_matches_kernel<<<svx_cdiv(n, 256), 256, 0, ctx->stream>>>(q, xyd, n, xyz_out);
    SVX_LAUNCH_CHECK(ctx, "project_matches_kernel");
    return SVX_OK;
}
