// exFAST feature detector (gfx950): sparsestereo/extendedfast.cpp:92-231, extendedfast.h:18,
// fast9-inl.h:76-173.  The reference filters a FAST-9 corner list at the minimum threshold through
// an adaptive-threshold test with a score, then runs a sparse 3x3 non-maximum suppression over the
// list.  Here every pixel is tested on its own: a score map (0 = no corner, score+1 otherwise) is
// written per frame, the suppression reads the 8 neighbours of the map, and the survivors are
// compacted in raster order, which is the order of the reference's list.
#include "svx_common.hpp"
#include "svx_compact.hpp"

namespace {

constexpr int FT_W = 64, FT_H = 4, FT_HALO = 3, FT_PITCH = FT_W + 2 * FT_HALO + 2;     // 72-byte LDS rows

// True when the 16-bit ring mask holds a run of at least 9 set bits (with wrap-around).
__device__ __forceinline__ bool ring_arc9(unsigned m) {
    unsigned r = m | (m << 16);
    unsigned a = r & (r >> 1);
    a &= a >> 2;
    a &= a >> 4;                // bit i: bits i..i+7 of r are set
    a &= r >> 8;                // ... and bit i+8
    return (a & 0xffffu) != 0;
}

__device__ __forceinline__ int sat_s8(int v) { return min(max(v, -128), 127); }
__device__ __forceinline__ int as_s8(int v) { return (int)(int8_t)(uint8_t)v; }

// Ring masks of "circle > hi" and "lo > circle" (signed-domain values).
__device__ __forceinline__ void ring_masks(const int (&c)[16], int hi, int lo, unsigned &brighter, unsigned &darker) {
    brighter = darker = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        brighter |= (c[i] > hi ? 1u : 0u) << i;
        darker |= (lo > c[i] ? 1u : 0u) << i;
    }
}

// A block scores a tile of FT_W x FT_H pixels in two steps: every pixel takes the FAST-9 test at the minimum threshold, and the
// few that pass it (a twentieth of a textured frame) queue up in LDS and are scored by as many lanes as there are of them:
// the adaptive test and the bisection, ten times the work of the first test, run in full wavefronts instead of in every
// wavefront that holds one candidate.
__global__ __launch_bounds__(256) void exfast_score_kernel(const uint8_t *__restrict__ img_all, int rows, int cols, int min_threshold,
                                                           float adaptivity, int round_sat, uint16_t *__restrict__ score_all) {
    __shared__ uint8_t tile[(FT_H + 2 * FT_HALO) * FT_PITCH];
    __shared__ uint16_t queue[FT_W * FT_H];
    __shared__ int queued;
    const long n = (long)rows * cols;
    const uint8_t *img = img_all + (long)blockIdx.z * n;
    uint16_t *score_map = score_all + (long)blockIdx.z * n;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    if (threadIdx.x == 0) queued = 0;
    for (int i = threadIdx.x; i < (FT_H + 2 * FT_HALO) * (FT_W + 2 * FT_HALO); i += 256) {
        const int ty = i / (FT_W + 2 * FT_HALO), tx = i - ty * (FT_W + 2 * FT_HALO);
        const int gy = y0 + ty - FT_HALO, gx = x0 + tx - FT_HALO;
        tile[ty * FT_PITCH + tx] = (gy >= 0 && gy < rows && gx >= 0 && gx < cols) ? img[(long)gy * cols + gx] : (uint8_t)0;
    }
    __syncthreads();
    // ring offsets in the order of extendedfast.cpp:73-90
    constexpr int DX[16] = {-1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2};
    constexpr int DY[16] = {-3, -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2};
    // stage 1: FAST-9 at the minimum threshold on unsigned values (extendedfast.cpp:102); everything else scores 0
    for (int p = threadIdx.x; p < FT_W * FT_H; p += 256) {
        const int lx = p & (FT_W - 1), ly = p / FT_W, x = x0 + lx, y = y0 + ly;
        if (x >= cols || y >= rows) continue;
        bool candidate = false;
        if (x >= 3 && x < cols - 3 && y >= 3 && y < rows - 3) {
            const uint8_t *c = &tile[(ly + FT_HALO) * FT_PITCH + lx + FT_HALO];
            int ring[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) ring[i] = c[DY[i] * FT_PITCH + DX[i]];
            const int v = c[0];
            unsigned br, dk;
            ring_masks(ring, v + min_threshold, v - min_threshold, br, dk);
            candidate = ring_arc9(br) || ring_arc9(dk);
        }
        if (candidate) queue[atomicAdd(&queued, 1)] = (uint16_t)p;
        else score_map[(long)y * cols + x] = 0;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < queued; q += 256) {
        const int p = queue[q], lx = p & (FT_W - 1), ly = p / FT_W, x = x0 + lx, y = y0 + ly;
        const uint8_t *c = &tile[(ly + FT_HALO) * FT_PITCH + lx + FT_HALO];
        int ring[16];
        int sum = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { ring[i] = c[DY[i] * FT_PITCH + DX[i]]; sum += ring[i]; }
        const int v = c[0];
        uint16_t result = 0;
        // stage 2: adaptive threshold from the ring's mean absolute deviation (extendedfast.cpp:156-165)
        const int avg = sum / 16;
        int sad = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) sad += abs(ring[i] - avg);
        const float tf = (float)sad * adaptivity / 16.f;
        const int thr_u = round_sat ? (sat_s8((int)rintf(tf)) & 0xff) : ((int)tf & 0xff);
        const int thr_s = as_s8(thr_u);
        const int mean5 = (v + c[-FT_PITCH] + c[FT_PITCH] + c[-1] + c[1]) / 5;      // extendedfast.h:18
        const int cen_s = mean5 - 128;
#pragma unroll
        for (int i = 0; i < 16; ++i) ring[i] -= 128;                                 // signed domain
        unsigned br, dk;
        ring_masks(ring, sat_s8(cen_s + thr_s), sat_s8(cen_s - thr_s), br, dk);      // extendedfast.cpp:177-195
        if (ring_arc9(br) || ring_arc9(dk)) {
            // score: largest threshold that still gives a corner, by bisection (extendedfast.cpp:197-231)
            int bmin = thr_u, bmax = 255, b = (bmax + bmin + 1) >> 1;
            for (;;) {
                ring_masks(ring, min(mean5 + b, 255) - 128, max(mean5 - b, 0) - 128, br, dk);
                if (ring_arc9(br) || ring_arc9(dk)) bmin = b;
                else bmax = b;
                if (bmin == bmax || bmin == bmax - 1) break;
                b = (bmax + bmin + 1) >> 1;
            }
            result = (uint16_t)(((bmin - thr_u) & 0xff) + 1);
        }
        score_map[(long)y * cols + x] = result;
    }
}

struct CornerPred {
    const uint16_t *score;
    int rows, cols, nonmax;
    __device__ bool operator()(int x, int y, int frame) const {
        const uint16_t *p = score + ((long)frame * rows + y) * cols + x;
        const unsigned s = *p;
        if (!s) return false;
        if (!nonmax) return true;
        // corners lie at least 3 pixels inside the image, so the 8 neighbours exist
        unsigned m = max(max((unsigned)p[-1], (unsigned)p[1]), max((unsigned)p[-cols], (unsigned)p[cols]));
        m = max(m, max(max((unsigned)p[-cols - 1], (unsigned)p[-cols + 1]), max((unsigned)p[cols - 1], (unsigned)p[cols + 1])));
        return m < s;                                       // a neighbour with an equal score suppresses too
    }
};

struct CornerEmit {
    const uint16_t *score;
    int32_t *xy;
    float *response;
    int rows, cols, nonmax;
    __device__ void operator()(unsigned slot, int x, int y, int frame) const {
        *reinterpret_cast<int2 *>(xy + 2 * (size_t)slot) = make_int2(x, y);
        response[slot] = nonmax ? (float)(score[((long)frame * rows + y) * cols + x] - 1) : -1000.f;
    }
};

}  // namespace

SVX_API int svx_exfast_detect(svx_ctx *ctx, const uint8_t *img, int rows, int cols, int n_frames, int min_threshold, float adaptivity,
                              int flags, int32_t *kp_xy, float *kp_response, int capacity, int *frame_offsets_host) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, img && frame_offsets_host, SVX_ERR_INVALID_ARG, "svx_exfast_detect: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_exfast_detect: empty image");
    SVX_REQUIRE(ctx, capacity >= 0 && (capacity == 0 || (kp_xy && kp_response)), SVX_ERR_INVALID_ARG, "svx_exfast_detect: bad output buffer");
    SVX_REQUIRE(ctx, (flags & ~(SVX_EXFAST_NONMAX | SVX_EXFAST_THRESHOLD_ROUND_SAT)) == 0, SVX_ERR_INVALID_ARG, "svx_exfast_detect: unknown flag");
    SVX_REQUIRE(ctx, n_frames <= 65535 && svx_cdiv(rows, FT_H) <= 65535, SVX_ERR_UNSUPPORTED, "svx_exfast_detect: too many frames or rows per call");
    const int thr = min_threshold < 0 ? 0 : (min_threshold > 255 ? 255 : min_threshold);
    const size_t map_bytes = (((size_t)rows * cols * n_frames * sizeof(uint16_t)) + 255) & ~(size_t)255;
    void *ws;
    rc = svx_ws_reserve(ctx, map_bytes + svx_compact_bytes(rows, n_frames), &ws);
    if (rc) return rc;
    uint16_t *score = (uint16_t *)ws;
    exfast_score_kernel<<<dim3(svx_cdiv(cols, FT_W), svx_cdiv(rows, FT_H), n_frames), 256, 0, ctx->stream>>>(
        img, rows, cols, thr, adaptivity, (flags & SVX_EXFAST_THRESHOLD_ROUND_SAT) ? 1 : 0, score);
    SVX_LAUNCH_CHECK(ctx, "exfast_score_kernel");
    const int nonmax = (flags & SVX_EXFAST_NONMAX) ? 1 : 0;
    CornerPred pred{score, rows, cols, nonmax};
    CornerEmit emit{score, kp_xy, kp_response, rows, cols, nonmax};
    return svx_compact_raster(ctx, (char *)ws + map_bytes, rows, cols, n_frames, pred, emit, capacity, frame_offsets_host, "exfast_detect");
}
