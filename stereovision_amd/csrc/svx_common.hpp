// Internal definitions shared by the libsvx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/svx.h"

struct svx_ctx {
    int device;
    hipStream_t stream;
    void *ws;             // device workspace, grown on demand
    size_t ws_bytes;
    void *pinned;         // pinned host staging buffer, grown on demand
    size_t pinned_bytes;
    char err[256];
    double stage_ms[SVX_PIPELINE_STAGES];   // wall-clock per stage of the last svx_high_perf_stereo call
    int opt[SVX_OPT_COUNT];                 // svx_set_option values (defaults: svx_option_default)
    void *arena;                            // svx_high_perf_stereo's device images, kept between calls and grown on demand
    size_t arena_bytes;
    void *tri;                              // triangle corners and planes of the mesh prior, kept between calls
    size_t tri_bytes;
    svx_ctx **workers;                      // svx_high_perf_stereo_batch: one context (and stream) per host thread, kept between calls
    int n_workers;
    hipStream_t own_stream;                 // a worker context's own stream (null for a caller's context)
    void *static_scratch;                   // this device's address of the code object's fixed scratch array (svx_dense.hip)
    hipEvent_t pinned_read;                 // recorded behind the last asynchronous copy out of `pinned`
    bool pinned_busy;
    int mesh_left_over, mesh_left_first;    // svx_mesh_dev.hip: points of the last device triangulation that needed the pass over all points / a second look
    struct svx_mail *mail_host, *mail_dev;  // a few words of host memory that kernels write to directly (svx_mail_next)
    unsigned mail_seq;
    hipStream_t side_stream;                // svx_high_perf_stereo: copies that run beside the frame's kernels (svx_side_copy_to_host)
    hipEvent_t side_ready, side_done;
    void *side_pin;                         // pinned host memory those copies land in
    size_t side_pin_bytes;
    bool side_busy;
};

// Counts a kernel leaves for the host without a copy: host memory the device writes straight into.  The host hands the kernel
// the device address and a sequence number (svx_mail_next); the kernel stores its values, then the number (svx_mail_post, after
// every storing thread's __threadfence_system and a barrier); svx_mail_wait returns when the number has arrived, which can be
// well before the stream drains.  It keeps an eye on the stream, so a launch that failed ends the wait.
struct svx_mail { unsigned seq; unsigned v[31]; };
int svx_mail_next(svx_ctx *ctx, svx_mail **dev, unsigned *seq);
int svx_mail_wait(svx_ctx *ctx, unsigned seq, const unsigned **values);
// A copy to the host beside the stream's work: pinned memory of its own (svx_side_reserve, which first waits for a copy still in
// flight), a copy that starts when the work queued on the context's stream so far is done and runs on a stream of its own
// (svx_side_copy_to_host), and the wait for it (svx_side_wait).
int svx_side_reserve(svx_ctx *ctx, size_t bytes, void **host);
int svx_side_copy_to_host(svx_ctx *ctx, void *host, const void *dev, size_t bytes);
int svx_side_wait(svx_ctx *ctx);

// Default of an option: its environment variable when set, else the built-in value.
int svx_option_default(int option);
// Option value of a context (nullptr: the defaults).
inline int svx_opt(const svx_ctx *ctx, int option) { return ctx ? ctx->opt[option] : svx_option_default(option); }

#define SVX_API extern "C" __attribute__((visibility("default")))

inline int svx_fail(svx_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define SVX_HIP(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return svx_fail((ctx), SVX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define SVX_LAUNCH_CHECK(ctx, name)                                                          \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess)                                                                \
            return svx_fail((ctx), SVX_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

#define SVX_REQUIRE(ctx, cond, code, msg)                          \
    do {                                                           \
        if (!(cond)) return svx_fail((ctx), (code), "%s", (msg));  \
    } while (0)

// Makes the context's device current and returns its workspace grown to at least `bytes`.
int svx_ws_reserve(svx_ctx *ctx, size_t bytes, void **out);
int svx_pinned_reserve(svx_ctx *ctx, size_t bytes, void **out);
// After enqueueing an asynchronous copy out of the pinned buffer: the next svx_pinned_reserve waits for it.
int svx_pinned_mark_read(svx_ctx *ctx);
// Grows a device block kept in the context (pointer and size fields passed in) to at least `bytes`.
int svx_block_reserve(svx_ctx *ctx, void **block, size_t *block_bytes, size_t bytes, const char *what);
int svx_enter(svx_ctx *ctx);
int svx_mesh_device_triangles(svx_ctx *ctx, const int32_t *pts_xy, int n, int rows, int cols, int32_t **tri_xy_out, int32_t **tri_idx_out,
                              float **abc_out, int *n_tri);      // svx_mesh_dev.hip
int svx_sparse_match_lists(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, const int32_t *left_xy,
                           int n_left, const int32_t *right_xy, int n_right, int max_disparity, float y_tolerance, float uniqueness,
                           int left_right_step, int *first_dev, int *cost_dev);      // svx_sparse.hip: device key-point lists in raster order
int svx_cost_eval_refine(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, const int32_t *pts_xy, int n_pts,
                         const float *disparities, int rows, int cols, int t_low, int t_high, unsigned occ_grid_size,
                         float *final_disparities, int8_t *final_costs, int32_t *conf_x, int32_t *conf_y, float *conf_d, int8_t *conf_c,
                         int32_t *inv_x, int32_t *inv_y, int8_t *inv_c);      // svx_prior.hip: svx_cost_eval + svx_refine without the cost map
int svx_static_scratch_locate(svx_ctx *ctx);   // svx_dense.hip: finds this device's copy of the fixed scratch array

// Matrix-core form of the dense search on two-word descriptors (svx_dense_mx.hip).
bool svx_mx_sweep_eligible(int rows, int cols, int max_disp);
size_t svx_mx_partial_words(const svx_ctx *ctx, int rows, int cols, int n_frames, int words, int max_disp);
// keep_keys: the left map is left as the winners' keys (below) instead of disparities, for a kernel that wants the winners'
// costs as well; svx_launch_mx_right's check then reads keys and marks a rejected pixel with SVX_MX_NO_KEY.
int svx_launch_mx_sweep(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int words, int max_disp,
                        float *disparity, int write_border, float *partial, bool keep_keys = false);
int svx_launch_mx_right(svx_ctx *ctx, const float *partial, int rows, int cols, int n_frames, int words, int max_disp,
                        float *disp_left, float *disp_right, float lr_threshold, bool keys = false);
// A winner's key on two-word descriptors: 1600 - 2 * (cost of the 5x5 window) + (1023 - d) / 1024, exact in a float; a pixel
// without an admissible pair holds a key below -2048 (disparity 0), and nothing reaches SVX_MX_NO_KEY.
constexpr float SVX_MX_NO_KEY = -4194304.0f;
__device__ __forceinline__ float svx_mx_key_disparity(float z) {
    const float frac = z - __builtin_floorf(z);
    return z < -2048.0f ? 0.0f : 1023.0f - frac * 1024.0f;
}
__device__ __forceinline__ int svx_mx_key_cost(float z) { return (1600 - (int)__builtin_floorf(z)) >> 1; }

// ---- device helpers -------------------------------------------------------

__device__ __forceinline__ void svx_mail_post(svx_mail *m, unsigned seq) {
    __threadfence_system();
    __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Flat-guard read: index into one contiguous frame of n elements, 0 outside.
__device__ __forceinline__ uint32_t ld_guard_u32(const uint32_t *__restrict__ img, long n, long idx) {
    return (idx >= 0 && idx < n) ? img[idx] : 0u;
}

__device__ __forceinline__ int popc32(uint32_t v) { return __builtin_popcount(v); }

// Grid-stride friendly ceil-div.
static inline unsigned svx_cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }
