// highPerfStereo(): the support-point / plane-prior pipeline that calls the census path end to end
// (highPerfStereo/highPerfStereoLib.cpp:177-645), as host glue over the libsvx entry points.  Every image and
// every list stays on the device (key points, matches, the support-point list, the mesh, the occupancy grid); the
// host sees counts, and the computed points at the end.
#include "svx_common.hpp"
#include "svx_compact.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {

// disparities(y,x) = d for the support points, in list order (highPerfStereoLib.cpp:436-445)
__global__ void scatter_support_kernel(const int32_t *__restrict__ xy, const float *__restrict__ d, int n, int cols, float *__restrict__ map) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) map[(long)xy[2 * i + 1] * cols + xy[2 * i]] = d[i];
}

// disparityMap(p) = int(disparities(p)) for every mesh vertex (highPerfStereoLib.cpp:628-637)
__global__ void vertex_disparities_kernel(const int32_t *__restrict__ xy, int n, int cols, const float *__restrict__ disparities,
                                          float *__restrict__ final_disp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const long p = (long)xy[2 * i + 1] * cols + xy[2 * i];
        final_disp[p] = (float)(int)disparities[p];
    }
}

// The support points the sparse matcher yields, in the order of the left key-point list (highPerfStereoLib.cpp:434-445):
// matched[i] = 1 when left feature i has a match; then, ranked by the scan of those flags, the point goes into the
// support list, its disparity x_left - x_right (SparseMatch::disparity(), sparsematch.h:27-29) into the map, and its
// pixel is marked as taken.  Key points are distinct pixels, so no position repeats here.
__global__ void support_flags_kernel(const int *__restrict__ first, int nl, unsigned *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nl) flag[i] = first[i] != -1 ? 1u : 0u;
}

__global__ void support_emit_kernel(const int32_t *__restrict__ kl, const int32_t *__restrict__ kr, const int *__restrict__ first, int nl,
                                    const unsigned *__restrict__ rank, int cols, int32_t *__restrict__ sup, float *__restrict__ disparities,
                                    uint8_t *__restrict__ taken) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nl || first[i] == -1) return;
    const int x = kl[2 * i], y = kl[2 * i + 1];
    const unsigned k = rank[i];
    sup[2 * k] = x; sup[2 * k + 1] = y;
    disparities[(long)y * cols + x] = (float)x - (float)kr[2 * first[i]];
    taken[(long)y * cols + x] = 1;
}

// The resampling step's points join the support list in slot order, a position that is taken already (by an earlier
// support point or an earlier slot of this step) only once (highPerfStereoLib.cpp:141-175 inserts into a mesh that
// ignores repeats): the smallest slot at a free pixel owns it.
__global__ void resample_owner_kernel(const int32_t *__restrict__ xy, const uint8_t *__restrict__ valid, int n_slots, int cols,
                                      const uint8_t *__restrict__ taken, int *__restrict__ owner) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots || !valid[s]) return;
    const long p = (long)xy[2 * s + 1] * cols + xy[2 * s];
    if (!taken[p]) atomicMin(&owner[p], s);
}

__global__ void resample_flags_kernel(const int32_t *__restrict__ xy, const uint8_t *__restrict__ valid, int n_slots, int cols,
                                      const uint8_t *__restrict__ taken, const int *__restrict__ owner, unsigned *__restrict__ flag) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    unsigned f = 0;
    if (valid[s]) {
        const long p = (long)xy[2 * s + 1] * cols + xy[2 * s];
        f = (!taken[p] && owner[p] == s) ? 1u : 0u;
    }
    flag[s] = f;
}

__global__ void resample_emit_kernel(const int32_t *__restrict__ xy, int n_slots, int cols, const unsigned *__restrict__ rank, int n_sup,
                                     int32_t *__restrict__ sup, uint8_t *__restrict__ taken, int *__restrict__ owner) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots || rank[s + 1] == rank[s]) return;
    const int x = xy[2 * s], y = xy[2 * s + 1];
    const unsigned k = (unsigned)n_sup + rank[s];
    sup[2 * k] = x; sup[2 * k + 1] = y;
    taken[(long)y * cols + x] = 1;
    owner[(long)y * cols + x] = 0x7fffffff;          // free again for the next step
}

// Up to four fills in one launch: range k is words[k] 32-bit words of value[k] at p[k] (16-byte aligned; 0 words: none).
struct FillRanges { uint32_t *p[4]; long words[4]; uint32_t value[4]; };
__global__ __launch_bounds__(256) void fill_ranges_kernel(FillRanges r) {
    const long stride = (long)gridDim.x * blockDim.x, t = (long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 v4 = make_uint4(r.value[k], r.value[k], r.value[k], r.value[k]);
        const long quads = r.words[k] >> 2;
        for (long i = t; i < quads; i += stride) reinterpret_cast<uint4 *>(r.p[k])[i] = v4;
        for (long i = 4 * quads + t; i < r.words[k]; i += stride) r.p[k][i] = r.value[k];
    }
}

// Bump allocator over one device allocation (the context's workspace belongs to the entry points called below).
struct Arena {
    char *base = nullptr;
    size_t used = 0, cap = 0;
    template <class T> T *take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(base + used);
        used += n * sizeof(T);
        return p;
    }
};

// Stage clock, active only under SVX_OPT_STAGE_TIMES: a mark waits for the stream and books the time since the
// previous mark to one stage.  Without the option a mark does nothing and the stage times read 0.
struct StageClock {
    svx_ctx *ctx;
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit StageClock(svx_ctx *c) : ctx(c), on(svx_opt(c, SVX_OPT_STAGE_TIMES) != 0), t(std::chrono::steady_clock::now()) {
        for (int i = 0; i < SVX_PIPELINE_STAGES; ++i) c->stage_ms[i] = 0.0;
    }
    void mark(int stage) {
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        ctx->stage_ms[stage] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
    }
};

}  // namespace

#define SVX_TRY(call) do { rc = (call); if (rc) return rc; } while (0)

SVX_API int svx_high_perf_stereo(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols,
                                 const svx_stereo_parameters *prm, float *disparity_map,
                                 int32_t *points_xy_host, int capacity, int *n_points) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, left && right, SVX_ERR_EMPTY, "Input images are empty!");                         // .cpp:184-185
    SVX_REQUIRE(ctx, prm && disparity_map && n_points, SVX_ERR_INVALID_ARG, "svx_high_perf_stereo: null pointer");
    SVX_REQUIRE(ctx, capacity >= 0 && (capacity == 0 || points_xy_host), SVX_ERR_INVALID_ARG, "svx_high_perf_stereo: bad point buffer");
    SVX_REQUIRE(ctx, rows >= 16 && cols >= 16, SVX_ERR_EMPTY, "svx_high_perf_stereo: images smaller than 16x16 crop to nothing");
    SVX_REQUIRE(ctx, prm->ddepth == 3, SVX_ERR_UNSUPPORTED, "svx_high_perf_stereo: ddepth must be CV_16S (3)");
    SVX_REQUIRE(ctx, prm->nIters >= 0 && prm->maxDisp >= 1, SVX_ERR_INVALID_ARG, "svx_high_perf_stereo: bad nIters or maxDisp");
    SVX_REQUIRE(ctx, !prm->showImages, SVX_ERR_UNSUPPORTED, "svx_high_perf_stereo: showImages needs a display; use recordFullDisp");
    *n_points = 0;

    const int R = 16 * (rows / 16), Cc = 16 * (cols / 16);                                              // .cpp:245
    const size_t n = (size_t)R * Cc;
    const int t_low = (signed char)prm->tLow, t_high = (signed char)prm->tHigh;
    unsigned occ = 64;                                                                                  // .cpp:449
    const size_t max_cells = (size_t)(R + 1) * (Cc + 1);     // the grid shrinks to 1-pixel cells after six halvings
    size_t cells_needed = (size_t)(R / 64 + 1) * (Cc / 64 + 1);
    {
        unsigned g = occ;
        for (int it = 1; it < prm->nIters; ++it) { g = std::max(1u, g / 2); cells_needed = std::max(cells_needed, (size_t)(R / g + 1) * (Cc / g + 1)); }
        cells_needed = std::min(cells_needed, max_cells);
    }

    // one device block for every intermediate image and list, kept in the context between calls and grown on demand;
    // laid out twice (measure, then place)
    uint8_t *L, *Rr, *La, *Ra, *grad;
    uint32_t *cL, *cR;
    int32_t *pts, *kpl, *kpr, *conf_x, *conf_y, *inv_x, *inv_y, *rs_xy, *sup_dev;
    int *m_first, *m_cost, *owner;
    unsigned *rank;
    uint8_t *taken;
    float *respl, *respr, *disparities, *final_disp, *conf_d, *rs_d;
    int8_t *costs, *final_costs, *conf_c, *inv_c;
    uint8_t *rs_valid;
    auto layout = [&](Arena &a) {
        L = a.take<uint8_t>(n); Rr = a.take<uint8_t>(n); La = a.take<uint8_t>(n); Ra = a.take<uint8_t>(n); grad = a.take<uint8_t>(n);
        cL = a.take<uint32_t>(n); cR = a.take<uint32_t>(n);
        pts = a.take<int32_t>(2 * n); kpl = a.take<int32_t>(2 * n); kpr = a.take<int32_t>(2 * n);
        respl = a.take<float>(n); respr = a.take<float>(n);
        disparities = a.take<float>(n); final_disp = a.take<float>(n);
        costs = a.take<int8_t>(n); final_costs = a.take<int8_t>(n);
        conf_x = a.take<int32_t>(cells_needed); conf_y = a.take<int32_t>(cells_needed);
        inv_x = a.take<int32_t>(cells_needed); inv_y = a.take<int32_t>(cells_needed);
        conf_d = a.take<float>(cells_needed);
        conf_c = a.take<int8_t>(cells_needed); inv_c = a.take<int8_t>(cells_needed);
        rs_xy = a.take<int32_t>(4 * cells_needed); rs_d = a.take<float>(2 * cells_needed); rs_valid = a.take<uint8_t>(2 * cells_needed);
        sup_dev = a.take<int32_t>(2 * n);                                   // the support-point list (at most one point a pixel)
        m_first = a.take<int>(n); m_cost = a.take<int>(n);                  // the sparse matcher's records, one per left key point
        rank = a.take<unsigned>(std::max(n, 2 * cells_needed) + 4);         // flags, scanned in place
        taken = a.take<uint8_t>(n); owner = a.take<int>(n);                 // pixels that hold a support point / that a resampling slot claims
    };
    Arena measure;
    layout(measure);
    Arena a;
    a.cap = measure.used + 256;
    if (a.cap > ctx->arena_bytes) {
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->arena) SVX_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
        if (hipMalloc(&ctx->arena, a.cap) != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "svx_high_perf_stereo: %zu bytes of device memory", a.cap);
        ctx->arena_bytes = a.cap;
    }
    a.base = (char *)ctx->arena;
    layout(a);

    StageClock clk(ctx);
    // crop to multiples of 16, equalise, gradient, high-gradient point set (.cpp:245-280)
    SVX_TRY(svx_crop16(ctx, left, rows, cols, 1, L));
    SVX_TRY(svx_crop16(ctx, right, rows, cols, 1, Rr));
    // the frame's size is a multiple of 256, so the pair's images lie one behind the other in the arena and one launch serves both
    if (Rr != L + n || Ra != La + n || cR != cL + n) return svx_fail(ctx, SVX_ERR_HIP, "svx_high_perf_stereo: arena layout");
    if (prm->applyHistEqualization) SVX_TRY(svx_equalize_hist(ctx, L, L, R, Cc, 2));
    clk.mark(0);
    SVX_TRY(svx_laplacian_abs(ctx, L, grad, R, Cc, 1, prm->kernelSize, (double)prm->scale, (double)prm->delta));
    int off[2] = {0, 0};
    SVX_TRY(svx_high_gradient_points(ctx, grad, R, Cc, 1, prm->gradThreshold, pts, (int)n, off));
    const int n_hg = off[1];
    // the high-gradient points are most of what goes back to the host (.cpp:624-637): their copy starts now, beside the frame's kernels
    const int take_hg = std::min(n_hg, capacity);
    int32_t *stage = nullptr;
    {
        void *pin = nullptr;
        const size_t room = (size_t)std::min((long)capacity, (long)n_hg + (long)n) * 8;
        if (room) SVX_TRY(svx_side_reserve(ctx, room, &pin));
        stage = (int32_t *)pin;
        if (take_hg) SVX_TRY(svx_side_copy_to_host(ctx, stage, pts, (size_t)take_hg * 8));
    }
    // .cpp:289-292: the blur kernel size is `(int)applyBlur` = 1 (.cpp:207), i.e. the blur is the identity

    clk.mark(1);
    // altered copies for the detector only (.cpp:296-335)
    SVX_HIP(ctx, hipMemcpyAsync(La, L, 2 * n, hipMemcpyDeviceToDevice, ctx->stream));
    if (prm->invertRows || prm->traceLines) {
        SVX_TRY(svx_alter_rows(ctx, La, R, Cc, 1, prm->invertRows, prm->nbRows, prm->traceLines, prm->nbLines, prm->lineSize));
        SVX_TRY(svx_alter_rows(ctx, Ra, R, Cc, 1, prm->invertRows, prm->nbRows, prm->traceLines, prm->nbLines, prm->lineSize));
    }

    // exFAST on the altered copies (left with suppression, right without, .cpp:358-359), census of the plain ones (.cpp:371-387)
    int offl[2], offr[2];
    SVX_TRY(svx_exfast_detect(ctx, La, R, Cc, 1, prm->minThreshold, (float)prm->adaptivity, SVX_EXFAST_NONMAX, kpl, respl, (int)n, offl));
    SVX_TRY(svx_exfast_detect(ctx, Ra, R, Cc, 1, prm->minThreshold, (float)prm->adaptivity, 0, kpr, respr, (int)n, offr));
    const int nl = offl[1], nr = offr[1];
    clk.mark(2);
    SVX_TRY(svx_census5x5_u8(ctx, L, cL, R, Cc, 2, SVX_CENSUS_SSE24 | SVX_CENSUS_ZERO_REST));

    clk.mark(3);
    // sparse matching of the key points (.cpp:354-355, :390): the detector's lists stay on the device, where they are in
    // raster order already; the matcher's records never leave it either
    if (nl && nr)
        SVX_TRY(svx_sparse_match_lists(ctx, cL, cR, R, Cc, kpl, nl, kpr, nr, prm->maxDisp, 1.0f, (float)prm->uniqueness, prm->leftRightStep, m_first, m_cost));

    clk.mark(4);
    // support points of the mesh, in insertion order, repeated positions dropped (.cpp:434-445): a ranked write on the device
    {   // what the frame starts from, in one launch: prior and claimed pixels zero, final costs 25 (.cpp:448), no pixel owned by a resampling slot
        FillRanges fr = {};
        fr.p[0] = (uint32_t *)disparities; fr.words[0] = (long)n; fr.value[0] = 0u;
        fr.p[1] = (uint32_t *)taken; fr.words[1] = (long)(n / 4); fr.value[1] = 0u;
        fr.p[2] = (uint32_t *)final_costs; fr.words[2] = (long)(n / 4); fr.value[2] = 0x19191919u;
        fr.p[3] = (uint32_t *)owner; fr.words[3] = prm->nIters > 1 ? (long)n : 0; fr.value[3] = 0x7fffffffu;
        fill_ranges_kernel<<<std::min(svx_cdiv((long)n / 4, 256), 2048u), 256, 0, ctx->stream>>>(fr);
        SVX_LAUNCH_CHECK(ctx, "fill_ranges_kernel");
    }
    int n_sup = 0;
    auto read_total = [&](int n_flags, int *total) -> int {       // scans rank[0 .. n_flags) in place and brings the total to the host
        svx_mail *mail = nullptr;
        unsigned seq = 0;
        int r = svx_mail_next(ctx, &mail, &seq);
        if (r) return r;
        scan4_kernel<<<1, 1024, 0, ctx->stream>>>(rank, n_flags, rank, mail, seq);
        SVX_LAUNCH_CHECK(ctx, "scan4_kernel");
        const unsigned *got = nullptr;
        r = svx_mail_wait(ctx, seq, &got);
        if (r) return r;
        *total = (int)got[0];
        return SVX_OK;
    };
    if (nl && nr) {
        support_flags_kernel<<<svx_cdiv(nl, 256), 256, 0, ctx->stream>>>(m_first, nl, rank);
        SVX_LAUNCH_CHECK(ctx, "support_flags_kernel");
        SVX_TRY(read_total(nl, &n_sup));
        support_emit_kernel<<<svx_cdiv(nl, 256), 256, 0, ctx->stream>>>(kpl, kpr, m_first, nl, rank, Cc, sup_dev, disparities, taken);
        SVX_LAUNCH_CHECK(ctx, "support_emit_kernel");
    }

    // final maps: the caller's disparity map (input-sized, .cpp:190) restricted to the cropped frame (the costs were started above)
    SVX_HIP(ctx, hipMemcpy2DAsync(final_disp, (size_t)Cc * 4, disparity_map, (size_t)cols * 4, (size_t)Cc * 4, R, hipMemcpyDeviceToDevice, ctx->stream));

    clk.mark(5);
    // By default the triangulation runs on the device, from the whole list each time (the mesh is a function of the point
    // list alone, so this gives the triangles a growing mesh would hold).  With the option off the mesh lives on the host
    // across the iterations, as the reference's Fade_2D object does: the sparse matcher's points are triangulated at once,
    // the resampling step's points are inserted into what is there.
    struct MeshHolder {
        svx_mesh *m = nullptr;
        ~MeshHolder() { svx_mesh_destroy(m); }
    } mesh;
    int in_mesh = 0;                         // support points already handed to the host mesh
    const bool mesh_on_device = svx_opt(ctx, SVX_OPT_MESH_ON_DEVICE) != 0 && R <= 8192 && Cc <= 8192;
    if (!mesh_on_device && prm->recordFullDisp && prm->nIters > 0 && svx_mesh_create(&mesh.m) != SVX_OK)
        return svx_fail(ctx, SVX_ERR_NOMEM, "svx_high_perf_stereo: mesh");
    for (int iter = 1; iter <= prm->nIters; ++iter) {                                                   // .cpp:451
        // plane table and interpolation of the prior over the whole frame (.cpp:454-466, :532-554)
        if (prm->recordFullDisp) {
            int n_tri = 0;
            if (mesh_on_device) SVX_TRY(svx_mesh_plane_prior_device(ctx, sup_dev, n_sup, disparities, R, Cc, disparities, &n_tri));
            else {
                std::vector<int32_t> fresh(2 * (size_t)(n_sup - in_mesh) + 2);
                if (n_sup > in_mesh) {
                    SVX_HIP(ctx, hipMemcpyAsync(fresh.data(), sup_dev + 2 * (size_t)in_mesh, (size_t)(n_sup - in_mesh) * 8, hipMemcpyDeviceToHost, ctx->stream));
                    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
                }
                if (svx_mesh_insert(mesh.m, fresh.data(), n_sup - in_mesh) != SVX_OK)
                    return svx_fail(ctx, SVX_ERR_UNSUPPORTED, "svx_high_perf_stereo: mesh insertion failed");
                in_mesh = n_sup;
                SVX_TRY(svx_mesh_prior(ctx, mesh.m, disparities, R, Cc, disparities, &n_tri));
            }
        }
        clk.mark(6);
        if (const int band = svx_opt(ctx, SVX_OPT_PRIOR_BAND_RADIUS); band > 0) {                       // build extension: search d0 - band .. d0 + band
            SVX_HIP(ctx, hipMemsetAsync(costs, t_high & 0xff, n, ctx->stream));                         // .cpp:573
            SVX_TRY(svx_cost_eval_band(ctx, cL, cR, R, Cc, pts, n_hg, disparities, costs, band, prm->maxDisp));
            SVX_TRY(svx_refine(ctx, pts, n_hg, disparities, costs, R, Cc, t_low, t_high, occ, final_disp, final_costs,
                               conf_x, conf_y, conf_d, conf_c, inv_x, inv_y, inv_c));
        } else                                                                                          // .cpp:573-576: the costs go from the evaluation
            SVX_TRY(svx_cost_eval_refine(ctx, cL, cR, pts, n_hg, disparities, R, Cc, t_low, t_high, occ, final_disp, final_costs,     // straight into the refinement
                                         conf_x, conf_y, conf_d, conf_c, inv_x, inv_y, inv_c));
        clk.mark(7);
        if (iter != prm->nIters) {                                                                      // .cpp:616-621
            const int gw = Cc / (int)occ + 1, gh = R / (int)occ + 1;
            const int n_slots = 2 * gw * gh;
            SVX_TRY(svx_support_resampling(ctx, cL, cR, R, Cc, 5, prm->costAggrWindowSize, gw, gh, conf_x, conf_y, conf_d, conf_c,
                                           inv_x, inv_y, inv_c, t_low, t_high, prm->maxDisp, disparities, rs_xy, rs_d, rs_valid));
            // the accepted slots join the support list in slot order
            resample_owner_kernel<<<svx_cdiv(n_slots, 256), 256, 0, ctx->stream>>>(rs_xy, rs_valid, n_slots, Cc, taken, owner);
            SVX_LAUNCH_CHECK(ctx, "resample_owner_kernel");
            resample_flags_kernel<<<svx_cdiv(n_slots, 256), 256, 0, ctx->stream>>>(rs_xy, rs_valid, n_slots, Cc, taken, owner, rank);
            SVX_LAUNCH_CHECK(ctx, "resample_flags_kernel");
            int added = 0;
            SVX_TRY(read_total(n_slots, &added));
            resample_emit_kernel<<<svx_cdiv(n_slots, 256), 256, 0, ctx->stream>>>(rs_xy, n_slots, Cc, rank, n_sup, sup_dev, taken, owner);
            SVX_LAUNCH_CHECK(ctx, "resample_emit_kernel");
            n_sup += added;
            occ = std::max(1u, occ / 2);
            clk.mark(8);
        }
    }

    // the support points themselves end up in the map, truncated to integers, and in the point list (.cpp:624-637)
    if (n_sup) {
        vertex_disparities_kernel<<<svx_cdiv(n_sup, 256), 256, 0, ctx->stream>>>(sup_dev, n_sup, Cc, disparities, final_disp);
        SVX_LAUNCH_CHECK(ctx, "vertex_disparities_kernel");
    }
    SVX_HIP(ctx, hipMemcpy2DAsync(disparity_map, (size_t)cols * 4, final_disp, (size_t)Cc * 4, (size_t)Cc * 4, R, hipMemcpyDeviceToDevice, ctx->stream));

    // computed points: the high-gradient set in raster order, then the mesh vertices
    *n_points = n_hg + n_sup;
    const int take_sup = std::max(0, std::min(n_sup, capacity - n_hg));
    // the mesh vertices follow the high-gradient points in the same pinned block (a transfer straight into the caller's pageable
    // memory is staged by the runtime in small pieces and takes three times as long)
    if (take_sup) SVX_HIP(ctx, hipMemcpyAsync(stage + 2 * (size_t)take_hg, sup_dev, (size_t)take_sup * 8, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SVX_TRY(svx_side_wait(ctx));
    if (take_hg) memcpy(points_xy_host, stage, (size_t)take_hg * 8);
    if (take_sup) memcpy(points_xy_host + 2 * (size_t)n_hg, stage + 2 * (size_t)take_hg, (size_t)take_sup * 8);
    clk.mark(9);
    return SVX_OK;
}

// Frames are independent, and a frame's host share (the triangulation above all) is what limits one host thread: the
// batch entry point runs svx_high_perf_stereo on n_threads host threads, each with a context and a stream of its own
// (kept in the caller's context between calls), which take the frames in turn.
SVX_API int svx_high_perf_stereo_batch(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                                       const svx_stereo_parameters *prm, float *disparity_maps, int32_t *points_xy_host,
                                       int capacity_per_frame, int *n_points_host, int n_threads) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, left && right, SVX_ERR_EMPTY, "Input images are empty!");
    SVX_REQUIRE(ctx, prm && disparity_maps && n_points_host, SVX_ERR_INVALID_ARG, "svx_high_perf_stereo_batch: null pointer");
    SVX_REQUIRE(ctx, n_frames >= 1 && capacity_per_frame >= 0 && (capacity_per_frame == 0 || points_xy_host), SVX_ERR_INVALID_ARG,
                "svx_high_perf_stereo_batch: bad frame count or point buffer");
    if (n_threads <= 0) n_threads = 4;
    n_threads = std::min(std::min(n_threads, n_frames), 64);
    if (n_threads > ctx->n_workers) {
        svx_ctx **w = (svx_ctx **)realloc(ctx->workers, sizeof(svx_ctx *) * (size_t)n_threads);
        if (!w) return svx_fail(ctx, SVX_ERR_NOMEM, "svx_high_perf_stereo_batch: worker table");
        ctx->workers = w;
        while (ctx->n_workers < n_threads) {
            svx_ctx *c = nullptr;
            rc = svx_create(ctx->device, &c);
            if (rc) return svx_fail(ctx, rc, "svx_high_perf_stereo_batch: worker context: %s", svx_strerror(rc));
            if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
                (void)svx_destroy(c);
                return svx_fail(ctx, SVX_ERR_HIP, "svx_high_perf_stereo_batch: worker stream");
            }
            c->stream = c->own_stream;
            ctx->workers[ctx->n_workers++] = c;
        }
    }
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));        // the inputs may still be in the making on the caller's stream
    const size_t n = (size_t)rows * cols;
    std::atomic<int> next(0), first_error(0);
    std::vector<int> codes(n_threads, SVX_OK);
    auto work = [&](int t) {
        svx_ctx *c = ctx->workers[t];
        for (int o = 0; o < SVX_OPT_COUNT; ++o) c->opt[o] = ctx->opt[o];
        for (int f = next.fetch_add(1); f < n_frames && !first_error.load(); f = next.fetch_add(1)) {
            const int r = svx_high_perf_stereo(c, left + f * n, right + f * n, rows, cols, prm, disparity_maps + f * n,
                                               points_xy_host ? points_xy_host + 2 * (size_t)f * capacity_per_frame : nullptr,
                                               capacity_per_frame, n_points_host + f);
            if (r) { codes[t] = r; first_error.store(1); }
        }
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; ++t) threads.emplace_back(work, t);
    work(0);
    for (auto &th : threads) th.join();
    for (int t = 0; t < n_threads; ++t)
        if (codes[t]) return svx_fail(ctx, codes[t], "svx_high_perf_stereo_batch: %s", ctx->workers[t]->err);
    return SVX_OK;
}

SVX_API int svx_pipeline_stage_times(const svx_ctx *ctx, double *ms, int n) {
    if (!ctx || !ms || n < 0) return SVX_ERR_INVALID_ARG;
    for (int i = 0; i < n && i < SVX_PIPELINE_STAGES; ++i) ms[i] = ctx->stage_ms[i];
    return SVX_OK;
}
