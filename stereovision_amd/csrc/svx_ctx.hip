// Context, error strings and device-memory helpers of libsvx.
#include "svx_common.hpp"
#include <cstdlib>
#include <new>

int svx_enter(svx_ctx *ctx) {
    if (!ctx) return SVX_ERR_INVALID_ARG;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    return SVX_OK;
}

int svx_ws_reserve(svx_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->ws_bytes) {
        // Growing synchronises with pending work that may still use the old block.
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ws) SVX_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
        size_t want = bytes + (bytes >> 2) + 4096;
        hipError_t e = hipMalloc(&ctx->ws, want);
        if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "workspace of %zu bytes: %s", want, hipGetErrorString(e));
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return SVX_OK;
}

int svx_block_reserve(svx_ctx *ctx, void **block, size_t *block_bytes, size_t bytes, const char *what) {
    if (bytes <= *block_bytes) return SVX_OK;
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));          // pending work may still use the old block
    if (*block) SVX_HIP(ctx, hipFree(*block));
    *block = nullptr;
    *block_bytes = 0;
    const size_t want = bytes + (bytes >> 2) + 4096;
    hipError_t e = hipMalloc(block, want);
    if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "%s: %zu bytes of device memory: %s", what, want, hipGetErrorString(e));
    *block_bytes = want;
    return SVX_OK;
}

int svx_pinned_mark_read(svx_ctx *ctx) {
    if (!ctx->pinned_read) SVX_HIP(ctx, hipEventCreateWithFlags(&ctx->pinned_read, hipEventDisableTiming));
    SVX_HIP(ctx, hipEventRecord(ctx->pinned_read, ctx->stream));
    ctx->pinned_busy = true;
    return SVX_OK;
}

int svx_pinned_reserve(svx_ctx *ctx, size_t bytes, void **out) {
    if (ctx->pinned_busy) {                                    // an earlier call's copy out of the buffer may still be queued
        SVX_HIP(ctx, hipEventSynchronize(ctx->pinned_read));
        ctx->pinned_busy = false;
    }
    if (bytes > ctx->pinned_bytes) {
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pinned) SVX_HIP(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        size_t want = bytes + (bytes >> 2) + 4096;
        hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
        if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "pinned staging of %zu bytes: %s", want, hipGetErrorString(e));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return SVX_OK;
}

int svx_mail_next(svx_ctx *ctx, svx_mail **dev, unsigned *seq) {
    if (!ctx->mail_host) {
        void *h = nullptr, *d = nullptr;
        hipError_t e = hipHostMalloc(&h, sizeof(svx_mail), hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
            if (h) (void)hipHostFree(h);
            return svx_fail(ctx, SVX_ERR_NOMEM, "host memory for the kernels' counts: %s", hipGetErrorString(e));
        }
        memset(h, 0, sizeof(svx_mail));
        ctx->mail_host = (svx_mail *)h;
        ctx->mail_dev = (svx_mail *)d;
        ctx->mail_seq = 0;
    }
    *dev = ctx->mail_dev;
    *seq = ++ctx->mail_seq;
    if (*seq == 0) *seq = ++ctx->mail_seq;          // (0 is what the memory starts with)
    return SVX_OK;
}

int svx_mail_wait(svx_ctx *ctx, unsigned seq, const unsigned **values) {
    const unsigned *at = &ctx->mail_host->seq;
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(at, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xfff) == 0) {                  // now and then: is the stream still at work?
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipErrorNotReady) continue;
            if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_HIP, "waiting for a kernel's counts: %s", hipGetErrorString(e));
            if (__atomic_load_n(at, __ATOMIC_ACQUIRE) == seq) break;
            return svx_fail(ctx, SVX_ERR_HIP, "a kernel ended without leaving its counts");
        }
    }
    *values = ctx->mail_host->v;
    return SVX_OK;
}

int svx_side_wait(svx_ctx *ctx) {
    if (ctx->side_busy) {
        SVX_HIP(ctx, hipEventSynchronize(ctx->side_done));
        ctx->side_busy = false;
    }
    return SVX_OK;
}

int svx_side_reserve(svx_ctx *ctx, size_t bytes, void **host) {
    int rc = svx_side_wait(ctx);
    if (rc) return rc;
    if (!ctx->side_stream) {
        SVX_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        SVX_HIP(ctx, hipEventCreateWithFlags(&ctx->side_ready, hipEventDisableTiming));
        SVX_HIP(ctx, hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming));
    }
    if (bytes > ctx->side_pin_bytes) {
        if (ctx->side_pin) SVX_HIP(ctx, hipHostFree(ctx->side_pin));
        ctx->side_pin = nullptr;
        ctx->side_pin_bytes = 0;
        const size_t want = bytes + (bytes >> 2) + 4096;
        const hipError_t e = hipHostMalloc(&ctx->side_pin, want, hipHostMallocDefault);
        if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "pinned memory of %zu bytes: %s", want, hipGetErrorString(e));
        ctx->side_pin_bytes = want;
    }
    *host = ctx->side_pin;
    return SVX_OK;
}

int svx_side_copy_to_host(svx_ctx *ctx, void *host, const void *dev, size_t bytes) {
    SVX_HIP(ctx, hipEventRecord(ctx->side_ready, ctx->stream));
    SVX_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_ready, 0));
    SVX_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->side_stream));
    SVX_HIP(ctx, hipEventRecord(ctx->side_done, ctx->side_stream));
    ctx->side_busy = true;
    return SVX_OK;
}

// Options: built-in default and the environment variable that overrides it when a context is created.
static const struct { int builtin; const char *env; } svx_option_table[SVX_OPT_COUNT] = {
    /* SVX_OPT_MATRIX_CORES */ {-1, "SVX_MX"},
    /* SVX_OPT_LR_TWO_SWEEPS */ {0, "SVX_LR_TWO_SWEEPS"},
    /* SVX_OPT_SWEEP_WAVES */ {3, "SVX_SWEEP_OCC"},
    /* SVX_OPT_SWEEP_WAVES_BOTH */ {3, "SVX_SWEEP_OCC_BOTH"},
    /* SVX_OPT_CENSUS97_PACKED */ {2, "SVX_CENSUS97_PACKED"},
    /* SVX_OPT_MX_STRIP_ROWS */ {0, "SVX_MX_STRIP_ROWS"},
    /* SVX_OPT_STAGE_TIMES */ {0, "SVX_STAGE_TIMES"},
    /* SVX_OPT_PRIOR_BAND_RADIUS */ {0, "SVX_PRIOR_BAND_RADIUS"},
    /* SVX_OPT_MX_WIDE_TILES */ {-1, "SVX_MX_WIDE_TILES"},
    /* SVX_OPT_MESH_ON_DEVICE */ {1, "SVX_MESH_ON_DEVICE"},
    /* SVX_OPT_SUBPIXEL_TILES */ {2, "SVX_SUBPIXEL_TILES"},
};

int svx_option_default(int option) {
    if (option < 0 || option >= SVX_OPT_COUNT) return 0;
    const char *e = getenv(svx_option_table[option].env);
    return e && e[0] ? atoi(e) : svx_option_table[option].builtin;
}

SVX_API int svx_set_option(svx_ctx *ctx, int option, int value) {
    if (!ctx) return SVX_ERR_INVALID_ARG;
    SVX_REQUIRE(ctx, option >= 0 && option < SVX_OPT_COUNT, SVX_ERR_INVALID_ARG, "svx_set_option: unknown option");
    ctx->opt[option] = value;
    return SVX_OK;
}

SVX_API int svx_get_option(const svx_ctx *ctx, int option, int *value) {
    if (!value || option < 0 || option >= SVX_OPT_COUNT) return SVX_ERR_INVALID_ARG;
    *value = svx_opt(ctx, option);
    return SVX_OK;
}

SVX_API int svx_version(void) { return SVX_VERSION; }

SVX_API const char *svx_strerror(int status) {
    switch (status) {
        case SVX_OK: return "ok";
        case SVX_ERR_INVALID_ARG: return "invalid argument";
        case SVX_ERR_SIZE_MISMATCH: return "Mismatching image sizes";
        case SVX_ERR_EMPTY: return "Input images are empty!";
        case SVX_ERR_HIP: return "HIP runtime error";
        case SVX_ERR_NO_DEVICE: return "no usable gfx950 device";
        case SVX_ERR_UNSUPPORTED: return "unsupported parameter";
        case SVX_ERR_NOMEM: return "device workspace allocation failed";
        default: return "unknown svx status";
    }
}

SVX_API const char *svx_last_error(const svx_ctx *ctx) { return ctx ? ctx->err : "null context"; }

SVX_API int svx_create(int device, svx_ctx **out) {
    if (!out) return SVX_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SVX_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return SVX_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SVX_ERR_NO_DEVICE;
    // The code object holds gfx950 kernels only.
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SVX_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SVX_ERR_NO_DEVICE;
    svx_ctx *c = new (std::nothrow) svx_ctx();
    if (!c) return SVX_ERR_NOMEM;
    c->device = device;
    c->stream = nullptr;
    c->ws = nullptr;
    c->ws_bytes = 0;
    c->pinned = nullptr;
    c->pinned_bytes = 0;
    c->err[0] = 0;
    c->arena = nullptr;
    c->arena_bytes = 0;
    c->tri = nullptr;
    c->tri_bytes = 0;
    c->pinned_read = nullptr;
    c->pinned_busy = false;
    c->mesh_left_over = c->mesh_left_first = 0;
    for (int o = 0; o < SVX_OPT_COUNT; ++o) c->opt[o] = svx_option_default(o);
    c->workers = nullptr;
    c->n_workers = 0;
    c->own_stream = nullptr;
    c->static_scratch = nullptr;
    (void)svx_static_scratch_locate(c);     // on failure the pointer stays null and only the allocation-free entry point refuses to run
    c->err[0] = 0;
    *out = c;
    return SVX_OK;
}

SVX_API int svx_destroy(svx_ctx *ctx) {
    if (!ctx) return SVX_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < ctx->n_workers; ++i) (void)svx_destroy(ctx->workers[i]);
    free(ctx->workers);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->tri) (void)hipFree(ctx->tri);
    if (ctx->pinned_read) (void)hipEventDestroy(ctx->pinned_read);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->mail_host) (void)hipHostFree(ctx->mail_host);
    if (ctx->side_stream) {
        (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamDestroy(ctx->side_stream);
        (void)hipEventDestroy(ctx->side_ready);
        (void)hipEventDestroy(ctx->side_done);
    }
    if (ctx->side_pin) (void)hipHostFree(ctx->side_pin);
    delete ctx;
    return SVX_OK;
}

SVX_API int svx_set_stream(svx_ctx *ctx, void *hip_stream) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    // The workspace is shared by every call of the context: work queued on the old stream must not overlap work on the new one.
    if ((hipStream_t)hip_stream != ctx->stream) SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    return SVX_OK;
}

SVX_API int svx_synchronize(svx_ctx *ctx) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}

SVX_API int svx_malloc(svx_ctx *ctx, size_t bytes, void **dptr) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, dptr != nullptr, SVX_ERR_INVALID_ARG, "svx_malloc: null result pointer");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return SVX_OK;
}

SVX_API int svx_free(svx_ctx *ctx, void *dptr) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_HIP(ctx, hipFree(dptr));
    return SVX_OK;
}

SVX_API int svx_memset(svx_ctx *ctx, void *dptr, int value, size_t bytes) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return SVX_OK;
}

SVX_API int svx_memcpy_h2d(svx_ctx *ctx, void *dst, const void *src_host, size_t bytes) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_HIP(ctx, hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}

SVX_API int svx_memcpy_d2h(svx_ctx *ctx, void *dst_host, const void *src, size_t bytes) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_HIP(ctx, hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}
