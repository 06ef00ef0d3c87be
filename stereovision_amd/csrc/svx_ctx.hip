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

int svx_pinned_reserve(svx_ctx *ctx, size_t bytes, void **out) {
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
    *out = c;
    return SVX_OK;
}

SVX_API int svx_destroy(svx_ctx *ctx) {
    if (!ctx) return SVX_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    delete ctx;
    return SVX_OK;
}

SVX_API int svx_set_stream(svx_ctx *ctx, void *hip_stream) {
    if (!ctx) return SVX_ERR_INVALID_ARG;
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
