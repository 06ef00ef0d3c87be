// Measurement aid: sustained issue rate of the VALU instructions the dense matcher is built from,
// the peak against which the popcount-bound kernel is priced.  No counterpart in the reference.
#include "svx_common.hpp"

// 8 independent chains, 16 VALU instructions per body, 8 bodies per loop trip; the loop counter and k live in SGPRs.
#define PROBE_BODY(I0, I1)                                                                              \
    asm volatile(I0(8, 16) I0(9, 17) I0(10, 18) I0(11, 19) I0(12, 20) I0(13, 21) I0(14, 22) I0(15, 23)    \
                 I1(0, 8) I1(1, 9) I1(2, 10) I1(3, 11) I1(4, 12) I1(5, 13) I1(6, 14) I1(7, 15)            \
                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7),        \
                   "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7)         \
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(k))

#define XOR_(t, a) "v_xor_b32 %" #t ", %" #a ", %24\n\t"
#define BCNT_(s, t) "v_bcnt_u32_b32 %" #s ", %" #t ", %" #s "\n\t"
#define ADD_(s, t) "v_add_u32 %" #s ", %" #t ", %" #s "\n\t"
#define ADD3_(s, t) "v_add3_u32 %" #s ", %" #t ", %" #s ", %" #s "\n\t"
#define MIN_(s, t) "v_min_u32 %" #s ", %" #t ", %" #s "\n\t"
#define DPPADD_(s, t) "v_add_u32_dpp %" #s ", %" #t ", %" #s " row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define PKADD_(s, t) "v_pk_add_u16 %" #s ", %" #t ", %" #s "\n\t"
#define MAD24_(s, t) "v_mad_u32_u24 %" #s ", %" #t ", 64, %" #s "\n\t"
#define SUBX_(t, a) "v_sub_u32 %" #t ", %" #a ", %24\n\t"
#define LSHLADD_(s, t) "v_lshl_add_u32 %" #s ", %" #t ", 8, %" #s "\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void valu_rate_kernel(uint32_t *__restrict__ sink, int iters) {
    uint32_t a0 = threadIdx.x * 2654435761u + 1u, a1 = a0 ^ 0x9e3779b9u, a2 = a0 * 3u, a3 = a0 + 77u;
    uint32_t a4 = ~a0, a5 = a0 >> 3, a6 = a0 << 5, a7 = a0 ^ 0x55555555u;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    uint32_t t0 = 1, t1 = 2, t2 = 3, t3 = 4, t4 = 5, t5 = 6, t6 = 7, t7 = 8;
    uint32_t k = blockIdx.x + 1u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // 128 VALU instructions per loop trip
            if (MODE == 0) PROBE_BODY(XOR_, BCNT_);
            else if (MODE == 1) PROBE_BODY(XOR_, ADD_);
            else if (MODE == 2) PROBE_BODY(SUBX_, ADD3_);
            else if (MODE == 3) PROBE_BODY(XOR_, MIN_);
            else if (MODE == 4) PROBE_BODY(XOR_, DPPADD_);
            else if (MODE == 5) PROBE_BODY(XOR_, PKADD_);
            else if (MODE == 6) PROBE_BODY(XOR_, MAD24_);
            else if (MODE == 7) PROBE_BODY(XOR_, LSHLADD_);
        }
        k += 0x01010101u;
    }
    const uint32_t s = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if (s == 0xdeadbeefu) sink[0] = s;   // keeps the chains live; practically never taken
}

// mode 0 = v_xor_b32 + v_bcnt_u32_b32 (the dense matcher's inner pair); modes 1..7 pair v_xor/v_sub with
// v_add_u32, v_add3_u32, v_min_u32, DPP v_add_u32, v_pk_add_u16, v_mad_u32_u24, v_lshl_add_u32.
// Returns the sustained rate in lane-instructions per second (64 per wavefront instruction) with
// `waves_per_simd` wavefronts resident per SIMD on every CU.  Synchronous.
SVX_API int svx_probe_valu_rate(svx_ctx *ctx, int mode, int waves_per_simd, int iters, double *lane_ops_per_s) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, lane_ops_per_s != nullptr, SVX_ERR_INVALID_ARG, "svx_probe_valu_rate: null result");
    SVX_REQUIRE(ctx, waves_per_simd >= 1 && waves_per_simd <= 8 && iters >= 1 && mode >= 0 && mode <= 7,
                SVX_ERR_INVALID_ARG, "svx_probe_valu_rate: bad parameters");
    hipDeviceProp_t prop;
    SVX_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    void *ws;
    rc = svx_ws_reserve(ctx, 256, &ws);
    if (rc) return rc;
    const int blocks = prop.multiProcessorCount * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    SVX_HIP(ctx, hipEventCreate(&e0));
    SVX_HIP(ctx, hipEventCreate(&e1));
    auto launch = [&](int n) {
        uint32_t *p = (uint32_t *)ws;
        switch (mode) {
            case 0: valu_rate_kernel<0><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
            case 1: valu_rate_kernel<1><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
            case 2: valu_rate_kernel<2><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
            case 3: valu_rate_kernel<3><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
            case 4: valu_rate_kernel<4><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
            case 5: valu_rate_kernel<5><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
            case 6: valu_rate_kernel<6><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
            default: valu_rate_kernel<7><<<blocks, 256, 0, ctx->stream>>>(p, n); break;
        }
    };
    launch(iters / 8 + 1);   // warm-up
    SVX_HIP(ctx, hipEventRecord(e0, ctx->stream));
    launch(iters);
    SVX_HIP(ctx, hipEventRecord(e1, ctx->stream));
    SVX_HIP(ctx, hipEventSynchronize(e1));
    SVX_LAUNCH_CHECK(ctx, "valu_rate_kernel");
    float ms = 0.f;
    SVX_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *lane_ops_per_s = (double)blocks * 256.0 * (double)iters * 128.0 / ((double)ms * 1e-3);
    return SVX_OK;
}
