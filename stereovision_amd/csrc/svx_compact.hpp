// Raster-order stream compaction over a batch of frames (gfx950).
//
// A predicate pred(x, y, frame) selects pixels; the selected ones are handed to emit(slot, x, y, frame)
// with slot = their rank in (frame, y, x) order, which is the order a sequential double loop over the
// image pushes them into a vector.  Three launches: per-line counts, one-block exclusive scan over the
// lines (which also hands the per-frame totals to the host), per-line ranked write.  The predicate is evaluated twice, so it should be a cheap read.
#pragma once
#include "svx_common.hpp"

namespace {          // one private copy per translation unit that includes this header

// Running total of x over the wavefront's lanes, this lane included: inside the rows of 16 lanes by DPP shifts, then the last
// lane of a row (of the first two rows) is added to the rows behind it.
__device__ __forceinline__ unsigned svx_wave_scan_u32(unsigned x) {
    const int v = (int)x;
    int t = v + __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);                    // row_shr:1
    t += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);                            // row_shr:2
    t += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, true);                            // row_shr:3
    t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xe, false);                           // row_shr:4, lanes 4..15 of a row
    t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xc, false);                           // row_shr:8, lanes 8..15
    t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xa, 0xf, false);                           // row_bcast:15, rows 1 and 3
    t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xc, 0xf, false);                           // row_bcast:31, rows 2 and 3
    return (unsigned)t;
}

template <class Pred>
__global__ __launch_bounds__(256) void compact_count_kernel(Pred pred, int rows, int cols, unsigned *__restrict__ line_count) {
    __shared__ unsigned wave_sum[4];
    const int line = blockIdx.x, frame = line / rows, y = line - frame * rows;
    unsigned c = 0;
    for (int x = threadIdx.x; x < cols; x += 256) c += pred(x, y, frame) ? 1u : 0u;
    for (int o = 32; o; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) line_count[line] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// In-place exclusive scan of line_count[0..n_lines) by one block; line_count[n_lines] receives the total.  Then the frames'
// offsets (frame f starts at line f * rows): frame_off[0 .. n_frames], and to the host as well when `mail` is given (n_frames < 31).
__global__ __launch_bounds__(1024) void compact_scan_kernel(unsigned *__restrict__ line_count, int n_lines, int rows, int n_frames,
                                                            unsigned *__restrict__ frame_off, svx_mail *mail, unsigned seq) {
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_lines; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned v = i < n_lines ? line_count[i] : 0u;
        const unsigned s = svx_wave_scan_u32(v);          // inclusive scan inside the wavefront
        if (lane == 63) wave_tot[wave] = s;
        __syncthreads();
        unsigned before = carry;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (i < n_lines) line_count[i] = before + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) line_count[n_lines] = carry;
    __syncthreads();
    for (int f = threadIdx.x; f <= n_frames; f += 1024) {
        const unsigned off = line_count[(long)f * rows];
        frame_off[f] = off;
        if (mail) { mail->v[f] = off; __threadfence_system(); }
    }
    if (mail) {
        __syncthreads();
        if (threadIdx.x == 0) svx_mail_post(mail, seq);
    }
}

// Exclusive scan of src[0..n) into v[0..n) by one workgroup (src == v: in place), sixteen consecutive values per thread and
// round (four 16-byte loads a lane, all in flight); v[n] receives the total, and so does the host (mail->v[0]) when `mail` is
// given.  v and src must be 16-byte aligned.
__global__ __launch_bounds__(1024) void scan4_kernel(unsigned *v, int n, const unsigned *src, svx_mail *mail, unsigned seq) {
    __shared__ unsigned wave_tot[2][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned carry = 0;
    int buf = 0;
    for (int base = 0; base < n; base += 16384, buf ^= 1) {
        const int i0 = base + 16 * (int)threadIdx.x;
        unsigned x[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + 4 * q;
            uint4 t = make_uint4(0, 0, 0, 0);
            if (i + 3 < n) t = *(const uint4 *)(src + i);
            else {
                if (i < n) t.x = src[i];
                if (i + 1 < n) t.y = src[i + 1];
                if (i + 2 < n) t.z = src[i + 2];
            }
            x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
        }
        unsigned sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += x[q];
        const unsigned incl = svx_wave_scan_u32(sum);     // inclusive scan of the threads' sums inside the wavefront
        if (lane == 63) wave_tot[buf][wave] = incl;
        __syncthreads();                                  // (the buffers alternate, so one meeting a round is enough)
        unsigned before = carry, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned t = wave_tot[buf][w];
            before += w < wave ? t : 0u;
            all += t;
        }
        unsigned e = before + incl - sum;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + 4 * q;
            const uint4 out = make_uint4(e, e + x[4 * q], e + x[4 * q] + x[4 * q + 1], e + x[4 * q] + x[4 * q + 1] + x[4 * q + 2]);
            e = out.w + x[4 * q + 3];
            if (i + 3 < n) *(uint4 *)(v + i) = out;
            else {
                if (i < n) v[i] = out.x;
                if (i + 1 < n) v[i + 1] = out.y;
                if (i + 2 < n) v[i + 2] = out.z;
            }
        }
        carry += all;
    }
    __syncthreads();                                      // every thread has read what it scans before the total lands behind it
    if (threadIdx.x == 0) {
        v[n] = carry;
        if (mail) { mail->v[0] = carry; svx_mail_post(mail, seq); }
    }
}

template <class Pred, class Emit>
__global__ __launch_bounds__(256) void compact_write_kernel(Pred pred, Emit emit, int rows, int cols,
                                                            const unsigned *__restrict__ line_off, unsigned capacity) {
    __shared__ unsigned wave_cnt[2][4];
    const int line = blockIdx.x, frame = line / rows, y = line - frame * rows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned base = line_off[line];
    if (line_off[line + 1] == base) return;               // nothing selected on this line
    int buf = 0;
    for (int x0 = 0; x0 < cols; x0 += 256, buf ^= 1) {
        const int x = x0 + threadIdx.x;
        const bool f = x < cols && pred(x, y, frame);
        const unsigned long long m = __ballot(f);
        if (lane == 0) wave_cnt[buf][wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = 0, all = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned c = wave_cnt[buf][w];
            before += w < wave ? c : 0u;
            all += c;
        }
        if (f) {
            const unsigned slot = base + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (slot < capacity) emit(slot, x, y, frame);
        }
        base += all;
    }
}

// Device bytes svx_compact_raster needs at `area`.
static inline size_t svx_compact_bytes(int rows, int n_frames) {
    return ((size_t)rows * n_frames + 1 + (size_t)n_frames + 1) * sizeof(unsigned);
}

// Runs the launches on the context's stream and returns the per-frame offsets
// (frame_offsets_host[f] .. frame_offsets_host[f+1] are frame f's slots; n_frames+1 entries) as soon as the
// scan has produced them (the ranked write may still be running on the stream).  Slots >= capacity are counted but not emitted.  `area` is device memory of
// svx_compact_bytes(rows, n_frames) bytes that the caller reserved from the context's workspace.
template <class Pred, class Emit>
int svx_compact_raster(svx_ctx *ctx, void *area, int rows, int cols, int n_frames, Pred pred, Emit emit, int capacity,
                       int *frame_offsets_host, const char *what) {
    const long n_lines = (long)rows * n_frames;
    if (n_lines > 0x7fffffffL / 2) return svx_fail(ctx, SVX_ERR_UNSUPPORTED, "%s: too many image lines", what);
    unsigned *line_off = (unsigned *)area;
    unsigned *frame_off = line_off + n_lines + 1;
    // the offsets reach the host straight from the scan (up to 30 frames), or by a copy behind the last launch
    svx_mail *mail = nullptr;
    unsigned seq = 0;
    void *pin = nullptr;
    int rc = n_frames < 31 ? svx_mail_next(ctx, &mail, &seq) : svx_pinned_reserve(ctx, (size_t)(n_frames + 1) * sizeof(unsigned), &pin);
    if (rc) return rc;
    compact_count_kernel<Pred><<<(unsigned)n_lines, 256, 0, ctx->stream>>>(pred, rows, cols, line_off);
    SVX_LAUNCH_CHECK(ctx, what);
    compact_scan_kernel<<<1, 1024, 0, ctx->stream>>>(line_off, (int)n_lines, rows, n_frames, frame_off, mail, seq);
    SVX_LAUNCH_CHECK(ctx, what);
    if (capacity > 0) {
        compact_write_kernel<Pred, Emit><<<(unsigned)n_lines, 256, 0, ctx->stream>>>(pred, emit, rows, cols, line_off, (unsigned)capacity);
        SVX_LAUNCH_CHECK(ctx, what);
    }
    const unsigned *got = nullptr;
    if (mail) {
        rc = svx_mail_wait(ctx, seq, &got);
        if (rc) return rc;
    } else {
        SVX_HIP(ctx, hipMemcpyAsync(pin, frame_off, (size_t)(n_frames + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        got = (const unsigned *)pin;
    }
    for (int f = 0; f <= n_frames; ++f) frame_offsets_host[f] = (int)got[f];
    return SVX_OK;
}

}  // namespace
