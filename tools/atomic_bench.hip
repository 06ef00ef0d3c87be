// Stand-alone micro-benchmark (measurement aid, not part of libsvx): throughput of no-return global
// atomic umin / plain stores with the access pattern of a per-wavefront-row flush (64 consecutive words).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/atomic_bench tools/atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// Every wavefront walks `rows` rows; per row it issues one 64-lane atomic to row*pitch + xoff + lane.
// `adders` wavefronts (spread over blocks) hit each address range, as neighbouring x-blocks would.
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *buf, int rows, int pitch, int n_xblocks, int frames, uint32_t seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xb = (blockIdx.x * 4 + wave) % n_xblocks;
    const int frame = ((blockIdx.x * 4 + wave) / n_xblocks) % frames;   // the two chunk parities of a frame share its rows
    uint32_t *base = buf + (size_t)frame * rows * pitch + xb * 16 + lane;
    uint32_t v = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
    for (int r = 0; r < rows; ++r) {
        v = v * 1664525u + 1013904223u;
        if (MODE == 0) __hip_atomic_fetch_min(base + (size_t)r * pitch, v >> 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) __hip_atomic_fetch_min(base + (size_t)r * pitch, v >> 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) base[(size_t)r * pitch] = v;
        else if (MODE == 3) { if ((v >> 4) < base[(size_t)r * pitch]) __hip_atomic_fetch_min(base + (size_t)r * pitch, v >> 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

int main() {
    const int rows = 368, pitch = 1232 + 256, n_xblocks = 77, frames = 64;
    uint32_t *buf;
    const size_t bytes = (size_t)frames * rows * pitch * 4;
    CHK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int waves = n_xblocks * frames * 2;     // two chunk parities
    const int blocks = waves / 4;
    // largest index touched: last frame, last row, last x-block, lane 63
    const size_t max_index = ((size_t)(frames - 1) * rows + (rows - 1)) * pitch + (n_xblocks - 1) * 16 + 63;
    if (waves % 4 != 0 || max_index * 4 >= bytes) { printf("bad geometry\n"); return 1; }
    const char *names[4] = {"atomic umin agent", "atomic umin workgroup", "plain store dword", "load-filtered atomic umin"};
    for (int mode = 0; mode < 4; ++mode) {
        CHK(hipMemset(buf, 0xff, bytes));
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0));
            if (mode == 0) k<0><<<blocks, 256>>>(buf, rows, pitch, n_xblocks, frames, rep);
            if (mode == 1) k<1><<<blocks, 256>>>(buf, rows, pitch, n_xblocks, frames, rep);
            if (mode == 2) k<2><<<blocks, 256>>>(buf, rows, pitch, n_xblocks, frames, rep);
            if (mode == 3) k<3><<<blocks, 256>>>(buf, rows, pitch, n_xblocks, frames, rep);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)waves * rows * 64;
            printf("%-28s rep %d: %.3f ms for %.0f M lane-ops (%.2f GB) -> %.2f TB/s\n", names[mode], rep, ms, n / 1e6, n * 4 / 1e9, n * 4 / ms / 1e9);
        }
    }
    return 0;
}
