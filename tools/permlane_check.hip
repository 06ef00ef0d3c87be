// Prints what v_permlane32_swap / v_permlane16_swap do to lane ids (measurement aid, not part of libsvx).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
    unsigned l = threadIdx.x;
    unsigned a = 100 + l, b = 200 + l;
    v2u r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = r.x; o[64 + l] = r.y;
    v2u s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + l] = s.x; o[192 + l] = s.y;
}
__global__ void t32(unsigned *o, int iters) {
    unsigned a = threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a * 11, f = a * 13, g = a * 17, h = a * 19;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
    }
    if (a + b + c + d + e + f + g + h == 0xdeadbeef) o[0] = a;
}
__global__ void t16(unsigned *o, int iters) {
    unsigned a = threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a * 11, f = a * 13, g = a * 17, h = a * 19;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
    }
    if (a + b + c + d + e + f + g + h == 0xdeadbeef) o[0] = a;
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"p32 vdst'", "p32 src0'", "p16 vdst'", "p16 src0'"};
    for (int r = 0; r < 4; ++r) {
        printf("%s:", names[r]);
        for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * r + l]);
        printf("\n");
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int w : {1, 2, 4, 8}) {
            int iters = 2000;
            if (which == 0) t32<<<256 * w, 256>>>(d, 100); else t16<<<256 * w, 256>>>(d, 100);
            hipEventRecord(e0);
            if (which == 0) t32<<<256 * w, 256>>>(d, iters); else t16<<<256 * w, 256>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s %dw: %.2f cycles/instr\n", which ? "permlane16_swap" : "permlane32_swap", w, ms * 1e-3 * 2.4e9 / ((double)w * iters * 64));
        }
    return 0;
}
