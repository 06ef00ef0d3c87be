// Stand-alone micro-benchmark (measurement aid, not part of libsvx): cycles per wave64 instruction for
// the VALU opcodes the dense matcher can be built from, at 1/2/4/8 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_bench tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// 16 independent chains: acc_i = OP(acc_i, src_i [, k]).  One body = 16 instructions; 8 bodies per trip.
#define BODY(OPSTR)                                                                                   \
    asm volatile(OPSTR(0, 16) OPSTR(1, 17) OPSTR(2, 18) OPSTR(3, 19) OPSTR(4, 20) OPSTR(5, 21) OPSTR(6, 22) OPSTR(7, 23) \
                 OPSTR(8, 24) OPSTR(9, 25) OPSTR(10, 26) OPSTR(11, 27) OPSTR(12, 28) OPSTR(13, 29) OPSTR(14, 30) OPSTR(15, 31) \
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]),   \
                   "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]) \
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),           \
                   "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "s"(k) \
                 : "vcc")

#define OP_XOR(s, a) "v_xor_b32_e32 %" #s ", %" #a ", %" #s "\n\t"
#define OP_XORS(s, a) "v_xor_b32_e64 %" #s ", %" #s ", %32\n\t"
#define OP_ADD(s, a) "v_add_u32_e32 %" #s ", %" #a ", %" #s "\n\t"
#define OP_SUB(s, a) "v_sub_u32_e32 %" #s ", %" #a ", %" #s "\n\t"
#define OP_MIN(s, a) "v_min_u32_e32 %" #s ", %" #a ", %" #s "\n\t"
#define OP_BCNT(s, a) "v_bcnt_u32_b32 %" #s ", %" #a ", %" #s "\n\t"
#define OP_ADD3(s, a) "v_add3_u32 %" #s ", %" #a ", %" #s ", %" #a "\n\t"
#define OP_MAD24(s, a) "v_mad_u32_u24 %" #s ", %" #a ", 64, %" #s "\n\t"
#define OP_LSHLADD(s, a) "v_lshl_add_u32 %" #s ", %" #a ", 8, %" #s "\n\t"
#define OP_LSHLOR(s, a) "v_lshl_or_b32 %" #s ", %" #a ", 8, %" #s "\n\t"
#define OP_DPPADD(s, a) "v_add_u32_dpp %" #s ", %" #a ", %" #s " row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define OP_DPPMIN(s, a) "v_min_u32_dpp %" #s ", %" #a ", %" #s " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_DPPMOV(s, a) "v_mov_b32_dpp %" #s ", %" #a " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_PKADD(s, a) "v_pk_add_u16 %" #s ", %" #a ", %" #s "\n\t"
#define OP_PKMIN(s, a) "v_pk_min_u16 %" #s ", %" #a ", %" #s "\n\t"
#define OP_PERM(s, a) "v_perm_b32 %" #s ", %" #a ", %" #s ", %" #a "\n\t"
#define OP_CNDMASK(s, a) "v_cndmask_b32_e32 %" #s ", %" #a ", %" #s ", vcc\n\t"
#define OP_MIN3(s, a) "v_min3_u32 %" #s ", %" #a ", %" #s ", %" #a "\n\t"
#define OP_SAD(s, a) "v_sad_u32 %" #s ", %" #a ", %" #s ", %" #a "\n\t"
#define OP_XAD(s, a) "v_xad_u32 %" #s ", %" #a ", %" #s ", %" #a "\n\t"
#define OP_ADDSDWA(s, a) "v_add_u32_sdwa %" #s ", %" #a ", %" #s " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define OP_AND(s, a) "v_and_b32_e32 %" #s ", %" #a ", %" #s "\n\t"
#define OP_LSHR(s, a) "v_lshrrev_b32_e32 %" #s ", 1, %" #s "\n\t"
#define OP_BFE(s, a) "v_bfe_u32 %" #s ", %" #s ", 1, 30\n\t"
#define OP_MOV(s, a) "v_mov_b32_e32 %" #s ", %" #a "\n\t"
#define OP_ALIGNBIT(s, a) "v_alignbit_b32 %" #s ", %" #a ", %" #s ", 8\n\t"
#define OP_PKSUB(s, a) "v_pk_sub_i16 %" #s ", %" #a ", %" #s "\n\t"
#define OP_MADI24(s, a) "v_mad_i32_i24 %" #s ", %" #a ", -1, %" #s "\n\t"
#define OP_CMPGT(s, a) "v_cmp_gt_i32_e32 vcc, %" #a ", %" #s "\n\t"
#define OP_ADDC(s, a) "v_addc_co_u32_e32 %" #s ", vcc, %" #s ", %" #s ", vcc\n\t"
#define OP_XORIP(s, a) "v_xor_b32_e32 %" #a ", %32, %" #a "\n\t"
#define OP_BCNTONLY(s, a) "v_bcnt_u32_b32 %" #s ", %" #a ", %" #s "\n\t"
#define OP_SUBIP(s, a) "v_sub_u32_e32 %" #a ", %" #a ", %" #s "\n\t"
#define OP_XORBCNT(s, a) "v_xor_b32_e32 %" #a ", %" #a ", %" #s "\n\tv_bcnt_u32_b32 %" #s ", %" #a ", %" #s "\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *sink, int iters) {
    uint32_t s[16], a[16];
    for (int i = 0; i < 16; ++i) { s[i] = threadIdx.x * 2654435761u + i * 77u; a[i] = s[i] ^ (0x9e3779b9u * (i + 1)); }
    uint32_t kk = blockIdx.x + 1u;
    uint32_t k = kk;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) BODY(OP_XOR); else if (MODE == 1) BODY(OP_XORS); else if (MODE == 2) BODY(OP_ADD);
            else if (MODE == 3) BODY(OP_SUB); else if (MODE == 4) BODY(OP_MIN); else if (MODE == 5) BODY(OP_BCNT);
            else if (MODE == 6) BODY(OP_ADD3); else if (MODE == 7) BODY(OP_MAD24); else if (MODE == 8) BODY(OP_LSHLADD);
            else if (MODE == 9) BODY(OP_LSHLOR); else if (MODE == 10) BODY(OP_DPPADD); else if (MODE == 11) BODY(OP_DPPMIN);
            else if (MODE == 12) BODY(OP_DPPMOV); else if (MODE == 13) BODY(OP_PKADD); else if (MODE == 14) BODY(OP_PKMIN);
            else if (MODE == 15) BODY(OP_PERM); else if (MODE == 16) BODY(OP_CNDMASK); else if (MODE == 17) BODY(OP_MIN3);
            else if (MODE == 18) BODY(OP_SAD); else if (MODE == 19) BODY(OP_XAD); else if (MODE == 20) BODY(OP_ADDSDWA);
            else if (MODE == 21) BODY(OP_AND); else if (MODE == 22) BODY(OP_LSHR); else if (MODE == 23) BODY(OP_BFE);
            else if (MODE == 24) BODY(OP_MOV); else if (MODE == 25) BODY(OP_ALIGNBIT); else if (MODE == 26) BODY(OP_PKSUB);
            else if (MODE == 27) BODY(OP_MADI24); else if (MODE == 28) BODY(OP_CMPGT); else if (MODE == 29) BODY(OP_ADDC);
            else if (MODE == 30) BODY(OP_XORBCNT);
            else if (MODE == 31) { BODY(OP_XORIP); BODY(OP_BCNTONLY); }
            else if (MODE == 32) { BODY(OP_SUBIP); BODY(OP_ADD3); }
            else if (MODE == 33) { BODY(OP_XORIP); BODY(OP_XORIP); BODY(OP_BCNTONLY); BODY(OP_BCNTONLY); }
        }
        k += 0x01010101u;
    }
    uint32_t t = 0;
    for (int i = 0; i < 16; ++i) t += s[i] + a[i];
    if (t == 0xdeadbeefu) sink[0] = t;
}

typedef void (*kern_t)(uint32_t *, int);
struct Entry { const char *name; kern_t fn; int per_body; };

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    uint32_t *sink;
    CHK(hipMalloc(&sink, 256));
    std::vector<Entry> es = {
        {"v_xor_b32_e32", k<0>, 16}, {"v_xor_b32_e64(sgpr)", k<1>, 16}, {"v_add_u32_e32", k<2>, 16}, {"v_sub_u32_e32", k<3>, 16},
        {"v_min_u32_e32", k<4>, 16}, {"v_bcnt_u32_b32", k<5>, 16}, {"v_add3_u32", k<6>, 16}, {"v_mad_u32_u24", k<7>, 16},
        {"v_lshl_add_u32", k<8>, 16}, {"v_lshl_or_b32", k<9>, 16}, {"v_add_u32_dpp row_ror", k<10>, 16}, {"v_min_u32_dpp row_shr", k<11>, 16},
        {"v_mov_b32_dpp wave_shr", k<12>, 16}, {"v_pk_add_u16", k<13>, 16}, {"v_pk_min_u16", k<14>, 16}, {"v_perm_b32", k<15>, 16},
        {"v_cndmask_b32_e32", k<16>, 16}, {"v_min3_u32", k<17>, 16}, {"v_sad_u32", k<18>, 16}, {"v_xad_u32", k<19>, 16},
        {"v_add_u32_sdwa", k<20>, 16}, {"v_and_b32_e32", k<21>, 16}, {"v_lshrrev_b32_e32", k<22>, 16}, {"v_bfe_u32", k<23>, 16},
        {"v_mov_b32_e32", k<24>, 16}, {"v_alignbit_b32", k<25>, 16}, {"v_pk_sub_i16", k<26>, 16}, {"v_mad_i32_i24", k<27>, 16},
        {"v_cmp_gt_i32_e32", k<28>, 16}, {"v_addc_co_u32_e32", k<29>, 16}, {"xor_e32+bcnt pair", k<30>, 32}, {"16 xor_e32(sgpr) ; 16 bcnt", k<31>, 32}, {"16 sub_e32 ; 16 add3", k<32>, 32}, {"32 xor ; 32 bcnt", k<33>, 64},
    };
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%-26s %8s %8s %8s %8s   (cycles per wave64 instruction per SIMD at 2.4 GHz)\n", "instruction", "1w", "2w", "4w", "8w");
    for (auto &e : es) {
        printf("%-26s", e.name);
        for (int w : {1, 2, 4, 8}) {
            const int blocks = prop.multiProcessorCount * w;
            e.fn<<<blocks, 256>>>(sink, iters / 4);
            CHK(hipEventRecord(e0));
            e.fn<<<blocks, 256>>>(sink, iters);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            // instructions issued per SIMD = w waves * iters * 8 bodies * per_body
            const double inst = (double)w * iters * 8.0 * e.per_body;
            const double cyc = ms * 1e-3 * 2.4e9 / inst;
            printf(" %8.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
