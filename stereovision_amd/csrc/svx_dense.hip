// a8: dense winner-takes-all over 5x5-aggregated census Hamming costs (gfx950).
//
// Two kernels:
//   dense_wta5x5_plain_kernel  one thread per pixel, 25 popcounts per disparity; the simplest
//                              correct statement of the search, used for sizes the sweep kernel
//                              does not cover (max_disp > 1024).
//   dense_wta5x5_sweep_kernel  the fast path, VALU-popcount bound.  One wavefront owns a block of
//                              16 output columns of one row strip; its 64 lanes are 64 consecutive
//                              disparities.  It sweeps down the rows and keeps everything in
//                              registers: per row and column one v_xor + v_bcnt accumulates the
//                              per-pixel Hamming distance into a running column sum, a prefix sum
//                              along x turns five columns into one subtraction, and a ring of the
//                              last five rows' window sums turns five rows into one addition.  The
//                              disparity is folded into the low bits of the cost so that the
//                              argmin over the 64 lanes is a plain min, done as a DPP butterfly that
//                              reduces the 16 columns at once.  No cost volume exists anywhere.
#include "svx_common.hpp"
#include <cstdlib>

// ---------------------------------------------------------------------------
template <int WORDS>
__global__ __launch_bounds__(256) void dense_wta5x5_plain_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                                 int rows, int cols, int max_disp, float *__restrict__ disp_all,
                                                                 int write_border, int mirror) {
    const long n = (long)rows * cols;
    const uint32_t *cl = cl_all + (long)blockIdx.z * n * WORDS;
    const uint32_t *cr = cr_all + (long)blockIdx.z * n * WORDS;
    float *disp = disp_all + (long)blockIdx.z * n;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    // logical column c lives at memory column cols-1-c in mirrored mode
    auto at = [&](int yy, int c) -> long { return (long)yy * cols + (mirror ? cols - 1 - c : c); };
    if (x < 2 || x >= cols - 2 || y < 2 || y >= rows - 2) {
        if (write_border) disp[at(y, x)] = 0.0f;
        return;
    }
    int best = 0x7fffffff, bestx = x;
    for (int i = x; i >= 5 && i > x - max_disp; --i) {
        int cost = 0;
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int w = 0; w < WORDS; ++w)
                    cost += popc32(cl[at(y + m - 2, x + k - 2) * WORDS + w] ^ cr[at(y + m - 2, i + k - 2) * WORDS + w]);
        if (cost < best) { best = cost; bestx = i; }
    }
    disp[at(y, x)] = (float)(x - bestx);
}

// ---------------------------------------------------------------------------
// Sweep kernel.
constexpr int SW_T = 16;             // output columns per wavefront
constexpr int SW_C = SW_T + 4;       // census columns touched per row (2-column halo each side)
constexpr int SW_SLOTS = 5;          // staged rows in the wavefront's private LDS ring (divides the 10-row group)
constexpr int SW_SLOT_WORDS = 96;    // right census words of columns xq .. xq+82 (83 used)
constexpr int SW_KEY_SHIFT = 10;     // key = cost << 10 | d  (d < 1024); the literal 10 also sits in sweep_row's asm
constexpr int SW_DMASK = (1 << SW_KEY_SHIFT) - 1;
constexpr uint32_t SW_INVALID = 0x40000000u;   // canonical key of "no admissible disparity": decodes to d = 0
constexpr int SW_MAX_STRIP = 96;     // rows per strip (best-key store in LDS: strip x 16 words per wavefront)

struct SweepState {
    uint32_t C[SW_C];          // running column sums of per-pixel Hamming distances (prefix over rows)
    uint32_t ring[5][SW_T];    // signed window sums of the five previous rows (sign alternates by row)
};

// Argmin / argmax butterfly over the 64 lanes for 16 registers at once.
// In : k[i] = key of column i in every lane (lane = disparity).
// Out: lane l holds the key of column (l & 15) reduced over its own 16-lane row (16 disparities); the four
//      rows are merged by the caller's LDS atomic.
template <bool IS_MAX>
__device__ __forceinline__ uint32_t reduce16(const uint32_t (&k)[SW_T], int lane) {
    uint32_t r[8], q[4], u[4];
    // stage A: lanes l and l+8 of each 16-lane row; banks 0,1 keep columns 0..7, banks 2,3 columns 8..15
#define SVX_STAGE_A(OP)                                                                                        \
    asm volatile("s_nop 1\n\t"                                                                                 \
                 OP " %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"  OP " %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   \
                 OP " %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t" OP " %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
                 OP " %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t" OP " %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
                 OP " %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t" OP " %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
                 OP " %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" OP " %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" \
                 OP " %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" OP " %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" \
                 OP " %4, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" OP " %5, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" \
                 OP " %6, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" OP " %7, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" \
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])            \
                 : "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]),                         \
                   "v"(k[8]), "v"(k[9]), "v"(k[10]), "v"(k[11]), "v"(k[12]), "v"(k[13]), "v"(k[14]), "v"(k[15]))
    // stage B: lanes l and l+-4; bank b ends up with column (i + 4b) of register i
#define SVX_STAGE_B(OP)                                                                                        \
    asm volatile("s_nop 1\n\t"                                                                                 \
                 OP " %0, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t" OP " %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"   \
                 OP " %2, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t" OP " %3, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"   \
                 OP " %0, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t" OP " %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"   \
                 OP " %2, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t" OP " %3, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t" \
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])                                                                \
                 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]))
    // stage C1: lanes l and l^1 inside every quad
#define SVX_STAGE_C1(OP)                                                                                       \
    asm volatile("s_nop 1\n\t"                                                                                 \
                 OP " %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" OP " %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                 OP " %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" OP " %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]))
    // stage C2: lanes l and l^2 inside every quad
#define SVX_STAGE_C2(OP, a, b)                                                                                 \
    asm volatile("s_nop 1\n\t"                                                                                 \
                 OP " %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" OP " %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
                 : "=&v"(a), "=&v"(b) : "v"(w0), "v"(w1))
    if (IS_MAX) { SVX_STAGE_A("v_max_i32_dpp"); SVX_STAGE_B("v_max_i32_dpp"); SVX_STAGE_C1("v_max_i32_dpp"); }
    else        { SVX_STAGE_A("v_min_u32_dpp"); SVX_STAGE_B("v_min_u32_dpp"); SVX_STAGE_C1("v_min_u32_dpp"); }
    // odd lanes take the pair result of the odd register
    const uint32_t w0 = (lane & 1) ? u[1] : u[0];
    const uint32_t w1 = (lane & 1) ? u[3] : u[2];
    uint32_t x0, x1;
    if (IS_MAX) { SVX_STAGE_C2("v_max_i32_dpp", x0, x1); } else { SVX_STAGE_C2("v_min_u32_dpp", x0, x1); }
    uint32_t z = (lane & 2) ? x1 : x0;     // lane (l & 15) now holds column (l & 15), reduced over its 16-lane row
    return z;
#undef SVX_STAGE_A
#undef SVX_STAGE_B
#undef SVX_STAGE_C1
#undef SVX_STAGE_C2
}

// One census row of the sweep, phase 1: fold the row into the running column sums.
// cl: the 20 left words of the row (wavefront-uniform, in SGPRs); rp: this lane's view of the staged right
// words, skewed by its disparity (lane l reads column xq + 63 - l + j).
// WORDS = 32-bit words per census descriptor; a staged row holds the descriptors of consecutive columns
// back to back (WORDS words each), so a two-word descriptor is one 64-bit LDS access.
// GROUPS > 1 folds the row in column groups (loads of a group, then its popcounts) so that fewer right words are live
// at once; used where registers, not latency, limit the kernel.
template <int WORDS, int GROUPS = 1>
__device__ __forceinline__ void sweep_row_accumulate(SweepState &st, const uint32_t (&cl)[SW_C * WORDS], const uint32_t *__restrict__ rp) {
    constexpr int PER = SW_C / GROUPS;
    static_assert(SW_C % GROUPS == 0, "column groups must divide the row");
#pragma unroll
    for (int g0 = 0; g0 < SW_C; g0 += PER) {
        uint32_t cr[PER][WORDS];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (WORDS == 1) {
                cr[j][0] = rp[g0 + j];
            } else {
                const uint2 v = *reinterpret_cast<const uint2 *>(rp + 2 * (g0 + j));
                cr[j][0] = v.x; cr[j][WORDS - 1] = v.y;
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            st.C[g0 + j] += (uint32_t)__builtin_popcount(cl[(g0 + j) * WORDS] ^ cr[j][0]);
#pragma unroll
            for (int w = 1; w < WORDS; ++w)   // keep the accumulating popcount: left to itself the compiler sums the counts with extra adds
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(st.C[g0 + j]) : "v"(cl[(g0 + j) * WORDS + w] ^ cr[j][w]), "v"(st.C[g0 + j]));
        }
        if (GROUPS > 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// Phase 2: form the 16 window keys and reduce them over the lanes.
// R = row counter modulo 10 (ring slot R % 5, sign (-1)^R).
// Right view from the same cost values: the cost of (left column x, disparity d) is also the cost of the right
// pixel x-d at disparity d, so the right image's winner is the minimum along a diagonal of the (column, lane)
// tile.  One DPP step per column carries the running minimum to the next lane (wave_shr:1; lane 0, which has no
// source, starts the next diagonal from its own key): afterwards k[15] holds, in lane l, the minimum of the
// diagonal that ends at (column 15, lane l), and lane 63 of k[i], i < 15, that of the diagonal leaving the tile
// at column i.  These 79 partial minima are stored for svx's merge pass (one tile covers only 16 of the up to
// max_disp columns of a diagonal).
#ifndef SVX_BASE_GROUPS
#define SVX_BASE_GROUPS 1
#endif
#ifndef SVX_BOTH_GROUPS
#define SVX_BOTH_GROUPS 2       // column groups of the row accumulation in the both-views kernel (measured best with 3 wavefronts/SIMD)
#endif
constexpr int SW_PART = 80;          // partial minima stored per (row, x-block, chunk): 64 + 15, padded
template <bool IS_MAX>
__device__ __forceinline__ uint32_t diag_step(uint32_t run, uint32_t key) {
    if (IS_MAX) {
        const int sh = __builtin_amdgcn_update_dpp((int)0x80000000, (int)run, 0x138, 0xf, 0xf, false);
        return (uint32_t)max(sh, (int)key);
    }
    const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)run, 0x138, 0xf, 0xf, false);
    return min(sh, key);
}

struct RightOut { uint32_t k[SW_T]; };   // canonical running minima: lane l of k[15] = slot l; lane 63 of k[i], i < 15, = slot 64 + i

template <int R, bool MASKED, bool BOTH>
__device__ __forceinline__ uint32_t sweep_row_reduce(SweepState &st, int lane, uint32_t d, int xlim_base, int dmax,
                                                     int ilim_right, RightOut &ro) {
    constexpr int S = R % 5;
    constexpr bool ODD = (R & 1) != 0;
    // prefix sum along x, scaled so that the disparity fits below the cost
    uint32_t ip[SW_C];
    ip[0] = st.C[0] << SW_KEY_SHIFT;
#pragma unroll
    for (int j = 1; j < SW_C; ++j)   // (C << 10) + previous in ONE instruction; opaque so that it is not re-associated
        asm("v_lshl_add_u32 %0, %1, 10, %2" : "=v"(ip[j]) : "v"(st.C[j]), "v"(ip[j - 1]));
    uint32_t key[SW_T];
#pragma unroll
    for (int i = 0; i < SW_T; ++i) {
        // window of columns i .. i+4  (output column i sits at census column i+2)
        const uint32_t hi = ip[i + 4], lo = i == 0 ? 0u : ip[i - 1];
        const uint32_t v = ODD ? lo - hi : hi - lo;
        key[i] = v + st.ring[S][i] + d;       // even rows: cost<<10 + d ; odd rows: -(cost<<10) + (1023 - d)
        st.ring[S][i] = v;
    }
    // right-view keys: the raw keys, or in masked blocks their own admissibility (right column x-d >= 2, left column
    // x <= cols-6, d < max_disp); the diagonal chain runs after the lane reduction so that it can reuse the key registers
    uint32_t kr[SW_T];
    if (BOTH && MASKED) {
#pragma unroll
        for (int i = 0; i < SW_T; ++i) {
            const bool ok = (int)(ODD ? SW_DMASK - d : d) <= min(xlim_base + 3 + i, dmax) && i <= ilim_right;
            kr[i] = ok ? key[i] : (ODD ? (uint32_t)SW_DMASK - SW_INVALID : SW_INVALID);
        }
    }
    if (MASKED) {
        // disparities that would put the right window left of column 5, or beyond max_disp, never win
#pragma unroll
        for (int i = 0; i < SW_T; ++i) {
            const bool ok = (int)(ODD ? SW_DMASK - d : d) <= min(xlim_base + i, dmax);
            key[i] = ok ? key[i] : (ODD ? (uint32_t)SW_DMASK - SW_INVALID : SW_INVALID);
        }
    }
    const uint32_t z = reduce16<ODD>(key, lane);
    if (BOTH) {
        if (!MASKED) {
#pragma unroll
            for (int i = 0; i < SW_T; ++i) kr[i] = key[i];
        }
#pragma unroll
        for (int i = 1; i < SW_T; ++i) kr[i] = diag_step<ODD>(kr[i - 1], kr[i]);
#pragma unroll
        for (int i = 0; i < SW_T; ++i) ro.k[i] = ODD ? (uint32_t)SW_DMASK - kr[i] : kr[i];     // canonical keys: cost << 10 | d
    }
    return ODD ? (uint32_t)SW_DMASK - z : z;    // canonical key: cost << 10 | d
}

// Sweeps one 64-disparity chunk of one (strip, 16-column block) and merges its winners into `best`.
// xb is the first output column; census columns xb-2 .. xb+17 are all inside the row (the caller
// guarantees 2 <= xb <= cols-18).
// MIRROR: logical column c lives at memory column cols-1-c (the right-view search is the left-view search on
// the mirrored pair with the images swapped).
template <bool MASKED, int WORDS, bool MIRROR, bool BOTH>
__device__ __forceinline__ void sweep_chunk(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int cols, int max_disp,
                                            int xb, int y_lo, int n_rows, int chunk, int lane,
                                            uint32_t (*stage)[WORDS * SW_SLOT_WORDS], uint32_t (*best)[SW_T],
                                            uint32_t *__restrict__ part, long part_row_stride) {
    const int dbase = chunk << 6;
    const uint32_t d = (uint32_t)(dbase + lane);
    const uint32_t d_odd = (uint32_t)SW_DMASK - d;
    const int xq = xb - 2 - dbase - 63;             // right column held in stage word 0
    const int xlim_base = xb - 5;                   // admissible: d <= xlim_base + i
    const int dmax = max_disp - 1;
    const int ilim_right = cols - 6 - xb;           // right view: left column xb + i must not exceed cols-6
    SweepState st;
#pragma unroll
    for (int j = 0; j < SW_C; ++j) st.C[j] = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int i = 0; i < SW_T; ++i) st.ring[s][i] = 0;

    // Right words: per row every lane fetches right column xq+l (g0) and, for l < 19, xq+64+l (g1).
    // Column indices are clamped into the row: a clamped word only ever feeds an inadmissible
    // (column, disparity) pair, which is masked out.
    const int lc0 = min(max(xq + lane, 0), cols - 1), lc1 = min(max(xq + 64 + (lane & 31), 0), cols - 1);
    const unsigned c0 = (unsigned)(MIRROR ? cols - 1 - lc0 : lc0) * WORDS;
    const unsigned c1 = (unsigned)(MIRROR ? cols - 1 - lc1 : lc1) * WORDS;
    struct Fetched { uint32_t a[WORDS], b[WORDS]; };
    auto fetch = [&](int r, Fetched &g) {
        const uint32_t *rowp = cr + (long)(y_lo - 2 + r) * cols * WORDS;   // uniform row base + per-lane column
        if (WORDS == 1) {
            g.a[0] = rowp[c0];
            g.b[0] = rowp[c1];
        } else {
            const uint2 va = *reinterpret_cast<const uint2 *>(rowp + c0), vb = *reinterpret_cast<const uint2 *>(rowp + c1);
            g.a[0] = va.x; g.a[WORDS - 1] = va.y;
            g.b[0] = vb.x; g.b[WORDS - 1] = vb.y;
        }
    };
    uint32_t *const st_w0 = &stage[0][lane * WORDS], *const st_w1 = &stage[0][(64 + (lane & 31)) * WORDS];
    const uint32_t *const st_r = &stage[0][(63 - lane) * WORDS];
    constexpr int SLOT_STRIDE = WORDS * SW_SLOT_WORDS;
    auto stash = [&](int slot, const Fetched &g) {
        if (WORDS == 1) {
            st_w0[slot * SLOT_STRIDE] = g.a[0];
            st_w1[slot * SLOT_STRIDE] = g.b[0];     // lanes l and l+32 store the same word
        } else {
            *reinterpret_cast<uint2 *>(st_w0 + slot * SLOT_STRIDE) = make_uint2(g.a[0], g.a[WORDS - 1]);
            *reinterpret_cast<uint2 *>(st_w1 + slot * SLOT_STRIDE) = make_uint2(g.b[0], g.b[WORDS - 1]);
        }
        __builtin_amdgcn_wave_barrier();            // the rows are read back by other lanes of this wavefront
    };
    // Left words: 20 consecutive descriptors of the row, the same for every lane -> scalar loads into SGPRs.
    // Mirrored, logical columns xb-2 .. xb+17 are memory columns cols-1-(xb-2) down to cols-1-(xb+17).
    const uint32_t *lbase = cl + (long)(MIRROR ? cols - 1 - (xb + SW_T + 1) : xb - 2) * WORDS;
    auto fetch_left = [&](int r, uint32_t (&w)[SW_C * WORDS]) {
        const uint32_t *lp = lbase + (long)(y_lo - 2 + r) * cols * WORDS;
#pragma unroll
        for (int j = 0; j < SW_C; ++j)
#pragma unroll
            for (int k = 0; k < WORDS; ++k) w[j * WORDS + k] = lp[(MIRROR ? SW_C - 1 - j : j) * WORDS + k];
    };

    // Software pipeline over rows: the global fetch of row r+3 and the scalar fetch of row r+1 are in
    // flight while row r is processed.  Rows run in unconditional groups of ten (ring slot and sign are
    // compile-time); rows past the end of the strip re-read the last census row and their winners are dropped.
    const int last = n_rows - 1;
    Fetched g;
    fetch(0, g);
    stash(0, g);
    fetch(min(1, last), g);
    stash(1, g);
    fetch(min(2, last), g);
    uint32_t lw[SW_C * WORDS];
    fetch_left(0, lw);

    // The scalar fetch of the next row's left words is issued right after the current row's words were
    // consumed: scalar loads and LDS reads share one counter, so it must not sit in front of an LDS wait.
    // Lane l holds column (l & 15) reduced over its 16-lane row; the LDS atomic merges the four rows.
#define SVX_ROW(RR)                                                                                   \
    {                                                                                                 \
        const int r = r0 + RR;                                                                        \
        sweep_row_accumulate<WORDS, (BOTH ? SVX_BOTH_GROUPS : SVX_BASE_GROUPS)>(st, lw, st_r + (RR % SW_SLOTS) * SLOT_STRIDE); \
        fetch_left(min(r + 1, last), lw);                                                             \
        stash((RR + 2) % SW_SLOTS, g);                                                                \
        fetch(min(r + 3, last), g);                                                                   \
        RightOut ro;                                                                                  \
        const uint32_t zk = sweep_row_reduce<RR, MASKED, BOTH>(st, lane, (RR & 1) ? d_odd : d, xlim_base, dmax, ilim_right, ro); \
        if (r >= 4 && r < n_rows) {                /* output row y_lo + r - 4 */                      \
            atomicMin(&best[r - 4][lane & 15], zk);                                                   \
            if (BOTH) {                /* pp_main walks down the rows */ \
                pp_main[0] = ro.k[SW_T - 1];                                                          \
                if (lane == 63) {      /* the diagonals that left the tile through the last lane: slots 64..78 (79 unused) */ \
                    uint4 *pd = reinterpret_cast<uint4 *>(pp_main + 1);                               \
                    pd[0] = make_uint4(ro.k[0], ro.k[1], ro.k[2], ro.k[3]);                           \
                    pd[1] = make_uint4(ro.k[4], ro.k[5], ro.k[6], ro.k[7]);                           \
                    pd[2] = make_uint4(ro.k[8], ro.k[9], ro.k[10], ro.k[11]);                         \
                    pd[3] = make_uint4(ro.k[12], ro.k[13], ro.k[14], ro.k[15]);                       \
                }                                                                                     \
                pp_main += part_row_stride;                                                           \
            }                                                                                         \
        }                                                                                             \
    }
    uint32_t *pp_main = part + lane;      // this lane's slot in the current output row
    for (int r0 = 0; r0 < n_rows; r0 += 10) {
        SVX_ROW(0) SVX_ROW(1) SVX_ROW(2) SVX_ROW(3) SVX_ROW(4)
        SVX_ROW(5) SVX_ROW(6) SVX_ROW(7) SVX_ROW(8) SVX_ROW(9)
    }
#undef SVX_ROW
}

// grid: (ceil(x-blocks / 2), strips, frames); block: 256 threads = 4 wavefronts = 2 x-blocks x 2 chunk
// parities: wavefronts 2k and 2k+1 sweep the even and the odd 64-disparity chunks of x-block k and merge
// their winners in one shared best-key array with LDS atomics (two barriers per block, none in the sweep).
// x-block b covers output columns [xb, xb+16), xb = min(2 + 16 b, cols - 18): the last block is shifted
// left so that it ends at column cols-3; the columns it shares with its neighbour get the same values
// from both.  Requires cols >= 20.
// BOTH: additionally stores the right view's partial minima (see diag_step): partial[frame][row][x-block][chunk][SW_PART].
template <int WAVES_PER_SIMD, int WORDS, bool MIRROR, bool BOTH>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void dense_wta5x5_sweep_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                                 int rows, int cols, int max_disp, int strip_rows, int n_xblocks,
                                                                 float *__restrict__ disp_all, int write_border, uint32_t *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_stage[4][SW_SLOTS][WORDS * SW_SLOT_WORDS];
    __shared__ uint32_t lds_best[2][SW_MAX_STRIP][SW_T];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, parity = wave & 1;
    const long n = (long)rows * cols;
    // Workgroups are handed to the eight XCDs (each with its own L2) round-robin in linear order: give every XCD a
    // contiguous run of work items (x-block pair fastest, then strip, then frame) so that the blocks that re-read the same
    // census rows share an L2.  Placement only affects speed; any mapping is correct.
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    unsigned w = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((total & 7u) == 0u) w = (w & 7u) * (total >> 3) + (w >> 3);
    const unsigned bx = w % gridDim.x, by = (w / gridDim.x) % gridDim.y, bz = w / (gridDim.x * gridDim.y);
    const uint32_t *cl = cl_all + (long)bz * n * WORDS;
    const uint32_t *cr = cr_all + (long)bz * n * WORDS;
    float *disp = disp_all + (long)bz * n;
    const int xblock = (int)bx * 2 + pair;
    const bool active = xblock < n_xblocks;
    const int xb = min(2 + xblock * SW_T, cols - 2 - SW_T);    // first output column of this wavefront
    // strip k produces output rows [2 + k*strip_rows, 2 + (k+1)*strip_rows) clipped to rows-2; for the
    // frame border the first strip also owns rows 0,1 and the last one rows rows-2, rows-1
    const int y_lo = 2 + (int)by * strip_rows, y_hi = min(y_lo + strip_rows, rows - 2);
    const int ys = by == 0 ? 0 : y_lo, y_end = by == gridDim.y - 1 ? rows : y_hi;
    uint32_t (*stage)[WORDS * SW_SLOT_WORDS] = lds_stage[wave];
    uint32_t (*best)[SW_T] = lds_best[pair];
    const int n_out = y_hi - y_lo;                          // >= 1 by construction of the grid
    const int half = parity * 64 + lane;                    // 0..127 over the two wavefronts of the pair

    for (int i = half; i < n_out * SW_T; i += 128) best[0][i] = 0xffffffffu;
    __syncthreads();
    if (active) {
        const int n_rows = n_out + 4;                       // census rows y_lo-2 .. y_hi+1
        const int n_chunks = (max_disp + 63) >> 6;
        for (int chunk = parity; chunk < n_chunks; chunk += 2) {
            const int dbase = chunk << 6;
            // no disparity of this chunk is admissible for any column of the block (left view: right column >= 5; right view: >= 2)
            if (chunk > 0 && dbase > xb + SW_T - 1 - (BOTH ? 2 : 5)) break;
            // masking is needed when some (column, lane) pair is inadmissible
            const bool masked = (dbase + 63 > xb - 5) || (dbase + 63 > max_disp - 1) || (BOTH && xb + SW_T - 1 > cols - 6);
            // partial minima of this (x-block, chunk), first output row of the strip; consecutive rows are row_stride apart
            const long row_stride = (long)n_xblocks * n_chunks * SW_PART;
            uint32_t *part = BOTH ? partial + ((long)bz * rows + y_lo) * row_stride + ((long)xblock * n_chunks + chunk) * SW_PART : nullptr;
            if (masked) sweep_chunk<true, WORDS, MIRROR, BOTH>(cl, cr, cols, max_disp, xb, y_lo, n_rows, chunk, lane, stage, best, part, row_stride);
            else sweep_chunk<false, WORDS, MIRROR, BOTH>(cl, cr, cols, max_disp, xb, y_lo, n_rows, chunk, lane, stage, best, part, row_stride);
        }
    }
    __syncthreads();
    if (!active) return;
    // write-out: 8 rows x 16 columns per pass over the pair
    const int xo = MIRROR ? cols - 1 - (xb + (lane & 15)) : xb + (lane & 15);
    for (int o = (half >> 4); o < n_out; o += 8)
        disp[(long)(y_lo + o) * cols + xo] = (float)(best[o][lane & 15] & SW_DMASK);
    if (write_border) {
        // the frame border is 0: rows outside [2, rows-2) over this block's columns, and the two
        // border columns on either side (written by the first and the last block; the set is its own mirror image)
        for (int y = ys + (half >> 4); y < y_end; y += 8)
            if (y < 2 || y >= rows - 2) disp[(long)y * cols + xo] = 0.0f;
        if (xblock == 0)
            for (int y = ys + (half >> 1); y < y_end; y += 64) disp[(long)y * cols + (lane & 1)] = 0.0f;
        if (xblock == n_xblocks - 1)
            for (int y = ys + (half >> 1); y < y_end; y += 64) disp[(long)y * cols + cols - 2 + (lane & 1)] = 0.0f;
    }
}

static int dense_args_ok(svx_ctx *ctx, const void *a, const void *b, const void *c, int rows, int cols, int n_frames, int max_disp) {
    SVX_REQUIRE(ctx, a && b && c, SVX_ERR_INVALID_ARG, "dense_wta: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "dense_wta: empty image");
    SVX_REQUIRE(ctx, max_disp >= 1, SVX_ERR_INVALID_ARG, "dense_wta: max_disp must be >= 1");
    SVX_REQUIRE(ctx, n_frames <= 65535, SVX_ERR_UNSUPPORTED, "dense_wta: more than 65535 frames per call");
    return SVX_OK;
}

// Right-view merge: the right image's disparity at (xr, y) is the minimum over the partial minima of every tile its
// diagonal crosses.  Tile (x-block b, chunk c) holds that diagonal in slot L = xb + 15 - dbase - xr when 0 <= L <= 63
// (it ends at the tile's last column) or in slot 142 - L when 64 <= L <= 78 (it left through lane 63).  The left
// columns xr + dbase .. xr + dbase + 63 of a chunk lie in five consecutive x-blocks at most.
__device__ __forceinline__ float right_disparity(const uint32_t *__restrict__ prow, int xr, int cols, int n_xblocks, int n_chunks) {
    uint32_t best = 0xffffffffu;
    for (int c = 0; c < n_chunks; ++c) {
        const int dbase = c << 6;
        const int b0 = (xr + dbase - 2) >> 4;                       // block of left column xr + dbase (floor; may be -1)
        if (xr >= 2 && b0 + 4 < n_xblocks - 1) {
            // All five blocks are regular ones (xb = 2 + 16 b) and none of their chunks was skipped (xb + 13 >= dbase
            // follows from xr >= 2): L = L0 + 16 k with L0 = 15 - ((xr - 2) & 15), so blocks k = 0..3 hold the diagonal
            // in slot L and block k = 4 holds it in slot 78 - L0 unless L0 = 15.
            const int l0 = 15 - ((xr - 2) & 15);
            const unsigned blk = (unsigned)n_chunks * SW_PART;
            const uint32_t *p = prow + ((unsigned)b0 * n_chunks + c) * SW_PART;
            const uint32_t v0 = p[l0], v1 = p[blk + 16 + l0], v2 = p[2 * blk + 32 + l0], v3 = p[3 * blk + 48 + l0];
            const uint32_t v4 = p[4 * blk + 78 - l0];               // slot 63 when l0 = 15: another diagonal, not used then
            best = min(best, min(min(v0, v1), min(v2, v3)));
            best = min(best, l0 == 15 ? 0xffffffffu : v4);
            continue;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int b = b0 + k;
            const int bc = min(max(b, 0), n_xblocks - 1);
            const int xb = min(2 + bc * SW_T, cols - 2 - SW_T);     // the last block is shifted left to end at cols-3
            const int L = xb + SW_T - 1 - dbase - xr;
            // the sweep skips chunks none of whose disparities is admissible for the block
            const bool ok = b == bc && L >= 0 && L <= 78 && !(c > 0 && dbase > xb + SW_T - 1 - 2);
            const int slot = L <= 63 ? L : 142 - L;
            const uint32_t v = prow[((long)bc * n_chunks + c) * SW_PART + min(max(slot, 0), SW_PART - 1)];
            best = min(best, ok ? v : 0xffffffffu);
        }
    }
    return best >= SW_INVALID ? 0.0f : (float)(best & SW_DMASK);
}

__global__ __launch_bounds__(256) void right_view_merge_kernel(const uint32_t *__restrict__ partial, int rows, int cols, int max_disp,
                                                               int n_xblocks, float *__restrict__ disp_right_all) {
    const int xr = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xr >= cols) return;
    float *out = disp_right_all + ((long)blockIdx.z * rows + y) * cols;
    if (xr < 2 || xr >= cols - 2 || y < 2 || y >= rows - 2) { out[xr] = 0.0f; return; }
    const int n_chunks = (max_disp + 63) >> 6;
    const uint32_t *prow = partial + ((long)blockIdx.z * rows + y) * ((long)n_xblocks * n_chunks * SW_PART);
    out[xr] = right_disparity(prow, xr, cols, n_xblocks, n_chunks);
}

// Left-right check straight from the partial minima: lr_check_kernel's rule with the right disparity merged on demand.
__global__ __launch_bounds__(256) void lr_check_partial_kernel(float *__restrict__ dl_all, const uint32_t *__restrict__ partial, int rows, int cols,
                                                               int max_disp, int n_xblocks, float threshold, float invalid) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < 2 || x >= cols - 2 || y < 2 || y >= rows - 2) return;
    float *dl = dl_all + ((long)blockIdx.z * rows + y) * cols;
    const float d1 = dl[x];
    if (!(d1 >= 0.0f) || !(d1 < (float)cols)) { dl[x] = invalid; return; }       // no disparity (negative, NaN, beyond the row): reads nothing
    const int xr = x - (int)d1;
    if (xr < 2) { dl[x] = invalid; return; }
    const int n_chunks = (max_disp + 63) >> 6;
    const uint32_t *prow = partial + ((long)blockIdx.z * rows + y) * ((long)n_xblocks * n_chunks * SW_PART);
    if (fabsf(right_disparity(prow, xr, cols, n_xblocks, n_chunks) - d1) > threshold) dl[x] = invalid;
}

template <int WORDS, bool MIRROR>
static int launch_sweep(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int max_disp,
                        float *disparity, int write_border, uint32_t *partial = nullptr) {
    // strips of SW_MAX_STRIP output rows from row 2 on: strip + 4 halo rows is a whole number of 10-row groups
    const int strip_rows = SW_MAX_STRIP;
    const int n_strips = (rows - 4 + strip_rows - 1) / strip_rows;
    const int n_xblocks = (cols - 4 + SW_T - 1) / SW_T;
    dim3 grid(svx_cdiv(n_xblocks, 2), n_strips, n_frames);
    // SVX_OPT_SWEEP_WAVES = 2 selects the build that may use up to 256 VGPRs (2 wavefronts per SIMD).
    const int occ = svx_opt(ctx, SVX_OPT_SWEEP_WAVES);
    if (partial) {
        if (MIRROR) return svx_fail(ctx, SVX_ERR_INVALID_ARG, "both-view sweep runs in the left view");
        const int occ_both = svx_opt(ctx, SVX_OPT_SWEEP_WAVES_BOTH);
        if (occ_both == 3)
            dense_wta5x5_sweep_kernel<3, WORDS, false, true><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, strip_rows, n_xblocks, disparity, write_border, partial);
        else
            dense_wta5x5_sweep_kernel<2, WORDS, false, true><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, strip_rows, n_xblocks, disparity, write_border, partial);
    } else if (occ == 2)
        dense_wta5x5_sweep_kernel<2, WORDS, MIRROR, false><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, strip_rows, n_xblocks, disparity, write_border, nullptr);
    else
        dense_wta5x5_sweep_kernel<3, WORDS, MIRROR, false><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, strip_rows, n_xblocks, disparity, write_border, nullptr);
    SVX_LAUNCH_CHECK(ctx, "dense_wta5x5_sweep_kernel");
    return SVX_OK;
}

// SVX_OPT_MATRIX_CORES: 0 = never the matrix cores; otherwise wherever they apply (every left-view and both-views search
// of at least 36 columns and 5 rows at up to 1024 disparities; the right view alone stays on the vector sweep).
static bool both_views_on_matrix_cores(const svx_ctx *ctx, int rows, int cols, int words, int max_disp) {
    (void)words;
    return svx_opt(ctx, SVX_OPT_MATRIX_CORES) != 0 && svx_mx_sweep_eligible(rows, cols, max_disp);
}

// words: 32-bit words per census descriptor (1 or 2); mirror: search in the other view (see sweep_chunk).
static int launch_dense(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int max_disp,
                        float *disparity, int write_border, int words = 1, int mirror = 0) {
    if (max_disp > SW_DMASK + 1 || cols < SW_T + 4 || rows < 5) {
        dim3 grid(svx_cdiv(cols, 64), svx_cdiv(rows, 4), n_frames);
        if (words == 1) dense_wta5x5_plain_kernel<1><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity, write_border, mirror);
        else dense_wta5x5_plain_kernel<2><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity, write_border, mirror);
        SVX_LAUNCH_CHECK(ctx, "dense_wta5x5_plain_kernel");
        return SVX_OK;
    }
    // the left-view search on the matrix cores (64 KITTI frames at 128 disparities: 0.68 against 0.90 ms with one-word, 1.07 against 1.18 ms with two-word descriptors)
    if (svx_opt(ctx, SVX_OPT_MATRIX_CORES) != 0 && !mirror && svx_mx_sweep_eligible(rows, cols, max_disp))
        return svx_launch_mx_sweep(ctx, cl, cr, rows, cols, n_frames, words, max_disp, disparity, write_border, nullptr);
    if (words == 1) return mirror ? launch_sweep<1, true>(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity, write_border)
                                  : launch_sweep<1, false>(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity, write_border);
    return mirror ? launch_sweep<2, true>(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity, write_border)
                  : launch_sweep<2, false>(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity, write_border);
}

SVX_API int svx_dense_wta5x5(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right,
                             int rows, int cols, int n_frames, int max_disp, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, census_left, census_right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    return launch_dense(ctx, census_left, census_right, rows, cols, n_frames, max_disp, disparity, 0);
}

SVX_API int svx_dense_census_wta_u8(svx_ctx *ctx, const uint8_t *left, const uint8_t *right,
                                    int rows, int cols, int n_frames, int max_disp, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, left, right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    const size_t n = (size_t)rows * cols * n_frames;
    void *ws;
    rc = svx_ws_reserve(ctx, 2 * n * sizeof(uint32_t), &ws);
    if (rc) return rc;
    uint32_t *cl = (uint32_t *)ws, *cr = cl + n;
    rc = svx_census5x5_u8(ctx, left, cl, rows, cols, n_frames, SVX_CENSUS_SSE24 | SVX_CENSUS_ZERO_REST);
    if (rc) return rc;
    rc = svx_census5x5_u8(ctx, right, cr, rows, cols, n_frames, SVX_CENSUS_SSE24 | SVX_CENSUS_ZERO_REST);
    if (rc) return rc;
    return launch_dense(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity, 1);
}

// ---------------------------------------------------------------------------
// svx_dense_census_wta_u8 for callers that cannot allocate (code under graph capture, serving loops): the two census
// batches, which the other entry point keeps in the context's workspace, live in a fixed array of the code object.
// Every device the library runs on holds its own copy of the array at its own address, found once per context (with
// the context's device current) and used as `ctx->static_scratch` from then on: in host code the array's NAME is a
// host-side stand-in, not a device address.
static __device__ __align__(256) unsigned char svx_static_scratch[(size_t)SVX_NOALLOC_MAX_PIXELS * 2 * sizeof(uint32_t)];

// Called by svx_create with the new context's device current.
int svx_static_scratch_locate(svx_ctx *ctx) {
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(svx_static_scratch));
    if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_HIP, "hipGetSymbolAddress(svx_static_scratch) failed: %s", hipGetErrorString(e));
    ctx->static_scratch = p;
    return SVX_OK;
}

SVX_API int svx_dense_census_wta_u8_noalloc(svx_ctx *ctx, const uint8_t *left, const uint8_t *right,
                                            int rows, int cols, int n_frames, int max_disp, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, left, right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    const size_t n = (size_t)rows * cols * n_frames;
    if (n > (size_t)SVX_NOALLOC_MAX_PIXELS)
        return svx_fail(ctx, SVX_ERR_UNSUPPORTED, "svx_dense_census_wta_u8_noalloc: %zu pixels per call, the fixed scratch holds %d",
                        n, SVX_NOALLOC_MAX_PIXELS);
    SVX_REQUIRE(ctx, ctx->static_scratch != nullptr, SVX_ERR_HIP, "svx_dense_census_wta_u8_noalloc: the fixed scratch was not found on this device");
    // every pixel of both census batches is written before the search reads it (SVX_CENSUS_ZERO_REST): nothing to clear
    uint32_t *cl = (uint32_t *)ctx->static_scratch, *cr = cl + n;
    rc = svx_census5x5_u8(ctx, left, cl, rows, cols, n_frames, SVX_CENSUS_SSE24 | SVX_CENSUS_ZERO_REST);
    if (rc) return rc;
    rc = svx_census5x5_u8(ctx, right, cr, rows, cols, n_frames, SVX_CENSUS_SSE24 | SVX_CENSUS_ZERO_REST);
    if (rc) return rc;
    return launch_dense(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity, 1);
}

// ---------------------------------------------------------------------------
// 64-bit descriptors: 9x7 census, dense search in either view, left-right check.
// The reference has no 9x7 census and no dense left-right check; these generalise its 5x5 dense
// matcher (denseCensusStereo.cpp:87-131) along the lines of its generic NxN census
// (denseCensusStereo/CensusTransform.cpp:33-67) and of libelas' consistency check (elas/elas.cpp:909-979).
constexpr int C97_W = 64, C97_H = 16, C97_HX = 4, C97_HY = 3, C97_PITCH = C97_W + 2 * C97_HX;

// Descriptor of pixel (x, y), 4 <= x < cols-4, 3 <= y < rows-3: neighbour (dy, dx), dy = -3..3, dx = -4..4, has
// index (dy+3)*9 + (dx+4) and bit 62 - index; the bit is set when the centre is greater than the neighbour
// (the centre's own bit, 31, stays 0).  Word 0 holds bits 0..31, word 1 bits 32..62.  All other pixels get 0.
__global__ __launch_bounds__(256) void census9x7_pixel_kernel(const uint8_t *__restrict__ img_all, int rows, int cols, uint2 *__restrict__ out_all) {
    __shared__ uint8_t tile[(C97_H + 2 * C97_HY) * C97_PITCH];
    const long n = (long)rows * cols;
    const uint8_t *img = img_all + (long)blockIdx.z * n;
    uint2 *out = out_all + (long)blockIdx.z * n;
    const int x0 = blockIdx.x * C97_W, y0 = blockIdx.y * C97_H;
    for (int i = threadIdx.x; i < (C97_H + 2 * C97_HY) * C97_PITCH; i += 256) {
        const int ty = i / C97_PITCH, tx = i - ty * C97_PITCH;
        const int gy = y0 + ty - C97_HY, gx = x0 + tx - C97_HX;
        tile[i] = (gy >= 0 && gy < rows && gx >= 0 && gx < cols) ? img[(long)gy * cols + gx] : (uint8_t)0;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, x = x0 + lx;
    if (x >= cols) return;
#pragma unroll 1
    for (int ly = threadIdx.x >> 6; ly < C97_H; ly += 4) {
        const int y = y0 + ly;
        if (y >= rows) break;
        uint32_t lo = 0, hi = 0;
        if (x >= C97_HX && x < cols - C97_HX && y >= C97_HY && y < rows - C97_HY) {
            const uint8_t *c = &tile[(ly + C97_HY) * C97_PITCH + lx + C97_HX];
            const uint32_t cv = c[0];
#pragma unroll
            for (int dy = -3; dy <= 3; ++dy)
#pragma unroll
                for (int dx = -4; dx <= 4; ++dx) {
                    const int bit = 62 - ((dy + 3) * 9 + (dx + 4));
                    if (dy == 0 && dx == 0) continue;
                    const uint32_t f = cv > (uint32_t)c[dy * C97_PITCH + dx] ? 1u : 0u;
                    if (bit < 32) lo |= f << bit; else hi |= f << (bit - 32);
                }
        }
        out[(long)y * cols + x] = make_uint2(lo, hi);
    }
}

// Fast variant: a lane owns four adjacent pixels of four rows.  The image is staged as dwords; the twelve bytes
// x-4..x+7 of a window row are three staged dwords, and each descriptor bit costs one SDWA byte compare plus one
// add-with-carry that shifts it in (generated blocks, tools/gen_census97.py).
// ALIGNED (cols % 4 == 0): every staged dword is one aligned load that lies wholly inside or wholly outside the frame.
// Otherwise a staged dword is cut out of the two aligned dwords of the batch that hold it; bytes that belong to a
// neighbouring row or frame land only in window positions that border pixels use, and those are written as 0.
#include "svx_census97_px.inc"
constexpr int C97T_W = 256, C97T_H = 16, C97T_LROW = C97T_W / 4 + 2 + 1;   // dwords per staged row (+1 pad)

// Bytes b and b+1 of the 12 bytes held in three dwords, each widened to a 16-bit field.
template <int B>
__device__ __forceinline__ uint32_t c97_pair(const uint32_t (&r)[3]) {
    constexpr int k = B / 4, k1 = (B + 1) / 4;
    constexpr uint32_t sel = (uint32_t)(B % 4) | (0x0cu << 8) | ((uint32_t)(k1 == k ? (B + 1) % 4 : 4 + (B + 1) % 4) << 16) | (0x0cu << 24);
    return __builtin_amdgcn_perm(r[k1 == k ? (k < 2 ? k + 1 : k) : k1], r[k], sel);
}

// The tile kernel's arithmetic on pixel pairs: two pixels sit in the 16-bit fields of a dword, `centre + 0x7fff - neighbour`
// has bit 15 of a field set exactly when the centre is greater, and the bit goes to its place in one of four
// 16-bit-per-pixel accumulators by a shift and a masked or: three instructions per two comparisons.
template <bool ALIGNED>
__device__ __forceinline__ void census9x7_tile_packed(const uint8_t *__restrict__ img_all, int rows, int cols, long total_bytes,
                                                      uint32_t *__restrict__ out_all, int frame) {
    __shared__ uint32_t tile[(C97T_H + 6) * C97T_LROW];
    const long n = (long)rows * cols;
    const uint8_t *img = img_all + (long)frame * n;
    uint32_t *out = out_all + (long)frame * n * 2;
    const int x0 = blockIdx.x * C97T_W, y0 = blockIdx.y * C97T_H;
    for (int i = threadIdx.x; i < (C97T_H + 6) * C97T_LROW; i += 256) {
        const int r = i / C97T_LROW, k = i - r * C97T_LROW;
        const int gy = y0 - 3 + r, gx = x0 - 4 + 4 * k;
        uint32_t v = 0u;
        if (ALIGNED) {
            if (gy >= 0 && gy < rows && gx >= 0 && gx < cols) v = *reinterpret_cast<const uint32_t *>(img + (long)gy * cols + gx);
        } else if (gy >= 0 && gy < rows && gx > -4 && gx < cols) {
            const long a = (long)frame * n + (long)gy * cols + gx;         // byte offset in the batch, may be negative by < 4
            const long a0 = a & ~3L;
            if (a0 >= 0 && a0 + 8 <= total_bytes) {
                const uint32_t d0 = *reinterpret_cast<const uint32_t *>(img_all + a0), d1 = *reinterpret_cast<const uint32_t *>(img_all + a0 + 4);
                v = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)(a & 3));
            } else {                                                            // the first and last bytes of the batch
                for (int b = 0; b < 4; ++b)
                    if (a + b >= 0 && a + b < total_bytes) v |= (uint32_t)img_all[a + b] << (8 * b);
            }
        }
        tile[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = x0 + lane * 4;
    if (x >= cols) return;
    uint32_t R[10][3];                 // staged rows wave*4 .. wave*4+9 = image rows y-3 .. y+6 of the lane's first row y
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t *q = tile + (wave * 4 + r) * C97T_LROW + lane;
        R[r][0] = q[0]; R[r][1] = q[1]; R[r][2] = q[2];
    }
    // acc[j][pair][w]: bits 16w .. 16w+15 of the descriptors of pixels 2 pair, 2 pair + 1 of output row j
    uint32_t acc[4][2][4], cpre[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cpre[j][0] = c97_pair<4>(R[j + 3]) + 0x7fff7fffu;
        cpre[j][1] = c97_pair<6>(R[j + 3]) + 0x7fff7fffu;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[j][0][w] = acc[j][1][w] = 0u;
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t pr[11];
        pr[0] = c97_pair<0>(R[r]); pr[1] = c97_pair<1>(R[r]); pr[2] = c97_pair<2>(R[r]); pr[3] = c97_pair<3>(R[r]);
        pr[4] = c97_pair<4>(R[r]); pr[5] = c97_pair<5>(R[r]); pr[6] = c97_pair<6>(R[r]); pr[7] = c97_pair<7>(R[r]);
        pr[8] = c97_pair<8>(R[r]); pr[9] = c97_pair<9>(R[r]); pr[10] = c97_pair<10>(R[r]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dy = r - j - 3;
            if (dy < -3 || dy > 3) continue;
#pragma unroll
            for (int dx = -4; dx <= 4; ++dx) {
                const int idx = (dy + 3) * 9 + dx + 4;
                if (idx == 31) continue;                       // the centre itself
                const int q = 62 - idx, w = q >> 4, pos = q & 15;
#pragma unroll
                for (int pair = 0; pair < 2; ++pair) {
                    const uint32_t u = cpre[j][pair] - pr[2 * pair + dx + 4];
                    // (shifted & mask) | acc in one instruction (left to itself the compiler pairs an and with half an or3)
                    asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc[j][pair][w]) : "v"(pos == 15 ? u : u >> (15 - pos)), "s"(0x00010001u << pos));
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + wave * 4 + j;
        if (y >= rows) break;
        uint32_t hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
        if (y >= 3 && y < rows - 3) {
#pragma unroll
            for (int pair = 0; pair < 2; ++pair) {
                lo[2 * pair] = __builtin_amdgcn_perm(acc[j][pair][1], acc[j][pair][0], 0x05040100u);
                hi[2 * pair] = __builtin_amdgcn_perm(acc[j][pair][3], acc[j][pair][2], 0x05040100u);
                lo[2 * pair + 1] = __builtin_amdgcn_perm(acc[j][pair][1], acc[j][pair][0], 0x07060302u);
                hi[2 * pair + 1] = __builtin_amdgcn_perm(acc[j][pair][3], acc[j][pair][2], 0x07060302u);
            }
            if (ALIGNED) {                                                                               // the border columns
                if (x < 4) hi[0] = lo[0] = hi[1] = lo[1] = hi[2] = lo[2] = hi[3] = lo[3] = 0u;            // are whole lanes
                if (x >= cols - 4) hi[0] = lo[0] = hi[1] = lo[1] = hi[2] = lo[2] = hi[3] = lo[3] = 0u;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (x + p < 4 || x + p >= cols - 4) hi[p] = lo[p] = 0u;
            }
        }
        if (ALIGNED || (((cols & 1) == 0) && x + 3 < cols)) {
            uint4 *o = reinterpret_cast<uint4 *>(out + ((long)y * cols + x) * 2);
            o[0] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            o[1] = make_uint4(lo[2], hi[2], lo[3], hi[3]);
        } else {
            uint2 *o = reinterpret_cast<uint2 *>(out + ((long)y * cols + x) * 2);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (x + p < cols) o[p] = make_uint2(lo[p], hi[p]);
        }
    }
}

// Third form of the tile kernel (widths that are multiples of 4): three comparisons per subtraction.  Three pixels of one
// COLUMN (image rows r, r+1, r+2) sit in 9-bit fields of a dword, 0x100 + pixel each ("triple", built once per tile in LDS for
// every row r and column); the centre is repeated in the three fields, and `triple - centres` leaves bit 8 of a field set exactly
// when that neighbour is >= the centre (a field stays within 1..511, nothing is borrowed across fields).  Shifted right by the
// column's place in the window and masked, the three bits land where descriptor order wants them: the nine columns of window
// rows -3..-1 (and of rows +1..+3) fill 27 consecutive descriptor bits, index 9 row + column, with three instructions per
// column (subtract, shift, and-or), i.e. one per comparison instead of the pair form's one and a half.  The window's own row
// (eight comparisons) takes a byte compare and an add-with-carry each.  The collected bits say "not greater": one exclusive-or
// per word turns them into the descriptor.
constexpr int C97F_H = 16;                     // rows of a tile (a multiple of 4: four wavefronts share them; 8 and 32 measured no faster)
constexpr int C97F_ROWS = C97F_H + 4;          // triples start at image rows y0-3 .. y0+C97F_H
constexpr int C97F_PITCH = C97T_W + 8;         // one triple per column x0-4 .. x0+C97T_W+3

#define C97_LE(SC, SN) asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:BYTE_" #SC " src1_sel:BYTE_" #SN "\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" \
                           : "+v"(acc) : "v"(c), "v"(n) : "vcc")
// acc = 2 acc + (byte SC of c <= byte SN of n)
template <int SC, int SN>
__device__ __forceinline__ void c97_le(uint32_t &acc, uint32_t c, uint32_t n) {
    if constexpr (SC == 0 && SN == 0) C97_LE(0, 0); else if constexpr (SC == 0 && SN == 1) C97_LE(0, 1);
    else if constexpr (SC == 0 && SN == 2) C97_LE(0, 2); else if constexpr (SC == 0 && SN == 3) C97_LE(0, 3);
    else if constexpr (SC == 1 && SN == 0) C97_LE(1, 0); else if constexpr (SC == 1 && SN == 1) C97_LE(1, 1);
    else if constexpr (SC == 1 && SN == 2) C97_LE(1, 2); else if constexpr (SC == 1 && SN == 3) C97_LE(1, 3);
    else if constexpr (SC == 2 && SN == 0) C97_LE(2, 0); else if constexpr (SC == 2 && SN == 1) C97_LE(2, 1);
    else if constexpr (SC == 2 && SN == 2) C97_LE(2, 2); else if constexpr (SC == 2 && SN == 3) C97_LE(2, 3);
    else if constexpr (SC == 3 && SN == 0) C97_LE(3, 0); else if constexpr (SC == 3 && SN == 1) C97_LE(3, 1);
    else if constexpr (SC == 3 && SN == 2) C97_LE(3, 2); else C97_LE(3, 3);
}

// The 27 "not greater" bits of one pixel against nine columns of three rows: t[P + j] is the triple of window column j.
template <int P>
__device__ __forceinline__ uint32_t c97_fields(const uint32_t (&t)[12], uint32_t centres) {
    uint32_t acc;
    asm("v_and_b32 %0, %1, %2" : "=v"(acc) : "s"(0x04020100u), "v"(t[P] - centres));
#pragma unroll
    for (int j = 1; j < 9; ++j)
        asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc) : "v"((t[P + j] - centres) >> j), "s"(0x04020100u >> j));
    return acc;
}

__device__ __forceinline__ void census9x7_tile_fields(const uint8_t *__restrict__ img_all, int rows, int cols, uint32_t *__restrict__ out_all, int frame) {
    __shared__ uint32_t tile[(C97F_H + 6) * C97T_LROW];
    __shared__ __attribute__((aligned(16))) uint32_t trip[C97F_ROWS * C97F_PITCH];
    const long n = (long)rows * cols;
    const uint8_t *img = img_all + (long)frame * n;
    uint32_t *out = out_all + (long)frame * n * 2;
    const int x0 = blockIdx.x * C97T_W, y0 = blockIdx.y * C97F_H;
    for (int i = threadIdx.x; i < (C97F_H + 6) * C97T_LROW; i += 256) {
        const int r = i / C97T_LROW, k = i - r * C97T_LROW;
        const int gy = y0 - 3 + r, gx = x0 - 4 + 4 * k;
        uint32_t v = 0u;
        if (gy >= 0 && gy < rows && gx >= 0 && gx < cols) v = *reinterpret_cast<const uint32_t *>(img + (long)gy * cols + gx);
        tile[i] = v;
    }
    __syncthreads();
    // triples of rows r, r+1, r+2 of the staged tile, four columns (one staged dword of each row) per step
    for (int u = threadIdx.x; u < C97F_ROWS * (C97F_PITCH / 4); u += 256) {
        const int r = u / (C97F_PITCH / 4), k = u - r * (C97F_PITCH / 4);
        const uint32_t a = tile[r * C97T_LROW + k], b = tile[(r + 1) * C97T_LROW + k], c = tile[(r + 2) * C97T_LROW + k];
        uint4 e;
        e.x = ((a & 0xffu) << 18) | ((b & 0xffu) << 9) | (c & 0xffu) | 0x04020100u;
        e.y = (((a >> 8) & 0xffu) << 18) | (((b >> 8) & 0xffu) << 9) | ((c >> 8) & 0xffu) | 0x04020100u;
        e.z = (((a >> 16) & 0xffu) << 18) | (((b >> 16) & 0xffu) << 9) | ((c >> 16) & 0xffu) | 0x04020100u;
        e.w = ((a >> 24) << 18) | ((b >> 24) << 9) | (c >> 24) | 0x04020100u;
        *reinterpret_cast<uint4 *>(trip + r * C97F_PITCH + 4 * k) = e;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = x0 + lane * 4;
    if (x >= cols) return;
#pragma unroll
    for (int j = 0; j < C97F_H / 4; ++j) {
        const int r = wave * (C97F_H / 4) + j, y = y0 + r;            // staged row r is image row y - 3
        if (y >= rows) break;
        uint32_t hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
        if (y >= 3 && y < rows - 3 && x >= 4 && x < cols - 4) {         // (the border columns are whole lanes: cols is a multiple of 4)
            uint32_t up[12], dn[12], mid[3];
            const uint4 *qa = reinterpret_cast<const uint4 *>(trip + r * C97F_PITCH + 4 * lane);          // rows y-3 .. y-1
            const uint4 *qb = reinterpret_cast<const uint4 *>(trip + (r + 4) * C97F_PITCH + 4 * lane);    // rows y+1 .. y+3
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint4 va = qa[k], vb = qb[k];
                up[4 * k] = va.x; up[4 * k + 1] = va.y; up[4 * k + 2] = va.z; up[4 * k + 3] = va.w;
                dn[4 * k] = vb.x; dn[4 * k + 1] = vb.y; dn[4 * k + 2] = vb.z; dn[4 * k + 3] = vb.w;
                mid[k] = tile[(r + 3) * C97T_LROW + lane + k];
            }
            auto pixel = [&](auto p_c) __attribute__((always_inline)) {
                constexpr int P = decltype(p_c)::value;
                const uint32_t cb = (mid[1] >> (8 * P)) & 0xffu;
                const uint32_t c2 = (cb << 9) | cb, centres = (c2 << 9) | cb;
                const uint32_t above = c97_fields<P>(up, centres), below = c97_fields<P>(dn, centres);
                uint32_t left = 0u, right = 0u;
                // the window's own row: columns x-4 .. x-1 and x+1 .. x+4 are bytes P .. P+3 and P+5 .. P+8 of the three staged dwords
                c97_le<P, (P + 0) % 4>(left, mid[1], mid[(P + 0) / 4]);
                c97_le<P, (P + 1) % 4>(left, mid[1], mid[(P + 1) / 4]);
                c97_le<P, (P + 2) % 4>(left, mid[1], mid[(P + 2) / 4]);
                c97_le<P, (P + 3) % 4>(left, mid[1], mid[(P + 3) / 4]);
                c97_le<P, (P + 5) % 4>(right, mid[1], mid[(P + 5) / 4]);
                c97_le<P, (P + 6) % 4>(right, mid[1], mid[(P + 6) / 4]);
                c97_le<P, (P + 7) % 4>(right, mid[1], mid[(P + 7) / 4]);
                c97_le<P, (P + 8) % 4>(right, mid[1], mid[(P + 8) / 4]);
                hi[P] = ((above << 4) | left) ^ 0x7fffffffu;
                lo[P] = ((right << 27) | below) ^ 0x7fffffffu;
            };
            pixel(std::integral_constant<int, 0>{}); pixel(std::integral_constant<int, 1>{});
            pixel(std::integral_constant<int, 2>{}); pixel(std::integral_constant<int, 3>{});
        }
        uint4 *o = reinterpret_cast<uint4 *>(out + ((long)y * cols + x) * 2);
        o[0] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
        o[1] = make_uint4(lo[2], hi[2], lo[3], hi[3]);
    }
}

// Both images of a batch of pairs in one launch: grid z = 2 x frames, the first half is the left batch.
// FIELDS: the three-comparisons-per-subtraction form (ALIGNED only).
template <bool ALIGNED, bool FIELDS = false>
__global__ __launch_bounds__(256) void census9x7_pair_kernel(const uint8_t *__restrict__ left, const uint8_t *__restrict__ right, int rows, int cols,
                                                             long total_bytes, uint32_t *__restrict__ out_left, uint32_t *__restrict__ out_right, int n_frames) {
    const int z = blockIdx.z;
    if constexpr (FIELDS) {
        if (z < n_frames) census9x7_tile_fields(left, rows, cols, out_left, z);
        else census9x7_tile_fields(right, rows, cols, out_right, z - n_frames);
    } else {
        if (z < n_frames) census9x7_tile_packed<ALIGNED>(left, rows, cols, total_bytes, out_left, z);
        else census9x7_tile_packed<ALIGNED>(right, rows, cols, total_bytes, out_right, z - n_frames);
    }
}

__global__ __launch_bounds__(256) void census9x7_fields_kernel(const uint8_t *__restrict__ img_all, int rows, int cols, uint32_t *__restrict__ out_all) {
    census9x7_tile_fields(img_all, rows, cols, out_all, blockIdx.z);
}

template <bool ALIGNED, bool PACKED>
__global__ __launch_bounds__(256) void census9x7_tile_kernel(const uint8_t *__restrict__ img_all, int rows, int cols, long total_bytes,
                                                             uint32_t *__restrict__ out_all) {
    if (PACKED) { census9x7_tile_packed<ALIGNED>(img_all, rows, cols, total_bytes, out_all, blockIdx.z); return; }
    __shared__ uint32_t tile[(C97T_H + 6) * C97T_LROW];
    const long n = (long)rows * cols;
    const uint8_t *img = img_all + (long)blockIdx.z * n;
    uint32_t *out = out_all + (long)blockIdx.z * n * 2;
    const int x0 = blockIdx.x * C97T_W, y0 = blockIdx.y * C97T_H;
    for (int i = threadIdx.x; i < (C97T_H + 6) * C97T_LROW; i += 256) {
        const int r = i / C97T_LROW, k = i - r * C97T_LROW;
        const int gy = y0 - 3 + r, gx = x0 - 4 + 4 * k;
        uint32_t v = 0u;
        if (ALIGNED) {
            if (gy >= 0 && gy < rows && gx >= 0 && gx < cols) v = *reinterpret_cast<const uint32_t *>(img + (long)gy * cols + gx);
        } else if (gy >= 0 && gy < rows && gx > -4 && gx < cols) {
            const long a = (long)blockIdx.z * n + (long)gy * cols + gx;         // byte offset in the batch, may be negative by < 4
            const long a0 = a & ~3L;
            if (a0 >= 0 && a0 + 8 <= total_bytes) {
                const uint32_t d0 = *reinterpret_cast<const uint32_t *>(img_all + a0), d1 = *reinterpret_cast<const uint32_t *>(img_all + a0 + 4);
                v = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)(a & 3));
            } else {                                                            // the first and last bytes of the batch
                for (int b = 0; b < 4; ++b)
                    if (a + b >= 0 && a + b < total_bytes) v |= (uint32_t)img_all[a + b] << (8 * b);
            }
        }
        tile[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = x0 + lane * 4;
    if (x >= cols) return;
    uint32_t R[10][3];                 // staged rows wave*4 .. wave*4+9 = image rows y-3 .. y+6 of the lane's first row y
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t *q = tile + (wave * 4 + r) * C97T_LROW + lane;
        R[r][0] = q[0]; R[r][1] = q[1]; R[r][2] = q[2];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + wave * 4 + j;
        if (y >= rows) break;
        uint32_t hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
        if (y >= 3 && y < rows - 3) {
            const uint32_t (*W)[3] = R + j;
            C97_PX0(hi[0], lo[0], W);
            C97_PX1(hi[1], lo[1], W);
            C97_PX2(hi[2], lo[2], W);
            C97_PX3(hi[3], lo[3], W);
            if (ALIGNED) {                                                                               // the border columns
                if (x < 4) hi[0] = lo[0] = hi[1] = lo[1] = hi[2] = lo[2] = hi[3] = lo[3] = 0u;            // are whole lanes
                if (x >= cols - 4) hi[0] = lo[0] = hi[1] = lo[1] = hi[2] = lo[2] = hi[3] = lo[3] = 0u;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (x + p < 4 || x + p >= cols - 4) hi[p] = lo[p] = 0u;
            }
        }
        if (ALIGNED || (((cols & 1) == 0) && x + 3 < cols)) {
            uint4 *o = reinterpret_cast<uint4 *>(out + ((long)y * cols + x) * 2);
            o[0] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            o[1] = make_uint4(lo[2], hi[2], lo[3], hi[3]);
        } else {
            uint2 *o = reinterpret_cast<uint2 *>(out + ((long)y * cols + x) * 2);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (x + p < cols) o[p] = make_uint2(lo[p], hi[p]);
        }
    }
}

SVX_API int svx_census9x7_u8(svx_ctx *ctx, const uint8_t *img, uint32_t *census, int rows, int cols, int n_frames) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, img && census, SVX_ERR_INVALID_ARG, "svx_census9x7_u8: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_census9x7_u8: empty image");
    SVX_REQUIRE(ctx, n_frames <= 65535 && svx_cdiv(rows, C97_H) <= 65535, SVX_ERR_UNSUPPORTED, "svx_census9x7_u8: too many frames or rows per call");
    const bool fast = (((uintptr_t)img & 3u) == 0) && (((uintptr_t)census & 15u) == 0);
    const long total_bytes = (long)rows * cols * n_frames;
    const dim3 tgrid(svx_cdiv(cols, C97T_W), svx_cdiv(rows, C97T_H), n_frames);
    // SVX_OPT_CENSUS97_PACKED: 2 (default) three comparisons per subtraction where the width is a multiple of 4, 1 two per
    // subtraction, 0 the one-comparison-per-instruction form of the tile kernel.
    const int packed = svx_opt(ctx, SVX_OPT_CENSUS97_PACKED);
    if (fast && cols % 4 == 0) {
        const dim3 fgrid(svx_cdiv(cols, C97T_W), svx_cdiv(rows, C97F_H), n_frames);
        if (packed >= 2) census9x7_fields_kernel<<<fgrid, 256, 0, ctx->stream>>>(img, rows, cols, census);
        else if (packed) census9x7_tile_kernel<true, true><<<tgrid, 256, 0, ctx->stream>>>(img, rows, cols, total_bytes, census);
        else census9x7_tile_kernel<true, false><<<tgrid, 256, 0, ctx->stream>>>(img, rows, cols, total_bytes, census);
        SVX_LAUNCH_CHECK(ctx, "census9x7_tile_kernel");
    } else if (fast && cols >= 9) {
        if (packed) census9x7_tile_kernel<false, true><<<tgrid, 256, 0, ctx->stream>>>(img, rows, cols, total_bytes, census);
        else census9x7_tile_kernel<false, false><<<tgrid, 256, 0, ctx->stream>>>(img, rows, cols, total_bytes, census);
        SVX_LAUNCH_CHECK(ctx, "census9x7_tile_kernel");
    } else {
        census9x7_pixel_kernel<<<dim3(svx_cdiv(cols, C97_W), svx_cdiv(rows, C97_H), n_frames), 256, 0, ctx->stream>>>(img, rows, cols, (uint2 *)census);
        SVX_LAUNCH_CHECK(ctx, "census9x7_pixel_kernel");
    }
    return SVX_OK;
}

// svx_census9x7_u8 of both images of a batch of pairs, in one launch where the tile kernel applies.
static int census9x7_pair(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, uint32_t *cl, uint32_t *cr, int rows, int cols, int n_frames) {
    const bool fast = (((uintptr_t)left & 3u) == 0) && (((uintptr_t)right & 3u) == 0) && (((uintptr_t)cl & 15u) == 0) && (((uintptr_t)cr & 15u) == 0);
    const int packed = svx_opt(ctx, SVX_OPT_CENSUS97_PACKED);
    if (!fast || !packed || cols < 9 || 2 * n_frames > 65535 || svx_cdiv(rows, C97T_H) > 65535) {
        const int rc = svx_census9x7_u8(ctx, left, cl, rows, cols, n_frames);
        return rc ? rc : svx_census9x7_u8(ctx, right, cr, rows, cols, n_frames);
    }
    const long total_bytes = (long)rows * cols * n_frames;
    const dim3 grid(svx_cdiv(cols, C97T_W), svx_cdiv(rows, C97T_H), 2 * n_frames);
    if (cols % 4 == 0 && packed >= 2)
        census9x7_pair_kernel<true, true><<<dim3(grid.x, svx_cdiv(rows, C97F_H), grid.z), 256, 0, ctx->stream>>>(left, right, rows, cols, total_bytes, cl, cr, n_frames);
    else if (cols % 4 == 0) census9x7_pair_kernel<true><<<grid, 256, 0, ctx->stream>>>(left, right, rows, cols, total_bytes, cl, cr, n_frames);
    else census9x7_pair_kernel<false><<<grid, 256, 0, ctx->stream>>>(left, right, rows, cols, total_bytes, cl, cr, n_frames);
    SVX_LAUNCH_CHECK(ctx, "census9x7_pair_kernel");
    return SVX_OK;
}

SVX_API int svx_dense_wta(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                          int words, int max_disp, int view, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, census_left, census_right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    SVX_REQUIRE(ctx, words == 1 || words == 2, SVX_ERR_UNSUPPORTED, "svx_dense_wta: words must be 1 or 2");
    SVX_REQUIRE(ctx, view == SVX_VIEW_LEFT || view == SVX_VIEW_RIGHT, SVX_ERR_INVALID_ARG, "svx_dense_wta: bad view");
    // the right view is the left-view search on the mirrored pair with the images swapped
    if (view == SVX_VIEW_RIGHT) return launch_dense(ctx, census_right, census_left, rows, cols, n_frames, max_disp, disparity, 1, words, 1);
    return launch_dense(ctx, census_left, census_right, rows, cols, n_frames, max_disp, disparity, 1, words, 0);
}

__global__ __launch_bounds__(256) void lr_check_kernel(float *__restrict__ dl_all, const float *__restrict__ dr_all, int rows, int cols,
                                                       float threshold, float invalid) {
    const long n = (long)rows * cols;
    float *dl = dl_all + (long)blockIdx.z * n;
    const float *dr = dr_all + (long)blockIdx.z * n;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < 2 || x >= cols - 2 || y < 2 || y >= rows - 2) return;
    const float d1 = dl[(long)y * cols + x];
    // a value that is no disparity (negative, such as an earlier pass's invalid marker, NaN, beyond the row) is invalid and reads nothing
    if (!(d1 >= 0.0f) || !(d1 < (float)cols)) { dl[(long)y * cols + x] = invalid; return; }
    const int xr = x - (int)d1;
    // a match whose right pixel falls outside the searched region of the right view cannot be confirmed
    if (xr < 2 || fabsf(dr[(long)y * cols + xr] - d1) > threshold) dl[(long)y * cols + x] = invalid;
}

SVX_API int svx_lr_check(svx_ctx *ctx, float *disp_left, const float *disp_right, int rows, int cols, int n_frames,
                         float threshold, float invalid_value) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, disp_left && disp_right, SVX_ERR_INVALID_ARG, "svx_lr_check: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0 && n_frames > 0, SVX_ERR_EMPTY, "svx_lr_check: empty image");
    SVX_REQUIRE(ctx, n_frames <= 65535 && rows <= 65535, SVX_ERR_UNSUPPORTED, "svx_lr_check: too many frames or rows per call");
    lr_check_kernel<<<dim3(svx_cdiv(cols, 256), rows, n_frames), 256, 0, ctx->stream>>>(disp_left, disp_right, rows, cols, threshold, invalid_value);
    SVX_LAUNCH_CHECK(ctx, "lr_check_kernel");
    return SVX_OK;
}

// Sub-pixel refinement (no counterpart in the reference): parabola through the aggregated costs at d-1, d, d+1.
template <int WORDS>
__global__ __launch_bounds__(256) void subpixel_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                       int rows, int cols, int max_disp, float *__restrict__ disp_all) {
    const long n = (long)rows * cols;
    const uint32_t *cl = cl_all + (long)blockIdx.z * n * WORDS;
    const uint32_t *cr = cr_all + (long)blockIdx.z * n * WORDS;
    float *disp = disp_all + (long)blockIdx.z * n;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < 2 || x >= cols - 2 || y < 2 || y >= rows - 2) return;
    const float f = disp[(long)y * cols + x];
    if (!(f >= 1.0f) || !(f < (float)max_disp)) return;                           // also NaN and values no int can hold
    const int d = (int)f;
    if ((float)d != f || d + 1 > max_disp - 1 || x - (d + 1) < 5) return;
    int c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (int m = -2; m <= 2; ++m) {
        const uint32_t *lrow = cl + ((long)(y + m) * cols + x) * WORDS;
        const uint32_t *rrow = cr + ((long)(y + m) * cols + x - d) * WORDS;
#pragma unroll
        for (int k = -2; k <= 2; ++k)
#pragma unroll
            for (int w = 0; w < WORDS; ++w) {
                const uint32_t l = lrow[k * WORDS + w];
                c0 += popc32(l ^ rrow[(k + 1) * WORDS + w]);      // disparity d-1: right column x-d+1
                c1 += popc32(l ^ rrow[k * WORDS + w]);
                c2 += popc32(l ^ rrow[(k - 1) * WORDS + w]);      // disparity d+1
            }
    }
    const int den = c0 - 2 * c1 + c2;
    if (den > 0) disp[(long)y * cols + x] = (float)d + __fdiv_rn((float)(c0 - c2), (float)(2 * den));
}

// The same step with the left neighbourhood of a block (SPX_ROWS output rows x 256 columns, 2 pixels of halo) staged in LDS:
// the left descriptors are the same for every disparity and shared by neighbouring pixels, so they are fetched once per
// block instead of 25 times per pixel, which takes 200 of the 480 bytes per pixel off the vector cache; the right
// descriptors depend on each pixel's own disparity and are gathered from memory as before.  Same arithmetic, same results.
constexpr int SPX_ROWS = 4, SPX_W = 256, SPX_PITCH = SPX_W + 4;
// KEYS (two-word descriptors): the map holds the keys of the matrix-core search's winners instead of disparities (svx_common.hpp),
// SVX_MX_NO_KEY where the check rejected the pixel: every pixel is decoded and written, and the winner's own cost comes from its key.
template <int WORDS, bool KEYS = false>
__global__ __launch_bounds__(256) void subpixel_tile_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                            int rows, int cols, int max_disp, float *__restrict__ disp_all) {
    typedef typename std::conditional<WORDS == 2, uint2, uint32_t>::type desc_t;
    __shared__ desc_t tile[(SPX_ROWS + 4) * SPX_PITCH];
    const long n = (long)rows * cols;
    const desc_t *cl = reinterpret_cast<const desc_t *>(cl_all) + (long)blockIdx.z * n;
    const uint32_t *cr = cr_all + (long)blockIdx.z * n * WORDS;
    float *disp = disp_all + (long)blockIdx.z * n;
    const int x0 = blockIdx.x * SPX_W, y0 = 2 + blockIdx.y * SPX_ROWS;       // first output pixel of the block (rows 2 .. rows-3 have outputs)
    for (int i = threadIdx.x; i < (SPX_ROWS + 4) * SPX_PITCH; i += 256) {
        const int ty = i / SPX_PITCH, tx = i - ty * SPX_PITCH;
        const int gy = min(y0 - 2 + ty, rows - 1), gx = min(max(x0 - 2 + tx, 0), cols - 1);   // (clamped pixels are only read by pixels without an output)
        tile[i] = cl[(long)gy * cols + gx];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x < 2 || x >= cols - 2) return;
#pragma unroll 1
    for (int j = 0; j < SPX_ROWS; ++j) {
        const int y = y0 + j;
        if (y >= rows - 2) break;
        float f = disp[(long)y * cols + x];
        int c0 = 0, c1 = 0, c2 = 0;
        if constexpr (KEYS) {
            if (f <= SVX_MX_NO_KEY) { disp[(long)y * cols + x] = -1.0f; continue; }
            c1 = svx_mx_key_cost(f);
            f = svx_mx_key_disparity(f);
        }
        bool refine = f >= 1.0f && f < (float)max_disp;                          // (false for NaN and values no int can hold)
        const int d = refine ? (int)f : 0;
        refine = refine && (float)d == f && d + 1 <= max_disp - 1 && x - (d + 1) >= 5;
        if (!refine) {
            if (KEYS) disp[(long)y * cols + x] = f;
            continue;
        }
#pragma unroll
        for (int m = -2; m <= 2; ++m) {
            const desc_t *lrow = tile + (j + 2 + m) * SPX_PITCH + threadIdx.x + 2;
            const uint32_t *rrow = cr + ((long)(y + m) * cols + x - d) * WORDS;
#pragma unroll
            for (int k = -2; k <= 2; ++k) {
                const desc_t lv = lrow[k];
#pragma unroll
                for (int w = 0; w < WORDS; ++w) {
                    uint32_t l;
                    if constexpr (WORDS == 2) l = w ? lv.y : lv.x; else l = lv;
                    c0 += popc32(l ^ rrow[(k + 1) * WORDS + w]);      // disparity d-1: right column x-d+1
                    if (!KEYS) c1 += popc32(l ^ rrow[k * WORDS + w]);
                    c2 += popc32(l ^ rrow[(k - 1) * WORDS + w]);      // disparity d+1
                }
            }
        }
        const int den = c0 - 2 * c1 + c2;
        if (den > 0) disp[(long)y * cols + x] = (float)d + __fdiv_rn((float)(c0 - c2), (float)(2 * den));
        else if (KEYS) disp[(long)y * cols + x] = f;
    }
}

// acc + popcount(v) as the one instruction it is (left to itself the compiler sums the counts of a long chain in a tree of
// three-way additions, a quarter more instructions).
__device__ __forceinline__ int popc_acc(uint32_t v, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(acc));
    return r;
}

// Both neighbourhoods of a block in LDS (8 output rows x 256 columns, 1024 threads: four row groups of 256): the right descriptors
// a pixel needs lie at the columns x - d - 3 .. x - d + 3 of five rows, so a block's pixels with disparities up to d_cap read them
// from the columns x0 - d_cap - 3 .. x0 + 258 of its 12 rows, staged once (about 4 descriptors a pixel, every thread's loads in
// flight together) instead of gathered 35 at a time through the vector cache, where neighbouring pixels of one disparity fetch the
// same descriptors seven times over.  Lanes of one disparity read consecutive LDS words (no bank conflict); a pixel whose disparity
// exceeds d_cap gathers from memory as before.  RCH = 256-column chunks of the right tile.  Two blocks a CU need 64 registers a
// thread at most and 80 KiB of LDS a block: two chunks, disparities up to 250.  Same arithmetic, same results.
constexpr int SPX2_ROWS = 8, SPX2_GROUPS = 4, SPX2_CHUNKS = 2, SPX2_DCAP = 256 * SPX2_CHUNKS - (SPX_W + 6);
template <int WORDS, int RCH, bool KEYS = false>
__global__ __launch_bounds__(256 * SPX2_GROUPS, 8) void subpixel_pair_tile_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                                                int rows, int cols, int max_disp, int d_cap, float *__restrict__ disp_all) {
    typedef typename std::conditional<WORDS == 2, uint2, uint32_t>::type desc_t;
    extern __shared__ __align__(16) unsigned char spx_lds[];
    constexpr int TROWS = SPX2_ROWS + 4, RPT = TROWS / SPX2_GROUPS;           // tile rows, and tile rows a row group stages
    desc_t *ltile = reinterpret_cast<desc_t *>(spx_lds);
    desc_t *rtile = ltile + TROWS * SPX_PITCH;
    const int r_pitch = SPX_W + 6 + d_cap;
    const long n = (long)rows * cols;
    const desc_t *cl = reinterpret_cast<const desc_t *>(cl_all) + (long)blockIdx.z * n;
    const desc_t *cr = reinterpret_cast<const desc_t *>(cr_all) + (long)blockIdx.z * n;
    float *disp = disp_all + (long)blockIdx.z * n;
    const int col = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int x0 = blockIdx.x * SPX_W, y0 = 2 + blockIdx.y * SPX2_ROWS;      // first output pixel of the block (rows 2 .. rows-3 have outputs)
    const int rx0 = x0 - d_cap - 3;                                          // image column of the right tile's column 0
    {
        desc_t lv[RPT][2], rv[RPT][RCH];                                      // (clamped pixels are only read by pixels without an output)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const long grow = (long)min(y0 - 2 + grp + SPX2_GROUPS * r, rows - 1) * cols;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (col + 256 * c < SPX_PITCH) lv[r][c] = cl[grow + min(max(x0 - 2 + col + 256 * c, 0), cols - 1)];
#pragma unroll
            for (int c = 0; c < RCH; ++c)
                if (col + 256 * c < r_pitch) rv[r][c] = cr[grow + min(max(rx0 + col + 256 * c, 0), cols - 1)];
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int ty = grp + SPX2_GROUPS * r;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (col + 256 * c < SPX_PITCH) ltile[ty * SPX_PITCH + col + 256 * c] = lv[r][c];
#pragma unroll
            for (int c = 0; c < RCH; ++c)
                if (col + 256 * c < r_pitch) rtile[ty * r_pitch + col + 256 * c] = rv[r][c];
        }
    }
    __syncthreads();
    const int x = x0 + col;
    if (x < 2 || x >= cols - 2) return;
#pragma unroll 1
    for (int j = grp; j < SPX2_ROWS; j += SPX2_GROUPS) {
        const int y = y0 + j;
        if (y >= rows - 2) break;
        float f = disp[(long)y * cols + x];
        int c0 = 0, c1 = 0, c2 = 0;
        if constexpr (KEYS) {                                                     // (see subpixel_tile_kernel)
            if (f <= SVX_MX_NO_KEY) { disp[(long)y * cols + x] = -1.0f; continue; }
            c1 = svx_mx_key_cost(f);
            f = svx_mx_key_disparity(f);
        }
        bool refine = f >= 1.0f && f < (float)max_disp;                          // (false for NaN and values no int can hold)
        const int d = refine ? (int)f : 0;
        refine = refine && (float)d == f && d + 1 <= max_disp - 1 && x - (d + 1) >= 5;
        if (!refine) {
            if (KEYS) disp[(long)y * cols + x] = f;
            continue;
        }
        if (d <= d_cap) {
            const desc_t *lbase = ltile + j * SPX_PITCH + col;                    // window column -2 of window row -2
            const desc_t *rbase = rtile + j * r_pitch + (x - d - 3 - rx0);        // right column x - d - 3 of window row -2
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                desc_t rw[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) rw[k] = rbase[m * r_pitch + k];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const desc_t lw = lbase[m * SPX_PITCH + k];
                    if constexpr (WORDS == 2) {
                        c0 = popc_acc(lw.y ^ rw[k + 2].y, popc_acc(lw.x ^ rw[k + 2].x, c0));      // disparity d-1: right column x-d+1
                        if (!KEYS) c1 = popc_acc(lw.y ^ rw[k + 1].y, popc_acc(lw.x ^ rw[k + 1].x, c1));
                        c2 = popc_acc(lw.y ^ rw[k].y, popc_acc(lw.x ^ rw[k].x, c2));              // disparity d+1
                    } else {
                        c0 = popc_acc(lw ^ rw[k + 2], c0);
                        c1 = popc_acc(lw ^ rw[k + 1], c1);
                        c2 = popc_acc(lw ^ rw[k], c2);
                    }
                }
            }
        } else {
            const uint32_t *crw = cr_all + (long)blockIdx.z * n * WORDS;
#pragma unroll
            for (int m = -2; m <= 2; ++m) {
                const desc_t *lrow = ltile + (j + 2 + m) * SPX_PITCH + col + 2;
                const uint32_t *rrow = crw + ((long)(y + m) * cols + x - d) * WORDS;
#pragma unroll
                for (int k = -2; k <= 2; ++k) {
                    const desc_t lw = lrow[k];
#pragma unroll
                    for (int w = 0; w < WORDS; ++w) {
                        uint32_t l;
                        if constexpr (WORDS == 2) l = w ? lw.y : lw.x; else l = lw;
                        c0 += popc32(l ^ rrow[(k + 1) * WORDS + w]);
                        if (!KEYS) c1 += popc32(l ^ rrow[k * WORDS + w]);
                        c2 += popc32(l ^ rrow[(k - 1) * WORDS + w]);
                    }
                }
            }
        }
        const int den = c0 - 2 * c1 + c2;
        if (den > 0) disp[(long)y * cols + x] = (float)d + __fdiv_rn((float)(c0 - c2), (float)(2 * den));
        else if (KEYS) disp[(long)y * cols + x] = f;
    }
}

template <int WORDS, bool KEYS = false>
static int launch_subpixel_pair_tiles(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int max_disp, float *disparity) {
    const int d_cap = max_disp - 2 < 0 ? 0 : max_disp - 2 < SPX2_DCAP ? max_disp - 2 : SPX2_DCAP;
    const size_t lds = (size_t)(SPX2_ROWS + 4) * 4 * WORDS * (size_t)(SPX_PITCH + SPX_W + 6 + d_cap);
    const dim3 grid(svx_cdiv(cols, SPX_W), svx_cdiv(rows - 4, SPX2_ROWS), n_frames);
    SVX_HIP(ctx, hipFuncSetAttribute((const void *)subpixel_pair_tile_kernel<WORDS, SPX2_CHUNKS, KEYS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    subpixel_pair_tile_kernel<WORDS, SPX2_CHUNKS, KEYS><<<grid, 256 * SPX2_GROUPS, lds, ctx->stream>>>(cl, cr, rows, cols, max_disp, d_cap, disparity);
    SVX_LAUNCH_CHECK(ctx, "subpixel_pair_tile_kernel");
    return SVX_OK;
}

// keys: the map holds the matrix-core search's keys (two-word descriptors, rows >= 5, SVX_OPT_SUBPIXEL_TILES not 0: subpixel_from_keys).
static bool subpixel_from_keys(const svx_ctx *ctx, int rows) { return rows >= 5 && svx_opt(ctx, SVX_OPT_SUBPIXEL_TILES) != 0; }
static int launch_subpixel(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int words, int max_disp, float *disparity,
                           bool keys = false) {
    if (keys) {
        if (max_disp - 2 <= SPX2_DCAP && svx_opt(ctx, SVX_OPT_SUBPIXEL_TILES) == 2)
            return launch_subpixel_pair_tiles<2, true>(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity);
        subpixel_tile_kernel<2, true><<<dim3(svx_cdiv(cols, SPX_W), svx_cdiv(rows - 4, SPX_ROWS), n_frames), 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity);
        SVX_LAUNCH_CHECK(ctx, "subpixel_tile_kernel");
        return SVX_OK;
    }
    // SVX_OPT_SUBPIXEL_TILES: 2 (default) = both neighbourhoods in LDS where the range of disparities lets two blocks share a CU (wider
    // ranges measured no faster that way, one block a CU or smaller blocks), else the left one; 1 = the left one; 0 = none.
    if (rows >= 5 && svx_opt(ctx, SVX_OPT_SUBPIXEL_TILES) == 2 && max_disp - 2 <= SPX2_DCAP)
        return words == 1 ? launch_subpixel_pair_tiles<1>(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity)
                          : launch_subpixel_pair_tiles<2>(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity);
    if (rows >= 5 && svx_opt(ctx, SVX_OPT_SUBPIXEL_TILES) != 0) {
        dim3 grid(svx_cdiv(cols, SPX_W), svx_cdiv(rows - 4, SPX_ROWS), n_frames);
        if (words == 1) subpixel_tile_kernel<1><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity);
        else subpixel_tile_kernel<2><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity);
        SVX_LAUNCH_CHECK(ctx, "subpixel_tile_kernel");
        return SVX_OK;
    }
    dim3 grid(svx_cdiv(cols, 256), rows, n_frames);
    if (words == 1) subpixel_kernel<1><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity);
    else subpixel_kernel<2><<<grid, 256, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, disparity);
    SVX_LAUNCH_CHECK(ctx, "subpixel_kernel");
    return SVX_OK;
}

SVX_API int svx_subpixel_refine(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                                int words, int max_disp, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, census_left, census_right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    SVX_REQUIRE(ctx, words == 1 || words == 2, SVX_ERR_UNSUPPORTED, "svx_subpixel_refine: words must be 1 or 2");
    SVX_REQUIRE(ctx, rows <= 65535, SVX_ERR_UNSUPPORTED, "svx_subpixel_refine: too many rows per call");
    return launch_subpixel(ctx, census_left, census_right, rows, cols, n_frames, words, max_disp, disparity);
}

static int dense_census9x7_impl(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                                int max_disp, float lr_threshold, int subpixel, float *disparity);

SVX_API int svx_dense_census9x7_lr_u8(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                                      int max_disp, float lr_threshold, float *disparity) {
    return dense_census9x7_impl(ctx, left, right, rows, cols, n_frames, max_disp, lr_threshold, 0, disparity);
}

SVX_API int svx_dense_census9x7_subpixel_u8(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                                            int max_disp, float lr_threshold, float *disparity) {
    return dense_census9x7_impl(ctx, left, right, rows, cols, n_frames, max_disp, lr_threshold, 1, disparity);
}

// Left and right disparity maps from one sweep over the cost values (sweep-eligible sizes only).
// `partial` must hold both_views_partial_words() words.
static bool both_views_eligible(int rows, int cols, int max_disp) { return !(max_disp > SW_DMASK + 1 || cols < SW_T + 4 || rows < 5); }
// Sized for whichever kernel the search runs: the vector sweep leaves SW_PART minima per (row, x-block, 64-disparity
// chunk), the matrix-core search one key per right column its workgroups cover (svx_mx_partial_words); on narrow frames
// the latter is the larger of the two, and an option may change the choice between calls, so the larger one is taken.
static size_t both_views_partial_words(const svx_ctx *ctx, int rows, int cols, int n_frames, int words, int max_disp) {
    const size_t n_xblocks = (cols - 4 + SW_T - 1) / SW_T, n_chunks = (max_disp + 63) >> 6;
    size_t need = (size_t)n_frames * rows * n_xblocks * n_chunks * SW_PART;
    if (both_views_on_matrix_cores(ctx, rows, cols, words, max_disp)) {
        const size_t mx = svx_mx_partial_words(ctx, rows, cols, n_frames, words, max_disp);
        if (mx > need) need = mx;
    }
    return need;
}
// The partial minima of a call are kept below 4 GiB: a batch whose minima would take more is swept in groups of frames
// that share one buffer.  0 = not even one frame fits (or the size is not covered): one search per view instead.
static int both_views_group_frames(const svx_ctx *ctx, int rows, int cols, int n_frames, int words, int max_disp) {
    if (!both_views_eligible(rows, cols, max_disp) || rows > 65535) return 0;
    const size_t per_frame = both_views_partial_words(ctx, rows, cols, 1, words, max_disp);
    const size_t fit = ((size_t)1 << 30) / per_frame;
    return (int)(fit < (size_t)n_frames ? fit : (size_t)n_frames);
}

// What follows the sweep of a group of frames: the right view's disparity map, or the check of the left map against the
// partial minima (disp_right == nullptr).
static int launch_both_views(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int words, int max_disp,
                             float *disp_left, float *disp_right, float lr_threshold, int group, uint32_t *partial, bool sweep_only = false,
                             bool keys = false) {
    const size_t n = (size_t)rows * cols;
    const int n_xblocks = (cols - 4 + SW_T - 1) / SW_T;
    // Two-word descriptors at up to 256 disparities: the cost values are formed on the matrix cores (at 128: 1.37 ms against
    // 1.57 ms per 64 KITTI frames, and a cheaper merge); SVX_MX=0 keeps the vector sweep.  `partial` holds
    // both_views_partial_words() words for `group` frames, which covers either kernel's layout.
    const bool mx = both_views_on_matrix_cores(ctx, rows, cols, words, max_disp);
    for (int f0 = 0; f0 < n_frames; f0 += group) {
        const int nf = n_frames - f0 < group ? n_frames - f0 : group;
        const uint32_t *gl = cl + (size_t)f0 * n * words, *gr = cr + (size_t)f0 * n * words;
        float *dl = disp_left + (size_t)f0 * n;
        if (mx) {
            int rc = svx_launch_mx_sweep(ctx, gl, gr, rows, cols, nf, words, max_disp, dl, 1, (float *)partial, keys);
            if (rc) return rc;
            if (sweep_only) continue;
            rc = svx_launch_mx_right(ctx, (const float *)partial, rows, cols, nf, words, max_disp, dl, disp_right ? disp_right + (size_t)f0 * n : nullptr, lr_threshold, keys);
            if (rc) return rc;
            continue;
        }
        int rc = words == 1 ? launch_sweep<1, false>(ctx, gl, gr, rows, cols, nf, max_disp, dl, 1, partial)
                            : launch_sweep<2, false>(ctx, gl, gr, rows, cols, nf, max_disp, dl, 1, partial);
        if (rc) return rc;
        if (sweep_only) continue;
        const dim3 grid(svx_cdiv(cols, 256), rows, nf);
        if (disp_right) {
            right_view_merge_kernel<<<grid, 256, 0, ctx->stream>>>(partial, rows, cols, max_disp, n_xblocks, disp_right + (size_t)f0 * n);
            SVX_LAUNCH_CHECK(ctx, "right_view_merge_kernel");
        } else {
            lr_check_partial_kernel<<<grid, 256, 0, ctx->stream>>>(dl, partial, rows, cols, max_disp, n_xblocks, lr_threshold, -1.0f);
            SVX_LAUNCH_CHECK(ctx, "lr_check_partial_kernel");
        }
    }
    return SVX_OK;
}

SVX_API int svx_dense_wta_both(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                               int words, int max_disp, float *disp_left, float *disp_right) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, census_left, census_right, disp_left, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    SVX_REQUIRE(ctx, disp_right != nullptr, SVX_ERR_INVALID_ARG, "svx_dense_wta_both: null pointer");
    SVX_REQUIRE(ctx, words == 1 || words == 2, SVX_ERR_UNSUPPORTED, "svx_dense_wta_both: words must be 1 or 2");
    SVX_REQUIRE(ctx, rows <= 65535, SVX_ERR_UNSUPPORTED, "svx_dense_wta_both: too many rows per call");
    // sizes the sweep kernel does not cover: one search per view
    const int group = both_views_group_frames(ctx, rows, cols, n_frames, words, max_disp);
    if (group == 0) {
        rc = launch_dense(ctx, census_left, census_right, rows, cols, n_frames, max_disp, disp_left, 1, words, 0);
        if (rc) return rc;
        return launch_dense(ctx, census_right, census_left, rows, cols, n_frames, max_disp, disp_right, 1, words, 1);
    }
    void *ws;
    rc = svx_ws_reserve(ctx, both_views_partial_words(ctx, rows, cols, group, words, max_disp) * sizeof(uint32_t), &ws);
    if (rc) return rc;
    return launch_both_views(ctx, census_left, census_right, rows, cols, n_frames, words, max_disp, disp_left, disp_right, 0.f, group, (uint32_t *)ws);
}

// Measurement aid: launches exactly the both-views sweep kernel (left map + the right view's partial minima in the
// workspace), the kernel the left-right entry points spend most of their time in, so that it can be timed alone.
SVX_API int svx_probe_both_views_sweep(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                                       int n_frames, int words, int max_disp, float *disp_left) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, census_left, census_right, disp_left, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    SVX_REQUIRE(ctx, words == 1 || words == 2, SVX_ERR_UNSUPPORTED, "svx_probe_both_views_sweep: words must be 1 or 2");
    const int group = both_views_group_frames(ctx, rows, cols, n_frames, words, max_disp);
    SVX_REQUIRE(ctx, group > 0, SVX_ERR_UNSUPPORTED, "svx_probe_both_views_sweep: size not covered by the both-views sweep");
    void *ws;
    rc = svx_ws_reserve(ctx, both_views_partial_words(ctx, rows, cols, group, words, max_disp) * sizeof(uint32_t), &ws);
    if (rc) return rc;
    return launch_both_views(ctx, census_left, census_right, rows, cols, n_frames, words, max_disp, disp_left, nullptr, 0.f, group, (uint32_t *)ws, true);
}

// Which kernel the dense search of this size runs: 0 = vector sweep, 1 = matrix-core search, 2 = one thread per pixel.
SVX_API int svx_dense_kernel_choice(const svx_ctx *ctx, int rows, int cols, int words, int max_disp, int both_views) {
    if (both_views) {
        if (!both_views_eligible(rows, cols, max_disp) || rows > 65535) return 2;
        return both_views_on_matrix_cores(ctx, rows, cols, words, max_disp) ? 1 : 0;
    }
    if (max_disp > SW_DMASK + 1 || cols < SW_T + 4 || rows < 5) return 2;
    return svx_opt(ctx, SVX_OPT_MATRIX_CORES) != 0 && svx_mx_sweep_eligible(rows, cols, max_disp) ? 1 : 0;
}

static int dense_census9x7_impl(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                                int max_disp, float lr_threshold, int subpixel, float *disparity) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    rc = dense_args_ok(ctx, left, right, disparity, rows, cols, n_frames, max_disp);
    if (rc) return rc;
    const size_t n = (size_t)rows * cols * n_frames;
    const bool lr = lr_threshold >= 0.f;
    // SVX_OPT_LR_TWO_SWEEPS = 1 computes the right view with a second (mirrored) sweep instead of from the first one's cost values.
    const bool two_sweeps = svx_opt(ctx, SVX_OPT_LR_TWO_SWEEPS) == 1;
    // the partial minima of the one-sweep path take 80 words per (row, x-block, 64-disparity chunk) and frame
    const int group = lr && !two_sweeps ? both_views_group_frames(ctx, rows, cols, n_frames, 2, max_disp) : 0;
    const bool both = group > 0;
    // with the sub-pixel step behind a matrix-core search and its check, the map carries the winners' keys until that step decodes them:
    // it takes each winner's own cost from its key instead of counting it again
    const bool keys = subpixel && both && both_views_on_matrix_cores(ctx, rows, cols, 2, max_disp) && subpixel_from_keys(ctx, rows);
    const size_t part_words = both ? both_views_partial_words(ctx, rows, cols, group, 2, max_disp) : 0;
    void *ws;
    rc = svx_ws_reserve(ctx, (4 * n + part_words) * sizeof(uint32_t) + (lr && !both ? n * sizeof(float) : 0), &ws);
    if (rc) return rc;
    uint32_t *cl = (uint32_t *)ws, *cr = cl + 2 * n, *partial = cr + 2 * n;
    float *dr = (float *)(partial + part_words);
    rc = census9x7_pair(ctx, left, right, cl, cr, rows, cols, n_frames);
    if (rc) return rc;
    if (both) {
        // one sweep; the check reads the right view's partial minima directly (no right disparity map is formed)
        rc = launch_both_views(ctx, cl, cr, rows, cols, n_frames, 2, max_disp, disparity, nullptr, lr_threshold, group, partial, false, keys);
        if (rc) return rc;
    } else {
        rc = launch_dense(ctx, cl, cr, rows, cols, n_frames, max_disp, disparity, 1, 2, 0);
        if (rc) return rc;
        if (lr) rc = launch_dense(ctx, cr, cl, rows, cols, n_frames, max_disp, dr, 1, 2, 1);
        if (rc) return rc;
    }
    if (lr && !both) {
        rc = svx_lr_check(ctx, disparity, dr, rows, cols, n_frames, lr_threshold, -1.0f);
        if (rc) return rc;
    }
    if (subpixel) {
        SVX_REQUIRE(ctx, rows <= 65535, SVX_ERR_UNSUPPORTED, "svx_dense_census9x7_subpixel_u8: too many rows per call");
        rc = launch_subpixel(ctx, cl, cr, rows, cols, n_frames, 2, max_disp, disparity, keys);
    }
    return rc;
}
