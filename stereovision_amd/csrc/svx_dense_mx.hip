// Dense 5x5-aggregated Hamming winner-takes-all on the matrix cores (two-word descriptors, left view and, from the same
// accumulators, the right view).
//
// Same results as the vector sweep of svx_dense.hip (search rule of denseCensusStereo/denseCensusStereo.cpp:87-131
// on 64-bit descriptors); the cost values are formed as dot products.  With every descriptor bit b written as the
// number 1 - 2b, the dot product of two 64-element descriptors is 64 - 2 * Hamming distance, and the cost of a
// 5-column window is the dot product of the two windows' concatenated descriptors (K = 5 x 64).  One wavefront
// owns 32 left columns and the 32*T right columns they can match (T tiles of 32 x 32 left/right pairs): per census
// row it adds the row that enters the 5-row window (5 k-steps per tile) and subtracts the row that leaves it (the
// same products with the left operand negated), so the accumulators always hold
//     acc[tile][left column m][right column n] = 1600 - 2 * window cost  (+ a tag, below).
// Operands are fp4 (E2M1: +1.0 = 0x2, -1.0 = 0xA) with unit block scales: every product and sum is an exact small
// integer in the f32 accumulators.  The disparity tag and the admissibility mask live in the accumulators too: an
// element starts at (1023 - d) / 1024, or near -2^20 when the pair is inadmissible, so the running sum is its own
// sort key (21 significant bits, exact) and no instruction is spent on forming keys.  The winner of a left column is
// the maximum over the tiles and over the 32 lanes that hold its right columns; the winner of a right column is the
// maximum over the 16 registers and the two lane halves that hold its left columns, merged over the x-blocks that
// reach it by a second kernel.
#include "svx_common.hpp"
#include <utility>

typedef int mx_v8i __attribute__((ext_vector_type(8)));
typedef float mx_v16f __attribute__((ext_vector_type(16)));

#ifndef MX_X
#define MX_X 0      // measurement aid: bit 0 drops the staging work, bit 1 the winner selection, bit 2 the meeting point, bit 3 the leaving row's staging (results are wrong then)
#endif
#ifndef MX_PLANE_PAD
#define MX_PLANE_PAD 0      // measurement aid: extra pixels between the two planes of a staged row
#endif
#ifndef MX_P
#define MX_P 5      // right operands fetched this many matrix instructions ahead; divides the instructions of a row (5 or 10 per tile), so a
#endif              // row's instruction s always finds its operand in slot s % MX_P, also when the previous row fetched it
constexpr int MX_M = 32;                 // left columns per wavefront
#ifndef MX_MAX_DISP
#define MX_MAX_DISP 1024  // the disparity tag in the keys has ten bits; wider ranges than one launch covers run in passes (mx_plan)
#endif

#ifndef MX_W
#define MX_W 4                           // wavefronts (x-blocks of 32 left columns) per workgroup: they share one staged image of the census rows
#endif
constexpr int MX_NBUF = 3;               // staged rows in flight: the one being read, the one being written, and one in between (see mx_strip)
constexpr int MX_KROWS = 3;              // right view: key rows in LDS (the one being written out, the one being filled, and the next one the wavefronts ahead fill)
constexpr float MX_NO_KEY = SVX_MX_NO_KEY; // below every key, admissible or not
constexpr int mx_pass_step(int tiles) { return 32 * tiles; }   // a range beyond one pass: pass p takes the right columns 32 T p further left (all columns of its T tiles, see mx_plan)

template <int T, int WORDS = 2>
struct MxGeom {
    static constexpr int RPX = 32 * (MX_W + T - 1) + 4;   // right pixels staged per row for the workgroup (2-column window halo each side)
    static constexpr int LPX = 32 * MX_W + 4;
    static constexpr int PX = RPX + LPX;
    // Staging rounds (every thread takes one census word per round): FR rounds of right words, then FL rounds of left
    // words, then the words left over, 64 per wavefront: ULT wavefronts' worth of left words followed by URT of right words,
    // MX_W of them per round.  A wavefront's words of a round thus come from ONE image: their addresses are a scalar row
    // base plus a 32-bit offset per lane, and whether the leaving row's words are complemented is a scalar.
    static constexpr int NT = 64 * MX_W;                   // threads
    static constexpr int RW = WORDS * RPX, LW = WORDS * LPX;
    static constexpr int FR = RW / NT, FL = LW / NT;
    static constexpr int ULT = (LW - FL * NT + 63) / 64, URT = (RW - FR * NT + 63) / 64;
    static constexpr int TAILS = (ULT + URT + MX_W - 1) / MX_W;
    static constexpr int ROUNDS = FR + FL + TAILS;         // words per thread and row
    static constexpr int ROUND_PX = NT / WORDS;            // pixels a full round covers
    // LDS image of a staged row, two-word descriptors: one plane per descriptor word (= per k-half of the matrix
    // instruction), a pixel's 32 fp4 values (16 bytes) at [word][pixel], right pixels first.  A 16-lane group of a
    // 16-byte read then covers 256 consecutive bytes (with the two words of a pixel side by side it would hit every
    // bank twice).  A write group of 8 lanes stores 4 pixels of each plane; the planes are PLANE pixels apart,
    // PLANE = 4 mod 8, so that the two runs of 64 bytes land in different halves of the 32 banks a store sees.
    // One-word descriptors: a single plane (the two k-halves of an instruction are two different pixels, see mx_row).
    static constexpr int PLANE = WORDS == 2 ? PX + (12 - PX % 8) % 8 + MX_PLANE_PAD : PX;
    static constexpr int SLOTS = WORDS * PLANE;
    // Right view: the workgroup's x-blocks merge their best keys per right column in LDS before they leave for memory.  A key
    // row holds the 32 (MX_W + T - 1) right columns the workgroup's tiles cover (column xl0_wg - 32 (T - 1) - d_base first);
    // MX_KROWS rows are in use at a time (see mx_keys_flush).
    static constexpr int NKEYS = 32 * (MX_W + T - 1);
    static constexpr int KEY_ROUNDS = (NKEYS + 64 * MX_W - 1) / (64 * MX_W);   // keys per thread when a row is written out
    static_assert(WORDS == 1 || MX_PLANE_PAD != 0 || PLANE % 8 == 4, "plane offset must alternate the bank halves of a store");
};

// 32 descriptor bits -> 32 fp4 values: dword j holds bits j, j+4, .. j+28 (the same permutation on both operands).
// c22 = 0x22222222 in a vector register (an instruction takes one scalar constant only).
__device__ __forceinline__ uint4 mx_expand(uint32_t w, uint32_t c22) {
    uint4 r;
    const uint32_t m88 = 0x88888888u;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.x) : "v"(w << 3), "s"(m88), "v"(c22));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.y) : "v"(w << 2), "s"(m88), "v"(c22));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.z) : "v"(w << 1), "s"(m88), "v"(c22));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.w) : "v"(w), "s"(m88), "v"(c22));
    return r;
}

__device__ __forceinline__ mx_v8i mx_frag(const uint4 v) {
    mx_v8i f = {(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    return f;
}

// Pieces of the maximum over the 16 lanes of every row for 16 registers at once (lane l ends up with register l & 15);
// they are issued between the matrix instructions of the following row.  A DPP operand must have been written at least
// two instructions earlier, and the compiler may place a register copy or reload right in front of an asm block: every
// piece starts with its own padding (measured: no cost).
#define MX_DPP_GAP() asm volatile("s_nop 1")
// The vector instructions that read the accumulators are asm blocks, which the compiler does not pad against the
// matrix instruction that wrote them (an 8-pass result may be read 11 wait states after issue at the earliest).  Inside a
// row other matrix instructions always lie between a tile's last product and its first reader; where a reader follows the
// last product directly (the last tile behind the last row of a strip), this padding does: it is tied to the tile's
// registers, so that no reader is moved in front of it.
#define MX_MATRIX_RESULT_GAP(tile) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(tile))
// stage A, pair i: lanes 0..7 of a row (banks 0,1) keep register i, lanes 8..15 register i + 8
#define MX_RED_A(i) asm volatile("s_nop 1\n\t"                                                       \
                                 "v_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
                                 "v_max_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"     \
                                 : "=&v"(st.ra[i]) : "v"(st.best[i]), "v"(st.best[i + 8]))
// stage B, pair i: banks 0,2 keep register i (partner lane + 4), banks 1,3 register i + 4 (partner lane - 4)
#define MX_RED_B(i) asm volatile("s_nop 1\n\t"                                                       \
                                 "v_max_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t" \
                                 "v_max_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"     \
                                 : "=&v"(st.rb[i]) : "v"(st.ra[i]), "v"(st.ra[i + 4]))
#define MX_RED_C() asm volatile("s_nop 1\n\t"                                                                  \
                                "v_max_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
                                "v_max_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
                                "v_max_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  \
                                "v_max_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"      \
                                : "=&v"(st.rc[0]), "=&v"(st.rc[1]), "=&v"(st.rc[2]), "=&v"(st.rc[3])           \
                                : "v"(st.rb[0]), "v"(st.rb[1]), "v"(st.rb[2]), "v"(st.rb[3]))

__device__ __forceinline__ float mx_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float mx_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression in the body.
template <int... Is, class F>
__device__ __forceinline__ void mx_static_for(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>{}), ...); }

// Matrix instruction of a row after which the workgroup meets: the first one of the last tile (PER instructions per
// tile).  From then on the row only needs operands it has fetched already or fetches from its own staged row, and the
// next row's image, which every wavefront has finished writing by then, may be read.
template <int T, int PER>
constexpr int mx_meet_step() { return PER * (T - 1); }
// Matrix instruction after which staging unit v (a round of the entering row, then the rounds of the leaving row) is
// issued: spread over the instructions before the meeting, early enough that the operand wait of the instruction the
// workgroup meets behind covers the unit's LDS write (LDS operations of a wavefront complete in order, and that
// instruction's operand was fetched MX_P instructions ahead).
template <int T, int TG, int WORDS, int PER>
constexpr int mx_unit_step(int v) { return 1 + v * (mx_meet_step<T, PER>() - MX_P - 1) / (2 * MxGeom<TG, WORDS>::ROUNDS); }

// Where the operands of a row are: two-word descriptors read the entering and the leaving image at this lane's pixel;
// with one-word descriptors each of the five instructions of a tile has a read position of its own (cur[j], nxt[j]).
// Where the staging of a row fetches from: the strip's census rows of either image as buffer resources (scalar), and the
// byte offsets of the row after next (entering) and of the one three rows up (leaving) in them.  A lane adds its 32-bit
// offset inside the row: no lane forms a 64-bit address.
struct MxRows { __amdgpu_buffer_rsrc_t left, right; uint32_t off_new, off_old; };
__device__ __forceinline__ uint32_t mx_word(__amdgpu_buffer_rsrc_t image, uint32_t row_off, uint32_t byte_off) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(image, byte_off, row_off, 0);
}
template <int WORDS> struct MxPtrs;
template <> struct MxPtrs<2> { const uint4 *cur_e, *cur_l, *nxt_e; };
template <> struct MxPtrs<1> { const uint4 *cur[5], *nxt[5]; };

// T: tiles this wavefront computes; TG: tiles of the workgroup's geometry (staged pixels, key rows), T <= TG: an x-block at
// the frame's left border whose first tiles hold no admissible pair computes the last T of the TG tiles only (mx_strip).
template <int T, int TG = T>
struct MxState {
    using G = MxGeom<TG, 2>;                                  // (sized for two-word descriptors; one-word ones use fewer rounds)
    uint32_t wn[G::ROUNDS], wo[G::ROUNDS];                   // census words of the next row to stage (entering, leaving)
    uint32_t voff[G::ROUNDS];                                // byte offset of this thread's word of staging round i in a census row
    int slot0;                                               // where this thread's word of a full round goes in a staged row (+ a constant per round)
    int slot_t[2];                                           // the same for the rounds of left-over words, which differ from wavefront to wavefront
    bool t_any[2], t_left[2], t_has[2];                      // those rounds: this wavefront takes part / stages left words / this lane holds a word
    bool own;                                                // this lane's left column belongs to this x-block alone (the last block is shifted over its neighbour)
    float best[16], ra[8], rb[4], rc[4];                     // left view: winners of a row being folded / reduced
    float lift_l;                                            // both views: what the left view adds to its winner (0 or 4096, see mx_strip)
    float *out_row;                                          // left map: the x-block's first pixel of the row being written (the same for every lane) ...
    uint32_t out_off;                                        // ... and this lane's byte offset from it
    long out_stride;
    // right view (see mx_keys_flush)
    float *key0;                                             // this lane's slot for tile 0 in key row 0 (LDS)
    float *kcur;                                             // the same in the key row being filled
    float *kflush0;                                          // this thread's first key of key row 0
    int krow;                                                // key row being filled
    float kv[G::KEY_ROUNDS];                                 // keys on their way out
    float *part_row;                                         // where the workgroup's keys of the next finished row go (the same for every thread; thread tid's first key 4 tid bytes behind)
    uint32_t tid4;
    long part_stride;
};

// The workgroup's meeting point.  No wait is attached: what a wavefront must have finished (its staging writes) is
// covered by the operand wait of the matrix instruction in front of it, and its pending global loads stay in flight.
__device__ __forceinline__ void mx_meet() {
    if (!(MX_X & 4)) asm volatile("s_barrier" ::: "memory");
}

// Accumulator element i of tile u as the left view (L) or the right view reads it: as it is (what a view must not use was
// lowered when the accumulators were started, mx_strip).
template <int T, int TG>
__device__ __forceinline__ float mx_elem_l(const mx_v16f (&acc)[T], const MxState<T, TG> &st, int u, int i) { return acc[u][i]; }
template <int T, int TG>
__device__ __forceinline__ float mx_elem_r(const mx_v16f (&acc)[T], const MxState<T, TG> &st, int u, int i) { return acc[u][i]; }

// Right view, tile u: the best key of each right column over the tile's 32 left columns, in three pieces.
template <int T, int U, int TG>
__device__ __forceinline__ void mx_right_a(const mx_v16f (&acc)[T], const MxState<T, TG> &st, float (&x)[5]) {
    x[0] = mx_max3(mx_elem_r<T>(acc, st, U, 0), mx_elem_r<T>(acc, st, U, 1), mx_elem_r<T>(acc, st, U, 2));
    x[1] = mx_max3(mx_elem_r<T>(acc, st, U, 3), mx_elem_r<T>(acc, st, U, 4), mx_elem_r<T>(acc, st, U, 5));
    x[2] = mx_max3(mx_elem_r<T>(acc, st, U, 6), mx_elem_r<T>(acc, st, U, 7), mx_elem_r<T>(acc, st, U, 8));
    x[3] = mx_max3(mx_elem_r<T>(acc, st, U, 9), mx_elem_r<T>(acc, st, U, 10), mx_elem_r<T>(acc, st, U, 11));
}
template <int T, int U, int TG>
__device__ __forceinline__ void mx_right_b(const mx_v16f (&acc)[T], const MxState<T, TG> &st, float (&x)[5]) {
    x[4] = mx_max3(mx_elem_r<T>(acc, st, U, 12), mx_elem_r<T>(acc, st, U, 13), mx_elem_r<T>(acc, st, U, 14));
    x[0] = mx_max3(x[0], x[1], x[2]);
    x[0] = mx_max3(x[0], x[3], x[4]);
    x[0] = mx_max(x[0], mx_elem_r<T>(acc, st, U, 15));       // (right columns 2..4 leave 4096 too low, see mx_strip: svx_mx_right_disparity lifts them)
}
// Both lane halves hold left columns of the same right column; the workgroup's other x-blocks cover it as well: all of
// them meet in the key row (a 4-byte LDS operation serves the two halves in different cycles, so equal addresses do not collide).
// KROW: the key row when the caller knows it at compile time (the address is then a constant offset), -1: the one st.kcur points to.
template <int T, int U, int KROW = -1, int TG = T>
__device__ __forceinline__ void mx_right_c(MxState<T, TG> &st, const float (&x)[5]) {
    float *const p = KROW < 0 ? st.kcur + U * 32 : st.key0 + KROW * MxState<T, TG>::G::NKEYS + U * 32;
    __hip_atomic_fetch_max(p, x[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The next output row's keys go to the next key row.
template <int T, int TG>
__device__ __forceinline__ void mx_keys_next_row(MxState<T, TG> &st) {
    st.krow = st.krow == MX_KROWS - 1 ? 0 : st.krow + 1;
    st.kcur = st.key0 + st.krow * MxState<T, TG>::G::NKEYS;
}
// Writing out a finished key row, in two pieces (LDS read; store and reset).  The keys of output row r are complete when
// every wavefront has passed the meeting point of the census row after it (its last tile is read during that row's first
// products): the row is written out behind that point, while the wavefronts fill the row after it and, those that are ahead,
// the one after that, which is why three key rows are in use.  Key row `which` (0: the one being filled, MX_KROWS - 1: the one before it).
template <int T, int KROW = -1, int TG = T>
__device__ __forceinline__ void mx_keys_flush_read(MxState<T, TG> &st, int which) {
    using G = typename MxState<T, TG>::G;
    const int kr = KROW >= 0 ? KROW : st.krow + which >= MX_KROWS ? st.krow + which - MX_KROWS : st.krow + which;
#pragma unroll
    for (int i = 0; i < G::KEY_ROUNDS; ++i) st.kv[i] = st.kflush0[kr * G::NKEYS + i * 64 * MX_W];   // (a thread without a key in the last round reads inside the LDS block)
}
template <int T, int KROW = -1, int TG = T>
__device__ __forceinline__ void mx_keys_flush_write(MxState<T, TG> &st, int which, int tid) {
    using G = typename MxState<T, TG>::G;
    const int kr = KROW >= 0 ? KROW : st.krow + which >= MX_KROWS ? st.krow + which - MX_KROWS : st.krow + which;
#pragma unroll
    for (int i = 0; i < G::KEY_ROUNDS; ++i)
        if ((i + 1) * 64 * MX_W <= G::NKEYS || tid + i * 64 * MX_W < G::NKEYS) {
            *reinterpret_cast<float *>(reinterpret_cast<char *>(st.part_row + i * 64 * MX_W) + st.tid4) = st.kv[i];
            st.kflush0[kr * G::NKEYS + i * 64 * MX_W] = MX_NO_KEY;
        }
    st.part_row += st.part_stride;
}

// Decodes a winner: key = window score + (1023 - d) / 1024 >= -1600; an inadmissible pair sits near -2^20, and a pair that
// only the other view admits, unlifted, below -2495 (mx_strip).
__device__ __forceinline__ float mx_decode(float z) { return svx_mx_key_disparity(z); }

// pass_flags: what a search over part of the disparity range does with the left map.  Bit 0: an earlier pass over smaller
// disparities left its winner's key in the map, which competes (keys carry their disparity: the larger key is the better
// pair, the smaller disparity among equal costs); bit 1: a later pass follows, so the key itself is stored, not decoded.
template <int T, bool BOTH, int TG>
__device__ __forceinline__ void mx_reduce_tail(MxState<T, TG> &st, int lane, int pass_flags) {
    const float w0 = (lane & 1) ? st.rc[1] : st.rc[0];
    const float w1 = (lane & 1) ? st.rc[3] : st.rc[2];
    float x0, x1;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(x0), "=&v"(x1) : "v"(w0), "v"(w1));
    float z = (lane & 2) ? x1 : x0;                          // lane l: register (l & 15) over its 16-lane row
    const float zo = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, z), 0x401f));   // lane ^ 16
    z = mx_max(z, zo);
    if (BOTH) z += st.lift_l;                                // left columns cols-5..cols-3: every pair of theirs starts 4096 lower (mx_strip)
    if (st.own) {                                            // (a column that two x-blocks cover is written by one of them: with passes the map is read back)
        float *const out = reinterpret_cast<float *>(reinterpret_cast<char *>(st.out_row) + st.out_off);
        if (pass_flags & 1) z = mx_max(z, *out);
        *out = (pass_flags & 2) ? z : mx_decode(z);          // lanes l and l ^ 16 write the same value to the same place
    }
}

// One census row r of a strip.  Its operands are in the workgroup's staged image of row r (entering the 5-row window)
// and of row r-5 (leaving it, left pixels negated); the pointers of `ptr` include this lane's pixel offset.  The first
// MX_P right operands and the left operands (an) were fetched while the previous row ran.  Between its matrix
// instructions the wavefront expands its share of rows r+1 and r-4 into the workgroup's next staged rows (stage_e,
// stage_l), fetches the words of rows r+2 and r-3 and, in the steady state (STEADY: the row five rows up leaves the
// window, its products with the left operand negated are added too, and every row has an output), picks the winners:
// a tile's accumulators are final from its last product until its first product of the next row, so tiles 0 .. T-2 are
// read while later tiles of the same row run and tile T-1 while tile 0 of the next row runs, followed by the lane
// reduction and the store of that previous row.  After the first product of the last tile the workgroup meets; behind
// that point the next row's left operands and first right operands are fetched from the next image.
//
// Two-word descriptors: an instruction's k = 64 is one pixel (its two words are the k-halves); a tile takes the five
// window columns of the entering row and, in the steady state, the five of the leaving row (PER = 10).
// One-word descriptors: k = 64 is TWO pixels, one per k-half (lanes 0..31 supply one, lanes 32..63 the other), so the
// ten window pixels of the entering and the leaving row make five instructions, (e0,e1) (e2,e3) (e4,l0) (l1,l2) (l3,l4),
// each with its own per-lane read position.  Before the steady state the leaving pixels are not there yet: the fourth
// and fifth instruction are skipped and the third multiplies its second half by zero.
template <int T, int WORDS, bool STEADY, bool BOTH, int KR = -1, int TG = T>
__device__ __forceinline__ void mx_row(mx_v16f (&acc)[T], const MxPtrs<WORDS> &ptr, uint4 *stage_e, uint4 *stage_l, MxState<T, TG> &st,
                                       mx_v8i (&an)[5], uint4 (&fb)[MX_P], const MxRows &rows_g, int lane, int tid, int pass_flags) {
    using G = MxGeom<TG, WORDS>;
    constexpr int PER = WORDS == 2 && STEADY ? 10 : 5, S = T * PER, P = MX_P;
    constexpr int TILE0 = 32 * (TG - T);             // the first computed tile in the staged right pixels
    constexpr int MEET = mx_meet_step<T, PER>();
    static_assert(S % P == 0 && S - P >= MEET && MEET - P - 1 >= 1, "operands of the next row are fetched behind the meeting point only, into the slots the row expects");
    static_assert(T >= 3, "the fold schedule needs three tiles");
    uint32_t c22;
    asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
    const uint32_t m88 = 0x88888888u;
    mx_v8i ao[5];                                    // two words: left operands of the leaving row; one word, before the steady state: an[2] with its second half zero
    if constexpr (WORDS == 2) {
        if (STEADY) {
#pragma unroll
            for (int k = 0; k < 5; ++k) ao[k] = mx_frag(ptr.cur_l[G::RPX + k]);
        }
    } else if constexpr (!STEADY) {
        ao[2] = an[2];
        if (lane >= 32) ao[2] = mx_v8i{0, 0, 0, 0, 0, 0, 0, 0};
    }
    auto right = [&](int s) -> uint4 {               // right operand of matrix instruction s; s >= S: of the next row's instruction s - S
        const bool next = s >= S;
        const int s2 = next ? s - S : s, t = s2 / PER, j = s2 % PER, k = j % 5;
        if constexpr (WORDS == 2) return (next ? ptr.nxt_e : j >= 5 ? ptr.cur_l : ptr.cur_e)[TILE0 + 32 * t + k];
        else return (next ? ptr.nxt[j] : ptr.cur[j])[TILE0 + 32 * t];
    };
    float rx[5];
    __builtin_amdgcn_sched_barrier(0);
    mx_static_for(std::make_integer_sequence<int, S>{}, [&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        constexpr int t = s / PER, j = s % PER, k = j % 5;
        if constexpr (WORDS == 2 || STEADY || j <= 2) {
            const mx_v8i b = mx_frag(fb[s % P]);
            const mx_v8i a = WORDS == 2 ? (j >= 5 ? ao[k] : an[k]) : (!STEADY && j == 2 ? ao[2] : an[j]);
            acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 4, 4, 0, 0, 0, 0);   // zero scale arguments select v_mfma_f32_32x32x64_f8f6f4, the form without scale operands (same products, shorter encoding, issues faster)
        }
        asm volatile("" : "+v"(acc[t]));      // keeps the instruction at this place among the ordered statements around it
        if constexpr (s == MEET) mx_meet();
        fb[s % P] = right(s + P);
        if constexpr (s >= MEET && s < MEET + 5) {   // the last use of this left operand was this instruction
            if constexpr (WORDS == 2) an[s - MEET] = mx_frag(ptr.nxt_e[G::RPX + (s - MEET)]);
            else an[s - MEET] = mx_frag(ptr.nxt[s - MEET][G::RPX]);
        }

        // this thread's share of the next staged rows
        mx_static_for(std::make_integer_sequence<int, 2 * G::ROUNDS>{}, [&](auto v_c) __attribute__((always_inline)) {
            constexpr int v = decltype(v_c)::value;
            if constexpr (mx_unit_step<T, TG, WORDS, PER>(v) == s && !(MX_X & 1) && !((MX_X & 8) && v >= G::ROUNDS)) {
                constexpr bool old = v >= G::ROUNDS;                  // entering row first, then the leaving row
                constexpr int i = old ? v - G::ROUNDS : v;
                constexpr int kind = i < G::FR ? 0 : i < G::FR + G::FL ? 1 : 2;     // right words, left words, left-over words
                constexpr int kt = kind == 2 ? i - G::FR - G::FL : 0;
                auto unit = [&]() __attribute__((always_inline)) {
                    uint32_t w = old ? st.wo[i] : st.wn[i];
                    if constexpr (old) {                               // the leaving row's left words are complemented
                        if constexpr (kind == 1) w = ~w;
                        else if constexpr (kind == 2) w ^= st.t_left[kt] ? 0xffffffffu : 0u;
                    }
                    uint4 e;
                    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(e.x) : "v"(w << 3), "s"(m88), "v"(c22));
                    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(e.y) : "v"(w << 2), "s"(m88), "v"(c22));
                    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(e.z) : "v"(w << 1), "s"(m88), "v"(c22));
                    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(e.w) : "v"(w), "s"(m88), "v"(c22));
                    uint4 *const img = old ? stage_l : stage_e;
                    if constexpr (kind == 0) img[st.slot0 + i * G::ROUND_PX] = e;
                    else if constexpr (kind == 1) img[st.slot0 + G::RPX + (i - G::FR) * G::ROUND_PX] = e;
                    else if (st.t_has[kt]) img[st.slot_t[kt]] = e;
                    const uint32_t row_off = old ? rows_g.off_old : rows_g.off_new;
                    uint32_t x;
                    if constexpr (kind == 0) x = mx_word(rows_g.right, row_off, st.voff[i]);
                    else if constexpr (kind == 1) x = mx_word(rows_g.left, row_off, st.voff[i]);
                    else x = st.t_left[kt] ? mx_word(rows_g.left, row_off, st.voff[i]) : mx_word(rows_g.right, row_off, st.voff[i]);
                    if constexpr (old) st.wo[i] = x; else st.wn[i] = x;
                };
                if constexpr (kind < 2) unit();
                else if (st.t_any[kt]) unit();
            }
        });

        // the winners of the previous output: a timetable of ten slots per tile; with five instructions per tile each takes two
        auto select = [&](auto j_c) __attribute__((always_inline)) {
            constexpr int j = decltype(j_c)::value;
            // left view
            // (five tiles: two three-way maxima per element; fewer tiles: tile by tile)
            // (a tile is first read from slot 2 on, i.e. behind at least two matrix instructions after its last product:
            // the readers are asm blocks, which the compiler does not pad against the matrix instruction, see MX_MATRIX_RESULT_GAP)
            constexpr int e = 2 * (j - 2);                             // slots 2 .. 9 take two accumulator elements each
            if constexpr (T == 5) {
                if constexpr (t == 0 && j >= 2) {                      // tiles 3 and 4 of the previous row
                    st.best[e] = mx_max3(st.best[e], mx_elem_l<T>(acc, st, 3, e), mx_elem_l<T>(acc, st, 4, e));
                    st.best[e + 1] = mx_max3(st.best[e + 1], mx_elem_l<T>(acc, st, 3, e + 1), mx_elem_l<T>(acc, st, 4, e + 1));
                }
                if constexpr (t == 3 && j >= 2) {                      // tiles 0, 1 and 2 of this row
                    st.best[e] = mx_max3(mx_elem_l<T>(acc, st, 0, e), mx_elem_l<T>(acc, st, 1, e), mx_elem_l<T>(acc, st, 2, e));
                    st.best[e + 1] = mx_max3(mx_elem_l<T>(acc, st, 0, e + 1), mx_elem_l<T>(acc, st, 1, e + 1), mx_elem_l<T>(acc, st, 2, e + 1));
                }
            } else {
                if constexpr (t == 0 && j >= 2) {                      // last tile of the previous row
                    st.best[e] = mx_max(st.best[e], mx_elem_l<T>(acc, st, T - 1, e));
                    st.best[e + 1] = mx_max(st.best[e + 1], mx_elem_l<T>(acc, st, T - 1, e + 1));
                }
                if constexpr (t == 2 && j >= 5 && j < 9) {             // tiles 0 and 1 of this row
#pragma unroll
                    for (int i = 4 * (j - 5); i < 4 * (j - 5) + 4; ++i)
                        st.best[i] = mx_max(mx_elem_l<T>(acc, st, 0, i), mx_elem_l<T>(acc, st, 1, i));
                }
                if constexpr (t >= 3 && j >= 2) {                      // tile t - 1 of this row
                    st.best[e] = mx_max(st.best[e], mx_elem_l<T>(acc, st, t - 1, e));
                    st.best[e + 1] = mx_max(st.best[e + 1], mx_elem_l<T>(acc, st, t - 1, e + 1));
                }
            }
            if constexpr (t == 1 && j < 8) { MX_RED_A(j); }
            if constexpr (t == 2 && j == 0) { MX_RED_B(0); MX_RED_B(1); }
            if constexpr (t == 2 && j == 1) { MX_RED_B(2); MX_RED_B(3); }
            if constexpr (t == 2 && j == 2) { MX_RED_C(); }
            if constexpr (t == 2 && j == 3) { mx_reduce_tail<T, BOTH>(st, lane, pass_flags); st.out_row += st.out_stride; }
            // right view: tile t - 1 of this row, or the last tile of the previous row
            if constexpr (BOTH) {
                constexpr int u = t == 0 ? T - 1 : t - 1;
                if constexpr (j == 4) mx_right_a<T, u>(acc, st, rx);
                if constexpr (j == 8) mx_right_b<T, u>(acc, st, rx);
                if constexpr (j == 9) {
                    if constexpr (KR < 0) {
                        mx_right_c<T, u>(st, rx);
                        if constexpr (t == 0) mx_keys_next_row<T>(st);    // the following tiles belong to this row
                    } else mx_right_c<T, u, (t == 0 ? KR : (KR + 1) % MX_KROWS)>(st, rx);
                }
            }
        };
        if constexpr (STEADY && !(MX_X & 2)) {
            if constexpr (PER == 10) select(std::integral_constant<int, j>{});
            else { select(std::integral_constant<int, 2 * j>{}); select(std::integral_constant<int, 2 * j + 1>{}); }
            if constexpr (BOTH) {                            // every wavefront has passed the meeting point: the previous output row's keys are complete
                if constexpr (s == MEET + 1) mx_keys_flush_read<T, KR>(st, MX_KROWS - 1);
                if constexpr (s == MEET + 3) mx_keys_flush_write<T, KR>(st, MX_KROWS - 1, tid);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Between the first five rows and the steady state: the first output row's tiles 0 .. T-2.
template <int T, bool BOTH, int TG>
__device__ __forceinline__ void mx_first_output(const mx_v16f (&acc)[T], MxState<T, TG> &st) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float b = mx_max(mx_elem_l<T>(acc, st, 0, i), mx_elem_l<T>(acc, st, 1, i));
#pragma unroll
        for (int u = 2; u < (T == 5 ? 3 : T - 1); ++u) b = mx_max(b, mx_elem_l<T>(acc, st, u, i));   // the rest follows with the next row
        st.best[i] = b;
    }
    if (BOTH) {
        mx_static_for(std::make_integer_sequence<int, T - 1>{}, [&](auto u_c) __attribute__((always_inline)) {
            constexpr int u = decltype(u_c)::value;
            float rx[5];
            mx_right_a<T, u>(acc, st, rx);
            mx_right_b<T, u>(acc, st, rx);
            mx_right_c<T, u>(st, rx);
        });
    }
}

// After the last row of a strip: its last tile, the lane reduction and the stores (nothing follows to hide them behind).
template <int T, bool BOTH, int TG>
__device__ __forceinline__ void mx_last_output(mx_v16f (&acc)[T], MxState<T, TG> &st, int lane, int tid, int pass_flags) {
    MX_MATRIX_RESULT_GAP(acc[T - 1]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (T == 5) st.best[i] = mx_max(st.best[i], mx_elem_l<T>(acc, st, 3, i));
        st.best[i] = mx_max(st.best[i], mx_elem_l<T>(acc, st, T - 1, i));
    }
    MX_DPP_GAP();
    MX_RED_A(0); MX_RED_A(1); MX_RED_A(2); MX_RED_A(3); MX_RED_A(4); MX_RED_A(5); MX_RED_A(6); MX_RED_A(7);
    MX_DPP_GAP();
    MX_RED_B(0); MX_RED_B(1); MX_RED_B(2); MX_RED_B(3);
    MX_DPP_GAP();
    MX_RED_C();
    mx_reduce_tail<T, BOTH>(st, lane, pass_flags);
    if (BOTH) {
        float rx[5];
        mx_right_a<T, T - 1>(acc, st, rx);
        mx_right_b<T, T - 1>(acc, st, rx);
        mx_right_c<T, T - 1>(st, rx);
        __syncthreads();                                     // the last output row's keys are complete
        mx_keys_flush_read<T>(st, 0);
        mx_keys_flush_write<T>(st, 0, tid);
    }
}

// Strip of the workgroup's x-blocks: census rows y_lo-2 .. y_hi+1, output rows y_lo .. y_hi-1; this wavefront owns the
// x-block whose first left column is xl0, the workgroup's first x-block starts at xl0_wg.
//
// Staged rows: MX_NBUF images, each an entering and a leaving row, used in turn.  Row r reads image r % 3 and
// writes image (r + 1) % 3; a wavefront that is ahead of the others has passed the meeting point of row r - 1 at
// least, so every wavefront has finished reading the image it overwrites (last read by row r - 2), and what it reads
// behind the meeting point of row r (the next image) is complete.
//
// TG > T: the x-block computes the last T of the geometry's TG tiles only (the others hold no admissible pair: their right
// columns lie left of the frame).  Staging, key rows and the searched disparity range stay those of TG tiles.
template <int T, int WORDS, bool BOTH, int TG = T>
__device__ __forceinline__ void mx_strip(uint4 *lds, const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr,
                                         int cols, int max_disp, int d_base, int pass_flags, int xl0, int xl0_wg, int own_from, int y_lo, int n_rows, int tid,
                                         float *__restrict__ disp, float *__restrict__ part_row0, long part_stride) {
    using G = MxGeom<TG, WORDS>;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int r0 = xl0 - 32 * (T - 1) - d_base;      // right column of (computed) tile 0, lane 0 (this pass's tiles hold the right columns xl0 - 32 (TG - 1) - d_base .. + 32 TG - 1)

    // Accumulator element i of tile t is (left column xl0 + m, right column r0 + 32 t + n).  It starts at
    // (1023 - d) / 1024, so that the running window score carries its disparity tag below the integer part
    // (21 significant bits: exact), or near -2^20 when the pair is inadmissible.  Left view: right column >= 5;
    // right view: right column >= 2 and left column <= cols-6 (the mirror image of the left view's rule).  A pair that
    // only ONE of the views admits (right columns 2..4; left columns cols-5..cols-3) starts 4096 lower (still exact: 23
    // significant bits, keys stay above -8192).  The other view never prefers it to a pair of its own (those lie above
    // -1600) and decodes it as "no disparity" when it has none (mx_decode).  The view that admits it gets the true key
    // back by adding 4096 AFTER its reduction: such a right column holds nothing but lowered or inadmissible pairs in every
    // x-block and pass, so its merged key is lifted where it is decoded (svx_mx_right_disparity: the search spends nothing on
    // it), and such a left column likewise, so the left view lifts its winner (lift_l, one addition per row).  No accumulator
    // read pays for the frame's borders.
    mx_v16f acc[T];
    MxState<T, TG> st;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int xr = r0 + 32 * t + n;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int d = xl0 + m - xr;
            const bool in_range = d >= 0 && d < max_disp;       // (the passes' right columns do not overlap: every pair is in one pass)
            const bool ok_l = xr >= 5, ok_r = xr >= 2 && xl0 + m <= cols - 6;          // what the left / the right view admits
            const bool ok = in_range && (BOTH ? ok_l || ok_r : ok_l);
            acc[t][i] = ok ? (float)(1023 - d) * (1.0f / 1024.0f) - (BOTH && !(ok_l && ok_r) ? 4096.0f : 0.0f) : -1048576.0f;
        }
    }

    // staging (rounds: MxGeom): right word q of a staged row is word (xl0_wg - 32 (T - 1) - d_base - 2) * WORDS + q of the right
    // census row, left word q is word (xl0_wg - 2) * WORDS + q of the left one.  Words outside the frame are clamped into it:
    // left of the frame they only feed inadmissible pairs, right of it (the columns behind the last x-block of a workgroup
    // that the frame does not fill) nothing reads them.
    const long row_words = (long)cols * WORDS;
    const uint32_t *const cl0 = cl + (long)(y_lo - 2) * row_words, *const cr0 = cr + (long)(y_lo - 2) * row_words;
    const uint32_t row_bytes = 4u * (uint32_t)row_words;
    MxRows rg;
    rg.left = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cl0), 0, (int)(row_bytes * (uint32_t)n_rows), 0x00020000);
    rg.right = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cr0), 0, (int)(row_bytes * (uint32_t)n_rows), 0x00020000);
    auto right_off = [&](int q) -> uint32_t { return 4u * (uint32_t)min(max((long)(xl0_wg - 32 * (TG - 1) - d_base - 2) * WORDS + q, 0L), row_words - 1); };
    auto left_off = [&](int q) -> uint32_t { return 4u * (uint32_t)min(max((long)(xl0_wg - 2) * WORDS + q, 0L), row_words - 1); };
    auto slot_of = [&](int q, bool left) -> int { return WORDS == 2 ? (q & 1) * G::PLANE + (left ? G::RPX : 0) + (q >> 1) : (left ? G::RPX : 0) + q; };
#pragma unroll
    for (int i = 0; i < G::FR; ++i) st.voff[i] = right_off(i * G::NT + tid);
#pragma unroll
    for (int i = 0; i < G::FL; ++i) st.voff[G::FR + i] = left_off(i * G::NT + tid);
    st.slot0 = slot_of(tid, false);
#pragma unroll
    for (int k = 0; k < G::TAILS; ++k) {
        const int v = k * MX_W + wave;               // this wavefront's unit of left-over words: ULT units of left words, then URT of right words
        const bool left = v < G::ULT;
        const int q = left ? G::FL * G::NT + 64 * v + lane : G::FR * G::NT + 64 * (v - G::ULT) + lane;
        st.t_any[k] = v < G::ULT + G::URT;
        st.t_left[k] = left;
        st.t_has[k] = q < (left ? G::LW : G::RW);
        st.slot_t[k] = slot_of(min(q, (left ? G::LW : G::RW) - 1), left);
        st.voff[G::FR + G::FL + k] = left ? left_off(q) : right_off(q);
    }
    // word of staging round i in census row r of the strip
    auto word = [&](int i, int r) -> uint32_t {
        const bool left = i < G::FR ? false : i < G::FR + G::FL ? true : st.t_left[i - G::FR - G::FL];
        return left ? mx_word(rg.left, r * row_bytes, st.voff[i]) : mx_word(rg.right, r * row_bytes, st.voff[i]);
    };
    auto put0 = [&](int i, uint4 e) {                // round i's word into the first staged row
        if (i < G::FR) lds[st.slot0 + i * G::ROUND_PX] = e;
        else if (i < G::FR + G::FL) lds[st.slot0 + G::RPX + (i - G::FR) * G::ROUND_PX] = e;
        else if (st.t_any[i - G::FR - G::FL] && st.t_has[i - G::FR - G::FL]) lds[st.slot_t[i - G::FR - G::FL]] = e;
    };
    const int last = n_rows - 1;
    const int i_out = lane & 15;
    st.out_row = disp + (long)y_lo * cols + xl0;
    st.out_off = 4u * (uint32_t)((i_out & 3) + 8 * (i_out >> 2) + 4 * h);
    st.own = xl0 + (i_out & 3) + 8 * (i_out >> 2) + 4 * h >= own_from;
    st.lift_l = BOTH && xl0 + (i_out & 3) + 8 * (i_out >> 2) + 4 * h > cols - 6 ? 4096.0f : 0.0f;
    st.out_stride = cols;
    if (BOTH) {
        // key rows behind the staged images; tile u of this wavefront covers right columns xl0 - 32 (T - 1) - d_base + 32 u + n
        float *const keys = reinterpret_cast<float *>(lds + MX_NBUF * 2 * G::SLOTS);
        for (int i = tid; i < MX_KROWS * G::NKEYS; i += 64 * MX_W) keys[i] = MX_NO_KEY;   // (the meeting behind the first staged row covers this)
        st.key0 = keys + (xl0 - xl0_wg) + 32 * (TG - T) + n;
        st.kcur = st.key0;
        st.krow = 0;
        st.kflush0 = keys + tid;
        st.part_row = part_row0;
        st.tid4 = 4u * (uint32_t)tid;
        st.part_stride = part_stride;
    }

    // image b: entering row at lds + 2 b SLOTS, leaving row behind it
    auto img_e = [&](int b) -> uint4 * { return lds + (2 * b) * G::SLOTS; };
    auto img_l = [&](int b) -> uint4 * { return lds + (2 * b + 1) * G::SLOTS; };
    // Read positions of this lane.  Two-word descriptors: its pixel in the plane of its k-half.  One-word descriptors: for
    // instruction j of a tile the window pixel its k-half takes: (e0,e1) (e2,e3) (e4,l0) (l1,l2) (l3,l4).
    const int rd = (WORDS == 2 ? h * G::PLANE : 0) + (xl0 - xl0_wg) + n;
    int rd1[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int slot = 2 * j + h;                  // 0..4: entering row's window column, 5..9: the leaving row's
        rd1[j] = (slot >= 5 ? G::SLOTS + slot - 5 : slot) + rd;
    }
    MxPtrs<WORDS> ptr;
    auto point = [&](int b_cur, int b_nxt) {
        if constexpr (WORDS == 2) {
            ptr.cur_e = img_e(b_cur) + rd; ptr.cur_l = img_l(b_cur) + rd; ptr.nxt_e = img_e(b_nxt) + rd;
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) { ptr.cur[j] = img_e(b_cur) + rd1[j]; ptr.nxt[j] = img_e(b_nxt) + rd1[j]; }
        }
    };

    // row 0 is staged here, every other row while its predecessor runs
    {
        uint32_t c22;
        asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i)
            put0(i, mx_expand(word(i, 0), c22));
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i) { st.wn[i] = word(i, 1); st.wo[i] = 0u; }
    }
    __syncthreads();
    int b_cur = 0, b_nxt = 1;
    point(b_cur, b_nxt);
    mx_v8i an[5];
    uint4 fb[MX_P];
    static_assert(MX_P == 5, "the first operands of a row are those of its first tile's five products");
#pragma unroll
    for (int k = 0; k < 5; ++k) {                    // tile 0: left operands and the first MX_P right operands
        if constexpr (WORDS == 2) { an[k] = mx_frag(ptr.cur_e[G::RPX + k]); fb[k] = ptr.cur_e[32 * (TG - T) + k]; }
        else { an[k] = mx_frag(ptr.cur[k][G::RPX]); fb[k] = ptr.cur[k][32 * (TG - T)]; }
    }

    // row r stages row r+1 (entering) with row r-4 (leaving) and fetches the words of rows r+2 and r-3
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
        rg.off_new = (uint32_t)min(r + 2, last) * row_bytes;
        rg.off_old = (uint32_t)max(r - 3, 0) * row_bytes;
        mx_row<T, WORDS, false, BOTH>(acc, ptr, img_e(b_nxt), img_l(b_nxt), st, an, fb, rg, lane, tid, pass_flags);
        b_cur = b_nxt;
        b_nxt = b_nxt == MX_NBUF - 1 ? 0 : b_nxt + 1;
        point(b_cur, b_nxt);
    }
    mx_first_output<T, BOTH>(acc, st);
    // Steady state, three rows per trip: census row r reads staged image r % 3 and writes image (r + 1) % 3, and the keys of
    // the output row that ends with it go to key row (r + 1) % 3, so with the phase of a row a compile-time constant every
    // LDS address of the row is a constant offset from a register that never changes.
    static_assert(MX_NBUF == 3 && MX_KROWS == 3, "the steady state is unrolled over three rows");
    auto steady = [&](auto b_c, int r) __attribute__((always_inline)) {
        constexpr int B = decltype(b_c)::value, N = (B + 1) % MX_NBUF;
        rg.off_new = (uint32_t)min(r + 2, last) * row_bytes;
        rg.off_old = (uint32_t)(r - 3) * row_bytes;
        point(B, N);
        mx_row<T, WORDS, true, BOTH, N>(acc, ptr, img_e(N), img_l(N), st, an, fb, rg, lane, tid, pass_flags);
    };
    if constexpr (T <= 5) {
        int r = 5;
#pragma unroll 1
        for (; r + 3 <= n_rows; r += 3) {
            steady(std::integral_constant<int, 2>{}, r);
            steady(std::integral_constant<int, 0>{}, r + 1);
            steady(std::integral_constant<int, 1>{}, r + 2);
        }
        if (r < n_rows) {
            steady(std::integral_constant<int, 2>{}, r);
            if (r + 1 < n_rows) steady(std::integral_constant<int, 0>{}, r + 1);
        }
        if (BOTH) {                                  // the key row the last output row's keys are in
            st.krow = (n_rows - 5) % MX_KROWS;
            st.kcur = st.key0 + st.krow * G::NKEYS;
        }
    } else {                                         // (nine tiles: the constant addresses of three unrolled rows do not fit in the registers)
#pragma unroll 1
        for (int r = 5; r < n_rows; ++r) {
            rg.off_new = (uint32_t)min(r + 2, last) * row_bytes;
            rg.off_old = (uint32_t)(r - 3) * row_bytes;
            mx_row<T, WORDS, true, BOTH>(acc, ptr, img_e(b_nxt), img_l(b_nxt), st, an, fb, rg, lane, tid, pass_flags);
            b_cur = b_nxt;
            b_nxt = b_nxt == MX_NBUF - 1 ? 0 : b_nxt + 1;
            point(b_cur, b_nxt);
        }
    }
    mx_last_output<T, BOTH>(acc, st, lane, tid, pass_flags);
}

// A wavefront behind the frame's last x-block has no pairs of its own.  It takes its share of the workgroup's staging and of
// writing out the key rows and meets the others once per row, in the same order as mx_strip: a barrier behind the first staged
// row, one per census row (behind its staging for the next row, which is complete by then: __syncthreads waits for it), and with
// BOTH one more in front of the last key row.  No matrix instruction, no winner selection.
template <int T, int WORDS, bool BOTH>
__device__ __forceinline__ void mx_strip_idle(uint4 *lds, const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cr, int cols, int d_base,
                                              int xl0_wg, int y_lo, int n_rows, int tid, float *__restrict__ part_row, long part_stride) {
    using G = MxGeom<T, WORDS>;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row_words = (long)cols * WORDS;
    const uint32_t *const cl0 = cl + (long)(y_lo - 2) * row_words, *const cr0 = cr + (long)(y_lo - 2) * row_words;
    const uint32_t row_bytes = 4u * (uint32_t)row_words;
    const __amdgpu_buffer_rsrc_t img_left = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cl0), 0, (int)(row_bytes * (uint32_t)n_rows), 0x00020000);
    const __amdgpu_buffer_rsrc_t img_right = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cr0), 0, (int)(row_bytes * (uint32_t)n_rows), 0x00020000);
    auto right_off = [&](int q) -> uint32_t { return 4u * (uint32_t)min(max((long)(xl0_wg - 32 * (T - 1) - d_base - 2) * WORDS + q, 0L), row_words - 1); };
    auto left_off = [&](int q) -> uint32_t { return 4u * (uint32_t)min(max((long)(xl0_wg - 2) * WORDS + q, 0L), row_words - 1); };
    auto slot_of = [&](int q, bool left) -> int { return WORDS == 2 ? (q & 1) * G::PLANE + (left ? G::RPX : 0) + (q >> 1) : (left ? G::RPX : 0) + q; };
    // this thread's word of every staging round (MxGeom): where it comes from, where it goes, whether there is one
    uint32_t voff[G::ROUNDS], wn[G::ROUNDS], wo[G::ROUNDS];
    int slot[G::ROUNDS];
    bool has[G::ROUNDS], left[G::ROUNDS];
#pragma unroll
    for (int i = 0; i < G::ROUNDS; ++i) {
        int q;
        if (i < G::FR) { q = i * G::NT + tid; left[i] = false; has[i] = true; }
        else if (i < G::FR + G::FL) { q = (i - G::FR) * G::NT + tid; left[i] = true; has[i] = true; }
        else {
            const int v = (i - G::FR - G::FL) * MX_W + wave;
            left[i] = v < G::ULT;
            q = left[i] ? G::FL * G::NT + 64 * v + lane : G::FR * G::NT + 64 * (v - G::ULT) + lane;
            has[i] = v < G::ULT + G::URT && q < (left[i] ? G::LW : G::RW);
            q = min(q, (left[i] ? G::LW : G::RW) - 1);
        }
        voff[i] = left[i] ? left_off(q) : right_off(q);
        slot[i] = slot_of(q, left[i]);
    }
    auto word = [&](int i, int r) -> uint32_t { return left[i] ? mx_word(img_left, (uint32_t)r * row_bytes, voff[i]) : mx_word(img_right, (uint32_t)r * row_bytes, voff[i]); };
    float *const keys = reinterpret_cast<float *>(lds + MX_NBUF * 2 * G::SLOTS);
    if (BOTH)
        for (int i = tid; i < MX_KROWS * G::NKEYS; i += G::NT) keys[i] = MX_NO_KEY;
    auto flush = [&](int kr) {                       // this thread's keys of key row kr (mx_keys_flush_read / _write)
#pragma unroll
        for (int i = 0; i < G::KEY_ROUNDS; ++i)
            if (tid + i * G::NT < G::NKEYS) {
                part_row[tid + i * G::NT] = keys[kr * G::NKEYS + tid + i * G::NT];
                keys[kr * G::NKEYS + tid + i * G::NT] = MX_NO_KEY;
            }
        part_row += part_stride;
    };
    uint32_t c22;
    asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
#pragma unroll
    for (int i = 0; i < G::ROUNDS; ++i) {
        if (has[i]) lds[slot[i]] = mx_expand(word(i, 0), c22);
        wn[i] = word(i, 1);
        wo[i] = 0u;
    }
    __syncthreads();
    const int last = n_rows - 1;
    int b_nxt = 1;
#pragma unroll 1
    for (int r = 0; r < n_rows; ++r) {
        uint4 *const img_e = lds + (2 * b_nxt) * G::SLOTS, *const img_l = lds + (2 * b_nxt + 1) * G::SLOTS;
        const int rn = min(r + 2, last), ro = max(r - 3, 0);
#pragma unroll
        for (int i = 0; i < G::ROUNDS; ++i) {
            if (has[i]) {
                img_e[slot[i]] = mx_expand(wn[i], c22);
                img_l[slot[i]] = mx_expand(left[i] ? ~wo[i] : wo[i], c22);       // the leaving row's left words are complemented
            }
            wn[i] = word(i, rn);
            wo[i] = word(i, ro);
        }
        __syncthreads();                             // the meeting point of row r
        if (BOTH && r >= 5) flush(b_nxt);            // the keys of the output row that ended with census row r - 1 (key row (r + 1) % 3)
        b_nxt = b_nxt == MX_NBUF - 1 ? 0 : b_nxt + 1;
    }
    if (BOTH) {
        __syncthreads();                             // the last output row's keys are complete
        flush((n_rows - 5) % MX_KROWS);
    }
}

// grid: (groups of MX_W x-blocks, strips, frames), MX_W wavefronts per block, one per x-block.  x-block b owns output
// columns [xl0, xl0 + 32), xl0 = min(2 + 32 b, cols - 34) (the last block is shifted left to end at column cols-3);
// a wavefront behind the last x-block only helps with the staging (mx_strip_idle).
// Searches every pair (left column x, right column x - d), 0 <= d < max_disp, whose right column lies in the x-block's T tiles,
// xl0 - 32 (T - 1) - d_base .. xl0 + 31 - d_base; a wider range takes several launches whose tiles lie side by side
// (d_base = 0, 160, ...; pass_flags: mx_reduce_tail).  Requires cols >= 36, rows >= 5.
// BOTH: also stores, for every output row and workgroup, the right view's best key of each of the 32 (MX_W + T - 1) right
// columns the workgroup's tiles cover, over the workgroup's left columns: partial[frame][row][workgroup][NKEYS]
// (svx_mx_right_disparity merges the workgroups that reach a right pixel).
template <int T, int WORDS, bool BOTH>
__global__ __launch_bounds__(64 * MX_W, T <= 5 ? 2 : 1) void dense_wta_mx_kernel(const uint32_t *__restrict__ cl_all, const uint32_t *__restrict__ cr_all,
                                                          int rows, int cols, int max_disp, int d_base, int pass_flags, int strip_rows, int n_xblocks,
                                                          float *__restrict__ disp_all, int write_border, float *__restrict__ partial) {
    using G = MxGeom<T, WORDS>;
    __shared__ uint4 lds[MX_NBUF * 2 * G::SLOTS + (BOTH ? MX_KROWS * G::NKEYS / 4 : 0)];    // [image][entering row, leaving row][see MxGeom], then the right view's key rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = lane & 31, h = lane >> 5;
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    unsigned wi = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((total & 7u) == 0u) wi = (wi & 7u) * (total >> 3) + (wi >> 3);      // contiguous run of work items per XCD
    const unsigned gx = wi % gridDim.x, by = (wi / gridDim.x) % gridDim.y, bz = wi / (gridDim.x * gridDim.y);
    const long npx = (long)rows * cols;
    const uint32_t *cl = cl_all + bz * npx * WORDS;
    const uint32_t *cr = cr_all + bz * npx * WORDS;
    float *disp = disp_all + bz * npx;
    const int bx_wave = (int)gx * MX_W + __builtin_amdgcn_readfirstlane(tid >> 6), bx = min(bx_wave, n_xblocks - 1);
    // left columns from own_from on are this wavefront's to write: the last x-block is shifted over its neighbour, and a
    // wavefront behind the last x-block writes nothing to the left map
    const int own_from = bx_wave > n_xblocks - 1 ? 0x7fffffff : 2 + bx * MX_M;
    const int xl0 = min(2 + bx * MX_M, cols - 2 - MX_M), xl0_wg = min(2 + (int)gx * MX_W * MX_M, cols - 2 - MX_M);
    const int y_lo = 2 + (int)by * strip_rows, y_hi = min(y_lo + strip_rows, rows - 2);
    const int n_rows = y_hi - y_lo + 4;              // census rows y_lo-2 .. y_hi+1
    const long part_stride = (long)gridDim.x * G::NKEYS;
    float *part_row0 = BOTH ? partial + ((long)bz * rows + y_lo) * part_stride + (long)gx * G::NKEYS : nullptr;
    // x-blocks at the frame's left border (and, in a later pass, left of its first disparity): tile u holds right columns
    // xl0 - 32 (T - 1) - d_base + 32 u .. + 31, and the first tiles, whose columns all lie left of column 2, hold no admissible pair.
    // With five tiles up to two of them are not computed (three is the fewest the row's timetable takes).
    // Likewise the first tiles of a search's last pass when every disparity of theirs is max_disp or more (tile u: at least d_base + 97 - 32 u).
    int dead = 0;
    if constexpr (T == 5) {
        dead = min(max((32 * (T - 1) - 29 + d_base - xl0 + 31) / 32, 0), 2);
        const int beyond = d_base + 97 - max_disp;
        if (beyond >= 0) dead = min(max(dead, beyond / 32 + 1), 2);
    }
    if (bx_wave > n_xblocks - 1) mx_strip_idle<T, WORDS, BOTH>(lds, cl, cr, cols, d_base, xl0_wg, y_lo, n_rows, tid, part_row0, part_stride);
    else if (T == 5 && dead == 2) {
        if constexpr (T == 5) mx_strip<3, WORDS, BOTH, 5>(lds, cl, cr, cols, max_disp, d_base, pass_flags, xl0, xl0_wg, own_from, y_lo, n_rows, tid, disp, part_row0, part_stride);
    } else if (T == 5 && dead == 1) {
        if constexpr (T == 5) mx_strip<4, WORDS, BOTH, 5>(lds, cl, cr, cols, max_disp, d_base, pass_flags, xl0, xl0_wg, own_from, y_lo, n_rows, tid, disp, part_row0, part_stride);
    } else mx_strip<T, WORDS, BOTH>(lds, cl, cr, cols, max_disp, d_base, pass_flags, xl0, xl0_wg, own_from, y_lo, n_rows, tid, disp, part_row0, part_stride);

    if (write_border) {
        // rows outside [2, rows-2) over this block's columns, and the two border columns on either side
        const int ys = by == 0 ? 0 : y_lo, y_end = by == gridDim.y - 1 ? rows : y_hi;
        for (int y = ys + h; y < y_end; y += 2)
            if (y < 2 || y >= rows - 2) disp[(long)y * cols + xl0 + n] = 0.0f;
        if (bx == 0)
            for (int y = ys + (lane >> 1); y < y_end; y += 32) disp[(long)y * cols + (lane & 1)] = 0.0f;
        if (bx == n_xblocks - 1)
            for (int y = ys + (lane >> 1); y < y_end; y += 32) disp[(long)y * cols + cols - 2 + (lane & 1)] = 0.0f;
    }
}

// Right view from the partial keys: right pixel xr can match left columns xr .. xr + max_disp - 1.  Within one pass
// (disparities d_base .. d_base + 32 (T - 1) - 1) they lie in the workgroups (groups of MX_W x-blocks) whose first left
// column X satisfies xs - 32 MX_W < X <= xs + 32 (T - 1) with xs = xr + d_base; group g keeps xr in slot xs - (X - 32 (T - 1))
// of its key row.  Returns the best key of that pass.
template <int T>
__device__ __forceinline__ float svx_mx_right_key(const float *__restrict__ prow, int xs, int cols, int n_groups) {
    constexpr int GW = 32 * MX_W, NK = MxGeom<T>::NKEYS, K = (NK + GW - 1) / GW;
    float best = MX_NO_KEY;
    const int a = xs >= 2 ? (xs - 2) / GW : -1;
    // (the last group is shifted left to end at the frame's last column, so it reaches further left than a regular group in its
    // place would: where it can hold xs, the general form below looks at it as well)
    const int last_from = min(2 + (n_groups - 1) * GW, cols - 2 - MX_M) - 32 * (T - 1);
    if (a >= 0 && a + K - 1 < n_groups - 1 && xs < last_from) {
        // all regular groups (X = 2 + GW g): group a + k holds xs in slot (xs - 2) % GW + 32 (T - 1) - GW k while that is not negative
        const int slot = (xs - 2) - a * GW + 32 * (T - 1);
        const float *p = prow + (long)a * NK + slot;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (slot - GW * k >= 0) best = fmaxf(best, p[k * (NK - GW)]);
        return best;
    }
#pragma unroll
    for (int k = 0; k < K + 1; ++k) {
        const int g = min(max(a, 0) + k, n_groups - 1);
        const int X = min(2 + g * GW, cols - 2 - MX_M);
        const int slot = xs - (X - 32 * (T - 1));
        const bool ok = slot >= 0 && slot < NK;
        const float v = prow[(long)g * NK + min(max(slot, 0), NK - 1)];
        best = fmaxf(best, ok ? v : MX_NO_KEY);
    }
    return best;
}

// The right disparity of (xr, row) over all passes: pass p's keys lie pass_stride floats behind pass p - 1's.
template <int T>
__device__ __forceinline__ float svx_mx_right_disparity(const float *__restrict__ prow, long pass_stride, int n_pass, int xr, int cols, int n_groups) {
    float best = svx_mx_right_key<T>(prow, xr, cols, n_groups);
    for (int p = 1; p < n_pass; ++p) best = fmaxf(best, svx_mx_right_key<T>(prow + p * pass_stride, xr + p * mx_pass_step(T), cols, n_groups));
    // right columns 2..4 are admitted by the right view only: each of their pairs started 4096 lower (mx_strip), so the
    // best of them is lifted here (a column without any pair stays far below the threshold of mx_decode)
    return mx_decode(xr < 5 ? best + 4096.0f : best);
}

template <int T>
__global__ __launch_bounds__(256) void mx_right_view_merge_kernel(const float *__restrict__ partial, long pass_stride, int n_pass, int rows, int cols,
                                                                  int n_groups, float *__restrict__ disp_right_all) {
    const int xr = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xr >= cols) return;
    float *out = disp_right_all + ((long)blockIdx.z * rows + y) * cols;
    if (xr < 2 || xr >= cols - 2 || y < 2 || y >= rows - 2) { out[xr] = 0.0f; return; }
    const float *prow = partial + ((long)blockIdx.z * rows + y) * ((long)n_groups * MxGeom<T>::NKEYS);
    out[xr] = svx_mx_right_disparity<T>(prow, pass_stride, n_pass, xr, cols, n_groups);
}

// Left-right check straight from the partial keys (the rule of svx_lr_check with the right disparity merged on demand).
// One block per row and frame; a thread checks the pixels tid, tid + 256, ... four at a time (their loads are in flight together).
// KEYS: the left map holds the winners' keys (svx_launch_mx_sweep, keep_keys); a rejected pixel gets SVX_MX_NO_KEY.
template <int T, bool KEYS = false>
__global__ __launch_bounds__(256) void mx_lr_check_kernel(float *__restrict__ dl_all, const float *__restrict__ partial, long pass_stride, int n_pass,
                                                          int rows, int cols, int n_groups, float threshold, float invalid) {
    const int y = blockIdx.x + 2;
    float *dl = dl_all + ((long)blockIdx.y * rows + y) * cols;
    const float *prow = partial + ((long)blockIdx.y * rows + y) * ((long)n_groups * MxGeom<T>::NKEYS);
    for (int x0 = 2 + threadIdx.x; x0 < cols - 2; x0 += 4 * 256) {
        float d1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d1[k] = x0 + 256 * k < cols - 2 ? dl[x0 + 256 * k] : 0.0f;
            if (KEYS) d1[k] = mx_decode(d1[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + 256 * k;
            if (x >= cols - 2) break;
            bool bad = !(d1[k] >= 0.0f) || !(d1[k] < (float)cols);                // no disparity (negative, NaN, beyond the row): reads nothing
            const int xr = bad ? 0 : x - (int)d1[k];
            bad = bad || xr < 2;
            if (!bad) bad = fabsf(svx_mx_right_disparity<T>(prow, pass_stride, n_pass, xr, cols, n_groups) - d1[k]) > threshold;
            if (bad) dl[x] = KEYS ? MX_NO_KEY : invalid;
        }
    }
}

bool svx_mx_sweep_eligible(int rows, int cols, int max_disp) { return cols >= MX_M + 4 && rows >= 5 && max_disp >= 1 && max_disp <= MX_MAX_DISP; }

// Tiles per wavefront and launches per search: up to 64 / 128 disparities three / five tiles in one pass; wider ranges in
// passes of T = five or nine tiles (mx_plan chooses).  The passes' tiles lie side by side: pass p holds the right columns
// xl0 - 32 (T - 1) - 32 T p .. xl0 + 31 - 32 T p, so only the first pass (d < 0) and the last (d >= max_disp) compute pairs
// outside the band, where passes of 32 (T - 1) disparities each would compute a half-empty first and last tile per pass: left
// column xl0 + m has reached disparity 32 (T - 1) + 32 T p + m after pass p, which makes 1 + ceil((max_disp - 32 (T - 1)) / 32 T)
// passes (five tiles, 256: 9 tiles instead of 10; 512: 18 instead of 20, three of the last pass's five being skipped or dead;
// 1024: 34 instead of 40).
struct MxPlan { int tiles, passes; };
static MxPlan mx_plan(const svx_ctx *ctx, int max_disp, int words = 2) {
    if (max_disp <= 64) return {3, 1};
    if (max_disp <= 128) return {5, 1};
    // More than 128 disparities: passes of five tiles (two wavefronts per SIMD; the last pass's leading tiles whose disparities all
    // lie beyond the range are not computed, at most two of five) or, on two-word descriptors, passes of nine (one wavefront per
    // SIMD, every tile computed): fewer launches, the left operands staged less often, fewer partial keys for the check to read.
    // SVX_OPT_MX_WIDE_TILES: 1 = nine tiles wherever they apply, 0 = never, -1 (default) = where one pass of nine computes no
    // more tiles than the passes of five (225..256 disparities: 3-4 % per step on four frame sizes).  Several passes of nine
    // measured within 1 % of the passes of five where the tile counts are equal (449..544, 737..832) and slower elsewhere.
    const int passes5 = 1 + (max_disp - 128 + mx_pass_step(5) - 1) / mx_pass_step(5);
    int tiles5 = 0;
    for (int p = 0; p < passes5; ++p) {
        const int beyond = p * mx_pass_step(5) + 97 - max_disp;
        tiles5 += 5 - (beyond >= 0 ? (beyond / 32 + 1 < 2 ? beyond / 32 + 1 : 2) : 0);
    }
    const int passes9 = max_disp <= 256 ? 1 : 1 + (max_disp - 256 + mx_pass_step(9) - 1) / mx_pass_step(9);
    const int wide = svx_opt(ctx, SVX_OPT_MX_WIDE_TILES);
    if (words == 2 && (wide == 1 || (wide < 0 && passes9 == 1 && 9 <= tiles5))) return {9, passes9};
    return {5, passes5};
}
static int mx_xblocks(int cols) { return (cols - 4 + MX_M - 1) / MX_M; }
static int mx_groups(int cols) { return (mx_xblocks(cols) + MX_W - 1) / MX_W; }                 // workgroups along x
static int mx_keys_per_group(int tiles) { return 32 * (MX_W + tiles - 1); }

size_t svx_mx_partial_words(const svx_ctx *ctx, int rows, int cols, int n_frames, int words, int max_disp) {
    const MxPlan plan = mx_plan(ctx, max_disp, words);                // the plan the launch makes for the same arguments
    return (size_t)plan.passes * n_frames * rows * mx_groups(cols) * mx_keys_per_group(plan.tiles);
}

// Left-view search on two-word descriptors; see svx_dense.hip for the meaning of the arguments.
// partial != nullptr: also the right view's partial keys (svx_mx_partial_words floats).
int svx_launch_mx_sweep(svx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, int rows, int cols, int n_frames, int words, int max_disp,
                        float *disparity, int write_border, float *partial, bool keep_keys) {
    const MxPlan plan = mx_plan(ctx, max_disp, words);
    const int T = plan.tiles;
    // Strips: every strip re-reads 4 halo rows and starts with 5 rows that fill the window, but the grid should fill the
    // chip a whole number of times (8 wavefronts or, with nine tiles, 4 fit a CU): take the
    // number of strips with the smallest (rounds of wavefronts) x (rows a wavefront walks).
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus <= 0) cus = 256;
    const int n_groups = mx_groups(cols);
    const long slots = (long)cus * ((T <= 5 ? 8 : 4) / MX_W > 0 ? (T <= 5 ? 8 : 4) / MX_W : 1), per_strip = (long)n_groups * n_frames;
    const int out_rows = rows - 4;
    int n_strips = 1;
    long best_cost = -1;
    for (int k = 1; k <= 16 && k <= out_rows; ++k) {
        const long rounds = (per_strip * k + slots - 1) / slots;
        const long cost = rounds * ((out_rows + k - 1) / k + 4 + 2);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; n_strips = k; }
    }
    // SVX_OPT_MX_STRIP_ROWS > 0 forces the output rows per strip.
    const int forced_rows = svx_opt(ctx, SVX_OPT_MX_STRIP_ROWS);
    const int strip_rows = forced_rows > 0 ? forced_rows : (out_rows + n_strips - 1) / n_strips;
    n_strips = (out_rows + strip_rows - 1) / strip_rows;
    const dim3 grid(n_groups, n_strips, n_frames), block(64 * MX_W);
    const int nb = mx_xblocks(cols);
    const size_t pass_words = (size_t)n_frames * rows * n_groups * mx_keys_per_group(T);
    for (int p = 0; p < plan.passes; ++p) {
        const int d_base = p * mx_pass_step(T), flags = (p > 0 ? 1 : 0) | (p + 1 < plan.passes || keep_keys ? 2 : 0);
        const int wb = p + 1 == plan.passes ? write_border : 0;
        float *part = partial ? partial + p * pass_words : nullptr;
#define MX_LAUNCH(TT, WW, BB, PART) dense_wta_mx_kernel<TT, WW, BB><<<grid, block, 0, ctx->stream>>>(cl, cr, rows, cols, max_disp, d_base, flags, strip_rows, nb, disparity, wb, PART)
        if (words == 1) {
            if (T == 3 && !partial) MX_LAUNCH(3, 1, false, nullptr);
            else if (T == 3) MX_LAUNCH(3, 1, true, part);
            else if (!partial) MX_LAUNCH(5, 1, false, nullptr);
            else MX_LAUNCH(5, 1, true, part);
        } else if (T == 9 && !partial) MX_LAUNCH(9, 2, false, nullptr);
        else if (T == 9) MX_LAUNCH(9, 2, true, part);
        else if (T == 3 && !partial) MX_LAUNCH(3, 2, false, nullptr);
        else if (T == 3) MX_LAUNCH(3, 2, true, part);
        else if (!partial) MX_LAUNCH(5, 2, false, nullptr);
        else MX_LAUNCH(5, 2, true, part);
#undef MX_LAUNCH
        SVX_LAUNCH_CHECK(ctx, "dense_wta_mx_kernel");
    }
    return SVX_OK;
}

// After svx_launch_mx_sweep with partial keys: the right disparity map (disp_right != nullptr) or the left-right
// check of disp_left in place.
int svx_launch_mx_right(svx_ctx *ctx, const float *partial, int rows, int cols, int n_frames, int words, int max_disp,
                        float *disp_left, float *disp_right, float lr_threshold, bool keys) {
    const dim3 grid(svx_cdiv(cols, 256), rows, n_frames), check_grid(rows - 4, n_frames);
    const int nb = mx_groups(cols);
    const MxPlan plan = mx_plan(ctx, max_disp, words);
    const long ps = (long)n_frames * rows * nb * mx_keys_per_group(plan.tiles);
    const int np = plan.passes;
    if (plan.tiles == 9) {
        if (disp_right) mx_right_view_merge_kernel<9><<<grid, 256, 0, ctx->stream>>>(partial, ps, np, rows, cols, nb, disp_right);
        else if (keys) mx_lr_check_kernel<9, true><<<check_grid, 256, 0, ctx->stream>>>(disp_left, partial, ps, np, rows, cols, nb, lr_threshold, -1.0f);
        else mx_lr_check_kernel<9><<<check_grid, 256, 0, ctx->stream>>>(disp_left, partial, ps, np, rows, cols, nb, lr_threshold, -1.0f);
    } else if (plan.tiles == 3) {
        if (disp_right) mx_right_view_merge_kernel<3><<<grid, 256, 0, ctx->stream>>>(partial, ps, np, rows, cols, nb, disp_right);
        else if (keys) mx_lr_check_kernel<3, true><<<check_grid, 256, 0, ctx->stream>>>(disp_left, partial, ps, np, rows, cols, nb, lr_threshold, -1.0f);
        else mx_lr_check_kernel<3><<<check_grid, 256, 0, ctx->stream>>>(disp_left, partial, ps, np, rows, cols, nb, lr_threshold, -1.0f);
    } else {
        if (disp_right) mx_right_view_merge_kernel<5><<<grid, 256, 0, ctx->stream>>>(partial, ps, np, rows, cols, nb, disp_right);
        else if (keys) mx_lr_check_kernel<5, true><<<check_grid, 256, 0, ctx->stream>>>(disp_left, partial, ps, np, rows, cols, nb, lr_threshold, -1.0f);
        else mx_lr_check_kernel<5><<<check_grid, 256, 0, ctx->stream>>>(disp_left, partial, ps, np, rows, cols, nb, lr_threshold, -1.0f);
    }
    SVX_LAUNCH_CHECK(ctx, "mx right view kernel");
    return SVX_OK;
}

// The matrix-core form of svx_dense_wta_both on two-word descriptors (max_disp <= 1024, cols >= 36, rows >= 5), run
// whatever the library's default choice of kernel is: left map, and with disp_right != NULL the right map from the
// same accumulators.  Same results as svx_dense_wta / svx_dense_wta_both.
SVX_API int svx_dense_wta_matrix(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                                 int max_disp, float *disp_left, float *disp_right) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, census_left && census_right && disp_left, SVX_ERR_INVALID_ARG, "svx_dense_wta_matrix: null pointer");
    SVX_REQUIRE(ctx, n_frames >= 1 && n_frames <= 65535 && rows <= 65535, SVX_ERR_INVALID_ARG, "svx_dense_wta_matrix: bad frame count or size");
    SVX_REQUIRE(ctx, svx_mx_sweep_eligible(rows, cols, max_disp), SVX_ERR_UNSUPPORTED,
                "svx_dense_wta_matrix: needs cols >= 36, rows >= 5, 1 <= max_disp <= 1024");
    if (!disp_right) return svx_launch_mx_sweep(ctx, census_left, census_right, rows, cols, n_frames, 2, max_disp, disp_left, 1, nullptr);
    const size_t per_frame = svx_mx_partial_words(ctx, rows, cols, 1, 2, max_disp);
    SVX_REQUIRE(ctx, per_frame * n_frames * sizeof(float) <= ((size_t)4 << 30), SVX_ERR_UNSUPPORTED, "svx_dense_wta_matrix: batch too large for one call");
    void *ws;
    rc = svx_ws_reserve(ctx, per_frame * n_frames * sizeof(float), &ws);
    if (rc) return rc;
    rc = svx_launch_mx_sweep(ctx, census_left, census_right, rows, cols, n_frames, 2, max_disp, disp_left, 1, (float *)ws);
    if (rc) return rc;
    return svx_launch_mx_right(ctx, (const float *)ws, rows, cols, n_frames, 2, max_disp, disp_left, disp_right, 0.0f);
}
