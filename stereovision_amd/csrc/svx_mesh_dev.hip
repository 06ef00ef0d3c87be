// Support-point mesh on the device: the Delaunay triangulation of svx_mesh.hip (the stand-in for the Fade_2D object of
// highPerfStereo/highPerfStereoLib.cpp:434-445, :457) computed point by point instead of by insertion.
//
// With the cocircular cases resolved by the symbolic perturbation of svx_mesh.hip (in_circle_tie: the point with the
// smallest index is lifted most) the triangulation is a function of the point list alone, so every point can find the
// triangles around it on its own, from exact integer predicates:
//   * its nearest neighbour a0 is joined to it by a Delaunay edge (the disc on that diameter holds no other point);
//   * given a Delaunay edge (p, a), the triangle to its left is (p, a, b) with b the point left of p -> a that no other
//     such point c beats (c beats b when it lies inside the circle through p, a, b): one pass over the candidates;
//   * walking a -> b until a0 comes back gives all triangles around p; when a side has no point (p is on the hull) the
//     walk is finished clockwise from a0.
// Candidates: the points are binned into square cells of about one point each (4 to 64 pixels a side); a point first
// looks at the 7 x 7 cells around its own, which hold every point within three cells of it (more towards the sides of the
// points' bounding box, outside which nothing lies).  A triangle found among them is the true one when its circumcircle
// lies within that range (p is on the circle, so twice the radius bounds how far the circle reaches from p), and a point
// whose triangles all pass owns its true ring.  Points that fail (next to empty areas, on the hull) look again at 17 x 17
// cells, and where that does not settle a triangle either, at the points of ALL cells the triangle's circle meets.  A
// triangle is reported by its vertex of smallest index, rotated to lead with it, and the list is ordered by (v0, v1, v2):
// the form svx_delaunay returns.
#include "svx_common.hpp"
#include "svx_compact.hpp"
#include <algorithm>
#include <vector>

namespace {

constexpr int MAX_RING = 256;             // triangles around one point (a loop bound; fans along the frame's border lines reach dozens); a longer ring sends the call back to the host triangulation
constexpr int MAX_LEAD = 32;              // triangles one point leads (it has the smallest index of the three)

struct DPt { int x, y, id; };

__device__ __forceinline__ DPt unpack(int2 c) { DPt p; p.x = c.x & 0xffff; p.y = (int)((unsigned)c.x >> 16); p.id = c.y; return p; }

// Coordinates lie in [0, 8192): differences below 2^13, cross products below 2^27 (32-bit), the circle determinant below 2^57.
// (the factors fit 24 bits: the full-rate multiplier)
__device__ __forceinline__ int d_orient(const DPt &a, const DPt &b, const DPt &c) { return __mul24(b.x - a.x, c.y - a.y) - __mul24(b.y - a.y, c.x - a.x); }

// > 0 when d lies inside the circle through a, b, c (counter-clockwise), cocircular cases by the rule of svx_mesh.hip
// (in_circle_small, in_circle_tie there): never 0 for four distinct positions.
__device__ __forceinline__ int d_in_circle(const DPt &a, const DPt &b, const DPt &c, const DPt &d) {
    const int ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    const long long a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    const long long det = a2 * (long long)(bx * cy - by * cx) - b2 * (long long)(ax * cy - ay * cx) + c2 * (long long)(ax * by - ay * bx);
    if (det != 0) return det > 0 ? 1 : -1;
    const int m = min(min(a.id, b.id), min(c.id, d.id));
    int o;
    if (m == a.id) o = d_orient(b, c, d);
    else if (m == b.id) o = -d_orient(a, c, d);
    else if (m == c.id) o = d_orient(a, b, d);
    else o = -d_orient(a, b, c);
    return o > 0 ? 1 : (o < 0 ? -1 : 0);
}

// The circle through p, a, b as centre (relative to p) and squared radius in double, a little enlarged: a candidate
// outside it cannot beat b, so the exact test is only evaluated for the few candidates inside (points ON the circle, the
// cocircular ties, stay inside the enlarged one: double arithmetic is exact here up to rounding of the quotient).
struct Circle {
    double ox, oy, r2;
    __device__ void set(const DPt &p, const DPt &a, const DPt &b) {
        const double ax = a.x - p.x, ay = a.y - p.y, bx = b.x - p.x, by = b.y - p.y;
        const double la = ax * ax + ay * ay, lb = bx * bx + by * by, inv = 0.5 / (ax * by - ay * bx);
        ox = (by * la - ay * lb) * inv; oy = (ax * lb - bx * la) * inv;
        r2 = (ox * ox + oy * oy) * (1.0 + 1e-9);
    }
    __device__ bool outside(const DPt &p, int cxy) const {
        const double dx = (double)((cxy & 0xffff) - p.x) - ox, dy = (double)((int)((unsigned)cxy >> 16) - p.y) - oy;
        return dx * dx + dy * dy > r2;
    }
};

// The smallest value of x over the wavefront, the same in every lane: inside the rows of 16 lanes by DPP (lane ^ 1, lane ^ 2,
// mirrored halves, mirrored row), the four rows through scalar registers.
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
    x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));     // quad_perm:[1,0,3,2]
    x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));     // quad_perm:[2,3,0,1]
    x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));    // row_half_mirror
    x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));    // row_mirror
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)x, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)x, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)x, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)x, 48);
    return min(min(r0, r1), min(r2, r3));
}
// float -> unsigned with the same order
__device__ __forceinline__ unsigned ordered_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int2 lane_value(int2 v, int src) {      // src: the same in every lane
    return make_int2(__builtin_amdgcn_readlane(v.x, src), __builtin_amdgcn_readlane(v.y, src));
}

// ---- binning ----
// (a point outside the image raises counters[2] and is left out; the call then fails, but nothing is indexed with it)
// counters[4..7]: the bounding box of the points (min x, min y, max x, max y; started at the opposite extremes by mesh_reset_kernel),
// one set of atomics per block of 1024 threads (they all hit one cache line)
__global__ __launch_bounds__(1024) void cell_count_kernel(const int32_t *__restrict__ xy, int n, int rows, int cols, int shift, int gw,
                                                          unsigned *__restrict__ cell_count, int *__restrict__ counters) {
    __shared__ int box[16][4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    if (i < n) {
        const int x = xy[2 * i], y = xy[2 * i + 1];
        if (x < 0 || x >= cols || y < 0 || y >= rows) atomicOr(&counters[2], 1);
        else {
            atomicAdd(&cell_count[(y >> shift) * gw + (x >> shift)], 1u);
            x0 = x1 = x; y0 = y1 = y;
        }
    }
    x0 = (int)wave_min_u32((unsigned)x0); y0 = (int)wave_min_u32((unsigned)y0);
    x1 = (int)~wave_min_u32(~(unsigned)(x1 + 1)) - 1; y1 = (int)~wave_min_u32(~(unsigned)(y1 + 1)) - 1;      // (maximum of values >= -1: the minimum of the complements of value + 1)
    if ((threadIdx.x & 63) == 0) { int *b = box[threadIdx.x >> 6]; b[0] = x0; b[1] = y0; b[2] = x1; b[3] = y1; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x, waves = (blockDim.x + 63) >> 6;
        int v = box[0][k];
        for (int w = 1; w < waves; ++w) v = k < 2 ? min(v, box[w][k]) : max(v, box[w][k]);
        if (k < 2) { if (v != 0x7fffffff) atomicMin(&counters[4 + k], v); }
        else if (v >= 0) atomicMax(&counters[4 + k], v);
    }
}

// What a triangulation starts from, in one launch: cell counts and fill counters, lead counts and counters zero, the bounding box
// inside out, every slot of the binned list unfilled (index -1).
__global__ void mesh_reset_kernel(unsigned *__restrict__ cells, int n_cell_words, int *__restrict__ lead_count, int n_lead, int2 *__restrict__ binned, int n,
                                  int *__restrict__ counters) {
    const int stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = t; i < n_cell_words; i += stride) cells[i] = 0u;
    for (int i = t; i < n_lead; i += stride) lead_count[i] = 0;
    for (int i = t; i < n; i += stride) binned[i] = make_int2(-1, -1);
    if (t < 8) counters[t] = t == 4 || t == 5 ? 0x7fffffff : (t >= 6 ? -1 : 0);
}

__global__ void cell_fill_kernel(const int32_t *__restrict__ xy, int n, int rows, int cols, int shift, int gw, const unsigned *__restrict__ cell_start,
                                 unsigned *__restrict__ cell_fill, int2 *__restrict__ binned) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = xy[2 * i], y = xy[2 * i + 1];
    if (x < 0 || x >= cols || y < 0 || y >= rows) return;
    const int cell = (y >> shift) * gw + (x >> shift);
    const unsigned slot = cell_start[cell] + atomicAdd(&cell_fill[cell], 1u);
    binned[slot] = make_int2(x | (y << 16), i);
}

// A repeated position is represented by its first index: the later copies lose theirs (the top bit of the index is set, which
// makes it negative) and take no part.  Copies are compared by their indices as first written (the top bit masked off), so
// it does not matter which of them have been marked already.
__global__ void mark_repeats_kernel(int n, int shift, int gw, const unsigned *__restrict__ cell_start, int2 *__restrict__ binned) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 me = binned[i];
    if (me.x < 0) return;                                  // an unfilled slot (the call fails: a point outside the image)
    const int my_id = me.y & 0x7fffffff;
    const int cell = ((int)((unsigned)me.x >> 16) >> shift) * gw + ((me.x & 0xffff) >> shift);
    bool repeat = false;
    for (unsigned j = cell_start[cell]; j < cell_start[cell + 1]; ++j) {
        const int ox = binned[j].x, oy = __hip_atomic_load(&binned[j].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        repeat = repeat || (ox == me.x && (oy & 0x7fffffff) < my_id);
    }
    if (repeat) atomicOr(&binned[i].y, (int)0x80000000);
}

// ---- the ring of one point ----
// A point is served by NW wavefronts whose lanes all hold the SAME winner so far.  For an edge (p, a) the candidates on the
// searched side are ordered by where the centre of their circle through p and a lies on the bisector of p a, which is the
// order of the quotient (|c|^2 - a.c) / |a x c| (coordinates relative to p).  The candidates held on chip sit in registers,
// lane l holding candidates l, l + 64, ...: every lane ranks its own by that quotient in single precision, the smallest of
// the wavefront (a DPP minimum of the ordered bit patterns) becomes the winner, and then every lane asks its candidates
// whether one beats the winner under the exact test: a lane whose candidate does raises its hand, the first such candidate
// becomes the winner of all lanes and the lanes ask again, until no hand goes up.  The exact test runs behind a test against
// the winner's circle, a little enlarged, in double, so it is rare; after the ranking it only settles ties and rounding.
// MODE 0, the first pass: a wavefront per point of the binned list; the points of the 7 x 7 cells around it are gathered
//         once into two registers per lane (more than 128 of them: left to the second pass).
// MODE 1, for what the first pass left: the points of the window of half-width W (of W / 2 or W / 4 when those are too many)
//         are gathered once through LDS into WIDE_CAP / 64 registers per lane.  A winner whose circle stays within what
//         the window guarantees is final; otherwise the points of the binned list are candidates: for 128 rows of cells at
//         a time, every lane works out which columns of ITS two rows can hold a better candidate (inside the winner's
//         circle, on the searched side of the line p a) and reads those rows' bounds, and the NW wavefronts share out the
//         candidates of all rows in chunks of 64 (ranked in double when there is no winner yet: then in growing boxes of
//         cells around p, so that the first winner comes from nearby); their winners meet in LDS.
constexpr int WIDE_CAP = 384;

#ifdef SVX_MESH_PROBE        // measurement aid: what the second pass spends on each point {slot, ring steps, searches of all points, 10 ns ticks, held candidates}
__device__ int mesh_probe[8192][6];
__device__ int mesh_probe_n;
#endif

template <int NW, int MODE>
__global__ __launch_bounds__(64 * NW) void ring_wave_kernel(const int2 *__restrict__ binned, int n, const unsigned *__restrict__ cell_start, int shift,
                                                            int gw, int gh, int W, const int *__restrict__ todo_in, int in_slot,
                                                            int2 *__restrict__ lead_all, int *__restrict__ lead_count, int *__restrict__ todo_out,
                                                            int *__restrict__ counters, int out_slot) {
    __shared__ int2 win[NW];
    __shared__ int win_d[NW];
    __shared__ double win_k[NW];
    __shared__ int2 wide[MODE == 1 ? WIDE_CAP : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_todo = MODE == 0 ? n : counters[in_slot];
    const int X0 = counters[4], Y0 = counters[5], X1 = counters[6], Y1 = counters[7];      // the points' bounding box
    const int2 none = make_int2(-1, -1);
    for (int w = blockIdx.x; w < n_todo; w += gridDim.x) {
        const int slot = MODE == 0 ? w : todo_in[w];
#ifdef SVX_MESH_PROBE
        const long long probe_t0 = wall_clock64();
        int probe_steps = 0, probe_far = 0;
#endif
        const int2 me = binned[slot];
        if (MODE == 0 && me.y < 0) continue;                 // a repeated position (or an unfilled slot)
        const DPt p = unpack(me);
        int2 *lead = lead_all + (size_t)p.id * MAX_LEAD;
        const int cx = p.x >> shift, cy = p.y >> shift;
        // what the candidates guarantee: every point within `reach` pixels of p (no limit on a side where the window
        // ends at the image border; -1: no limit at all)
        // The held window as a pixel box, and around it the box that holds every point (X0..X1, Y0..Y1: their bounding box,
        // which the key points' borders keep well inside the frame).  A circle (or half-plane) that meets no part of that box
        // outside the window can hold no point that is not among the held candidates: the four rectangles outside the window
        // are tested one by one (this also passes the large circles of the flat triangles along the hull, which lie outside
        // the points' box, and the hull edges along its sides, where many points lie on one line).
        // mode 1: the window of half-width W, or the largest of W / 2, W / 4 whose points fit on chip (lane r looks at row r of the window)
        int We = W, n_wide = 0, w_s = 0, w_cnt = 0, w_incl = 0;
        if (MODE == 1) {
            for (int attempt = 0;; ++attempt, We >>= 1) {
                const int xa = max(cx - We, 0), xb = min(cx + We + 1, gw), ya = max(cy - We, 0), yb = min(cy + We + 1, gh);
                w_s = 0; w_cnt = 0;
                if (ya + lane < yb) { w_s = (int)cell_start[(ya + lane) * gw + xa]; w_cnt = (int)cell_start[(ya + lane) * gw + xb] - w_s; }
                w_incl = (int)svx_wave_scan_u32((unsigned)w_cnt);  // running total over the rows
                n_wide = __builtin_amdgcn_readlane(w_incl, 63);
                if (n_wide <= WIDE_CAP && yb - ya <= 64) break;
                if (attempt == 2) { n_wide = -1; break; }     // still too many: nothing held on chip, nothing guaranteed
            }
        }
        const int bx0 = max(cx - We, 0) << shift, bx1 = (min(cx + We + 1, gw) << shift) - 1;
        const int by0 = max(cy - We, 0) << shift, by1 = (min(cy + We + 1, gh) << shift) - 1;
        int reach = 0x7fffffff;
        if (bx0 > X0) reach = min(reach, p.x - bx0);
        if (bx1 < X1) reach = min(reach, bx1 - p.x);
        if (by0 > Y0) reach = min(reach, p.y - by0);
        if (by1 < Y1) reach = min(reach, by1 - p.y);
        if (reach == 0x7fffffff) reach = -1;
        auto circle_in_box = [&](const Circle &c) -> bool {
            if (reach < 0) return true;
            if (reach == 0) return false;                     // (mode 1 with nothing held)
            if (4.0 * c.r2 <= (double)reach * (double)reach) return true;      // p lies on the circle: no point of it is further from p than the diameter
            const double mx = (double)p.x + c.ox, my = (double)p.y + c.oy;
            auto clear = [&](int x0, int y0, int x1, int y1) {
                if (x0 > x1 || y0 > y1) return true;
                const double dx = fmax(fmax((double)x0 - mx, mx - (double)x1), 0.0), dy = fmax(fmax((double)y0 - my, my - (double)y1), 0.0);
                return dx * dx + dy * dy > c.r2 + 1e-6;
            };
            return clear(X0, Y0, bx0 - 1, Y1) && clear(bx1 + 1, Y0, X1, Y1) && clear(max(bx0, X0), Y0, min(bx1, X1), by0 - 1) && clear(max(bx0, X0), by1 + 1, min(bx1, X1), Y1);
        };
        // the 7 x 7 cells around the point into registers (mode 0): the row bounds are read by independent loads
        int2 c0 = none, c1 = none;
        int n_near = 0;
        bool overfull = false;
        if (MODE == 0) {
            const int xa = max(cx - 3, 0), xb = min(cx + 4, gw), ya = max(cy - 3, 0), yb = min(cy + 4, gh);
            int rs[7], re[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const bool in = ya + r < yb;
                rs[r] = in ? (int)cell_start[(ya + r) * gw + xa] : 0;
                re[r] = in ? (int)cell_start[(ya + r) * gw + xb] : 0;
            }
            int j0 = -1, j1 = -1;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int cnt = re[r] - rs[r], i0 = lane - n_near, i1 = lane + 64 - n_near;
                if (i0 >= 0 && i0 < cnt) j0 = rs[r] + i0;
                if (i1 >= 0 && i1 < cnt) j1 = rs[r] + i1;
                n_near += cnt;
            }
            if (j0 >= 0) c0 = binned[j0];
            if (j1 >= 0) c1 = binned[j1];
            overfull = n_near > 128;
        }
        // the window's points into LDS (mode 1), row by row
        if (MODE == 1) {
            __syncthreads();                                  // the last point's reads of the LDS window are done
            if (n_wide < 0) { n_wide = 0; reach = 0; }
            else {
                const int n_rows = min(cy + We + 1, gh) - max(cy - We, 0);
                for (int r = 0; r < n_rows; ++r) {
                    const int rs = __shfl(w_s, r, 64), rc = __shfl(w_cnt, r, 64), at = __shfl(w_incl - w_cnt, r, 64);
                    for (int i = threadIdx.x; i < rc; i += 64 * NW) wide[at + i] = binned[rs + i];
                }
            }
            __syncthreads();
        }
        // the candidates held on chip, in registers: lane l holds candidates l, l + 64, ... (n_chunks registers in use)
        constexpr int NC = MODE == 0 ? 2 : WIDE_CAP / 64;
        int2 cand[NC];
        int n_chunks;
        if (MODE == 0) { cand[0] = c0; cand[1] = c1; n_chunks = n_near > 64 ? 2 : 1; }
        else {
#pragma unroll
            for (int j = 0; j < NC; ++j) cand[j] = 64 * j + lane < n_wide ? wide[64 * j + lane] : none;
            n_chunks = (n_wide + 63) >> 6;
        }
        // far(f, cols): mode 1 only, f(candidate or none) for every lane of every chunk this wavefront takes of the binned
        // list; cols(y0, y1, x0, x1) narrows the pixel columns [x0, x1] worth reading in the pixel rows y0..y1 (x0 > x1: none)
        // (a lane per row of cells that walks its own row is faster when a search meets a handful of points and twice as slow
        // when it meets hundreds, which is what the slowest points do)
        auto far = [&](auto f, auto cols) {
            if (MODE != 1) return;
#ifdef SVX_MESH_PROBE
            ++probe_far;
#endif
            for (int g = 0; g < gh; g += 128) {                // 128 rows of cells a round: two per lane, their loads in flight together
                int s[2] = {0, 0}, cnt[2] = {0, 0};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int y = g + 64 * h + lane;
                    if (y < gh) {
                        double x0 = 0.0, x1 = (double)((gw << shift) - 1);
                        cols((double)(y << shift), (double)(((y + 1) << shift) - 1), x0, x1);
                        if (x0 <= x1) {
                            const int xa = (int)x0 >> shift, xb = min(((int)x1 >> shift) + 1, gw);
                            s[h] = (int)cell_start[y * gw + xa];
                            cnt[h] = (int)cell_start[y * gw + xb] - s[h];
                        }
                    }
                }
                const int incl0 = (int)svx_wave_scan_u32((unsigned)cnt[0]), incl1 = (int)svx_wave_scan_u32((unsigned)cnt[1]);
                const int total0 = __builtin_amdgcn_readlane(incl0, 63), total = total0 + __builtin_amdgcn_readlane(incl1, 63);
                const int at0 = incl0 - cnt[0], at1 = total0 + incl1 - cnt[1];          // where each row starts among the round's candidates
                for (int base = 64 * wave; base < total; base += 256 * NW) {          // four loads in flight
                    int2 c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = base + 64 * NW * u + lane;
                        c[u] = none;
                        if (base + 64 * NW * u < total) {                              // (uniform: the shuffles below need every lane)
                            const bool second = idx >= total0;
                            int lo = 0, hi = 63;                                        // the last row that starts at or before idx
                            for (int it = 0; it < 6; ++it) {
                                const int mid = (lo + hi + 1) >> 1;
                                const int m0 = __shfl(at0, mid, 64), m1 = __shfl(at1, mid, 64);
                                if ((second ? m1 : m0) <= idx) lo = mid; else hi = mid - 1;
                            }
                            const int rs0 = __shfl(s[0], lo, 64), rs1 = __shfl(s[1], lo, 64), ra0 = __shfl(at0, lo, 64), ra1 = __shfl(at1, lo, 64);
                            if (idx < total) c[u] = binned[(second ? rs1 : rs0) + idx - (second ? ra1 : ra0)];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (base + 64 * NW * u < total) f(c[u]);
                }
            }
        };
        // columns of the disc around (mx, my) of radius r within the pixel rows y0..y1 (doubles: r may be huge)
        auto disc_cols = [&](double mx, double my, double r, double y0, double y1, double &x0, double &x1) {
            const double d = my < y0 ? y0 - my : (my > y1 ? my - y1 : 0.0);
            if (d > r) { x0 = 1.0; x1 = 0.0; return; }
            const double half = sqrt(r * r - d * d) + 1.0;
            x0 = fmax(x0, mx - half); x1 = fmin(x1, mx + half);
        };
        // nearest neighbour (ties: the smaller index, so that every wavefront agrees)
        int bd = 0x7fffffff; int2 bc = none;
        auto nearer = [&](int od, int2 oc) { if (oc.y >= 0 && (od < bd || (od == bd && oc.y < bc.y))) { bd = od; bc = oc; } };
        auto by_distance = [&](int2 c) {
            if (c.y < 0 || c.x == me.x) return;
            const int dx = (c.x & 0xffff) - p.x, dy = (int)((unsigned)c.x >> 16) - p.y;
            nearer(dx * dx + dy * dy, c);
        };
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (j < n_chunks) by_distance(cand[j]);
        {   // (which of several nearest points is taken does not matter: each of them is a neighbour)
            const unsigned dmin = wave_min_u32((unsigned)bd);
            const unsigned long long who = __ballot((unsigned)bd == dmin);
            bc = dmin == 0x7fffffffu ? none : lane_value(bc, __ffsll((long long)who) - 1);
            bd = (int)dmin;
        }
        if (MODE == 1 && !(bc.y >= 0 && (reach < 0 || (long long)bd <= (long long)reach * reach))) {
            const double r_near = bd == 0x7fffffff ? 1e30 : sqrt((double)bd) + 1.0;     // a nearer point lies within the nearest found in the window
            far(by_distance, [&](double y0, double y1, double &x0, double &x1) { disc_cols((double)p.x, (double)p.y, r_near, y0, y1, x0, x1); });
            for (int o = 32; o; o >>= 1) nearer(__shfl_xor(bd, o, 64), make_int2(__shfl_xor(bc.x, o, 64), __shfl_xor(bc.y, o, 64)));
            if (NW > 1) {
                __syncthreads();
                if (lane == 0) { win[wave] = bc; win_d[wave] = bd; }
                __syncthreads();
                for (int v = 0; v < NW; ++v) nearer(win_d[v], win[v]);
            }
        }
        int n_lead = 0;
        int verdict = 0;                                      // 0 done, -1 left over again, -2 beyond the limits
        if (overfull) verdict = -1;
        else if (bc.y < 0) verdict = (MODE == 1 || reach < 0) ? 0 : -1;          // (mode 1 has seen every point: there is no other)
        else if (MODE == 0 && reach >= 0 && (long long)bd > (long long)reach * reach) verdict = -1;
        else {
            const int2 a0c = bc;
            bool open = false;
            for (int dir = 0; dir < 2 && verdict == 0; ++dir) {          // 0: counter-clockwise from a0; 1: clockwise, for an open ring
                if (dir == 1 && !open) break;
                DPt a = unpack(a0c);
                int axy = a0c.x;
                for (int step = 0;; ++step) {
                    if (step >= MAX_RING) { verdict = -2; break; }
#ifdef SVX_MESH_PROBE
                    ++probe_steps;
#endif
                    DPt b; int2 bcc = none;
                    Circle circ;
                    // the nearby candidates ranked in double; when none of them lies on the searched side (mode 2), all that do
                    const int ax = a.x - p.x, ay = a.y - p.y;
                    const double A = dir == 0 ? (double)ay : (double)-ay, B = dir == 0 ? (double)ax : (double)-ax;
                    // the searched side of the line p -> a in the pixel rows y0..y1: A (x - px) < B (y - py)
                    auto side_cols = [&](double y0, double y1, double &x0, double &x1) {
                        if (A > 0.0) x1 = fmin(x1, (double)p.x + fmax(B * (y0 - p.y), B * (y1 - p.y)) / A + 1.0);
                        else if (A < 0.0) x0 = fmax(x0, (double)p.x + fmax(B * (y0 - p.y), B * (y1 - p.y)) / A - 1.0);   // (the larger numerator: A is negative)
                        else if (B > 0.0 ? !(y1 > p.y) : !(y0 < p.y)) { x0 = 1.0; x1 = 0.0; }
                    };
                    double bk = 1e300; int2 bkc = none;
                    auto rank = [&](int2 cc) {
                        if (cc.y < 0 || cc.x == me.x || cc.x == axy) return;
                        const int ccx = (cc.x & 0xffff) - p.x, ccy = (int)((unsigned)cc.x >> 16) - p.y;
                        const int cr = ax * ccy - ay * ccx;
                        if (dir == 0 ? cr <= 0 : cr >= 0) return;
                        const double k = (double)(ccx * ccx + ccy * ccy - (ax * ccx + ay * ccy)) / fabs((double)cr);
                        if (k < bk || (k == bk && cc.y > bkc.y)) { bk = k; bkc = cc; }
                    };
                    auto rank_meet = [&]() {
                        for (int o = 32; o; o >>= 1) {
                            const double ok = __shfl_xor(bk, o, 64);
                            const int ox = __shfl_xor(bkc.x, o, 64), oy = __shfl_xor(bkc.y, o, 64);
                            if (ok < bk || (ok == bk && oy > bkc.y)) { bk = ok; bkc = make_int2(ox, oy); }
                        }
                    };
                    {   // the held candidates in single precision (what the order of the rounded quotients leaves open, the exact pass settles)
                        float kb = __builtin_inff();
                        int2 kc = none;
#pragma unroll
                        for (int j = 0; j < NC; ++j)
                            if (j < n_chunks) {
                                const int2 cc = cand[j];
                                const int ccx = (cc.x & 0xffff) - p.x, ccy = (int)((unsigned)cc.x >> 16) - p.y;
                                const int cr = __mul24(ax, ccy) - __mul24(ay, ccx);
                                const bool on = cc.y >= 0 && cc.x != me.x && cc.x != axy && (dir == 0 ? cr > 0 : cr < 0);
                                const float k = (float)(__mul24(ccx, ccx) + __mul24(ccy, ccy) - (__mul24(ax, ccx) + __mul24(ay, ccy))) * __builtin_amdgcn_rcpf((float)abs(cr));
                                if (on && k < kb) { kb = k; kc = cc; }
                            }
                        const unsigned key = ordered_bits(kb), kmin = wave_min_u32(key);
                        if (kmin != ordered_bits(__builtin_inff())) bkc = lane_value(kc, __ffsll((long long)__ballot(key == kmin)) - 1);
                    }
                    // No held candidate on the searched side: p -> a is a hull edge if no point outside the held window lies there
                    // either, which is certain when the part of the frame outside the window's box (up to four rectangles) lies
                    // wholly on the other side of the line: a rectangle does when its four corners do.
                    bool side_empty = false;
                    if (bkc.y < 0 && !(MODE == 1 && n_wide == 0)) {
                        auto away = [&](int x, int y) { const int o = ax * (y - p.y) - ay * (x - p.x); return dir == 0 ? o <= 0 : o >= 0; };
                        auto clear = [&](int x0, int y0, int x1, int y1) {
                            return x0 > x1 || y0 > y1 || (away(x0, y0) && away(x1, y0) && away(x0, y1) && away(x1, y1));
                        };
                        side_empty = clear(X0, Y0, bx0 - 1, Y1) && clear(bx1 + 1, Y0, X1, Y1) && clear(max(bx0, X0), Y0, min(bx1, X1), by0 - 1) && clear(max(bx0, X0), by1 + 1, min(bx1, X1), Y1);
                    }
                    // No held candidate on the searched side, and points may lie there: a first winner is looked for in growing
                    // boxes of cells around p (ranking the whole half-plane would read half of all points to tell what the nearest
                    // ones tell as well); the exact pass below then only reads what lies inside that winner's circle.
                    bool all_ranked = false;                              // every point on the searched side has been ranked
                    if (MODE == 1 && bkc.y < 0 && !side_empty) {
                        for (int wk = 4 * max(W, 1);; wk *= 4) {
                            const bool all = cx - wk <= 0 && cx + wk + 1 >= gw && cy - wk <= 0 && cy + wk + 1 >= gh;
                            const double fx0 = (double)(max(cx - wk, 0) << shift), fx1 = (double)((min(cx + wk + 1, gw) << shift) - 1);
                            const double fy0 = (double)(max(cy - wk, 0) << shift), fy1 = (double)((min(cy + wk + 1, gh) << shift) - 1);
                            far(rank, [&](double y0, double y1, double &x0, double &x1) {
                                if (y1 < fy0 || y0 > fy1) { x0 = 1.0; x1 = 0.0; return; }
                                side_cols(y0, y1, x0, x1);
                                x0 = fmax(x0, fx0); x1 = fmin(x1, fx1);
                            });
                            rank_meet();
                            if (NW > 1) {
                                __syncthreads();
                                if (lane == 0) { win[wave] = bkc; win_k[wave] = bk; }
                                __syncthreads();
                                for (int v = 0; v < NW; ++v)
                                    if (win_k[v] < bk || (win_k[v] == bk && win[v].y > bkc.y)) { bk = win_k[v]; bkc = win[v]; }
                            }
                            if (bkc.y >= 0 || all) { all_ranked = all; break; }
                        }
                    }
                    if (bkc.y >= 0) { bcc = bkc; b = unpack(bcc); circ.set(p, a, b); }
                    auto beats = [&](int2 cc) -> bool {                   // exact: does cc (on the searched side) beat the winner so far?
                        if (cc.y < 0 || cc.x == me.x || cc.x == axy || cc.x == bcc.x) return false;
                        if (bcc.y >= 0 && circ.outside(p, cc.x)) return false;
                        const DPt c = unpack(cc);
                        const int o = d_orient(p, a, c);
                        if (dir == 0 ? o <= 0 : o >= 0) return false;
                        return bcc.y < 0 || (dir == 0 ? d_in_circle(p, a, b, c) : d_in_circle(p, b, a, c)) > 0;
                    };
                    auto hands = [&](int2 cc) {
                        for (;;) {
                            const unsigned long long up = __ballot(beats(cc));
                            if (!up) break;
                            const int src = __ffsll((long long)up) - 1;
                            bcc = make_int2(__shfl(cc.x, src, 64), __shfl(cc.y, src, 64));
                            b = unpack(bcc);
                            circ.set(p, a, b);
                        }
                    };
                    for (;;) {                                            // the held candidates: every lane asks its own until one beats the winner
                        bool up = false;
                        int2 mine = none;
#pragma unroll
                        for (int j = 0; j < NC; ++j)
                            if (j < n_chunks && !up && beats(cand[j])) { up = true; mine = cand[j]; }
                        const unsigned long long who = __ballot(up);
                        if (!who) break;
                        bcc = lane_value(mine, __ffsll((long long)who) - 1);
                        b = unpack(bcc);
                        circ.set(p, a, b);
                    }
                    if (MODE == 1 && !side_empty && !(all_ranked && bkc.y < 0) && !(bcc.y >= 0 && circle_in_box(circ))) {
                        // the searched side of the line p -> a and, if there is a winner, its circle
                        const bool has = bcc.y >= 0;
                        const double mx = (double)p.x + circ.ox, my = (double)p.y + circ.oy, mr = has ? sqrt(circ.r2) : 0.0;
                        far(hands, [&](double y0, double y1, double &x0, double &x1) {
                            side_cols(y0, y1, x0, x1);
                            if (x0 > x1) return;
                            if (has) disc_cols(mx, my, mr, y0, y1, x0, x1);
                        });
                        if (NW > 1) {
                            __syncthreads();
                            if (lane == 0) win[wave] = bcc;
                            __syncthreads();
                            bcc = none;
                            for (int v = 0; v < NW; ++v)
                                if (beats(win[v])) { bcc = win[v]; b = unpack(bcc); circ.set(p, a, b); }
                        }
                    }
                    if (bcc.y < 0) {
                        if (MODE == 0 && reach >= 0 && !side_empty) verdict = -1;      // only more points can tell whether this is a hull edge
                        if (dir == 0) open = true;
                        break;
                    }
                    if (MODE == 0 && !circle_in_box(circ)) { verdict = -1; break; }
                    const DPt u = dir == 0 ? a : b, v2 = dir == 0 ? b : a;     // triangle (p, u, v2), counter-clockwise
                    if (p.id < u.id && p.id < v2.id) {
                        if (n_lead >= MAX_LEAD) { verdict = -2; break; }
                        if (threadIdx.x == 0) lead[n_lead] = make_int2(u.id, v2.id);
                        ++n_lead;
                    }
                    a = b; axy = bcc.x;
                    if (dir == 0 && axy == a0c.x) break;
                }
            }
        }
#ifdef SVX_MESH_PROBE
        if (MODE == 1 && threadIdx.x == 0) {
            const int k = atomicAdd(&mesh_probe_n, 1);
            if (k < 8192) { mesh_probe[k][0] = slot; mesh_probe[k][1] = probe_steps; mesh_probe[k][2] = probe_far; mesh_probe[k][3] = (int)(wall_clock64() - probe_t0); mesh_probe[k][4] = n_wide; mesh_probe[k][5] = me.x; }
        }
#endif
        if (threadIdx.x == 0) {
            if (verdict == 0) lead_count[p.id] = n_lead;
            else if (verdict == -1 && todo_out) todo_out[atomicAdd(&counters[out_slot], 1)] = slot;
            else atomicOr(&counters[1], 1);
        }
        if (NW > 1 || MODE == 1) __syncthreads();
    }
}

// Triangles in their final order: point by point, a point's triangles by their other two indices.
__global__ void write_triangles_kernel(const int32_t *__restrict__ xy, int n, const int2 *__restrict__ lead_all, const int *__restrict__ lead_count,
                                       const unsigned *__restrict__ lead_start, int capacity, int32_t *__restrict__ tri_idx, int32_t *__restrict__ tri_xy,
                                       const int *__restrict__ counters, svx_mail *mail, unsigned seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && mail) {                                  // the host sizes the next launches by the number of triangles
        mail->v[0] = lead_start[n];
        mail->v[1] = (unsigned)counters[0]; mail->v[2] = (unsigned)counters[1]; mail->v[3] = (unsigned)counters[2];
        svx_mail_post(mail, seq);
    }
    if (i >= n) return;
    const int cnt = lead_count[i];
    const int2 *mine = lead_all + (size_t)i * MAX_LEAD;
    unsigned slot = lead_start[i];
    long long last = -1;
    for (int r = 0; r < cnt; ++r, ++slot) {                 // selection in ascending (v1, v2): the pairs of one point are distinct
        long long best = 0x7fffffffffffffffLL;
        for (int j = 0; j < cnt; ++j) {
            const long long key = ((long long)mine[j].x << 32) | (unsigned)mine[j].y;
            if (key > last && key < best) best = key;
        }
        last = best;
        if ((int)slot >= capacity) continue;
        const int v1 = (int)(best >> 32), v2 = (int)(best & 0xffffffffLL);
        if (tri_idx) { tri_idx[3 * (size_t)slot] = i; tri_idx[3 * (size_t)slot + 1] = v1; tri_idx[3 * (size_t)slot + 2] = v2; }
        if (tri_xy) {
            int32_t *o = tri_xy + 6 * (size_t)slot;
            o[0] = xy[2 * i]; o[1] = xy[2 * i + 1]; o[2] = xy[2 * v1]; o[3] = xy[2 * v1 + 1]; o[4] = xy[2 * v2]; o[5] = xy[2 * v2 + 1];
        }
    }
}

template <class T>
T *carve(char *&at, size_t count) {
    T *p = (T *)at;
    at += (count * sizeof(T) + 255) / 256 * 256;
    return p;
}

}  // namespace

// Device triangulation into the context's mesh block.  On success *tri_xy_out (corner coordinates, six per triangle) and
// *tri_idx_out (index triples) point into that block, *n_tri is the number of triangles, and the block has room for the
// planes behind them (*abc_out).  Returns SVX_ERR_UNSUPPORTED (no message change) when the point set is one the device
// form does not take (a ring beyond its limits): the caller then uses the host triangulation.
int svx_mesh_device_triangles(svx_ctx *ctx, const int32_t *pts_xy, int n, int rows, int cols, int32_t **tri_xy_out, int32_t **tri_idx_out,
                              float **abc_out, int *n_tri) {
    *n_tri = 0;
    if (n < 3) return SVX_OK;
    if (rows > 8192 || cols > 8192) return SVX_ERR_UNSUPPORTED;
    // cells of about one point each: side 2^shift nearest the square root of the area per point, 4 .. 64 pixels
    int shift = 2;
    while (shift < 6 && (double)(1 << shift) * (1 << shift) * 2.0 < (double)rows * cols / n) ++shift;
    const int gw = ((cols - 1) >> shift) + 1, gh = ((rows - 1) >> shift) + 1, n_cells = gw * gh;
    const int W1 = 3, W2 = 8;                     // half-widths, in cells, of the two windowed passes
    const size_t max_tri = 2 * (size_t)n;
    size_t bytes = 0;
    {
        char *at = nullptr;
        carve<int32_t>(at, 6 * max_tri); carve<int32_t>(at, 3 * max_tri); carve<float>(at, 3 * max_tri);
        carve<unsigned>(at, 2 * (size_t)n_cells + 2); carve<int2>(at, n); carve<int>(at, n);
        carve<unsigned>(at, (size_t)n + 1); carve<int>(at, (size_t)n + 1); carve<int2>(at, (size_t)n * MAX_LEAD); carve<int>(at, 8);
        bytes = (size_t)(at - (char *)nullptr);
    }
    int rc = svx_block_reserve(ctx, &ctx->tri, &ctx->tri_bytes, bytes, "mesh on the device");
    if (rc) return rc;
    char *at = (char *)ctx->tri;
    int32_t *tri_xy = carve<int32_t>(at, 6 * max_tri);
    int32_t *tri_idx = carve<int32_t>(at, 3 * max_tri);
    float *abc = carve<float>(at, 3 * max_tri);
    unsigned *cell_start = carve<unsigned>(at, 2 * (size_t)n_cells + 2);   // counts, scanned in place; then the fill counters
    unsigned *cell_fill = cell_start + n_cells + 1;
    int2 *binned = carve<int2>(at, n);
    int *todo = carve<int>(at, n);
    unsigned *lead_start = carve<unsigned>(at, (size_t)n + 1);
    int *lead_count = carve<int>(at, (size_t)n + 1);
    int2 *lead_all = carve<int2>(at, (size_t)n * MAX_LEAD);
    int *counters = carve<int>(at, 8);        // [0] points left by the first pass, [1] limits exceeded, [2] point outside the image, [3] unused, [4..7] the points' bounding box
    hipStream_t st = ctx->stream;
    const unsigned nb = svx_cdiv(n, 256);
    mesh_reset_kernel<<<std::min(svx_cdiv(std::max(2 * n_cells + 2, n + 1), 256), 1024u), 256, 0, st>>>(cell_start, 2 * n_cells + 2, lead_count, n + 1, binned, n, counters);
    SVX_LAUNCH_CHECK(ctx, "mesh_reset_kernel");
    int h_counters[4];
    cell_count_kernel<<<svx_cdiv(n, 1024), 1024, 0, st>>>(pts_xy, n, rows, cols, shift, gw, cell_start, counters);
    SVX_LAUNCH_CHECK(ctx, "cell_count_kernel");
    scan4_kernel<<<1, 1024, 0, st>>>(cell_start, n_cells, cell_start, nullptr, 0);
    SVX_LAUNCH_CHECK(ctx, "scan4_kernel");
    cell_fill_kernel<<<nb, 256, 0, st>>>(pts_xy, n, rows, cols, shift, gw, cell_start, cell_fill, binned);
    SVX_LAUNCH_CHECK(ctx, "cell_fill_kernel");
    mark_repeats_kernel<<<nb, 256, 0, st>>>(n, shift, gw, cell_start, binned);
    SVX_LAUNCH_CHECK(ctx, "mark_repeats_kernel");
    ring_wave_kernel<1, 0><<<std::min(n, 65536), 64, 0, st>>>(binned, n, cell_start, shift, gw, gh, W1, nullptr, 0, lead_all, lead_count, todo, counters, 0);
    SVX_LAUNCH_CHECK(ctx, "ring_wave_kernel");
    ring_wave_kernel<2, 1><<<std::min(n, 8192), 128, 0, st>>>(binned, n, cell_start, shift, gw, gh, W2, todo, 0, lead_all, lead_count, nullptr, counters, 3);
    SVX_LAUNCH_CHECK(ctx, "ring_wave_kernel");
    svx_mail *mail = nullptr;
    unsigned seq = 0;
    rc = svx_mail_next(ctx, &mail, &seq);
    if (rc) return rc;
    scan4_kernel<<<1, 1024, 0, st>>>(lead_start, n, (const unsigned *)lead_count, nullptr, 0);
    SVX_LAUNCH_CHECK(ctx, "scan4_kernel");
    write_triangles_kernel<<<nb, 256, 0, st>>>(pts_xy, n, lead_all, lead_count, lead_start, (int)max_tri, tri_idx, tri_xy, counters, mail, seq);
    SVX_LAUNCH_CHECK(ctx, "write_triangles_kernel");
    const unsigned *got = nullptr;                        // the number of triangles, then counters[0..2], as soon as they are known
    rc = svx_mail_wait(ctx, seq, &got);
    if (rc) return rc;
    const unsigned total = got[0];
    for (int k = 0; k < 3; ++k) h_counters[k] = (int)got[1 + k];
    if (h_counters[2]) return svx_fail(ctx, SVX_ERR_INVALID_ARG, "mesh on the device: support point outside the image");
    if (h_counters[1] || total > max_tri) return SVX_ERR_UNSUPPORTED;
    ctx->mesh_left_over = 0;
    ctx->mesh_left_first = h_counters[0];
    *n_tri = (int)total;
    *tri_xy_out = tri_xy;
    *tri_idx_out = tri_idx;
    *abc_out = abc;
    return SVX_OK;
}

#ifdef SVX_MESH_PROBE
// copies the probe records of the second pass to the host and clears them; returns their number
SVX_API int svx_mesh_probe_dump(int *out, int capacity) {
    int n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(mesh_probe_n), sizeof(int));
    n = std::min(std::min(n, capacity), 8192);
    if (n) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mesh_probe), (size_t)n * 6 * sizeof(int));
    const int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mesh_probe_n), &zero, sizeof(int));
    return n;
}
#endif

SVX_API int svx_mesh_device_stats(const svx_ctx *ctx, int *left_by_first_pass, int *left_by_second_pass) {
    if (!ctx) return SVX_ERR_INVALID_ARG;
    if (left_by_first_pass) *left_by_first_pass = ctx->mesh_left_first;
    if (left_by_second_pass) *left_by_second_pass = ctx->mesh_left_over;
    return SVX_OK;
}

SVX_API int svx_delaunay_device(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, int rows, int cols, int32_t *tri_idx, int capacity, int *n_tri_host) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n_pts >= 0 && capacity >= 0 && n_tri_host, SVX_ERR_INVALID_ARG, "svx_delaunay_device: bad count or null result");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_delaunay_device: empty image");
    SVX_REQUIRE(ctx, (n_pts == 0 || pts_xy) && (capacity == 0 || tri_idx), SVX_ERR_INVALID_ARG, "svx_delaunay_device: null pointer");
    int32_t *d_xy = nullptr, *d_idx = nullptr;
    float *d_abc = nullptr;
    rc = svx_mesh_device_triangles(ctx, pts_xy, n_pts, rows, cols, &d_xy, &d_idx, &d_abc, n_tri_host);
    if (rc == SVX_ERR_UNSUPPORTED)
        return svx_fail(ctx, rc, "svx_delaunay_device: frames beyond 8192 x 8192 or a point with more than %d triangles around it: use svx_delaunay", MAX_RING);
    if (rc) return rc;
    const int m = std::min(*n_tri_host, capacity);
    if (m) SVX_HIP(ctx, hipMemcpyAsync(tri_idx, d_idx, (size_t)m * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return SVX_OK;
}

SVX_API int svx_mesh_plane_prior_device(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, const float *disparities, int rows, int cols, float *prior,
                                        int *n_tri_out) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n_pts >= 0, SVX_ERR_INVALID_ARG, "svx_mesh_plane_prior_device: negative point count");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_mesh_plane_prior_device: empty image");
    SVX_REQUIRE(ctx, disparities && prior && (n_pts == 0 || pts_xy), SVX_ERR_INVALID_ARG, "svx_mesh_plane_prior_device: null pointer");
    int32_t *d_xy = nullptr, *d_idx = nullptr;
    float *d_abc = nullptr;
    int nt = 0;
    rc = svx_mesh_device_triangles(ctx, pts_xy, n_pts, rows, cols, &d_xy, &d_idx, &d_abc, &nt);
    if (rc == SVX_ERR_UNSUPPORTED) {                     // the host triangulation takes every point set
        std::vector<int32_t> h(2 * (size_t)n_pts);
        SVX_HIP(ctx, hipMemcpyAsync(h.data(), pts_xy, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return svx_mesh_plane_prior(ctx, h.data(), n_pts, disparities, rows, cols, prior, n_tri_out);
    }
    if (rc) return rc;
    if (n_tri_out) *n_tri_out = nt;
    if (nt == 0) return SVX_OK;
    rc = svx_plane_fit(ctx, d_xy, nt, disparities, rows, cols, d_abc);
    if (!rc) rc = svx_plane_prior_raster(ctx, d_xy, d_abc, nt, rows, cols, prior);
    return rc;
}
