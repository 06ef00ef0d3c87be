/*
 * svx_oracle.c -- CPU restatement of the reference's census stereo-matching
 * hot path.  TEST INFRASTRUCTURE ONLY: this file is the checker that the HIP
 * kernels are compared against (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing in the product path links, loads or
 * calls it.
 *
 * Every function restates one reference routine in plain scalar C and cites
 * the reference file:line it follows (paths relative to the reference root).
 * Nothing here is copied from the reference: the SSE / OpenCV formulations
 * are re-expressed as scalar loops over flat row-major buffers.
 *
 * Parity pinning: the reference cannot be built in this container (every hot
 * translation unit includes OpenCV 2.4 / Boost headers that are absent), so
 * the oracle is pinned against the known-answer values that were captured
 * from the reference at survey time (tests/golden/survey_kat.json) and, for
 * the popcount primitive, against the reference's own hammingdistance.cpp
 * compiled from where it lies (oracle/_ref).
 *
 * Memory model shared by all functions ("flat guard" semantics): images are
 * contiguous row-major with step == cols.  The reference does pointer
 * arithmetic that leaves the row (wrapping into the neighbouring row of the
 * same contiguous buffer) and, at four corner pixels, the buffer itself.  A
 * read at a flat index outside [0, rows*cols) is DEFINED here to return 0,
 * which is what the reference returns when its buffers sit between zeroed
 * guard bands.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- helpers */

static inline int8_t rd_s8(const int8_t *img, long n, long idx) {
    return (idx >= 0 && idx < n) ? img[idx] : (int8_t)0;
}
static inline uint32_t rd_u32(const uint32_t *img, long n, long idx) {
    return (idx >= 0 && idx < n) ? img[idx] : 0u;
}

/* FNV-1a 64 over a byte range (offset 1469598103934665603, prime 1099511628211). */
ORC_API uint64_t orc_fnv1a64(const void *data, size_t nbytes, uint64_t h) {
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < nbytes; ++i) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}
ORC_API uint64_t orc_fnv1a64_init(void) { return 1469598103934665603ULL; }

/* ------------------------------------------------------------------ a1 --
 * sparsestereo/imageconversion.cpp:63-78  ImageConversion::unsignedToSigned
 * dst = src - 128 with 8-bit wrap-around, i.e. src ^ 0x80. */
ORC_API void orc_u8_to_s8(const uint8_t *src, int8_t *dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = (int8_t)(uint8_t)(src[i] - 128u);
}

/* ------------------------------------------------------------------ a3 --
 * sparsestereo/hammingdistance.h:18-46 (LUT) and :58-76 (parallel count).
 * Restated as the classic mask-and-add reduction; equals hardware popcount. */
ORC_API unsigned orc_popcount32(uint32_t v) {
    v = (v & 0x55555555u) + ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    v = (v & 0x0f0f0f0fu) + ((v >> 4) & 0x0f0f0f0fu);
    v = (v & 0x00ff00ffu) + ((v >> 8) & 0x00ff00ffu);
    return (v & 0xffffu) + (v >> 16);
}
ORC_API unsigned orc_hamming32(uint32_t a, uint32_t b) { return orc_popcount32(a ^ b); }

/* ------------------------------------------------------------------ a2 --
 * sparsestereo/census.cpp:17-94  Census::transform5x5SSE (+ :96-108 store).
 * 24-bit layout.  For rows y in [2, rows-2) and ALL x in [0, cols):
 *   neighbour (dy,dx), idx=(dy+2)*5+(dx+2), g=24-idx, bit = g<12 ? g : g-1,
 *   bit set when centre > neighbour (signed 8-bit compare).
 * The neighbour is fetched at flat index (y+dy)*cols + (x+dx): columns 0,1 and
 * cols-2,cols-1 therefore see bytes of the adjacent row (census.cpp:28,41).
 * Rows 0,1,rows-2,rows-1 of `out` are never written (census.cpp:22). */
ORC_API void orc_census5x5_sse24(const int8_t *in, uint32_t *out, int rows, int cols) {
    const long n = (long)rows * cols;
    for (int y = 2; y < rows - 2; ++y)
        for (int x = 0; x < cols; ++x) {
            const int8_t c = in[(long)y * cols + x];
            uint32_t w = 0;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const int idx = (dy + 2) * 5 + (dx + 2);
                    const int g = 24 - idx;
                    const int bit = g < 12 ? g : g - 1;
                    const int8_t nb = rd_s8(in, n, (long)(y + dy) * cols + (x + dx));
                    if (c > nb) w |= 1u << bit;
                }
            out[(long)y * cols + x] = w;
        }
}

/* ----------------------------------------------------------------- a2' --
 * sparsestereo/census-inl.h:13-53  generic Census::transform5x5<T>.
 * 25-bit layout bit = 24-idx (centre bit 12 always 0), interior pixels only
 * (x in [2,cols-2), y in [2,rows-2)); everything else is left untouched. */
ORC_API void orc_census5x5_generic25(const int8_t *in, uint32_t *out, int rows, int cols) {
    for (int y = 2; y < rows - 2; ++y)
        for (int x = 2; x < cols - 2; ++x) {
            const int8_t c = in[(long)y * cols + x];
            uint32_t w = 0;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const int idx = (dy + 2) * 5 + (dx + 2);
                    if (c > in[(long)(y + dy) * cols + (x + dx)]) w |= 1u << (24 - idx);
                }
            out[(long)y * cols + x] = w;
        }
}

/* ------------------------------------------------------------------ a4 --
 * sparsestereo/censuswindow.h:96-107  CensusWindow<5>::match, popcount build.
 * Three 64-bit loads per row starting at column x-2 => 5 rows x 6 columns
 * (x-2 .. x+3).  Flat addressing: column x+3 may be the next row's first
 * word.  Result is returned through `short` in the reference (max 720). */
ORC_API int orc_window_cost_5x6(const uint32_t *ref, const uint32_t *cmp, int rows, int cols,
                                int rx, int ry, int px, int py) {
    const long n = (long)rows * cols;
    int cost = 0;
    for (int dy = -2; dy <= 2; ++dy)
        for (int k = 0; k < 6; ++k) {
            const uint32_t a = rd_u32(ref, n, (long)(ry + dy) * cols + (rx - 2 + k));
            const uint32_t b = rd_u32(cmp, n, (long)(py + dy) * cols + (px - 2 + k));
            cost += (int)orc_popcount32(a ^ b);
        }
    return (int)(short)cost;
}

/* ----------------------------------------------------------------- a4' --
 * sparsestereo/censuswindow.h:108-113 (and the LUT/SSE forms :119-232): the
 * exact 5x5 window.  Not the behaviour of the reference as it compiles on
 * x86-64 (that is a4); offered as an option. */
ORC_API int orc_window_cost_5x5(const uint32_t *ref, const uint32_t *cmp, int rows, int cols,
                                int rx, int ry, int px, int py) {
    const long n = (long)rows * cols;
    int cost = 0;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            const uint32_t a = rd_u32(ref, n, (long)(ry + dy) * cols + (rx + dx));
            const uint32_t b = rd_u32(cmp, n, (long)(py + dy) * cols + (px + dx));
            cost += (int)orc_popcount32(a ^ b);
        }
    return (int)(short)cost;
}

/* ------------------------------------------------------------------ a8 --
 * denseCensusStereo/denseCensusStereo.cpp:87-131 (epipolarMatching, 5x5) and
 * the driver loop :257-266.  For every pixel 2<=x<cols-2, 2<=y<rows-2 scan
 * x_r = x, x-1, ... while x_r >= 5 and x_r > x-maxDisp; cost = sum over the
 * 5x5 neighbourhood of per-pixel Hamming distances; strict '<' keeps the
 * first minimum => smallest disparity on ties.  disparity = float(x - argmin);
 * x < 5 never enters the loop => 0.  Pixels outside the loop bounds are not
 * written. */
ORC_API void orc_dense_wta5x5(const uint32_t *cl, const uint32_t *cr, int rows, int cols,
                              int maxDisp, float *disp) {
    for (int y = 2; y < rows - 2; ++y)
        for (int x = 2; x < cols - 2; ++x) {
            int minCost = INT_MAX, matchingX = x;
            for (int i = x; i >= 5 && i > x - maxDisp; --i) {
                int cost = 0;
                for (int m = -2; m <= 2; ++m) {
                    const uint32_t *l = cl + (long)(y + m) * cols;
                    const uint32_t *r = cr + (long)(y + m) * cols;
                    for (int n = -2; n <= 2; ++n) cost += (int)orc_popcount32(l[x + n] ^ r[i + n]);
                }
                if (cost < minCost) { minCost = cost; matchingX = i; }
            }
            disp[(long)y * cols + x] = (float)(x - matchingX);
        }
}

/* Row-range variant used to time a bounded sample of a8 on the host and to
 * split the work over threads; identical arithmetic. */
ORC_API void orc_dense_wta5x5_rows(const uint32_t *cl, const uint32_t *cr, int rows, int cols,
                                   int maxDisp, float *disp, int y0, int y1) {
    if (y0 < 2) y0 = 2;
    if (y1 > rows - 2) y1 = rows - 2;
    for (int y = y0; y < y1; ++y)
        for (int x = 2; x < cols - 2; ++x) {
            int minCost = INT_MAX, matchingX = x;
            for (int i = x; i >= 5 && i > x - maxDisp; --i) {
                int cost = 0;
                for (int m = -2; m <= 2; ++m) {
                    const uint32_t *l = cl + (long)(y + m) * cols;
                    const uint32_t *r = cr + (long)(y + m) * cols;
                    for (int n = -2; n <= 2; ++n)
                        cost += __builtin_popcount(l[x + n] ^ r[i + n]);
                }
                if (cost < minCost) { minCost = cost; matchingX = i; }
            }
            disp[(long)y * cols + x] = (float)(x - matchingX);
        }
}

/* ------------------------------------------------------------------ a9 --
 * highPerfStereo/highPerfStereoLib.cpp:88-139  epipolarMatching (A x A).
 * half = costAggrWindowSize/2, margin = censusSize/2.  Scan x_r = x downwards
 * while x_r >= half+margin and x_r > x-maxDisp.  Rows are clipped by
 * y+m < rows (loop end) and y+m < 0 (skip); columns only by x_r+n < cols
 * (loop end), so the LEFT index x+n is unclipped and addressed flat.  The
 * returned cost is the single-pixel Hamming distance at the winner (:136),
 * converted to `char`. */
ORC_API void orc_epipolar_match(const uint32_t *cl, const uint32_t *cr, int rows, int cols,
                                int censusSize, int costAggrWindowSize, int x, int y,
                                int maxDisp, float *out_disp, int8_t *out_cost) {
    const long n = (long)rows * cols;
    const int half = costAggrWindowSize / 2, margin = censusSize / 2;
    int minCost = INT_MAX, matchingX = x;
    for (int i = x; i >= half + margin && i > x - maxDisp; --i) {
        int cost = 0;
        for (int m = -half; m <= half && y + m < rows; ++m) {
            if (y + m < 0) continue;
            const long rowoff = (long)(y + m) * cols;
            for (int k = -half; k <= half && i + k < cols; ++k)
                cost += (int)orc_popcount32(rd_u32(cl, n, rowoff + x + k) ^ rd_u32(cr, n, rowoff + i + k));
        }
        if (cost < minCost) { minCost = cost; matchingX = i; }
    }
    *out_disp = (float)(x - matchingX);
    *out_cost = (int8_t)orc_popcount32(rd_u32(cl, n, (long)y * cols + x) ^ rd_u32(cr, n, (long)y * cols + matchingX));
}

/* ----------------------------------------------------------------- a10 --
 * highPerfStereo/highPerfStereoLib.cpp:20-41  costEvaluation.
 * For each point (x,y): xR = x - (int)floor(double(d)+0.5); xR < 2 => skip
 * (the caller's pre-fill survives); else cost = (char)hamming(cl(y,x),
 * cr(y,xR)).  xR has no upper bound check: the right word is addressed flat. */
ORC_API void orc_cost_eval(const uint32_t *cl, const uint32_t *cr, int rows, int cols,
                           const int32_t *pts_xy, int npts, const float *disp, int8_t *costs) {
    const long n = (long)rows * cols;
    for (int p = 0; p < npts; ++p) {
        const int x = pts_xy[2 * p], y = pts_xy[2 * p + 1];
        const float d = disp[(long)y * cols + x];
        const int xr = x - (int)floor((double)d + 0.5);
        if (xr < 2) continue;
        costs[(long)y * cols + x] =
            (int8_t)orc_popcount32(cl[(long)y * cols + x] ^ rd_u32(cr, n, (long)y * cols + xr));
    }
}

/* Banded plane-prior search (build-defined extension of a10; the reference evaluates one disparity per point,
 * highPerfStereoLib.cpp:20-41, and its only banded search around a plane prior is libelas' SAD matcher,
 * elas/elas.cpp:683-780, which is the model for the band).  d0 = (int)floor(double(prior)+0.5) is evaluated exactly as
 * orc_cost_eval does (admitted when x - d0 >= 2, no other bound).  The other candidates d0 + k, 0 < |k| <= radius, are
 * admitted when 0 <= d0 + k, d0 + k < maxDisp (when maxDisp > 0) and x - (d0 + k) >= 2.  Candidates are visited in
 * ascending disparity with a strict '<', so ties go to the smallest disparity.  No admitted candidate: the point is
 * skipped.  Otherwise costs(y,x) = the winner's single-pixel Hamming distance, and when the winner is not d0 the prior
 * becomes (float)winner; a winning d0 leaves the (fractional) prior as it is.  radius 0 is orc_cost_eval. */
ORC_API void orc_cost_eval_band(const uint32_t *cl, const uint32_t *cr, int rows, int cols, const int32_t *pts_xy, int npts,
                                float *disp, int8_t *costs, int radius, int maxDisp) {
    const long n = (long)rows * cols;
    for (int p = 0; p < npts; ++p) {
        const int x = pts_xy[2 * p], y = pts_xy[2 * p + 1];
        const long o = (long)y * cols + x;
        const int d0 = (int)floor((double)disp[o] + 0.5);
        int best = 1 << 30, best_d = 0, found = 0;
        for (int k = -radius; k <= radius; ++k) {
            const long d = (long)d0 + k;
            const long xr = (long)x - d;
            if (xr < 2) continue;
            if (k != 0 && (d < 0 || (maxDisp > 0 && d >= maxDisp))) continue;
            const int c = orc_popcount32(cl[o] ^ rd_u32(cr, n, (long)y * cols + xr));
            if (c < best) { best = c; best_d = (int)d; found = 1; }
        }
        if (!found) continue;
        costs[o] = (int8_t)best;
        if (best_d != d0) disp[o] = (float)best_d;
    }
}

/* ----------------------------------------------------------------- a11 --
 * highPerfStereo/highPerfStereoLib.cpp:44-86  disparityRefinement, with the
 * grid container of highPerfStereoLib.h:114-151.
 * grid is (rows/g + 1) x (cols/g + 1); every cell starts as
 * conf=(0,0,0.0f,tLow), inv=(0,0,tHigh).  Points are visited in the given
 * order; all compares are strict, so the first point wins ties.  The
 * finalCosts compare reads the stored byte as UNSIGNED (Mat_::ptr() without a
 * type yields uchar*, :70) while mc is a signed char.
 * Grid outputs are struct-of-arrays, cell index = j*gridW + i. */
ORC_API void orc_refine(const int32_t *pts_xy, int npts, const float *disp, const int8_t *costs,
                        int rows, int cols, int tLow, int tHigh, unsigned occGridSize,
                        float *finalDisp, int8_t *finalCosts,
                        int32_t *conf_x, int32_t *conf_y, float *conf_d, int8_t *conf_c,
                        int32_t *inv_x, int32_t *inv_y, int8_t *inv_c) {
    const unsigned gh = (unsigned)(rows / occGridSize) + 1, gw = (unsigned)(cols / occGridSize) + 1;
    for (unsigned c = 0; c < gh * gw; ++c) {
        conf_x[c] = 0; conf_y[c] = 0; conf_d[c] = 0.0f; conf_c[c] = (int8_t)tLow;
        inv_x[c] = 0; inv_y[c] = 0; inv_c[c] = (int8_t)tHigh;
    }
    const int8_t tl = (int8_t)tLow, th = (int8_t)tHigh;
    for (int p = 0; p < npts; ++p) {
        const int u = pts_xy[2 * p], v = pts_xy[2 * p + 1];
        const long o = (long)v * cols + u;
        const float d = disp[o];
        const int8_t mc = costs[o];
        const unsigned cell = (unsigned)(v / (int)occGridSize) * gw + (unsigned)(u / (int)occGridSize);
        if ((int)mc < (int)(uint8_t)finalCosts[o]) { finalDisp[o] = d; finalCosts[o] = mc; }
        if (mc < tl && mc < conf_c[cell]) { conf_x[cell] = u; conf_y[cell] = v; conf_d[cell] = d; conf_c[cell] = mc; }
        if (mc > th && mc > inv_c[cell]) { inv_x[cell] = u; inv_y[cell] = v; inv_c[cell] = mc; }
    }
}

/* ------------------------------------------------------------ a5 + a6 ---
 * sparsestereo/sparsestereo-inl.h:37-81 (match), :83-140 (calcCosts),
 * :142-174 (getRowOffsets), :176-227 (denseConsistencyCheck), with rect==NULL:
 * sparserectification.cpp:37-45,61-69 (copy + sort by (y,x),
 * sparserectification.h:35-38), :106-108 (identity epiline-start table stored
 * as short) and the dummy Epiline of stereorectification.h:40-50.
 *
 * Inputs are the caller's keypoint coordinates (float x,y).  Outputs, one per
 * LEFT feature in (y,x)-sorted order:
 *   out_left[i]  original index of the i-th sorted left feature,
 *   out_right[i] original index of the matched right feature, or -1,
 *   out_cost[i]  window cost of the match.
 * Ties in the sort (identical coordinates) are broken by original index.
 * A fresh SparseStereo instance is modelled: features skipped by the `ly < 2`
 * continue (:99-101) keep the value-initialised pair (0,0), i.e. they come out
 * matched to sorted right feature 0 with cost 0, exactly as the reference
 * would emit them. */
typedef struct { float x, y; int idx; } orc_feat;
static int feat_cmp(const void *a, const void *b) {
    const orc_feat *p = (const orc_feat *)a, *q = (const orc_feat *)b;
    if (p->y < q->y) return -1;
    if (p->y > q->y) return 1;
    if (p->x < q->x) return -1;
    if (p->x > q->x) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx);
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

ORC_API int orc_sparse_match(const uint32_t *cl, const uint32_t *cr, int rows, int cols,
                             const float *left_xy, int nl, const float *right_xy, int nr,
                             int maxDisparity, float yTolerance, float uniqueness, int leftRightStep,
                             int32_t *out_left, int32_t *out_right, int16_t *out_cost) {
    if (nl == 0 || nr == 0) return 0;            /* :44-45 */
    orc_feat *L = (orc_feat *)malloc(sizeof(orc_feat) * (size_t)nl);
    orc_feat *R = (orc_feat *)malloc(sizeof(orc_feat) * (size_t)nr);
    unsigned *offsets = (unsigned *)malloc(sizeof(unsigned) * (size_t)rows);
    int *mfirst = (int *)calloc((size_t)nl, sizeof(int));
    short *msecond = (short *)calloc((size_t)nl, sizeof(short));
    for (int i = 0; i < nl; ++i) { L[i].x = left_xy[2 * i]; L[i].y = left_xy[2 * i + 1]; L[i].idx = i; }
    for (int i = 0; i < nr; ++i) { R[i].x = right_xy[2 * i]; R[i].y = right_xy[2 * i + 1]; R[i].idx = i; }
    qsort(L, (size_t)nl, sizeof(orc_feat), feat_cmp);
    qsort(R, (size_t)nr, sizeof(orc_feat), feat_cmp);

    /* getRowOffsets :142-174 */
    {
        int lastRow = -1;
        for (int i = 0; i < nr; ++i) {
            const int y = (int)(R[i].y + 0.5);
            if (y >= rows - 1) break;
            else if (y < 0) continue;
            if (y != lastRow) {
                for (int j = lastRow + 1; j <= y; ++j) offsets[j] = (unsigned)i;
                lastRow = y;
            }
        }
        for (int i = lastRow + 1; i < rows; ++i) offsets[i] = (unsigned)nr;
    }

    /* calcCosts :83-140 */
    for (int l = 0; l < nl; ++l) {
        const int ly = (int)(L[l].y + 0.5);
        if (ly < 2) continue;                                   /* :99-101 */
        if (ly >= rows - 2 - 1) {                               /* :102-107 */
            for (; l < nl; ++l) mfirst[l] = -1;
            break;
        }
        const int rightStart = (int)offsets[clampi((int)(ly - yTolerance), 0, rows - 1)];
        const int rightEnd = (int)offsets[clampi((int)(ly + yTolerance) + 2, 0, rows - 1)];
        short minCost = SHRT_MAX;
        int minRight = -1;
        const int rx = (int)(L[l].x + 0.5), ry = (int)(L[l].y + 0.5);
        for (int r = rightStart; r < rightEnd; ++r) {
            if (R[r].x <= L[l].x && R[r].x >= L[l].x - maxDisparity &&
                fabs(L[l].y - R[r].y) <= yTolerance && R[r].x >= 2 && R[r].x < cols - 2) {
                const short c = (short)orc_window_cost_5x6(cl, cr, rows, cols, rx, ry,
                                                           (int)(R[r].x + 0.5), (int)(R[r].y + 0.5));
                if (c < minCost) { minCost = c; minRight = r; }
            }
        }
        mfirst[l] = minRight;
        msecond[l] = minCost;
    }

    /* denseConsistencyCheck :176-227 (reference image = RIGHT census) */
    for (int l = 0; l < nl; ++l) {
        const int ly = (int)(L[l].y + 0.5);
        if (ly < 2) continue;
        if (ly >= rows - 2 - 1) break;
        const int r = mfirst[l];
        const short minCost = msecond[l];
        if (r == -1) continue;
        /* identity table lookup precompEpilinesStart(int(pt.y), clamp(int(rx+.5),0,cols)) :203-204;
         * column == cols addresses the next row's column 0, whose entry is 0. */
        int col = clampi((int)(R[r].x + 0.5), 0, cols);
        const float leftStartX = (float)(short)(col == cols ? 0 : col);
        const int startX = 2 > (int)leftStartX ? 2 : (int)leftStartX;
        const int endLim = (int)(leftStartX + maxDisparity);
        const int endX = cols - 2 < endLim ? cols - 2 : endLim;
        const int y = (int)(L[l].y + 0.5);                      /* dummy epiline: at(x) = pt.y */
        const int rpx = (int)(R[r].x + 0.5), rpy = (int)(R[r].y + 0.5);
        const int lxi = (int)(L[l].x + 0.5);
        for (int x = startX; x < endX; x += leftRightStep) {
            if (y < 2 || y >= rows - 2) continue;
            const short c = (short)orc_window_cost_5x6(cr, cl, rows, cols, rpx, rpy, x, y);
            if ((float)c < (float)minCost / uniqueness && fabs((double)(x - lxi)) > leftRightStep) {
                mfirst[l] = -1;
                break;
            }
        }
    }

    for (int l = 0; l < nl; ++l) {
        out_left[l] = L[l].idx;
        out_right[l] = mfirst[l] == -1 ? -1 : R[mfirst[l]].idx;
        out_cost[l] = mfirst[l] == -1 ? 0 : msecond[l];
    }
    free(L); free(R); free(offsets); free(mfirst); free(msecond);
    return nl;
}

/* ----------------------------------------------------------------- a12 --
 * Plane fit: plane/Plane.cpp:27-65 with linear-system-solving/
 * CramerTheorem.cpp:24-48 -- solve [x y 1][a b c]^T = d for three vertices by
 * Cramer's rule in double, results narrowed to float (Plane.h members are
 * float).  A singular system leaves the plane at (0,0,0) (Plane.cpp:15-19,
 * :48-53).  The cofactor expansion order follows Matrix::determinant
 * (linear-system-solving/Matrix.cpp:38-54): first-row expansion. */
static double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1])
         - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
         + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}
ORC_API void orc_plane_fit(const double xy_d[9], float abc[3]) {
    double A[3][3], M[3][3];
    for (int i = 0; i < 3; ++i) { A[i][0] = xy_d[3 * i]; A[i][1] = xy_d[3 * i + 1]; A[i][2] = 1.0; }
    const double dA = det3(A);
    abc[0] = abc[1] = abc[2] = 0.0f;
    if (dA == 0.0) return;
    for (int k = 0; k < 3; ++k) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M[i][j] = (j == k) ? xy_d[3 * i + 2] : A[i][j];
        abc[k] = (float)(det3(M) / dA);
    }
}

/* Prior interpolation: highPerfStereo/highPerfStereoLib.cpp:532-554 with
 * Plane::getDepth(Point2) plane/Plane.cpp:71-73:
 *   prior(j,i) = float( double(a)*double(i) + double(b)*double(j) + double(c) )
 * (float members promoted against Point2's double coordinates).  The
 * reference asks Fade2D which triangle holds the pixel; that library is not
 * available, so the point-in-triangle rule is build-defined ("parity
 * unpinned" for pixels lying exactly on shared edges): a pixel belongs to the
 * FIRST non-degenerate triangle in list order whose closed area contains it
 * (three edge functions all >= 0 or all <= 0, in exact 64-bit integers).
 * Pixels covered by no triangle keep their previous value. */
ORC_API void orc_plane_prior_raster(const int32_t *tri_xy /* ntri*6 */, const float *tri_abc /* ntri*3 */,
                                    int ntri, int rows, int cols, float *prior) {
    /* triangles in list order, each over its bounding box; a pixel already owned is skipped, which is
     * the same as asking every pixel for its first containing triangle */
    unsigned char *owned = (unsigned char *)calloc((size_t)rows * cols, 1);
    if (!owned) return;
    for (int t = 0; t < ntri; ++t) {
        const int32_t *v = tri_xy + 6 * t;
        const int64_t area2 = (int64_t)(v[2] - v[0]) * (v[5] - v[1]) - (int64_t)(v[3] - v[1]) * (v[4] - v[0]);
        if (area2 == 0) continue;   /* degenerate triangles own nothing */
        int x0 = v[0], x1 = v[0], y0 = v[1], y1 = v[1];
        for (int k = 1; k < 3; ++k) {
            if (v[2 * k] < x0) x0 = v[2 * k];
            if (v[2 * k] > x1) x1 = v[2 * k];
            if (v[2 * k + 1] < y0) y0 = v[2 * k + 1];
            if (v[2 * k + 1] > y1) y1 = v[2 * k + 1];
        }
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        if (x1 > cols - 1) x1 = cols - 1;
        if (y1 > rows - 1) y1 = rows - 1;
        const float *p = tri_abc + 3 * t;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                if (owned[(long)y * cols + x]) continue;
                const int64_t e0 = (int64_t)(v[2] - v[0]) * (y - v[1]) - (int64_t)(v[3] - v[1]) * (x - v[0]);
                const int64_t e1 = (int64_t)(v[4] - v[2]) * (y - v[3]) - (int64_t)(v[5] - v[3]) * (x - v[2]);
                const int64_t e2 = (int64_t)(v[0] - v[4]) * (y - v[5]) - (int64_t)(v[1] - v[5]) * (x - v[4]);
                if ((e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0)) {
                    owned[(long)y * cols + x] = 1;
                    prior[(long)y * cols + x] = (float)((double)p[0] * (double)x + (double)p[1] * (double)y + (double)p[2]);
                }
            }
    }
    free(owned);
}

/* ----------------------------------------------------------------- a13 --
 * highPerfStereo/highPerfStereoLib.cpp:141-175  supportResampling, without the
 * mesh: cells are visited row-major; an invalid match with cost > tHigh is
 * re-matched over the full range (a9) and becomes a support point when the
 * winner's single-pixel cost is < tLow; then the cell's confident support (cost
 * < tLow) becomes one.  Each new support writes its disparity into the prior
 * map and is appended to the output list (the order in which the reference
 * inserts them into the mesh).  Returns the number of points.  No known-answer
 * value of the reference covers this routine: parity unpinned beyond a9. */
ORC_API int orc_support_resampling(const uint32_t *cl, const uint32_t *cr, int rows, int cols,
                                   int censusSize, int costAggrWindowSize, int gridW, int gridH,
                                   const int32_t *conf_x, const int32_t *conf_y, const float *conf_d, const int8_t *conf_c,
                                   const int32_t *inv_x, const int32_t *inv_y, const int8_t *inv_c,
                                   int tLow, int tHigh, int maxDisp, float *disparities,
                                   int32_t *out_xy, float *out_d) {
    int n = 0;
    const int8_t tl = (int8_t)tLow, th = (int8_t)tHigh;
    for (int j = 0; j < gridH; ++j)
        for (int i = 0; i < gridW; ++i) {
            const int c = j * gridW + i;
            if (inv_c[c] > th) {
                float d; int8_t cost;
                orc_epipolar_match(cl, cr, rows, cols, censusSize, costAggrWindowSize, inv_x[c], inv_y[c], maxDisp, &d, &cost);
                if (cost < tl) {
                    disparities[(long)inv_y[c] * cols + inv_x[c]] = d;
                    out_xy[2 * n] = inv_x[c]; out_xy[2 * n + 1] = inv_y[c]; out_d[n] = d; ++n;
                }
            }
            if (conf_c[c] < tl) {
                disparities[(long)conf_y[c] * cols + conf_x[c]] = conf_d[c];
                out_xy[2 * n] = conf_x[c]; out_xy[2 * n + 1] = conf_y[c]; out_d[n] = conf_d[c]; ++n;
            }
        }
    return n;
}

/* ----------------------------------------------------------------- a14 --
 * denseCensusStereo/CensusTransform.cpp:9-67  CensusTransform (generic N x N).
 * maskSize = size made odd (:10-13); nWords = maskSize^2/32 + 1 (:17); a pixel
 * within maskSize/2 of any border gets all-zero words (:36-37).  Bit j*m+i
 * (+pass) is set when neighbour >= centre (:56) -- the opposite polarity of
 * a2, and the centre itself is compared, so its bit is set.  The skip test
 * (:47) compares the WINDOW index (i,j) with the IMAGE coordinate (u,v): it
 * fires only for pixels with u,v < m, skips that (non-centre) tap and shifts
 * every later bit down by one (pass = -1).  Restated exactly.
 * out holds nWords words per pixel, pixel-major.  The reference's callers hand
 * an unsigned image to the signed parameter (denseCensusStereo.cpp:307-308),
 * which OpenCV converts with saturation: orc_u8_saturate_s8 restates that. */
ORC_API int orc_census_nxn_words(int size) { const int m = (size % 2 == 0) ? size + 1 : size; return (m * m) / 32 + 1; }
ORC_API void orc_u8_saturate_s8(const uint8_t *src, int8_t *dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = (int8_t)(src[i] > 127 ? 127 : src[i]);
}
ORC_API void orc_census_nxn(const int8_t *in, int rows, int cols, int size, uint32_t *out) {
    const int m = (size % 2 == 0) ? size + 1 : size, h = m / 2;
    const int nw = (m * m) / 32 + 1;
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            uint32_t *w = out + ((long)v * cols + u) * nw;
            for (int k = 0; k < nw; ++k) w[k] = 0;
            if (u < h || v < h || u >= cols - h || v >= rows - h) continue;
            const int8_t ref = in[(long)v * cols + u];
            int pass = 0;
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < m; ++i) {
                    if (j == v && i == u) { pass = -1; continue; }
                    if (in[(long)(v - h + j) * cols + (u - h + i)] >= ref) {
                        const int bit = j * m + i + pass;
                        w[bit / 32] |= 1u << (bit % 32);
                    }
                }
        }
}

/* ===================== "next" rows: the steps either side of the path ===================== */

/* ------------------------------------------------------------------ f4 --
 * Pre-processing of highPerfStereo/highPerfStereoLib.cpp:245-280.  The
 * arithmetic lives in OpenCV (not vendored, no version pinned), restated here
 * from its documented behaviour: parity unpinned.
 *
 * equalizeHist (highPerfStereoLib.cpp:251-252): 256-bin histogram; i0 = first
 * non-empty bin; if it holds every pixel the image becomes i0; else
 * scale = 255.f/(total - hist[i0]), lut[i0] = 0 and for i > i0
 * lut[i] = saturate_u8(round_half_even(float(sum_{k=i0+1..i} hist[k]) * scale)). */
static int round_half_even_f(float v) { return (int)lrintf(v); }
static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
ORC_API void orc_equalize_hist(const uint8_t *src, uint8_t *dst, int rows, int cols) {
    const long n = (long)rows * cols;
    int hist[256] = {0};
    for (long i = 0; i < n; ++i) hist[src[i]]++;
    int i0 = 0;
    while (i0 < 256 && !hist[i0]) ++i0;
    if (i0 == 256) return;
    if (hist[i0] == n) { for (long i = 0; i < n; ++i) dst[i] = (uint8_t)i0; return; }
    uint8_t lut[256] = {0};
    const float scale = 255.f / (float)(n - hist[i0]);
    int sum = 0;
    for (int i = i0 + 1; i < 256; ++i) { sum += hist[i]; lut[i] = sat_u8(round_half_even_f((float)sum * scale)); }
    for (long i = 0; i < n; ++i) dst[i] = lut[src[i]];
}

/* Laplacian(ddepth = CV_16S, ksize 1 or 3, scale, delta, BORDER_REFLECT_101) followed by
 * convertScaleAbs (highPerfStereoLib.cpp:258-259): kernel {0,1,0,1,-4,1,0,1,0} for ksize 1,
 * {2,0,2,0,-8,0,2,0,2} for ksize 3; v = saturate_s16(round(sum*scale + delta)); out = saturate_u8(|v|). */
static int refl101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; }
ORC_API void orc_laplacian_abs(const uint8_t *src, uint8_t *dst, int rows, int cols, int ksize, double scale, double delta) {
    static const int k1[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0}, k3[9] = {2, 0, 2, 0, -8, 0, 2, 0, 2};
    const int *k = ksize == 1 ? k1 : k3;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int sum = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    sum += k[(dy + 1) * 3 + dx + 1] * (int)src[(long)refl101(y + dy, rows) * cols + refl101(x + dx, cols)];
            long v = lrint((double)sum * scale + delta);
            if (v < -32768) v = -32768;
            if (v > 32767) v = 32767;
            dst[(long)y * cols + x] = sat_u8((int)(v < 0 ? -v : v));
        }
}

/* High-gradient point set, highPerfStereoLib.cpp:269-280: raster order over 2<=y<rows-2,
 * 2<=x<cols-2, pixel > threshold.  Returns the count; pts_xy may be NULL to count only. */
ORC_API int orc_high_gradient_points(const uint8_t *abs_grad, int rows, int cols, int threshold, int32_t *pts_xy) {
    int n = 0;
    for (int y = 2; y < rows - 2; ++y)
        for (int x = 2; x < cols - 2; ++x)
            if ((int)abs_grad[(long)y * cols + x] > threshold) {
                if (pts_xy) { pts_xy[2 * n] = x; pts_xy[2 * n + 1] = y; }
                ++n;
            }
    return n;
}

/* Crop to multiples of 16 from the top-left corner (highPerfStereoLib.cpp:245-247). dst is
 * 16*(rows/16) x 16*(cols/16), densely packed. */
ORC_API void orc_crop16(const uint8_t *src, int rows, int cols, uint8_t *dst) {
    const int r2 = 16 * (rows / 16), c2 = 16 * (cols / 16);
    for (int y = 0; y < r2; ++y) memcpy(dst + (long)y * c2, src + (long)y * cols, (size_t)c2);
}

/* GaussianBlur(img, img, Size(k,k), 0, 0) (highPerfStereoLib.cpp:289-292) for k in {1,3,5}: with
 * sigma 0 OpenCV takes the binomial kernels {1}, {1,2,1}/4, {1,4,6,4,1}/16; all weights are
 * dyadic, so the 8-bit result is the exact 2-D weighted sum rounded half up; BORDER_REFLECT_101.
 * The reference passes blurSize = (int)applyBlur = 1 (highPerfStereoLib.cpp:207), i.e. a copy. */
ORC_API void orc_gaussian_blur(const uint8_t *src, uint8_t *dst, int rows, int cols, int ksize) {
    static const int w1[1] = {1}, w3[3] = {1, 2, 1}, w5[5] = {1, 4, 6, 4, 1};
    const int *w = ksize == 1 ? w1 : (ksize == 3 ? w3 : w5);
    const int h = ksize / 2, norm = ksize == 1 ? 1 : (ksize == 3 ? 16 : 256);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int sum = 0;
            for (int dy = -h; dy <= h; ++dy)
                for (int dx = -h; dx <= h; ++dx)
                    sum += w[dy + h] * w[dx + h] * (int)src[(long)refl101(y + dy, rows) * cols + refl101(x + dx, cols)];
            dst[(long)y * cols + x] = (uint8_t)((sum + norm / 2) / norm);
        }
}

/* Row-band inversion and line tracing on the copies handed to the feature detector
 * (highPerfStereoLib.cpp:300-335), in place.  Returns -1 where the reference would not terminate
 * (nb_lines > rows makes its step 0). */
ORC_API int orc_alter_rows(uint8_t *img, int rows, int cols, int invert_rows, int nb_rows, int trace_lines, int nb_lines, int line_size) {
    if (invert_rows) {
        if (nb_rows <= 0) return -1;
        const int row_size = rows / nb_rows + 1;
        const int nb = rows / row_size;
        for (int i = 0; i < nb - 1; ++i)
            if (i % 2 == 0)
                for (long k = (long)i * row_size * cols; k < (long)(i + 1) * row_size * cols; ++k) img[k] = (uint8_t)(255 - img[k]);
        if ((nb - 1) % 2 == 0)
            for (long k = (long)(nb - 1) * row_size * cols; k < (long)rows * cols; ++k) img[k] = (uint8_t)(255 - img[k]);
    }
    if (trace_lines) {
        if (nb_lines <= 0 || rows / nb_lines == 0) return -1;
        const int step = rows / nb_lines;
        for (int k = step / 2; k < rows; k += step)
            for (int j = k; j < k + line_size && j < rows; ++j) memset(img + (long)j * cols, 0, (size_t)cols);
    }
    return 0;
}

/* ------------------------------------------------------------------ f1 --
 * exFAST feature detector, sparsestereo/extendedfast.cpp:92-231 with the centre value of
 * extendedfast.h:18 and the sparse non-maximum suppression of fast9-inl.h:76-173.  The first
 * stage, cv::FAST(image, corners, minThreshold, false) (extendedfast.cpp:102), lives in OpenCV
 * (absent here): restated from its published definition (FAST-9 on the 16-pixel radius-3 circle,
 * strict comparisons against centre +- threshold, 3-pixel border, raster order): parity unpinned.
 *
 * cv::saturate_cast<char>(float) (extendedfast.cpp:164) has no specialisation for plain char, so the
 * generic template applies: a C++ float->char conversion, i.e. truncation toward zero, of which x86
 * keeps the low byte.  round_sat != 0 selects the other reading (round half even, saturate to
 * [-128,127]) for comparison. */
static int arc9(unsigned m) {                      /* extendedfast.cpp:42-71 with the >=9 test of :34 */
    int best = 0, start = -1, cur = 0;
    for (int b = 0; b < 16; ++b) {
        if (m & (1u << b)) ++cur;
        else { if (cur > best) best = cur; if (start < 0) start = cur; cur = 0; }
    }
    if (start >= 0) cur += start;
    if (cur > best) best = cur;
    return best >= 9;
}
static const int CIRC_DX[16] = {-1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2};   /* extendedfast.cpp:73-90 */
static const int CIRC_DY[16] = {-3, -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2};
static int sat_s8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
static int sat_u8i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static int fast9_min(const uint8_t *img, int cols, int x, int y, int t) {
    const int v = img[(long)y * cols + x];
    unsigned br = 0, dk = 0;
    for (int i = 0; i < 16; ++i) {
        const int p = img[(long)(y + CIRC_DY[i]) * cols + x + CIRC_DX[i]];
        if (p > v + t) br |= 1u << i;
        if (p < v - t) dk |= 1u << i;
    }
    return arc9(br) || arc9(dk);
}

/* extendedfast.cpp:153-175 and :177-231.  Returns 1 for a corner and stores its score. */
static int exfast_test(const uint8_t *img, int cols, int x, int y, float adaptivity, int round_sat, uint8_t *score) {
    int circ_u[16], circ_s[16], sum = 0, sad = 0;
    for (int i = 0; i < 16; ++i) {
        circ_u[i] = img[(long)(y + CIRC_DY[i]) * cols + x + CIRC_DX[i]];
        circ_s[i] = (int)(int8_t)(uint8_t)(circ_u[i] - 128);
        sum += circ_u[i];
    }
    const int avg = (int)(uint8_t)(sum / 16);
    for (int i = 0; i < 16; ++i) sad += abs(circ_u[i] - avg);
    const float tf = (float)sad * adaptivity / 16;
    int thr_u;
    if (round_sat) thr_u = (int)(uint8_t)(int8_t)sat_s8((int)lrintf(tf));
    else thr_u = (int)tf & 0xff;
    const int thr_s = (int)(int8_t)(uint8_t)thr_u;
    const long c = (long)y * cols + x;
    const int mean5 = ((int)img[c] + (int)img[c - cols] + (int)img[c + cols] + (int)img[c - 1] + (int)img[c + 1]) / 5;
    const int cen_s = (int)(int8_t)(uint8_t)(mean5 - 128);
    const int plus = sat_s8(cen_s + thr_s), minus = sat_s8(cen_s - thr_s);
    unsigned br = 0, dk = 0;
    for (int i = 0; i < 16; ++i) { if (circ_s[i] > plus) br |= 1u << i; if (minus > circ_s[i]) dk |= 1u << i; }
    if (!arc9(br) && !arc9(dk)) return 0;
    /* score: binary search between the threshold and 255 in the unsigned domain */
    const int cen_u = (int)(uint8_t)(cen_s + 128);
    int bmin = thr_u, bmax = 255, b = (bmax + bmin + 1) >> 1;
    for (;;) {
        const int cp = (int)(int8_t)(uint8_t)(sat_u8i(cen_u + b) - 128);
        unsigned m = 0;
        for (int i = 0; i < 16; ++i) if (circ_s[i] > cp) m |= 1u << i;
        if (arc9(m)) bmin = b;
        else {
            const int cm = (int)(int8_t)(uint8_t)(sat_u8i(cen_u - b) - 128);
            m = 0;
            for (int i = 0; i < 16; ++i) if (cm > circ_s[i]) m |= 1u << i;
            if (arc9(m)) bmin = b; else bmax = b;
        }
        if (bmin == bmax || bmin == bmax - 1) break;
        b = (bmax + bmin + 1) >> 1;
    }
    *score = (uint8_t)(bmin - thr_u);
    return 1;
}

/* Sparse non-maximum suppression over a raster-ordered corner list (fast9-inl.h:76-173): a corner
 * is dropped when a listed 8-neighbour has a score >= its own.  Two cursors walk the rows above and
 * below as the reference does.  keep[i] = 1 for survivors. */
static void sparse_nonmax(const int32_t *xy, const uint8_t *score, int n, uint8_t *keep) {
    if (n < 1) return;
    const int last_row = xy[2 * (n - 1) + 1];
    int *row_start = (int *)malloc(sizeof(int) * (size_t)(last_row + 1));
    for (int r = 0; r <= last_row; ++r) row_start[r] = -1;
    for (int i = 0, prev = -1; i < n; ++i) if (xy[2 * i + 1] != prev) { prev = xy[2 * i + 1]; row_start[prev] = i; }
    int above = 0, below = 0;
#define CX(i) xy[2 * (i)]
#define CY(i) xy[2 * (i) + 1]
    for (int i = 0; i < n; ++i) {
        const int px = CX(i), py = CY(i), s = score[i];
        keep[i] = 0;
        if (i > 0 && CX(i - 1) == px - 1 && CY(i - 1) == py && score[i - 1] >= s) continue;
        if (i < n - 1 && CX(i + 1) == px + 1 && CY(i + 1) == py && score[i + 1] >= s) continue;
        int hit = 0;
        if (py != 0 && row_start[py - 1] != -1) {
            if (CY(above) < py - 1) above = row_start[py - 1];
            while (CY(above) < py && CX(above) < px - 1) ++above;
            for (int j = above; CY(j) < py && CX(j) <= px + 1; ++j)
                if (CX(j) >= px - 1 && score[j] >= s) { hit = 1; break; }
            if (hit) continue;
        }
        if (py != last_row && row_start[py + 1] != -1 && below < n) {
            if (CY(below) < py + 1) below = row_start[py + 1];
            while (below < n && CY(below) == py + 1 && CX(below) < px - 1) ++below;
            for (int j = below; j < n && CY(j) == py + 1 && CX(j) <= px + 1; ++j)
                if (CX(j) >= px - 1 && score[j] >= s) { hit = 1; break; }
            if (hit) continue;
        }
        keep[i] = 1;
    }
#undef CX
#undef CY
    free(row_start);
}

/* Exported for the tests of the suppression rule itself. */
ORC_API void orc_sparse_nonmax(const int32_t *xy, const uint8_t *score, int n, uint8_t *keep) { sparse_nonmax(xy, score, n, keep); }
ORC_API int orc_arc9(unsigned mask16) { return arc9(mask16); }

/* detectImpl, extendedfast.cpp:92-130.  Key points come out in raster order with response = score
 * (nonmax) or -1000 (no nonmax); size 6 and angle -1 are constants of the reference and not stored.
 * Returns the number found; at most `capacity` are written. */
ORC_API int orc_exfast_detect(const uint8_t *img, int rows, int cols, int min_threshold, float adaptivity, int nonmax,
                              int round_sat, int32_t *kp_xy, float *kp_response, int capacity) {
    const long cap0 = (long)rows * cols;
    int32_t *xy = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(cap0 ? cap0 : 1));
    uint8_t *sc = (uint8_t *)malloc((size_t)(cap0 ? cap0 : 1));
    int n = 0;
    const int t = min_threshold < 0 ? 0 : (min_threshold > 255 ? 255 : min_threshold);
    for (int y = 3; y < rows - 3; ++y)
        for (int x = 3; x < cols - 3; ++x) {
            if (!fast9_min(img, cols, x, y, t)) continue;
            uint8_t s = 0;
            if (!exfast_test(img, cols, x, y, adaptivity, round_sat, &s)) continue;
            xy[2 * n] = x; xy[2 * n + 1] = y; sc[n] = s; ++n;
        }
    int out = 0;
    if (nonmax) {
        uint8_t *keep = (uint8_t *)malloc((size_t)(n ? n : 1));
        sparse_nonmax(xy, sc, n, keep);
        for (int i = 0; i < n; ++i)
            if (keep[i]) {
                if (out < capacity) { kp_xy[2 * out] = xy[2 * i]; kp_xy[2 * out + 1] = xy[2 * i + 1]; kp_response[out] = (float)sc[i]; }
                ++out;
            }
        free(keep);
    } else {
        for (int i = 0; i < n; ++i) {
            if (out < capacity) { kp_xy[2 * out] = xy[2 * i]; kp_xy[2 * out + 1] = xy[2 * i + 1]; kp_response[out] = -1000.f; }
            ++out;
        }
    }
    free(xy); free(sc);
    return out;
}

/* ------------------------------------------------------------------ f2 --
 * Support-point mesh.  The reference triangulates with Fade_2D (highPerfStereoLib.cpp:434-445,
 * :457), a closed library that is not part of the reference tree: there is no algorithm to restate
 * and no output to pin on, parity unpinned.  What can be checked is the definition of a Delaunay
 * triangulation, done here by brute force with exact integer predicates.
 *
 * orc_delaunay_check returns 0 when `tri` (index triples into pts) is a Delaunay triangulation of
 * the distinct positions in pts, else a positive code:
 *   1 index out of range / repeated vertex   2 degenerate triangle
 *   3 a point strictly inside a circumcircle 4 a directed edge used twice / unmatched inner edge
 *   5 triangle areas do not add up to the hull area   6 a point is no vertex although triangles exist
 */
typedef __int128 orc_i128;
static long long orient2(const int32_t *p, int a, int b, int c) {
    return (long long)(p[2 * b] - p[2 * a]) * (p[2 * c + 1] - p[2 * a + 1]) - (long long)(p[2 * b + 1] - p[2 * a + 1]) * (p[2 * c] - p[2 * a]);
}
static int incircle_sign(const int32_t *p, int a, int b, int c, int d) {   /* a,b,c counter-clockwise */
    const long long ax = p[2 * a] - p[2 * d], ay = p[2 * a + 1] - p[2 * d + 1], bx = p[2 * b] - p[2 * d], by = p[2 * b + 1] - p[2 * d + 1];
    const long long cx = p[2 * c] - p[2 * d], cy = p[2 * c + 1] - p[2 * d + 1];
    const orc_i128 a2 = (orc_i128)ax * ax + (orc_i128)ay * ay, b2 = (orc_i128)bx * bx + (orc_i128)by * by, c2 = (orc_i128)cx * cx + (orc_i128)cy * cy;
    const orc_i128 det = a2 * ((orc_i128)bx * cy - (orc_i128)by * cx) - b2 * ((orc_i128)ax * cy - (orc_i128)ay * cx) + c2 * ((orc_i128)ax * by - (orc_i128)ay * bx);
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
}
static const int32_t *g_sort_pts;
static int cmp_pt_idx(const void *a, const void *b) {
    const int i = *(const int *)a, j = *(const int *)b;
    if (g_sort_pts[2 * i] != g_sort_pts[2 * j]) return g_sort_pts[2 * i] < g_sort_pts[2 * j] ? -1 : 1;
    if (g_sort_pts[2 * i + 1] != g_sort_pts[2 * j + 1]) return g_sort_pts[2 * i + 1] < g_sort_pts[2 * j + 1] ? -1 : 1;
    return i < j ? -1 : (i > j);
}
static int cmp_edge(const void *a, const void *b) {
    const long long x = *(const long long *)a, y = *(const long long *)b;
    return x < y ? -1 : (x > y);
}
ORC_API int orc_delaunay_check(const int32_t *pts, int n, const int32_t *tri, int nt) {
    long long area2 = 0;
    /* canonical representative of each position: the first index holding it */
    int *order = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1)), *canon = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; ++i) order[i] = i;
    g_sort_pts = pts;
    qsort(order, (size_t)n, sizeof(int), cmp_pt_idx);
    for (int i = 0; i < n; ++i)
        canon[order[i]] = (i > 0 && pts[2 * order[i]] == pts[2 * order[i - 1]] && pts[2 * order[i] + 1] == pts[2 * order[i - 1] + 1]) ? canon[order[i - 1]] : order[i];
    int rc = 0;
    uint8_t *used = (uint8_t *)calloc((size_t)(n ? n : 1), 1);
    long long *edges = (long long *)malloc(sizeof(long long) * 3 * (size_t)(nt ? nt : 1));
    for (int t = 0; t < nt && !rc; ++t) {
        const int a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
        if (a < 0 || b < 0 || c < 0 || a >= n || b >= n || c >= n || a == b || b == c || a == c) { rc = 1; break; }
        const long long o = orient2(pts, a, b, c);
        if (o == 0) { rc = 2; break; }
        area2 += o < 0 ? -o : o;
        used[canon[a]] = used[canon[b]] = used[canon[c]] = 1;
        const int v[3] = {canon[a], o > 0 ? canon[b] : canon[c], o > 0 ? canon[c] : canon[b]};   /* counter-clockwise */
        for (int k = 0; k < 3; ++k) edges[3 * t + k] = (long long)v[k] * n + v[(k + 1) % 3];
        for (int d = 0; d < n; ++d)
            if (incircle_sign(pts, v[0], v[1], v[2], d) > 0) { rc = 3; break; }
    }
    /* hull of the distinct positions (monotone chain, collinear points dropped) and its doubled area */
    long long hull2 = 0;
    int n_hull_edges = 0;
    if (!rc) {
        int *h = (int *)malloc(sizeof(int) * (size_t)(2 * n + 2));
        int m = 0, nu = 0;
        int *u = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
        for (int i = 0; i < n; ++i) if (canon[order[i]] == order[i]) u[nu++] = order[i];
        for (int i = 0; i < nu; ++i) { while (m >= 2 && orient2(pts, h[m - 2], h[m - 1], u[i]) <= 0) --m; h[m++] = u[i]; }
        for (int i = nu - 2, lo = m + 1; i >= 0; --i) { while (m >= lo && orient2(pts, h[m - 2], h[m - 1], u[i]) <= 0) --m; h[m++] = u[i]; }
        if (m > 1) --m;
        for (int i = 1; i + 1 < m; ++i) hull2 += orient2(pts, h[0], h[i], h[i + 1]);
        if (nt > 0) for (int i = 0; i < nu; ++i) if (!used[u[i]]) rc = 6;
        if (!rc && hull2 != area2) rc = 5;
        if (!rc && nt == 0 && hull2 != 0) rc = 5;
        free(h); free(u);
    }
    if (!rc && nt > 0) {
        qsort(edges, (size_t)(3 * nt), sizeof(long long), cmp_edge);
        for (int i = 1; i < 3 * nt; ++i) if (edges[i] == edges[i - 1]) { rc = 4; break; }
        /* every directed edge without its reverse is a boundary edge; a triangulation of a convex region has
         * as many of those as the hull polygon has (possibly split) sides, and they must enclose hull2 */
        for (int i = 0; i < 3 * nt && !rc; ++i) {
            const long long a = edges[i] / n, b = edges[i] % n, rev = b * n + a;
            if (!bsearch(&rev, edges, (size_t)(3 * nt), sizeof(long long), cmp_edge)) ++n_hull_edges;
        }
        /* Euler: for a triangulated disc, 3*nt = 2*E - B and V - E + nt = 1 */
        int nv = 0;
        for (int i = 0; i < n; ++i) nv += used[i];
        const long long E = (3LL * nt + n_hull_edges) / 2;
        if (!rc && ((3LL * nt + n_hull_edges) % 2 != 0 || nv - E + nt != 1)) rc = 4;
    }
    free(order); free(canon); free(used); free(edges);
    return rc;
}

/* ------------------------------------------------------------------ f3 --
 * Point cloud, highPerfStereo/highPerfStereo.cpp:689-723.  cv::perspectiveTransform (OpenCV, absent)
 * restated from its documented formula for 3-channel double input and a 4x4 matrix:
 * w = m12 x + m13 y + m14 z + m15; if |w| > FLT_EPSILON the result is the first three rows times
 * 1/w, else 0.  Parity unpinned.  A point is kept when z > 0 and sqrt(x^2+y^2+z^2) < max_range
 * (:720); colours are read at the ROI-shifted position in r,g,b order (:715-718).
 * Returns the number kept. */
ORC_API int orc_point_cloud(const int32_t *pts_xy, int n, const float *disp, int cols, int roi_x, int roi_y, const double *q,
                            const uint8_t *bgr, int img_cols, double max_range, double *xyz, uint8_t *rgb) {
    int out = 0;
    for (int i = 0; i < n; ++i) {
        const int px = pts_xy[2 * i], py = pts_xy[2 * i + 1];
        const double x = px + roi_x, y = py + roi_y, z = (double)disp[(long)py * cols + px];
        double w = x * q[12] + y * q[13] + z * q[14] + q[15], X = 0, Y = 0, Z = 0;
        if (fabs(w) > (double)FLT_EPSILON) {
            w = 1. / w;
            X = (x * q[0] + y * q[1] + z * q[2] + q[3]) * w;
            Y = (x * q[4] + y * q[5] + z * q[6] + q[7]) * w;
            Z = (x * q[8] + y * q[9] + z * q[10] + q[11]) * w;
        }
        if (Z > 0 && sqrt(pow(X, 2) + pow(Y, 2) + pow(Z, 2)) < max_range) {
            xyz[3 * out] = X; xyz[3 * out + 1] = Y; xyz[3 * out + 2] = Z;
            if (rgb) {
                const uint8_t *c = bgr + 3 * ((long)(py + roi_y) * img_cols + px + roi_x);
                rgb[3 * out] = c[2]; rgb[3 * out + 1] = c[1]; rgb[3 * out + 2] = c[0];
            }
            ++out;
        }
    }
    return out;
}

/* SparseMatch::projectMatches, sparsestereo/sparsematch.h:35-50: Q converted to float, the 4x4 by 4x1 float product of
 * (rectLeft.x, rectLeft.y, disparity, 1) summed left to right in float (OpenCV's product, absent: parity unpinned), the
 * first three components divided by the fourth with no test for zero. */
ORC_API void orc_project_matches(const float *xyd, int n, const double *q, float *xyz) {
    float qf[16];
    for (int i = 0; i < 16; ++i) qf[i] = (float)q[i];
    for (int i = 0; i < n; ++i) {
        const float v[4] = {xyd[3 * i], xyd[3 * i + 1], xyd[3 * i + 2], 1.0f};
        float p[4];
        for (int r = 0; r < 4; ++r) p[r] = qf[4 * r] * v[0] + qf[4 * r + 1] * v[1] + qf[4 * r + 2] * v[2] + qf[4 * r + 3] * v[3];
        xyz[3 * i] = p[0] / p[3]; xyz[3 * i + 1] = p[1] / p[3]; xyz[3 * i + 2] = p[2] / p[3];
    }
}

/* Stereo against Kinect depth, highPerfStereo/PerformanceEvaluator.cpp:132-328 (called from
 * highPerfStereo/highPerfStereo.cpp:792).  OpenCV and Boost.Accumulators are absent; their parts are
 * restated from their documented behaviour, so parity is unpinned:
 *  - the considered pixels become Point3f (x+xOffset, y+yOffset, disparity) and go through
 *    cv::perspectiveTransform with Q: double arithmetic, float result (:132-147);
 *  - the 4x4 float matrix [R|T] (the double R and T converted to float, :53-59) times (p,1) in float, summed left to
 *    right, gives the point in the Kinect frame (:183-200);
 *  - cv::projectPoints without distortion (the member kinectDistortion is never assigned, :16-27,:109):
 *    X = R p + T in double, u = fx X/Z + cx, v = fy Y/Z + cy, float result.  The reference hands the rotation over as a
 *    Rodrigues vector and OpenCV rebuilds the matrix from it; here R itself is used;
 *  - a point counts when Rect(0,0,640,480) contains the pixel rounded to nearest even (Point2f -> Point), and
 *    0 < z <= 5000 in the Kinect frame (:224); its Kinect partner is the 3-D point of the pixel TRUNCATED to int (:226,
 *    :202-204), which must have z > 0 (:228); the bin is the raw depth at the ROUNDED pixel (:236-239);
 *  - Kinect raw depth to millimetres: 1000/(raw * -0.0030711016 + 3.3309495161) as float for raw < 2047, else 0 (:176-181);
 *    back-projection with the constants of :161-174;
 *  - errors: stereo minus Kinect per axis in float, distance = sqrt of the sum of squares in double, as float (:231-234);
 *  - per bin and over all bins (visited in bin order, each bin in insertion order): Boost's lazy variance on float:
 *    sum and sum of squares accumulate in float, mean = sum / n, variance = sumsq / n - mean * mean, sigma = sqrt;
 *    an empty set gives NaN (:258-310).  The relative error list holds mean distance error / millimetres of the bin for
 *    every non-empty bin but 2047 (:292-294).
 * bin_mean / bin_sigma: [4][2048] in the order x, y, z, distance; totals: mean x, y, z, distance, sigma x, y, z, distance,
 * mean and sigma of the relative errors.  corr_bin / corr_err (4 per correspondence) list the correspondences in input
 * order.  Returns their number, or -1 when the depth image is smaller than 640x480 or holds a value outside [0, 2048). */
static float kinect_mm(int raw) {
    if (raw < 2047) return (float)(1000 * (1.0 / ((double)raw * -0.0030711016 + 3.3309495161)));
    return 0.0f;
}
typedef struct { float sum, sumsq; size_t n; } lazy_var;
static void lv_add(lazy_var *a, float v) { a->sum += v; a->sumsq += v * v; a->n += 1; }
static float lv_mean(const lazy_var *a) { return a->sum / (float)a->n; }
static float lv_var(const lazy_var *a) { const float m = lv_mean(a); return a->sumsq / (float)a->n - m * m; }

ORC_API int orc_kinect_evaluate(const int32_t *raw_depth, int depth_rows, int depth_cols, const float *disp, int cols,
                                const int32_t *pts_xy, int n, int x_off, int y_off, const double *q, const double *cam,
                                const double *r, const double *t, int32_t *corr_bin, float *corr_err, float *bin_mean,
                                float *bin_sigma, int32_t *bin_count, float *totals) {
    if (depth_rows < 480 || depth_cols < 640) return -1;
    for (long i = 0; i < (long)depth_rows * depth_cols; ++i)
        if (raw_depth[i] < 0 || raw_depth[i] >= 2048) return -1;
    const double fx_d = 1.0 / 5.9421434211923247e+02, fy_d = 1.0 / 5.9104053696870778e+02;
    const double cx_d = 3.3930780975300314e+02, cy_d = 2.4273913761751615e+02;
    float m[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) m[4 * i + j] = (float)r[3 * i + j];
        m[4 * i + 3] = (float)t[i];
    }
    int nc = 0;
    for (int i = 0; i < n; ++i) {
        const int px = pts_xy[2 * i], py = pts_xy[2 * i + 1];
        const float fx = (float)(px + x_off), fy = (float)(py + y_off), fd = disp[(long)py * cols + px];
        const double x = fx, y = fy, z = fd;
        double w = x * q[12] + y * q[13] + z * q[14] + q[15];
        float sx = 0, sy = 0, sz = 0;
        if (fabs(w) > (double)FLT_EPSILON) {
            w = 1. / w;
            sx = (float)((x * q[0] + y * q[1] + z * q[2] + q[3]) * w);
            sy = (float)((x * q[4] + y * q[5] + z * q[6] + q[7]) * w);
            sz = (float)((x * q[8] + y * q[9] + z * q[10] + q[11]) * w);
        }
        /* Kinect frame, float */
        const float kx = m[0] * sx + m[1] * sy + m[2] * sz + m[3] * 1.0f;
        const float ky = m[4] * sx + m[5] * sy + m[6] * sz + m[7] * 1.0f;
        const float kz = m[8] * sx + m[9] * sy + m[10] * sz + m[11] * 1.0f;
        /* Kinect image, double */
        const double X = r[0] * sx + r[1] * sy + r[2] * sz + t[0];
        const double Y = r[3] * sx + r[4] * sy + r[5] * sz + t[1];
        const double Z = r[6] * sx + r[7] * sy + r[8] * sz + t[2];
        const double iz = Z != 0 ? 1. / Z : 1.;
        const float u = (float)(X * iz * cam[0] + cam[2]), v = (float)(Y * iz * cam[4] + cam[5]);
        if (!(u > -1e6f && u < 1e6f && v > -1e6f && v < 1e6f)) continue;    /* far outside the image (or NaN) */
        const int ur = round_half_even_f(u), vr = round_half_even_f(v);
        if (!(ur >= 0 && ur < 640 && vr >= 0 && vr < 480 && kz > 0 && kz <= 5000)) continue;
        const int ut = (int)u, vt = (int)v;
        const float mm = kinect_mm(raw_depth[(long)vt * depth_cols + ut]);
        const float qx = (float)((ut - cx_d) * (double)mm * fx_d), qy = (float)((vt - cy_d) * (double)mm * fy_d), qz = mm;
        if (!(qz > 0)) continue;
        const float ex = kx - qx, ey = ky - qy, ez = kz - qz;
        const float ed = (float)sqrt((double)ex * ex + (double)ey * ey + (double)ez * ez);
        corr_bin[nc] = raw_depth[(long)vr * depth_cols + ur];
        corr_err[4 * nc] = ex; corr_err[4 * nc + 1] = ey; corr_err[4 * nc + 2] = ez; corr_err[4 * nc + 3] = ed;
        ++nc;
    }
    /* statistics: bins in order, each in insertion order */
    int *start = (int *)calloc(2049, sizeof(int)), *order = (int *)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
    for (int i = 0; i < nc; ++i) start[corr_bin[i] + 1]++;
    for (int b = 0; b < 2048; ++b) { bin_count[b] = start[b + 1]; start[b + 1] += start[b]; }
    int *fill = (int *)malloc(sizeof(int) * 2048);
    memcpy(fill, start, sizeof(int) * 2048);
    for (int i = 0; i < nc; ++i) order[fill[corr_bin[i]]++] = i;
    lazy_var total[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, rel = {0, 0, 0};
    for (int b = 0; b < 2048; ++b) {
        lazy_var acc[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int k = 0; k < 4; ++k) {
            for (int j = start[b]; j < start[b + 1]; ++j) { lv_add(&acc[k], corr_err[4 * order[j] + k]); lv_add(&total[k], corr_err[4 * order[j] + k]); }
            bin_mean[k * 2048 + b] = lv_mean(&acc[k]);
            bin_sigma[k * 2048 + b] = sqrtf(lv_var(&acc[k]));
        }
        if (b != 2047 && bin_count[b] > 0) lv_add(&rel, bin_mean[3 * 2048 + b] / kinect_mm(b));
    }
    for (int k = 0; k < 4; ++k) { totals[k] = lv_mean(&total[k]); totals[4 + k] = sqrtf(lv_var(&total[k])); }
    totals[8] = lv_mean(&rel); totals[9] = sqrtf(lv_var(&rel));
    free(start); free(order); free(fill);
    return nc;
}

/* ------------------------------------------------------- 64-bit descriptors --
 * Build-defined generalisations for the 9x7 / left-right-check configuration.  The reference has
 * no 9x7 census and no dense left-right check (its generic census is square,
 * denseCensusStereo/CensusTransform.cpp:33-67; the only dense consistency check is libelas',
 * elas/elas.cpp:909-979): nothing to pin on, these functions ARE the definition.
 *
 * orc_census9x7: pixel (x,y), 4 <= x < cols-4, 3 <= y < rows-3: neighbour (dy,dx), dy=-3..3,
 * dx=-4..4, index (dy+3)*9+(dx+4), bit 62-index set when centre > neighbour; out[2p] = bits 0..31,
 * out[2p+1] = bits 32..62; every other pixel 0. */
ORC_API void orc_census9x7(const uint8_t *img, int rows, int cols, uint32_t *out) {
    memset(out, 0, sizeof(uint32_t) * 2 * (size_t)rows * cols);
    for (int y = 3; y < rows - 3; ++y)
        for (int x = 4; x < cols - 4; ++x) {
            uint64_t w = 0;
            const int c = img[(long)y * cols + x];
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -4; dx <= 4; ++dx) {
                    if (!dy && !dx) continue;
                    if (c > (int)img[(long)(y + dy) * cols + x + dx]) w |= 1ull << (62 - ((dy + 3) * 9 + (dx + 4)));
                }
            out[2 * ((long)y * cols + x)] = (uint32_t)w;
            out[2 * ((long)y * cols + x) + 1] = (uint32_t)(w >> 32);
        }
}

/* The search of orc_dense_wta5x5 (denseCensusStereo.cpp:87-131) on descriptors of `words` words,
 * whole map written (border 0), rows y0 <= y < y1 only (for timing a bounded sample).
 * view 0: left image's map.  view 1: right image's map: for (xr,y) scan xl = xr, xr+1, ... while
 * xl <= cols-6 and xl < xr+maxDisp; first minimum wins; disparity = xl - xr. */
ORC_API void orc_dense_wta_words(const uint32_t *cl, const uint32_t *cr, int rows, int cols, int words, int maxDisp, int view,
                                 float *disp, int y0, int y1) {
    for (int y = y0 < 0 ? 0 : y0; y < (y1 > rows ? rows : y1); ++y)
        for (int x = 0; x < cols; ++x) {
            float d = 0.f;
            if (y >= 2 && y < rows - 2 && x >= 2 && x < cols - 2) {
                int minCost = INT_MAX, best = x;
                for (int k = 0; k < maxDisp; ++k) {
                    const int xl = view ? x + k : x, xr = view ? x : x - k;
                    if (view ? xl > cols - 6 : xr < 5) break;
                    int cost = 0;
                    for (int m = -2; m <= 2; ++m)
                        for (int n = -2; n <= 2; ++n)
                            for (int w = 0; w < words; ++w)
                                cost += __builtin_popcount(cl[((long)(y + m) * cols + xl + n) * words + w] ^ cr[((long)(y + m) * cols + xr + n) * words + w]);
                    if (cost < minCost) { minCost = cost; best = k; }
                }
                d = minCost == INT_MAX ? 0.f : (float)best;
            }
            disp[(long)y * cols + x] = d;
        }
}

/* Left-right check: 2 <= x < cols-2, 2 <= y < rows-2: d = dl(y,x), xr = x - (int)d; dl(y,x) = invalid when
 * xr < 2 or |dr(y,xr) - d| > threshold. */
ORC_API void orc_lr_check(float *dl, const float *dr, int rows, int cols, float threshold, float invalid) {
    for (int y = 2; y < rows - 2; ++y)
        for (int x = 2; x < cols - 2; ++x) {
            const float d = dl[(long)y * cols + x];
            /* a value that is no disparity (negative, e.g. an earlier pass's invalid marker, NaN, beyond the row) is invalid and reads nothing */
            if (!(d >= 0.0f) || !(d < (float)cols)) { dl[(long)y * cols + x] = invalid; continue; }
            const int xr = x - (int)d;
            if (xr < 2 || fabsf(dr[(long)y * cols + xr] - d) > threshold) dl[(long)y * cols + x] = invalid;
        }
}

/* Sub-pixel refinement of a left-view disparity map (build-defined: the reference has no sub-pixel step
 * for census matches; its only fractional disparities are the plane-interpolated floats of
 * highPerfStereoLib.cpp:69-73).  For 2 <= x < cols-2, 2 <= y < rows-2 with an integer disparity d >= 1 whose
 * neighbours d-1 and d+1 are both admissible for the search of orc_dense_wta_words (d+1 <= maxDisp-1 and
 * x-(d+1) >= 5): with c(k) the 5x5-aggregated Hamming cost at disparity k,
 *   den = c(d-1) - 2 c(d) + c(d+1);   den > 0:  disp = (float)d + (float)(c(d-1) - c(d+1)) / (float)(2 den)
 * (one float division, one float addition); otherwise, and for every other pixel (negative = invalidated
 * values included), the value is left as it is.  Values that are not integers are left alone too. */
ORC_API void orc_subpixel_refine(const uint32_t *cl, const uint32_t *cr, int rows, int cols, int words, int maxDisp, float *disp) {
    for (int y = 2; y < rows - 2; ++y)
        for (int x = 2; x < cols - 2; ++x) {
            const float f = disp[(long)y * cols + x];
            if (!(f >= 1.0f) || !(f < (float)maxDisp)) continue;          /* also NaN and values no int can hold */
            const int d = (int)f;
            if ((float)d != f || d + 1 > maxDisp - 1 || x - (d + 1) < 5) continue;
            int c[3];
            for (int k = 0; k < 3; ++k) {
                const int xr = x - (d - 1 + k);
                int cost = 0;
                for (int m = -2; m <= 2; ++m)
                    for (int n = -2; n <= 2; ++n)
                        for (int w = 0; w < words; ++w)
                            cost += __builtin_popcount(cl[((long)(y + m) * cols + x + n) * words + w] ^ cr[((long)(y + m) * cols + xr + n) * words + w]);
                c[k] = cost;
            }
            const int den = c[0] - 2 * c[1] + c[2];
            if (den > 0) disp[(long)y * cols + x] = (float)d + (float)(c[0] - c[2]) / (float)(2 * den);
        }
}
