/*
 * svx_cpu_simd.c -- the dense census matcher on the host in the reference's own style of optimisation (SIMD
 * intrinsics + OpenMP; the reference uses SSE in sparsestereo/census.cpp:17-94 and OpenMP sections in
 * denseCensusStereo/denseCensusStereo.cpp:237-249), as the CPU baseline that bench.py times beside the GPU.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY, like svx_oracle.c: nothing in the product (libsvx.so) links, loads or calls
 * it.  Its results are checked against the scalar oracle (tests/test_oracle_simd.py), which is what pins it.
 *
 * Same definitions as orc_census9x7 / orc_census5x5_sse24 (interior), orc_dense_wta_words (both views) and
 * orc_lr_check of svx_oracle.c; the arithmetic is organised differently: per disparity the Hamming distances of a
 * census row are formed 4 or 8 pixels at a time (AVX2 nibble-table popcount), kept in a ring of five rows, summed into
 * running column sums and box-filtered, so that every cost value is formed once and serves both views.
 */
#include <immintrin.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* 9x7 census (definition: orc_census9x7), rows in parallel. */
ORC_API void simd_census9x7(const uint8_t *img, int rows, int cols, uint32_t *out, int n_threads) {
    memset(out, 0, sizeof(uint32_t) * 2 * (size_t)rows * cols);
    if (rows < 7 || cols < 9) return;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int y = 3; y < rows - 3; ++y) {
        uint64_t *w = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)cols);
        memset(w, 0, sizeof(uint64_t) * (size_t)cols);
        const uint8_t *c = img + (long)y * cols;
        for (int dy = -3; dy <= 3; ++dy)
            for (int dx = -4; dx <= 4; ++dx) {
                if (!dy && !dx) continue;
                const int bit = 62 - ((dy + 3) * 9 + (dx + 4));
                const uint8_t *q = img + (long)(y + dy) * cols + dx;
                for (int x = 4; x < cols - 4; ++x) w[x] |= (uint64_t)(c[x] > q[x]) << bit;
            }
        memcpy(out + 2 * ((long)y * cols + 4), w + 4, sizeof(uint64_t) * (size_t)(cols - 8));
        free(w);
    }
}

/* h[i] = popcount(l[i] ^ r[i]) for n_px descriptors of `words` 32-bit words (AVX2: nibble table + byte sums, four 64-bit descriptors at a time). */
static inline int popcnt_row_u16(const uint8_t *l, const uint8_t *r, int n_px, int words, uint16_t *h) {
    const __m256i lut = _mm256_setr_epi8(0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4, 0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4);
    const __m256i low = _mm256_set1_epi8(0x0f);
    int i = 0;
    if (words == 2) {
        for (; i + 4 <= n_px; i += 4) {
            const __m256i v = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)(l + 8 * (size_t)i)), _mm256_loadu_si256((const __m256i *)(r + 8 * (size_t)i)));
            const __m256i c = _mm256_add_epi8(_mm256_shuffle_epi8(lut, _mm256_and_si256(v, low)),
                                              _mm256_shuffle_epi8(lut, _mm256_and_si256(_mm256_srli_epi16(v, 4), low)));
            const __m256i s = _mm256_sad_epu8(c, _mm256_setzero_si256());          /* four 64-bit sums */
            h[i] = (uint16_t)_mm256_extract_epi16(s, 0);
            h[i + 1] = (uint16_t)_mm256_extract_epi16(s, 4);
            h[i + 2] = (uint16_t)_mm256_extract_epi16(s, 8);
            h[i + 3] = (uint16_t)_mm256_extract_epi16(s, 12);
        }
        for (; i < n_px; ++i) {
            uint64_t a, b;
            memcpy(&a, l + 8 * (size_t)i, 8);
            memcpy(&b, r + 8 * (size_t)i, 8);
            h[i] = (uint16_t)__builtin_popcountll(a ^ b);
        }
    } else {
        for (; i < n_px; ++i) {
            uint32_t a, b;
            memcpy(&a, l + 4 * (size_t)i, 4);
            memcpy(&b, r + 4 * (size_t)i, 4);
            h[i] = (uint16_t)__builtin_popcount(a ^ b);
        }
    }
    return n_px;
}

/* Both disparity maps of orc_dense_wta_words (view 0 into disp_left, view 1 into disp_right, either may be NULL) for
 * output rows y0 <= y < y1, bands of rows in parallel.  Every cost value cost(x, d) = 5x5-aggregated Hamming distance
 * of left pixel x and right pixel x - d is formed once: it competes for left pixel x (admitted when x - d >= 5) and for
 * right pixel x - d (admitted when x <= cols - 6); disparities are visited in ascending order with a strict '<', which
 * is the first minimum of either view's own scan. */
ORC_API void simd_dense_wta_words(const uint32_t *cl, const uint32_t *cr, int rows, int cols, int words, int maxDisp,
                                  float *disp_left, float *disp_right, int y0, int y1, int n_threads) {
    if (y0 < 0) y0 = 0;
    if (y1 > rows) y1 = rows;
    for (int y = y0; y < y1; ++y) {                         /* border pixels and pixels without a candidate are 0 */
        if (disp_left) memset(disp_left + (long)y * cols, 0, sizeof(float) * (size_t)cols);
        if (disp_right) memset(disp_right + (long)y * cols, 0, sizeof(float) * (size_t)cols);
    }
    const int ya = y0 < 2 ? 2 : y0, yb = y1 > rows - 2 ? rows - 2 : y1;
    if (yb <= ya || cols < 5) return;
    const int band = 32;
    const int n_bands = (yb - ya + band - 1) / band;
    const size_t px_bytes = 4 * (size_t)words;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int bi = 0; bi < n_bands; ++bi) {
        const int b0 = ya + bi * band, b1 = b0 + band < yb ? b0 + band : yb, nb = b1 - b0;
        uint16_t *best_l = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)nb * cols), *best_r = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)nb * cols);
        uint16_t *arg_l = (uint16_t *)calloc((size_t)nb * cols, sizeof(uint16_t)), *arg_r = (uint16_t *)calloc((size_t)nb * cols, sizeof(uint16_t));
        uint16_t *ring = (uint16_t *)malloc(sizeof(uint16_t) * 5 * (size_t)cols), *colsum = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)cols);
        uint16_t *cost = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)cols);
        memset(best_l, 0xff, sizeof(uint16_t) * (size_t)nb * cols);
        memset(best_r, 0xff, sizeof(uint16_t) * (size_t)nb * cols);
        for (int d = 0; d < maxDisp && d <= cols - 5; ++d) {
            /* pairs (x, x - d) exist for x >= d; window columns x-2 .. x+2 need x - 2 >= d */
            const int x_first = d, n_px = cols - d;
            memset(colsum, 0, sizeof(uint16_t) * (size_t)cols);
            for (int r = b0 - 2; r < b1 + 2; ++r) {          /* census row entering the window of output row r - 2 */
                uint16_t *h = ring + (size_t)((r - (b0 - 2)) % 5) * cols;
                if (r - (b0 - 2) >= 5)                        /* the row that leaves: h holds it */
                    for (int x = x_first; x < cols; ++x) colsum[x] = (uint16_t)(colsum[x] - h[x]);
                popcnt_row_u16((const uint8_t *)cl + ((size_t)r * cols + x_first) * px_bytes, (const uint8_t *)cr + ((size_t)r * cols) * px_bytes, n_px, words, h + x_first);
                for (int x = x_first; x < cols; ++x) colsum[x] = (uint16_t)(colsum[x] + h[x]);
                const int y = r - 2;
                if (y < b0) continue;
                const int xs = d + 2 > 2 ? d + 2 : 2, xe = cols - 2;   /* left pixels whose whole window has pairs */
                if (xs >= xe) continue;
                for (int x = xs; x < xe; ++x) cost[x] = (uint16_t)(colsum[x - 2] + colsum[x - 1] + colsum[x] + colsum[x + 1] + colsum[x + 2]);
                uint16_t *bl = best_l + (size_t)(y - b0) * cols, *al = arg_l + (size_t)(y - b0) * cols;
                uint16_t *br = best_r + (size_t)(y - b0) * cols - d, *ar = arg_r + (size_t)(y - b0) * cols - d;   /* indexed by the left column */
                const int xl0 = d + 5 > xs ? d + 5 : xs;      /* left view: right pixel x - d >= 5 */
                for (int x = xl0; x < xe; ++x)
                    if (cost[x] < bl[x]) { bl[x] = cost[x]; al[x] = (uint16_t)d; }
                const int xr1 = cols - 5 < xe ? cols - 5 : xe; /* right view: left pixel x <= cols - 6, right pixel x - d >= 2 */
                for (int x = xs; x < xr1; ++x)
                    if (cost[x] < br[x]) { br[x] = cost[x]; ar[x] = (uint16_t)d; }
            }
        }
        for (int y = b0; y < b1; ++y)
            for (int x = 2; x < cols - 2; ++x) {
                if (disp_left) disp_left[(long)y * cols + x] = (float)arg_l[(size_t)(y - b0) * cols + x];
                if (disp_right) disp_right[(long)y * cols + x] = (float)arg_r[(size_t)(y - b0) * cols + x];
            }
        free(best_l); free(best_r); free(arg_l); free(arg_r); free(ring); free(colsum); free(cost);
    }
}

/* orc_lr_check, rows in parallel. */
ORC_API void simd_lr_check(float *dl, const float *dr, int rows, int cols, float threshold, float invalid, int n_threads) {
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int y = 2; y < rows - 2; ++y)
        for (int x = 2; x < cols - 2; ++x) {
            const float d = dl[(long)y * cols + x];
            if (!(d >= 0.0f) || !(d < (float)cols)) { dl[(long)y * cols + x] = invalid; continue; }
            const int xr = x - (int)d;
            if (xr < 2 || fabsf(dr[(long)y * cols + xr] - d) > threshold) dl[(long)y * cols + x] = invalid;
        }
}

ORC_API int simd_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
