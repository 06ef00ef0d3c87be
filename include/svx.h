/*
 * svx.h -- C ABI of libsvx: the census stereo-matching hot path of
 * TomEwbank/StereoVision as hand-written HIP kernels for AMD MI355X (gfx950).
 *
 * The reference has no FFI layer: its boundary is a set of C++ signatures over
 * cv::Mat_.  Each entry point below names the reference routine it replaces
 * (file:line relative to the reference root).  Conventions:
 *
 *   - every image is row-major and contiguous (step == cols), one frame after
 *     another when n_frames > 1 (frame stride == rows*cols elements);
 *   - pointers are DEVICE pointers unless the parameter name ends in _host;
 *   - work is enqueued on the context's stream and is asynchronous with
 *     respect to the host unless the entry point has _host parameters (those
 *     return after their _host results are on the host; what the same call
 *     leaves in device memory is still ordered by the stream only: read it on
 *     that stream, or after svx_synchronize);
 *   - return value: SVX_OK (0) or a negative svx_status; svx_last_error()
 *     holds a message for the most recent failure of that context;
 *   - a context belongs to one device and one host thread at a time; distinct
 *     contexts are independent.
 *
 * Out-of-image addressing: the reference leaves rows through pointer
 * arithmetic (census borders, 6-column window, unclipped disparity).  libsvx
 * defines such an access as a flat index into the contiguous frame; an index
 * outside the frame reads 0.  This equals the reference whenever its buffers
 * lie between zeroed guard bands.
 */
#ifndef SVX_H
#define SVX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVX_VERSION 100

typedef enum svx_status {
    SVX_OK = 0,
    SVX_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad enum */
    SVX_ERR_SIZE_MISMATCH = -2, /* "Mismatching image sizes" (sparsestereo-inl.h:42-43) */
    SVX_ERR_EMPTY = -3,         /* "Input images are empty!" (highPerfStereoLib.cpp:184-185) */
    SVX_ERR_HIP = -4,           /* a HIP runtime call or kernel launch failed */
    SVX_ERR_NO_DEVICE = -5,     /* no usable gfx950 device */
    SVX_ERR_UNSUPPORTED = -6,   /* parameter outside what the kernels were built for */
    SVX_ERR_NOMEM = -7          /* device workspace allocation failed */
} svx_status;

typedef struct svx_ctx svx_ctx;

/* ---- context ---------------------------------------------------------- */
int svx_create(int device, svx_ctx **out);
int svx_destroy(svx_ctx *ctx);
/* hip_stream is a hipStream_t (NULL = the default stream).  A context works on ONE stream at a time: its
 * workspace is shared by all its calls, so switching streams first waits for the work queued on the old one.
 * Callers that want concurrent streams create one context per stream. */
int svx_set_stream(svx_ctx *ctx, void *hip_stream);
int svx_synchronize(svx_ctx *ctx);
/* Per-context options (no counterpart in the reference).  Except for the last one results never depend on them, only
 * the choice of kernel.  A new context starts from the built-in defaults, or from the environment variable named below when it is
 * set at svx_create time; svx_set_option overrides it for that context only.
 *   SVX_OPT_MATRIX_CORES     (SVX_MX)              0 = the vector sweep everywhere; any other value (default -1) = the
 *                                                  matrix-core search wherever it applies
 *   SVX_OPT_LR_TWO_SWEEPS    (SVX_LR_TWO_SWEEPS)   1 = the left-right entry points search each view with its own sweep
 *   SVX_OPT_SWEEP_WAVES      (SVX_SWEEP_OCC)       2 or 3 wavefronts per SIMD for the vector sweep (default 3)
 *   SVX_OPT_SWEEP_WAVES_BOTH (SVX_SWEEP_OCC_BOTH)  the same for the both-views vector sweep
 *   SVX_OPT_CENSUS97_PACKED  (SVX_CENSUS97_PACKED) form of the 9x7 census tile kernel: 2 (default) three comparisons per subtraction (widths
 *                                                  that are multiples of 4; others run form 1), 1 two per subtraction, 0 one comparison per instruction
 *   SVX_OPT_MX_STRIP_ROWS    (SVX_MX_STRIP_ROWS)   > 0 forces the output rows per strip of the matrix-core search
 *   SVX_OPT_STAGE_TIMES      (SVX_STAGE_TIMES)     1 = svx_high_perf_stereo synchronises at its stage marks and fills
 *                                                  svx_pipeline_stage_times (default 0: no synchronisation, times 0)
 *   SVX_OPT_MX_WIDE_TILES    (SVX_MX_WIDE_TILES)   more than 128 disparities on two-word descriptors: 1 = passes of the nine-tile
 *                                                  matrix-core kernel, 0 = passes of the five-tile one; default -1: nine tiles where
 *                                                  one pass of them covers the range with no more tiles than passes of five (225..256)
 *   SVX_OPT_MESH_ON_DEVICE   (SVX_MESH_ON_DEVICE)  0 = svx_high_perf_stereo triangulates its support points on the host
 *                                                  (svx_mesh_*); default 1 = on the device (svx_mesh_plane_prior_device);
 *                                                  the triangles are the same
 *   SVX_OPT_SUBPIXEL_TILES   (SVX_SUBPIXEL_TILES)  0 = the sub-pixel step reads every descriptor from memory; 1 = the left
 *                                                  descriptors of a block of pixels are staged in LDS; default 2 = the right ones
 *                                                  too, up to 252 disparities (beyond: as 1)
 * The one option that changes results:
 *   SVX_OPT_PRIOR_BAND_RADIUS (SVX_PRIOR_BAND_RADIUS) radius of svx_high_perf_stereo's search around the plane prior
 *                                                  (svx_cost_eval_band); default 0 = the reference's single disparity */
enum svx_option {
    SVX_OPT_MATRIX_CORES = 0, SVX_OPT_LR_TWO_SWEEPS = 1, SVX_OPT_SWEEP_WAVES = 2, SVX_OPT_SWEEP_WAVES_BOTH = 3,
    SVX_OPT_CENSUS97_PACKED = 4, SVX_OPT_MX_STRIP_ROWS = 5, SVX_OPT_STAGE_TIMES = 6, SVX_OPT_PRIOR_BAND_RADIUS = 7, SVX_OPT_MX_WIDE_TILES = 8,
    SVX_OPT_MESH_ON_DEVICE = 9, SVX_OPT_SUBPIXEL_TILES = 10, SVX_OPT_COUNT = 11
};
int svx_set_option(svx_ctx *ctx, int option, int value);
int svx_get_option(const svx_ctx *ctx, int option, int *value);   /* ctx NULL: the default a new context would get */
int svx_version(void);
const char *svx_strerror(int status);
const char *svx_last_error(const svx_ctx *ctx);

/* Device-memory helpers for bindings that cannot call HIP themselves. */
int svx_malloc(svx_ctx *ctx, size_t bytes, void **dptr);
int svx_free(svx_ctx *ctx, void *dptr);
int svx_memset(svx_ctx *ctx, void *dptr, int value, size_t bytes);
int svx_memcpy_h2d(svx_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int svx_memcpy_d2h(svx_ctx *ctx, void *dst_host, const void *src, size_t bytes);

/* ---- a1: unsigned -> signed pixels -------------------------------------
 * Replaces ImageConversion::unsignedToSigned, sparsestereo/imageconversion.h:24,
 * imageconversion.cpp:63-78.  dst = src - 128 (mod 256). */
int svx_u8_to_s8(svx_ctx *ctx, const uint8_t *src, int8_t *dst, int rows, int cols, int n_frames);

/* ---- a2: 5x5 census transform ------------------------------------------
 * Replaces Census::transform5x5, sparsestereo/census.h:17-18.
 *   SVX_CENSUS_SSE24     the `char` specialisation, census.cpp:17-94: 24-bit
 *                        words, rows [2,rows-2) x ALL columns written, border
 *                        columns see the adjacent row (flat addressing).
 *   SVX_CENSUS_GENERIC25 the generic template, census-inl.h:13-53: 25-bit
 *                        words (centre bit 0), interior pixels only.
 * Pixels the reference does not write are left untouched unless
 * SVX_CENSUS_ZERO_REST is or-ed in, which writes 0 there.
 * svx_census5x5_u8 fuses a1 and takes the unsigned image directly. */
enum { SVX_CENSUS_SSE24 = 0, SVX_CENSUS_GENERIC25 = 1, SVX_CENSUS_ZERO_REST = 16 };
int svx_census5x5_s8(svx_ctx *ctx, const int8_t *in, uint32_t *out, int rows, int cols, int n_frames, int flags);
int svx_census5x5_u8(svx_ctx *ctx, const uint8_t *in, uint32_t *out, int rows, int cols, int n_frames, int flags);

/* ---- a14: generic N x N census -------------------------------------------
 * Replaces CensusTransform::CensusTransform / computeCensus, denseCensusStereo/
 * CensusTransform.h, CensusTransform.cpp:9-67: size is made odd, every pixel
 * gets svx_census_nxn_words(size) = m*m/32 + 1 words (pixel-major), bit j*m+i
 * set when neighbour >= centre, all-zero within m/2 of the border, and the
 * reference's skip test (window index compared with the image coordinate,
 * CensusTransform.cpp:47) reproduced.  svx_census_nxn_u8 first narrows the
 * unsigned pixels with saturation to 127, as the reference's callers do
 * implicitly (denseCensusStereo.cpp:307-308). */
int svx_census_nxn_words(int size);
int svx_census_nxn_s8(svx_ctx *ctx, const int8_t *in, int rows, int cols, int n_frames, int size, uint32_t *out);
int svx_census_nxn_u8(svx_ctx *ctx, const uint8_t *in, int rows, int cols, int n_frames, int size, uint32_t *out);

/* ---- a4: census window cost ---------------------------------------------
 * Replaces CensusWindow<5>::setReferencePoint + match, sparsestereo/
 * censuswindow.h:46-49, :96-107.  For pair p: cost[p] = window Hamming cost of
 * ref at (ref_xy[2p], ref_xy[2p+1]) against cmp at (cmp_xy[2p], cmp_xy[2p+1]).
 *   SVX_WINDOW_5X6  what the reference computes on x86-64 (5 rows x columns
 *                   x-2..x+3);  SVX_WINDOW_5X5  the exact 5x5 window. */
enum { SVX_WINDOW_5X6 = 0, SVX_WINDOW_5X5 = 1 };
int svx_census_window_cost(svx_ctx *ctx, const uint32_t *ref, const uint32_t *cmp, int rows, int cols,
                           const int32_t *ref_xy, const int32_t *cmp_xy, int n_pairs, int16_t *cost, int mode);

/* ---- a5 + a6 + a7: sparse feature matching ------------------------------
 * Replaces SparseStereo<CensusWindow<5>,short>::match, sparsestereo/
 * sparsestereo.h:24-30, sparsestereo-inl.h:37-227, for rect == NULL (inputs
 * already rectified).  Keypoints are float (x,y) pairs on the HOST; results
 * come back on the host, one record per left feature in (y,x)-sorted order:
 * out_left = index into left_xy, out_right = index into right_xy or -1
 * (rejected / unmatched), out_cost = 5x6 window cost.  *n_out = number of
 * records (0 when either list is empty, sparsestereo-inl.h:44-45). */
int svx_sparse_match(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                     const float *left_xy_host, int n_left, const float *right_xy_host, int n_right,
                     int max_disparity, float y_tolerance, float uniqueness, int left_right_step,
                     int32_t *out_left_host, int32_t *out_right_host, int16_t *out_cost_host, int *n_out);

/* ---- a8: dense winner-takes-all ------------------------------------------
 * Replaces epipolarMatching + the driver loop of denseCensusStereo/
 * denseCensusStereo.cpp:87-131, :257-266: 5x5-aggregated Hamming cost, every
 * disparity d in [0, min(max_disp-1, x-5)], strict '<' (ties -> smallest d).
 * svx_dense_wta5x5 writes pixels 2<=x<cols-2, 2<=y<rows-2 only (as the
 * reference does).  svx_dense_census_wta_u8 is the fused hot path: a1 + a2 on
 * both images and a8 in one call.  The two census batches (4 bytes a pixel and
 * image) live in the context's workspace between the census and the search
 * launches; a cost volume never exists.  It writes the whole frame (0 outside
 * the interior, the reference's initial value). */
int svx_dense_wta5x5(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right,
                     int rows, int cols, int n_frames, int max_disp, float *disparity);
int svx_dense_census_wta_u8(svx_ctx *ctx, const uint8_t *left, const uint8_t *right,
                            int rows, int cols, int n_frames, int max_disp, float *disparity);

/* svx_dense_census_wta_u8 without any allocation.  No counterpart in the reference: it is for callers that cannot
 * call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it).  Same arguments and
 * results; the two census batches live in a fixed 32 MiB array of the library's code object (one copy per device,
 * held until the process ends) instead of the context's workspace, so n_frames*rows*cols is limited to
 * SVX_NOALLOC_MAX_PIXELS (SVX_ERR_UNSUPPORTED beyond it).  It calls no hipMalloc/hipFree.  All calls on one device share
 * that array: calls that overlap on one device (two contexts or streams) race. */
#define SVX_NOALLOC_MAX_PIXELS 4194304
int svx_dense_census_wta_u8_noalloc(svx_ctx *ctx, const uint8_t *left, const uint8_t *right,
                                    int rows, int cols, int n_frames, int max_disp, float *disparity);

/* ---- a9: batched full-range epipolar search ------------------------------
 * Replaces epipolarMatching, highPerfStereo/highPerfStereoLib.h:211-214,
 * highPerfStereoLib.cpp:88-139, for n points at once.  out_cost is the
 * single-pixel Hamming distance at the winner (:136). */
int svx_epipolar_match_batch(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right,
                             int rows, int cols, int census_size, int cost_aggr_window_size,
                             const int32_t *pts_xy, int n_pts, int max_disp, float *out_disp, int8_t *out_cost);

/* ---- a13: support resampling (device part) ---------------------------------
 * Replaces the loop of supportResampling, highPerfStereoLib.h:216-222,
 * highPerfStereoLib.cpp:141-175, without the mesh: for cell c (row-major) slot
 * 2c is its invalid match re-matched over the full range (a9) and accepted when
 * the winner's single-pixel cost is < t_low; slot 2c+1 is its confident
 * support.  Accepted slots get out_valid = 1, their disparity is written into
 * `disparities`, and (out_xy[2s], out_xy[2s+1], out_d[s]) is the point to insert
 * into the mesh; the reference's insertion order is slot order.  Grid arrays
 * are those returned by svx_refine. */
int svx_support_resampling(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                           int census_size, int cost_aggr_window_size, int grid_w, int grid_h,
                           const int32_t *conf_x, const int32_t *conf_y, const float *conf_d, const int8_t *conf_c,
                           const int32_t *inv_x, const int32_t *inv_y, const int8_t *inv_c,
                           int t_low, int t_high, int max_disp, float *disparities,
                           int32_t *out_xy, float *out_d, uint8_t *out_valid);

/* ---- a10: cost of the plane prior ----------------------------------------
 * Replaces costEvaluation, highPerfStereoLib.h:196-200, .cpp:20-41.
 * costs is in/out (the caller pre-fills it with tHigh, .cpp:573). */
int svx_cost_eval(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                  const int32_t *pts_xy, int n_pts, const float *disparities, int8_t *costs);

/* Banded search around the plane prior (no counterpart in the reference: costEvaluation evaluates exactly one
 * disparity per point; the only banded plane-prior search in its tree is libelas' SAD matcher, elas/elas.cpp:683-780,
 * the model for this one).  d0 = (int)floor(prior + 0.5), the disparity svx_cost_eval evaluates, is admitted as there
 * (x - d0 >= 2); the other candidates d0 + k, 0 < |k| <= radius, when 0 <= d0 + k, d0 + k < max_disp (max_disp <= 0:
 * no upper bound) and x - (d0 + k) >= 2.  Ascending disparity, strict '<' (ties: smallest disparity).  costs(y,x) =
 * the winner's single-pixel Hamming distance; disparities(y,x) becomes (float)winner when the winner is not d0 and is
 * left as it is (fractional) otherwise, so svx_refine takes its result unchanged.  radius 0 is svx_cost_eval.
 * SVX_OPT_PRIOR_BAND_RADIUS makes svx_high_perf_stereo use it with max_disp = maxDisp. */
int svx_cost_eval_band(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                       const int32_t *pts_xy, int n_pts, float *disparities, int8_t *costs, int radius, int max_disp);

/* ---- a11: accept / refine + occupancy grid --------------------------------
 * Replaces disparityRefinement, highPerfStereoLib.h:203-209, .cpp:44-86, and
 * the PotentialSupports container, highPerfStereoLib.h:114-151, returned as
 * struct-of-arrays with cell index j*grid_w + i, grid_w = cols/g + 1,
 * grid_h = rows/g + 1.  pts_xy must be in the caller's visiting order (raster
 * order in the reference): ties go to the earliest point. */
int svx_refine(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, const float *disparities, const int8_t *costs,
               int rows, int cols, int t_low, int t_high, unsigned occ_grid_size,
               float *final_disparities, int8_t *final_costs,
               int32_t *conf_x, int32_t *conf_y, float *conf_d, int8_t *conf_c,
               int32_t *inv_x, int32_t *inv_y, int8_t *inv_c);

/* ---- a12: plane prior ------------------------------------------------------
 * svx_plane_fit replaces Plane::Plane(Triangle2*, Mat_<float>), plane/
 * Plane.cpp:27-65 (Cramer's rule in double, CramerTheorem.cpp:24-48): for
 * triangle t with integer vertices tri_xy[6t..6t+5] it reads the three vertex
 * disparities from `disparities` and writes float a,b,c to tri_abc[3t..].
 * svx_plane_prior_raster replaces the interpolation loop highPerfStereoLib.cpp:
 * 532-554 with Plane::getDepth, Plane.cpp:71-73: every pixel inside a
 * triangle gets float(a*x + b*y + c) evaluated in double; pixels outside all
 * triangles keep their value.  A pixel on a shared edge takes the triangle
 * that comes first in the list (the reference delegates this choice to
 * Fade2D's locate()). */
int svx_plane_fit(svx_ctx *ctx, const int32_t *tri_xy, int n_tri, const float *disparities, int rows, int cols,
                  float *tri_abc);
int svx_plane_prior_raster(svx_ctx *ctx, const int32_t *tri_xy, const float *tri_abc, int n_tri,
                           int rows, int cols, float *prior);

/* ---- 64-bit descriptors, right view, left-right check (no counterpart in the reference) ----
 * Generalisations of svx_dense_wta5x5 for the 9x7 / 128-disparity configuration; the reference
 * has 5x5 (sparsestereo/census.cpp) and a square NxN census (denseCensusStereo/
 * CensusTransform.cpp:33-67) only, and a dense consistency check only in libelas
 * (elas/elas.cpp:909-979), which these follow in spirit.
 * svx_census9x7_u8: descriptor of pixel (x,y), 4 <= x < cols-4, 3 <= y < rows-3: neighbour
 *   (dy,dx), dy=-3..3, dx=-4..4, index (dy+3)*9+(dx+4), bit 62-index, set when centre >
 *   neighbour; census[2p] = bits 0..31, census[2p+1] = bits 32..62; other pixels 0.
 * svx_dense_wta: svx_dense_wta5x5's search (5x5 aggregation, x_r >= 5, strict '<') on
 *   descriptors of `words` 32-bit words (1 or 2, interleaved per pixel), the whole map written
 *   (border 0).  SVX_VIEW_RIGHT gives the right image's map: for (x_r, y) scan
 *   x_l = x_r, x_r+1, ... while x_l <= cols-6 and x_l < x_r + max_disp, disparity = x_l - x_r
 *   of the first minimum.
 * svx_dense_wta_both: both maps of svx_dense_wta in one call.  The cost of (left column x, disparity d) is also the cost
 *   of right pixel x-d at disparity d, so one sweep over the cost values serves both views: the results are identical
 *   to the two svx_dense_wta calls.
 * svx_lr_check: for 2 <= x < cols-2, 2 <= y < rows-2: d = disp_left(y,x), x_r = x - (int)d;
 *   disp_left(y,x) = invalid_value when d is no disparity (d < 0, NaN or d >= cols: nothing is read then), when
 *   x_r < 2, or when |disp_right(y,x_r) - d| > threshold.
 * svx_dense_census9x7_lr_u8: census of both images, left-view search, and for
 *   lr_threshold >= 0 the right-view search and the check (invalid = -1). */
enum svx_view { SVX_VIEW_LEFT = 0, SVX_VIEW_RIGHT = 1 };
int svx_census9x7_u8(svx_ctx *ctx, const uint8_t *img, uint32_t *census, int rows, int cols, int n_frames);
int svx_dense_wta(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                  int words, int max_disp, int view, float *disparity);
int svx_dense_wta_both(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                       int words, int max_disp, float *disp_left, float *disp_right);
int svx_lr_check(svx_ctx *ctx, float *disp_left, const float *disp_right, int rows, int cols, int n_frames,
                 float threshold, float invalid_value);
int svx_dense_census9x7_lr_u8(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                              int max_disp, float lr_threshold, float *disparity);

/* Sub-pixel refinement of a left-view map (no counterpart in the reference, which has no sub-pixel step
 * for census matches).  For 2 <= x < cols-2, 2 <= y < rows-2 holding an integer disparity d >= 1 with
 * d+1 <= max_disp-1 and x-(d+1) >= 5: c(k) = the 5x5-aggregated Hamming cost at disparity k,
 * den = c(d-1) - 2c(d) + c(d+1); when den > 0 the value becomes (float)d + (float)(c(d-1)-c(d+1)) / (float)(2 den).
 * Every other value (invalidated pixels, d = 0, range ends, flat cost curves) stays as it is.
 * svx_dense_census9x7_subpixel_u8 = svx_dense_census9x7_lr_u8 followed by this step on the same descriptors. */
int svx_subpixel_refine(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                        int words, int max_disp, float *disparity);
int svx_dense_census9x7_subpixel_u8(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                                    int max_disp, float lr_threshold, float *disparity);

/* ---- steps either side of the census path ----------------------------------
 * Pre-processing of highPerfStereo(), highPerfStereoLib.cpp:245-335.  Its
 * arithmetic is OpenCV's (not part of the reference tree); the entry points
 * restate OpenCV's documented behaviour for 8-bit single-channel images.
 *
 * svx_crop16: newROI(0,0,16*(cols/16),16*(rows/16)), .cpp:245-247; dst is
 *   densely packed [n_frames][16*(rows/16)][16*(cols/16)].
 * svx_equalize_hist: cv::equalizeHist, .cpp:251-252 (src == dst allowed).
 * svx_laplacian_abs: cv::Laplacian(ddepth CV_16S, ksize 1|3, scale, delta,
 *   BORDER_DEFAULT) followed by cv::convertScaleAbs, .cpp:258-259.
 * svx_high_gradient_points: the point set of .cpp:269-280 (2 <= y < rows-2,
 *   2 <= x < cols-2, pixel > threshold) in the reference's push order, frames
 *   concatenated; frame_offsets_host[f]..[f+1] (n_frames+1 host ints) delimit
 *   frame f.  Points past `capacity` are counted but not stored.  Synchronous.
 * svx_gaussian_blur: cv::GaussianBlur(Size(k,k), 0, 0), .cpp:289-292, k in
 *   {1,3,5}; the reference passes k = (int)applyBlur = 1 (.cpp:207), a copy.
 * svx_alter_rows: the row-band inversion and line tracing applied to the
 *   detector's input copies, .cpp:300-335, in place. */
int svx_crop16(svx_ctx *ctx, const uint8_t *src, int rows, int cols, int n_frames, uint8_t *dst);
int svx_equalize_hist(svx_ctx *ctx, const uint8_t *src, uint8_t *dst, int rows, int cols, int n_frames);
int svx_laplacian_abs(svx_ctx *ctx, const uint8_t *src, uint8_t *abs_grad, int rows, int cols, int n_frames,
                      int ksize, double scale, double delta);
int svx_high_gradient_points(svx_ctx *ctx, const uint8_t *abs_grad, int rows, int cols, int n_frames, int threshold,
                             int32_t *pts_xy, int capacity, int *frame_offsets_host);
int svx_gaussian_blur(svx_ctx *ctx, const uint8_t *src, uint8_t *dst, int rows, int cols, int n_frames, int ksize);
int svx_alter_rows(svx_ctx *ctx, uint8_t *img, int rows, int cols, int n_frames, int invert_rows, int nb_rows,
                   int trace_lines, int nb_lines, int line_size);

/* exFAST feature detector: replaces ExtendedFAST::detectImpl, sparsestereo/
 * extendedfast.cpp:92-130 (FAST-9 at min_threshold :102, adaptive test and
 * score :153-231, centre value extendedfast.h:18, non-maximum suppression
 * fast9-inl.h:76-173).  Key points of all frames are written in raster order
 * (kp_xy[2s], kp_xy[2s+1]) with kp_response[s] = score, or -1000 without
 * SVX_EXFAST_NONMAX (.cpp:125); size 6 and angle -1 are constants of the
 * reference (.cpp:116).  frame_offsets_host as for svx_high_gradient_points.
 * SVX_EXFAST_THRESHOLD_ROUND_SAT reads cv::saturate_cast<char>(float) (.cpp:164)
 * as round + saturate to [-128,127]; the default is the C++ conversion the
 * generic template performs (truncation, low byte).  Synchronous. */
enum svx_exfast_flags { SVX_EXFAST_NONMAX = 1, SVX_EXFAST_THRESHOLD_ROUND_SAT = 2 };
int svx_exfast_detect(svx_ctx *ctx, const uint8_t *img, int rows, int cols, int n_frames, int min_threshold, float adaptivity,
                      int flags, int32_t *kp_xy, float *kp_response, int capacity, int *frame_offsets_host);

/* Support-point mesh: stands in for the Fade_2D triangulation of highPerfStereoLib.cpp:434-445
 * (insert) and :457 (getTrianglePointers).  Fade2D is a closed library outside the reference
 * tree; the mesh built here satisfies the Delaunay property exactly (integer predicates); where four or more
 * points are cocircular the diagonals follow a fixed rule on the point indices and may differ from Fade2D's.
 * svx_delaunay (host in, host out): triangles as index triples into pts_xy_host, every triple
 *   rotated to lead with its smallest index, the list sorted lexicographically; repeated
 *   positions are represented by their first index; all-collinear input gives no triangle;
 *   *n_tri is the number found, at most `capacity` are stored.  Touches no device: ctx may be
 *   NULL.
 * svx_mesh_plane_prior: triangulation, plane table (.cpp:454-466) and interpolation loop
 *   (.cpp:532-554) in one call: host support points, device disparities, device prior. */
int svx_delaunay(svx_ctx *ctx, const int32_t *pts_xy_host, int n_pts, int32_t *tri_idx_host, int capacity, int *n_tri);
/* The same triangulation computed on the device (device points in, device index triples out, in svx_delaunay's
 * order; *n_tri_host is on the host when the call returns, the triples follow in stream order): every point finds the triangles around it from the points
 * near it, and the points that cannot be sure (next to large empty areas, on the hull) from all points.  Points
 * must lie inside a rows x cols frame of at most 8192 x 8192.  SVX_ERR_UNSUPPORTED for a point set with more than 256
 * triangles around one point (or 32 led by one point): svx_delaunay takes those.
 * svx_mesh_plane_prior_device is svx_mesh_plane_prior with the points on the device and the triangulation done there
 * (it falls back to the host triangulation by itself where svx_delaunay_device would refuse).
 * svx_mesh_device_stats: how many points of the context's last device triangulation needed the second pass (a wider
 * neighbourhood and, where that cannot tell, all points); the second number is always 0 (a tuning aid; either pointer
 * may be NULL). */
int svx_delaunay_device(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, int rows, int cols, int32_t *tri_idx, int capacity, int *n_tri_host);
int svx_mesh_plane_prior_device(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, const float *disparities, int rows, int cols,
                                float *prior, int *n_tri_out);
int svx_mesh_device_stats(const svx_ctx *ctx, int *left_by_first_pass, int *left_by_second_pass);

/* The mesh as an object that grows, the role of the Fade_2D instance that highPerfStereo() keeps across its iterations
 * (insert: highPerfStereoLib.cpp:434-445 and, through supportResampling, :141-175; getTrianglePointers: :457).  Host
 * only.  svx_mesh_insert appends points to the mesh's list (indices continue across calls; a repeated position keeps its
 * first index); the first points are triangulated at once, later ones are located in the mesh and inserted one by one.
 * Cocircular points are resolved by a fixed rule on the indices, so svx_mesh_triangles returns exactly what svx_delaunay
 * returns for the whole list, however the list was split over the calls.  svx_mesh_prior is svx_mesh_plane_prior on the
 * mesh as it stands.  A mesh belongs to one host thread at a time. */
typedef struct svx_mesh svx_mesh;
int svx_mesh_create(svx_mesh **out);
void svx_mesh_destroy(svx_mesh *mesh);
int svx_mesh_insert(svx_mesh *mesh, const int32_t *pts_xy_host, int n_pts);
int svx_mesh_counts(const svx_mesh *mesh, int *n_points, int *n_triangles);
int svx_mesh_triangles(const svx_mesh *mesh, int32_t *tri_idx_host, int capacity, int *n_tri);
int svx_mesh_prior(svx_ctx *ctx, const svx_mesh *mesh, const float *disparities, int rows, int cols, float *prior, int *n_tri_out);
int svx_mesh_plane_prior(svx_ctx *ctx, const int32_t *pts_xy_host, int n_pts, const float *disparities, int rows, int cols,
                         float *prior, int *n_tri_out);

/* Point cloud: replaces the reprojection loop of highPerfStereo/highPerfStereo.cpp:689-723
 * (cv::perspectiveTransform of (x+roi_x, y+roi_y, disparity(y,x)) by the 4x4 double matrix Q,
 * row-major in q_host; keep when z > 0 and sqrt(x^2+y^2+z^2) < max_range, 8000 in the
 * reference).  pts_xy (device) are positions in the disparity map, e.g. the high-gradient list;
 * kept points are written in input order: xyz_out[3s..] doubles and, when rgb_out is given,
 * r,g,b bytes read from the 3-channel BGR image `bgr` ([img_rows][img_cols][3], device) at the
 * ROI-shifted position.  *n_out_host = number kept; at most `capacity` are stored.  Synchronous. */
int svx_point_cloud(svx_ctx *ctx, const int32_t *pts_xy, int n_pts, const float *disparity_map, int rows, int cols,
                    int roi_x, int roi_y, const double *q_host, const uint8_t *bgr, int img_rows, int img_cols,
                    double max_range, double *xyz_out, uint8_t *rgb_out, int capacity, int *n_out_host);

/* Dense variant of svx_point_cloud for whole disparity maps (the consumer of the dense matcher's output):
 * every pixel of every frame whose disparity is > min_disparity (0 skips unmatched and invalidated
 * pixels), in raster order, frames concatenated, reprojected and filtered as above.  xyz_out gets three
 * doubles a point and pix_xy_out (optional) its pixel; frame_offsets_host[f]..[f+1] (n_frames+1 host
 * ints) delimit frame f; points past `capacity` are counted but not stored.  Synchronous. */
int svx_dense_point_cloud(svx_ctx *ctx, const float *disparity_map, int rows, int cols, int n_frames, int roi_x, int roi_y,
                          const double *q_host, float min_disparity, double max_range, double *xyz_out, int32_t *pix_xy_out,
                          int capacity, int *frame_offsets_host);

/* Replaces SparseMatch::projectMatches, sparsestereo/sparsematch.h:35-50: xyd (device) holds rectLeft.x, rectLeft.y and
 * disparity() of every match, q_host the 4x4 Q (doubles, converted to float as the reference does); xyz_out (device)
 * receives the three float coordinates Q.(x, y, d, 1) / w per match, with no test for w == 0, like the reference.
 * Enqueued on the context's stream. */
int svx_project_matches(svx_ctx *ctx, const float *xyd, int n, const double *q_host, float *xyz_out);

/* Stereo against Kinect depth: replaces PerformanceEvaluator, highPerfStereo/PerformanceEvaluator.cpp:16-328 (constructed at
 * highPerfStereo/highPerfStereo.cpp:792), without its text files.  raw_depth (device, [depth_rows][depth_cols] int32, values
 * in [0, 2048), at least 640x480) is the Kinect's raw depth; pts_xy (device) the considered pixels of the rows x cols
 * disparity_map (device); x_offset / y_offset the common ROI's origin; q_host the 4x4 Q, cam_host the Kinect's 3x3
 * camera matrix, r_host / t_host the rotation matrix and translation from the left stereo camera to the Kinect (all
 * row-major doubles on the host).  The distortion coefficients the reference's constructor receives never reach
 * projectPoints there (:16-27, :109), so there are none here.  Every considered pixel is reprojected, moved to the
 * Kinect frame and projected into the Kinect image on the device; the correspondences (pixel inside 640x480,
 * 0 < z <= 5000 mm, Kinect depth present) come back in input order and the statistics are accumulated on the host
 * in the reference's order (float sums, bins ascending).
 * Host outputs: bin_mean_host / bin_sigma_host [4][2048] (x, y, z, distance error per raw depth value; NaN where a bin
 * is empty, as Boost's accumulators give), bin_count_host [2048], totals_host [10] = mean x, y, z, distance, sigma x, y,
 * z, distance over all correspondences, mean and sigma of the per-bin relative distance errors; *n_corr_host =
 * nbCorrespondences.  Synchronous. */
int svx_kinect_evaluate(svx_ctx *ctx, const int32_t *raw_depth, int depth_rows, int depth_cols, const float *disparity_map,
                        int rows, int cols, const int32_t *pts_xy, int n_pts, int x_offset, int y_offset, const double *q_host,
                        const double *cam_host, const double *r_host, const double *t_host, float *bin_mean_host,
                        float *bin_sigma_host, int32_t *bin_count_host, float *totals_host, int *n_corr_host);

/* Point cloud text file, the format of highPerfStereo/highPerfStereo.cpp:703-723 ("x y z r g b" per line, numbers as
 * the default C++ stream formatting prints doubles).  Host pointers: xyz_host 3 doubles and rgb_host 3 bytes a point;
 * rgb_host NULL writes the constant colour const_rgb (as PerformanceEvaluator.cpp:85-106 does).  Touches no device:
 * ctx may be NULL (then no message is kept on failure). */
int svx_write_point_cloud_txt(svx_ctx *ctx, const char *path, const double *xyz_host, const uint8_t *rgb_host, int n_pts,
                              const uint8_t *const_rgb);

/* ---- the pipeline that calls the census path end to end ----------------------
 * svx_stereo_parameters mirrors StereoParameters, highPerfStereo/highPerfStereoLib.h:35-70, field for
 * field (bool -> int, uchar/char kept).  resizeFactor, blurSize and colorMapSliding are carried but unused,
 * as in the reference (its blur kernel size is (int)applyBlur = 1, highPerfStereoLib.cpp:207, :289-292:
 * the identity).
 * svx_high_perf_stereo replaces highPerfStereo(), highPerfStereoLib.h:224-228, .cpp:177-645, for one
 * rectified pair on the device: crop to multiples of 16, [equalise], Laplacian + |.|, high-gradient
 * point set, row alterations for the detector, exFAST on both images, 5x5 census, sparse matching,
 * support-point mesh, then nIters rounds of {plane prior, costEvaluation, disparityRefinement,
 * supportResampling}, and finally the support points themselves.  disparity_map is rows x cols
 * (the INPUT size, .cpp:190), in/out: only pixels the reference writes change.  The computed points
 * (high-gradient points in raster order, then the mesh vertices in insertion order) are written to
 * points_xy_host; *n_points is their number, at most `capacity` are stored.
 * Differences from the reference, all outside its own arithmetic: the mesh is libsvx's Delaunay
 * triangulation instead of Fade2D's (see svx_delaunay); the prior is interpolated when recordFullDisp
 * is set, but no "disparity<iter>.png" is written (.cpp:565-570); showImages (a display) is refused;
 * ddepth must be CV_16S (3), the value every caller in the reference passes.  Synchronous. */
typedef struct svx_stereo_parameters {
    double uniqueness;
    int maxDisp;
    int leftRightStep;
    int costAggrWindowSize;
    unsigned char gradThreshold;
    signed char tLow;
    signed char tHigh;
    int nIters;
    double resizeFactor;
    int applyBlur;
    int applyHistEqualization;
    int blurSize;
    int kernelSize;
    int scale;
    int delta;
    int ddepth;
    double adaptivity;
    int minThreshold;
    int traceLines;
    int nbLines;
    int lineSize;
    int invertRows;
    int nbRows;
    int recordFullDisp;
    int showImages;
    int colorMapSliding;
} svx_stereo_parameters;
int svx_high_perf_stereo(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols,
                         const svx_stereo_parameters *parameters, float *disparity_map,
                         int32_t *points_xy_host, int capacity, int *n_points);
/* svx_high_perf_stereo for n_frames pairs (frames back to back in left, right and disparity_maps; frame f's points at
 * points_xy_host + 2 f capacity_per_frame, its count in n_points_host[f]) on n_threads host threads (<= 0: 4), each with
 * a context and stream of its own inside `ctx`: frames are independent and a frame's host share (the triangulation)
 * limits one thread, so this is how one GPU is kept busy.  Same results as n_frames single calls.  No counterpart in the
 * reference (its callers loop over pairs on one thread).  Synchronous. */
int svx_high_perf_stereo_batch(svx_ctx *ctx, const uint8_t *left, const uint8_t *right, int rows, int cols, int n_frames,
                               const svx_stereo_parameters *parameters, float *disparity_maps, int32_t *points_xy_host,
                               int capacity_per_frame, int *n_points_host, int n_threads);
/* Wall-clock milliseconds of the stages of the context's last svx_high_perf_stereo call, in the spirit of the
 * per-stage table the reference's demo prints (highPerfStereo/highPerfStereo.cpp:97-99, :591-595):
 * [0] crop + equalisation, [1] gradient + high-gradient set, [2] row alterations + exFAST (both images),
 * [3] census (both images), [4] sparse matching, [5] support points to the device, [6] mesh + plane prior (all
 * iterations), [7] cost evaluation + refinement (all iterations), [8] support resampling (all iterations),
 * [9] final vertex pass + results to the caller.  Up to n values are written. */
#define SVX_PIPELINE_STAGES 10
int svx_pipeline_stage_times(const svx_ctx *ctx, double *ms, int n);

/* ---- measurement aid (no counterpart in the reference) ---------------------
 * Sustained issue rate, in lane-instructions per second, of the VALU pairs the
 * dense matcher is built from, with `waves_per_simd` wavefronts resident per
 * SIMD.  mode 0 = v_xor_b32 + v_bcnt_u32_b32, the measured peak the
 * popcount-bound kernel is priced against; modes 1..7 pair v_xor/v_sub with
 * v_add_u32, v_add3_u32, v_min_u32, DPP v_add_u32, v_pk_add_u16,
 * v_mad_u32_u24 and v_lshl_add_u32.  Synchronous. */
int svx_probe_valu_rate(svx_ctx *ctx, int mode, int waves_per_simd, int iters, double *lane_ops_per_s);
/* Launches exactly the kernel the left-right entry points spend most of their time in (one sweep over the
 * cost values that writes the left map and leaves the right view's partial minima in the workspace), so
 * that a benchmark can bracket it with stream events.  Asynchronous. */
int svx_probe_both_views_sweep(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols,
                               int n_frames, int words, int max_disp, float *disp_left);
/* Which kernel the dense search of this size runs in this context (for benchmarks that name the kernel they time):
 * 0 = the vector sweep, 1 = the matrix-core search, 2 = one thread per pixel.  both_views: the search of
 * svx_dense_wta_both and of the fused left-right entry points.  Follows SVX_OPT_MATRIX_CORES; the default uses the
 * matrix cores for the left-view and the both-views search (one- or two-word descriptors, at least 36 columns and 5
 * rows, up to 1024 disparities).  ctx NULL: the choice of a new context.  No counterpart in the reference.  Host only. */
int svx_dense_kernel_choice(const svx_ctx *ctx, int rows, int cols, int words, int max_disp, int both_views);
/* The search of svx_dense_wta / svx_dense_wta_both on two-word descriptors with the cost values formed on the matrix
 * cores (fp4 dot products of the expanded descriptors; denseCensusStereo/denseCensusStereo.cpp:87-131 generalised as
 * there): identical results.  Covers cols >= 36, rows >= 5, 1 <= max_disp <= 1024 (SVX_ERR_UNSUPPORTED otherwise); more
 * than 128 disparities are searched in passes (160 or 288 right columns each, side by side) whose winners' keys are merged.
 * disp_right may be NULL (left view only).  Border pixels are written as 0.  Asynchronous. */
int svx_dense_wta_matrix(svx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int rows, int cols, int n_frames,
                         int max_disp, float *disp_left, float *disp_right);

#ifdef __cplusplus
}
#endif
#endif /* SVX_H */
