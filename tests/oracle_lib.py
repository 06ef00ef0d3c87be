"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")


def _load():
    so = os.path.join(_ORACLE_DIR, "liboracle.so")
    src = os.path.join(_ORACLE_DIR, "svx_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return C.CDLL(so)


lib = _load()
_p = C.c_void_p
lib.orc_fnv1a64.restype = C.c_uint64
lib.orc_fnv1a64.argtypes = [_p, C.c_size_t, C.c_uint64]
lib.orc_fnv1a64_init.restype = C.c_uint64
lib.orc_popcount32.restype = C.c_uint
lib.orc_popcount32.argtypes = [C.c_uint32]
lib.orc_hamming32.restype = C.c_uint
lib.orc_hamming32.argtypes = [C.c_uint32, C.c_uint32]
lib.orc_u8_to_s8.argtypes = [_p, _p, C.c_size_t]
lib.orc_census5x5_sse24.argtypes = [_p, _p, C.c_int, C.c_int]
lib.orc_census5x5_generic25.argtypes = [_p, _p, C.c_int, C.c_int]
for _n in ("orc_window_cost_5x6", "orc_window_cost_5x5"):
    getattr(lib, _n).restype = C.c_int
    getattr(lib, _n).argtypes = [_p, _p] + [C.c_int] * 6
lib.orc_dense_wta5x5.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, _p]
lib.orc_dense_wta5x5_rows.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, _p, C.c_int, C.c_int]
lib.orc_epipolar_match.argtypes = [_p, _p] + [C.c_int] * 7 + [_p, _p]
lib.orc_cost_eval.argtypes = [_p, _p, C.c_int, C.c_int, _p, C.c_int, _p, _p]
lib.orc_cost_eval_band.argtypes = [_p, _p, C.c_int, C.c_int, _p, C.c_int, _p, _p, C.c_int, C.c_int]
lib.orc_refine.argtypes = [_p, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint] + [_p] * 9
lib.orc_sparse_match.restype = C.c_int
lib.orc_sparse_match.argtypes = [_p, _p, C.c_int, C.c_int, _p, C.c_int, _p, C.c_int,
                                 C.c_int, C.c_float, C.c_float, C.c_int, _p, _p, _p]
lib.orc_plane_fit.argtypes = [_p, _p]
lib.orc_support_resampling.restype = C.c_int
lib.orc_support_resampling.argtypes = [_p, _p] + [C.c_int] * 6 + [_p] * 7 + [C.c_int] * 3 + [_p, _p, _p]
lib.orc_census_nxn_words.restype = C.c_int
lib.orc_census_nxn_words.argtypes = [C.c_int]
lib.orc_u8_saturate_s8.argtypes = [_p, _p, C.c_size_t]
lib.orc_census_nxn.argtypes = [_p, C.c_int, C.c_int, C.c_int, _p]
lib.orc_plane_prior_raster.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, _p]
lib.orc_crop16.argtypes = [_p, C.c_int, C.c_int, _p]
lib.orc_equalize_hist.argtypes = [_p, _p, C.c_int, C.c_int]
lib.orc_laplacian_abs.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
lib.orc_high_gradient_points.restype = C.c_int
lib.orc_high_gradient_points.argtypes = [_p, C.c_int, C.c_int, C.c_int, _p]
lib.orc_gaussian_blur.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int]
lib.orc_alter_rows.restype = C.c_int
lib.orc_alter_rows.argtypes = [_p] + [C.c_int] * 7
lib.orc_sparse_nonmax.argtypes = [_p, _p, C.c_int, _p]
lib.orc_arc9.restype = C.c_int
lib.orc_arc9.argtypes = [C.c_uint]
lib.orc_delaunay_check.restype = C.c_int
lib.orc_delaunay_check.argtypes = [_p, C.c_int, _p, C.c_int]
lib.orc_point_cloud.restype = C.c_int
lib.orc_point_cloud.argtypes = [_p, C.c_int, _p, C.c_int, C.c_int, C.c_int, _p, _p, C.c_int, C.c_double, _p, _p]
lib.orc_census9x7.argtypes = [_p, C.c_int, C.c_int, _p]
lib.orc_dense_wta_words.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, C.c_int, C.c_int]
lib.orc_lr_check.argtypes = [_p, _p, C.c_int, C.c_int, C.c_float, C.c_float]
lib.orc_subpixel_refine.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]
lib.orc_exfast_detect.restype = C.c_int
lib.orc_exfast_detect.argtypes = [_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _p, _p, C.c_int]


def _ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % lib.orc_fnv1a64(_ptr(a), a.nbytes, lib.orc_fnv1a64_init())


def u8_to_s8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(img.shape, dtype=np.int8)
    lib.orc_u8_to_s8(_ptr(img), _ptr(out), img.size)
    return out


def census5x5(img_s8, layout="sse24", out=None):
    img_s8 = np.ascontiguousarray(img_s8, dtype=np.int8)
    rows, cols = img_s8.shape
    if out is None:
        out = np.zeros((rows, cols), dtype=np.uint32)
    fn = lib.orc_census5x5_sse24 if layout == "sse24" else lib.orc_census5x5_generic25
    fn(_ptr(img_s8), _ptr(out), rows, cols)
    return out


def window_cost(ref, cmp_, rx, ry, px, py, mode="5x6"):
    rows, cols = ref.shape
    fn = lib.orc_window_cost_5x6 if mode == "5x6" else lib.orc_window_cost_5x5
    return fn(_ptr(ref), _ptr(cmp_), rows, cols, rx, ry, px, py)


def dense_wta5x5(cl, cr, max_disp, out=None):
    rows, cols = cl.shape
    if out is None:
        out = np.zeros((rows, cols), dtype=np.float32)
    lib.orc_dense_wta5x5(_ptr(cl), _ptr(cr), rows, cols, max_disp, _ptr(out))
    return out


def epipolar_match(cl, cr, census_size, aggr, x, y, max_disp):
    rows, cols = cl.shape
    d = C.c_float()
    c = C.c_int8()
    lib.orc_epipolar_match(_ptr(cl), _ptr(cr), rows, cols, census_size, aggr, x, y, max_disp,
                           C.byref(d), C.byref(c))
    return d.value, c.value


def cost_eval(cl, cr, pts_xy, disp, costs):
    rows, cols = cl.shape
    pts_xy = np.ascontiguousarray(pts_xy, dtype=np.int32)
    lib.orc_cost_eval(_ptr(cl), _ptr(cr), rows, cols, _ptr(pts_xy), len(pts_xy), _ptr(disp), _ptr(costs))
    return costs


def cost_eval_band(cl, cr, pts_xy, disp, costs, radius, max_disp):
    """In place on disp (the prior) and costs."""
    rows, cols = cl.shape
    pts_xy = np.ascontiguousarray(pts_xy, dtype=np.int32)
    lib.orc_cost_eval_band(_ptr(cl), _ptr(cr), rows, cols, _ptr(pts_xy), len(pts_xy), _ptr(disp), _ptr(costs), int(radius), int(max_disp))
    return disp, costs


def refine(pts_xy, disp, costs, t_low, t_high, grid, final_disp, final_costs):
    rows, cols = disp.shape
    pts_xy = np.ascontiguousarray(pts_xy, dtype=np.int32)
    gh, gw = rows // grid + 1, cols // grid + 1
    g = dict(conf_x=np.zeros(gh * gw, np.int32), conf_y=np.zeros(gh * gw, np.int32),
             conf_d=np.zeros(gh * gw, np.float32), conf_c=np.zeros(gh * gw, np.int8),
             inv_x=np.zeros(gh * gw, np.int32), inv_y=np.zeros(gh * gw, np.int32),
             inv_c=np.zeros(gh * gw, np.int8))
    lib.orc_refine(_ptr(pts_xy), len(pts_xy), _ptr(disp), _ptr(costs), rows, cols, t_low, t_high, grid,
                   _ptr(final_disp), _ptr(final_costs), _ptr(g["conf_x"]), _ptr(g["conf_y"]),
                   _ptr(g["conf_d"]), _ptr(g["conf_c"]), _ptr(g["inv_x"]), _ptr(g["inv_y"]), _ptr(g["inv_c"]))
    g["gw"], g["gh"] = gw, gh
    return g


def sparse_match(cl, cr, left_xy, right_xy, max_disp, y_tol=1.0, uniqueness=0.6, lr_step=1):
    rows, cols = cl.shape
    left_xy = np.ascontiguousarray(left_xy, dtype=np.float32).reshape(-1, 2)
    right_xy = np.ascontiguousarray(right_xy, dtype=np.float32).reshape(-1, 2)
    nl, nr = len(left_xy), len(right_xy)
    ol = np.zeros(max(nl, 1), np.int32)
    orr = np.zeros(max(nl, 1), np.int32)
    oc = np.zeros(max(nl, 1), np.int16)
    n = lib.orc_sparse_match(_ptr(cl), _ptr(cr), rows, cols, _ptr(left_xy), nl, _ptr(right_xy), nr,
                             max_disp, y_tol, uniqueness, lr_step, _ptr(ol), _ptr(orr), _ptr(oc))
    return ol[:n], orr[:n], oc[:n]


def support_resampling(cl, cr, g, census_size, aggr, t_low, t_high, max_disp, disparities):
    rows, cols = cl.shape
    nc = g["gw"] * g["gh"]
    oxy = np.zeros((2 * nc, 2), np.int32)
    od = np.zeros(2 * nc, np.float32)
    n = lib.orc_support_resampling(_ptr(cl), _ptr(cr), rows, cols, census_size, aggr, g["gw"], g["gh"],
                                   _ptr(g["conf_x"]), _ptr(g["conf_y"]), _ptr(g["conf_d"]), _ptr(g["conf_c"]),
                                   _ptr(g["inv_x"]), _ptr(g["inv_y"]), _ptr(g["inv_c"]), t_low, t_high, max_disp,
                                   _ptr(disparities), _ptr(oxy), _ptr(od))
    return oxy[:n], od[:n]


def u8_saturate_s8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(img.shape, np.int8)
    lib.orc_u8_saturate_s8(_ptr(img), _ptr(out), img.size)
    return out


def census_nxn(img_s8, size):
    img_s8 = np.ascontiguousarray(img_s8, dtype=np.int8)
    rows, cols = img_s8.shape
    nw = lib.orc_census_nxn_words(size)
    out = np.empty((rows, cols, nw), np.uint32)
    lib.orc_census_nxn(_ptr(img_s8), rows, cols, size, _ptr(out))
    return out


def plane_fit(xyd):
    xyd = np.ascontiguousarray(xyd, dtype=np.float64).reshape(9)
    out = np.zeros(3, np.float32)
    lib.orc_plane_fit(_ptr(xyd), _ptr(out))
    return out


def plane_prior_raster(tri_xy, tri_abc, rows, cols, prior):
    tri_xy = np.ascontiguousarray(tri_xy, dtype=np.int32).reshape(-1, 6)
    tri_abc = np.ascontiguousarray(tri_abc, dtype=np.float32).reshape(-1, 3)
    lib.orc_plane_prior_raster(_ptr(tri_xy), _ptr(tri_abc), len(tri_xy), rows, cols, _ptr(prior))
    return prior


# ---- pre-processing and feature detection ----
def crop16(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    out = np.empty((16 * (rows // 16), 16 * (cols // 16)), np.uint8)
    lib.orc_crop16(_ptr(img), rows, cols, _ptr(out))
    return out


def equalize_hist(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = img.copy()
    lib.orc_equalize_hist(_ptr(img), _ptr(out), img.shape[0], img.shape[1])
    return out


def laplacian_abs(img, ksize=3, scale=1.0, delta=0.0):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    lib.orc_laplacian_abs(_ptr(img), _ptr(out), img.shape[0], img.shape[1], ksize, scale, delta)
    return out


def high_gradient_points(abs_grad, threshold):
    abs_grad = np.ascontiguousarray(abs_grad, dtype=np.uint8)
    rows, cols = abs_grad.shape
    n = lib.orc_high_gradient_points(_ptr(abs_grad), rows, cols, threshold, None)
    pts = np.zeros((max(n, 1), 2), np.int32)
    lib.orc_high_gradient_points(_ptr(abs_grad), rows, cols, threshold, _ptr(pts))
    return pts[:n]


def gaussian_blur(img, ksize):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    lib.orc_gaussian_blur(_ptr(img), _ptr(out), img.shape[0], img.shape[1], ksize)
    return out


def alter_rows(img, invert_rows, nb_rows, trace_lines, nb_lines, line_size):
    out = np.ascontiguousarray(img, dtype=np.uint8).copy()
    rc = lib.orc_alter_rows(_ptr(out), out.shape[0], out.shape[1], int(invert_rows), nb_rows, int(trace_lines), nb_lines, line_size)
    assert rc == 0
    return out


def sparse_nonmax(xy, score):
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    score = np.ascontiguousarray(score, dtype=np.uint8)
    keep = np.zeros(max(len(xy), 1), np.uint8)
    lib.orc_sparse_nonmax(_ptr(xy), _ptr(score), len(xy), _ptr(keep))
    return keep[:len(xy)].astype(bool)


def exfast_detect(img, min_threshold, adaptivity, nonmax=True, round_sat=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    cap = rows * cols
    xy = np.zeros((cap, 2), np.int32)
    resp = np.zeros(cap, np.float32)
    n = lib.orc_exfast_detect(_ptr(img), rows, cols, min_threshold, adaptivity, int(nonmax), int(round_sat), _ptr(xy), _ptr(resp), cap)
    return xy[:n].copy(), resp[:n].copy()


def delaunay_check(pts_xy, tri):
    pts = np.ascontiguousarray(pts_xy, dtype=np.int32).reshape(-1, 2)
    tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
    return lib.orc_delaunay_check(_ptr(pts), len(pts), _ptr(tri) if len(tri) else None, len(tri))


def point_cloud(pts_xy, disp, q, roi=(0, 0), bgr=None, max_range=8000.0):
    pts = np.ascontiguousarray(pts_xy, dtype=np.int32).reshape(-1, 2)
    disp = np.ascontiguousarray(disp, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(16)
    xyz = np.zeros((max(len(pts), 1), 3), np.float64)
    rgb = np.zeros((max(len(pts), 1), 3), np.uint8) if bgr is not None else None
    n = lib.orc_point_cloud(_ptr(pts), len(pts), _ptr(disp), disp.shape[1], roi[0], roi[1], _ptr(q),
                            _ptr(bgr) if bgr is not None else None, bgr.shape[1] if bgr is not None else 0, max_range,
                            _ptr(xyz), _ptr(rgb) if rgb is not None else None)
    return xyz[:n], (rgb[:n] if rgb is not None else None)


lib.orc_project_matches.restype = None
lib.orc_project_matches.argtypes = [_p, C.c_int, _p, _p]


def project_matches(xyd, q):
    xyd = np.ascontiguousarray(xyd, dtype=np.float32).reshape(-1, 3)
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(16)
    out = np.zeros((max(len(xyd), 1), 3), np.float32)
    lib.orc_project_matches(_ptr(xyd), len(xyd), _ptr(q), _ptr(out))
    return out[:len(xyd)]


lib.orc_kinect_evaluate.restype = C.c_int
lib.orc_kinect_evaluate.argtypes = [_p, C.c_int, C.c_int, _p, C.c_int, _p, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]


def kinect_evaluate(raw_depth, disp, pts_xy, q, cam, rotation, translation, roi=(0, 0)):
    """PerformanceEvaluator restated: dict with n_corr, the correspondences (bin, err[4]) in input order, the per-bin
    tables and the totals."""
    raw = np.ascontiguousarray(raw_depth, dtype=np.int32)
    disp = np.ascontiguousarray(disp, dtype=np.float32)
    pts = np.ascontiguousarray(pts_xy, dtype=np.int32).reshape(-1, 2)
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(16)
    cam = np.ascontiguousarray(cam, dtype=np.float64).reshape(9)
    r = np.ascontiguousarray(rotation, dtype=np.float64).reshape(9)
    t = np.ascontiguousarray(translation, dtype=np.float64).reshape(3)
    n = len(pts)
    cbin = np.zeros(max(n, 1), np.int32); cerr = np.zeros((max(n, 1), 4), np.float32)
    mean = np.zeros((4, 2048), np.float32); sigma = np.zeros((4, 2048), np.float32)
    count = np.zeros(2048, np.int32); totals = np.zeros(10, np.float32)
    nc = lib.orc_kinect_evaluate(_ptr(raw), raw.shape[0], raw.shape[1], _ptr(disp), disp.shape[1], _ptr(pts), n, roi[0], roi[1],
                                 _ptr(q), _ptr(cam), _ptr(r), _ptr(t), _ptr(cbin), _ptr(cerr), _ptr(mean), _ptr(sigma), _ptr(count),
                                 _ptr(totals))
    if nc < 0:
        raise ValueError("raw depth image too small or out of range")
    return {"n_corr": nc, "corr_bin": cbin[:nc], "corr_err": cerr[:nc], "bin_mean": mean, "bin_sigma": sigma, "bin_count": count,
            "mean": totals[0:4].copy(), "sigma": totals[4:8].copy(), "mean_rel": float(totals[8]), "sigma_rel": float(totals[9])}


# ---- 64-bit descriptors, right view, left-right check ----
def census9x7(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(img.shape + (2,), np.uint32)
    lib.orc_census9x7(_ptr(img), img.shape[0], img.shape[1], _ptr(out))
    return out


def dense_wta_words(cl, cr, max_disp, view=0, y0=0, y1=None):
    cl = np.ascontiguousarray(cl, dtype=np.uint32); cr = np.ascontiguousarray(cr, dtype=np.uint32)
    rows, cols = cl.shape[0], cl.shape[1]
    words = cl.shape[2] if cl.ndim == 3 else 1
    out = np.zeros((rows, cols), np.float32)
    lib.orc_dense_wta_words(_ptr(cl), _ptr(cr), rows, cols, words, max_disp, view, _ptr(out), y0, rows if y1 is None else y1)
    return out


def lr_check(dl, dr, threshold, invalid=-1.0):
    out = np.ascontiguousarray(dl, dtype=np.float32).copy()
    dr = np.ascontiguousarray(dr, dtype=np.float32)
    lib.orc_lr_check(_ptr(out), _ptr(dr), out.shape[0], out.shape[1], threshold, invalid)
    return out


def subpixel_refine(cl, cr, max_disp, disp):
    cl = np.ascontiguousarray(cl, dtype=np.uint32); cr = np.ascontiguousarray(cr, dtype=np.uint32)
    out = np.ascontiguousarray(disp, dtype=np.float32).copy()
    words = cl.shape[2] if cl.ndim == 3 else 1
    lib.orc_subpixel_refine(_ptr(cl), _ptr(cr), out.shape[0], out.shape[1], words, max_disp, _ptr(out))
    return out


# ---- the formula image of tests/golden/survey_kat.json ----
def kat_images():
    R, Cc = 32, 64

    def pix(y, x):
        return (y * 37 + x * 101 + ((x * y) % 7) * 13 + ((x * x + 3 * y) % 11) * 29) & 255

    right = np.zeros((R, Cc), np.uint8)
    left = np.zeros((R, Cc), np.uint8)
    for y in range(R):
        d = 3 + y // 8
        for x in range(Cc):
            right[y, x] = pix(y, x)
            left[y, x] = pix(y, x - d) if x >= d else pix(y + 1, x + 50)
    return left, right


# ---- the SIMD + OpenMP CPU baseline (oracle/svx_cpu_simd.c), checked against the routines above ----
def _load_simd():
    so = os.path.join(_ORACLE_DIR, "liboracle_simd.so")
    src = os.path.join(_ORACLE_DIR, "svx_cpu_simd.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "liboracle_simd.so"], stdout=subprocess.DEVNULL)
    s = C.CDLL(so)
    s.simd_census9x7.argtypes = [_p, C.c_int, C.c_int, _p, C.c_int]
    s.simd_dense_wta_words.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int]
    s.simd_lr_check.argtypes = [_p, _p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    s.simd_max_threads.restype = C.c_int
    return s


_simd = None


def simd():
    global _simd
    if _simd is None:
        _simd = _load_simd()
    return _simd


def simd_census9x7(img, threads=1):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros(img.shape + (2,), np.uint32)
    simd().simd_census9x7(_ptr(img), img.shape[0], img.shape[1], _ptr(out), threads)
    return out


def simd_dense_wta_words(cl, cr, max_disp, threads=1, left=True, right=True, y0=0, y1=None):
    cl = np.ascontiguousarray(cl, dtype=np.uint32); cr = np.ascontiguousarray(cr, dtype=np.uint32)
    rows, cols = cl.shape[0], cl.shape[1]
    words = cl.shape[2] if cl.ndim == 3 else 1
    dl = np.zeros((rows, cols), np.float32) if left else None
    dr = np.zeros((rows, cols), np.float32) if right else None
    simd().simd_dense_wta_words(_ptr(cl), _ptr(cr), rows, cols, words, max_disp, _ptr(dl) if left else None, _ptr(dr) if right else None,
                                y0, rows if y1 is None else y1, threads)
    return dl, dr


def simd_lr_check(dl, dr, threshold, invalid=-1.0, threads=1):
    out = np.ascontiguousarray(dl, dtype=np.float32).copy()
    simd().simd_lr_check(_ptr(out), _ptr(np.ascontiguousarray(dr, dtype=np.float32)), out.shape[0], out.shape[1], threshold, invalid, threads)
    return out
