"""GPU parity tests of the steps either side of the census path (pre-processing and the exFAST
detector), through the C ABI, bit-exact against the CPU oracle.  The oracle's versions of the OpenCV
calls (equalizeHist, Laplacian, convertScaleAbs, GaussianBlur, FAST) restate documented behaviour and
are not pinned on reference outputs: parity unpinned for those, pinned GPU == oracle only."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    import stereovision_amd as S
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    c = S.Context(0)
    yield c
    c.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def textured(rng, rows, cols, blocks=True):
    """Piecewise-constant blocks + ramps + noise: plenty of corners and large ring contrasts."""
    y, x = np.mgrid[0:rows, 0:cols]
    img = (x * 3 + y * 5) % 256
    if blocks:
        img = np.where(((x // 7) + (y // 5)) % 2 == 0, img, 255 - img)
        img = np.where(((x // 23) + (y // 17)) % 3 == 0, (img // 128) * 255, img)
    img = img + rng.integers(-12, 13, size=(rows, cols))
    return np.clip(img, 0, 255).astype(np.uint8)


SHAPES = [(375, 1242), (64, 64), (37, 129), (16, 16), (7, 7), (6, 40), (5, 5), (1, 1), (3, 300)]


@pytest.mark.parametrize("rows,cols", SHAPES)
def test_crop16(ctx, rows, cols):
    rng = np.random.default_rng(rows + cols)
    img = rng.integers(0, 256, size=(2, rows, cols), dtype=np.uint8)
    if rows < 16 or cols < 16:
        import stereovision_amd as S
        with pytest.raises(S.SvxError):
            ctx.crop16(dev(img))
        return
    got = ctx.crop16(dev(img)).cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], O.crop16(img[f]))


@pytest.mark.parametrize("rows,cols", SHAPES)
def test_equalize_hist(ctx, rows, cols):
    rng = np.random.default_rng(rows * 7 + cols)
    frames = [textured(rng, rows, cols), rng.integers(40, 90, size=(rows, cols), dtype=np.uint8),
              np.full((rows, cols), 77, np.uint8), rng.integers(0, 2, size=(rows, cols), dtype=np.uint8) * 255]
    batch = np.stack(frames)
    got = ctx.equalize_hist(dev(batch)).cpu().numpy()
    for f in range(len(frames)):
        assert np.array_equal(got[f], O.equalize_hist(frames[f])), f
    # in place, as the reference calls it (highPerfStereoLib.cpp:251)
    d = dev(frames[0])
    ctx.equalize_hist(d, out=d)
    assert np.array_equal(d.cpu().numpy(), O.equalize_hist(frames[0]))


@pytest.mark.parametrize("rows,cols", SHAPES)
@pytest.mark.parametrize("ksize,scale,delta", [(3, 1, 0), (1, 1, 0), (3, 2, -7), (1, 0.5, 3), (3, 40, 0)])
def test_laplacian_abs(ctx, rows, cols, ksize, scale, delta):
    rng = np.random.default_rng(rows * 3 + cols + ksize)
    batch = np.stack([textured(rng, rows, cols), rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)])
    got = ctx.laplacian_abs(dev(batch), ksize, scale, delta).cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], O.laplacian_abs(batch[f], ksize, scale, delta))


@pytest.mark.parametrize("rows,cols", SHAPES)
@pytest.mark.parametrize("threshold", [0, 50, 200, 255])
def test_high_gradient_points(ctx, rows, cols, threshold):
    rng = np.random.default_rng(rows + 31 * cols + threshold)
    batch = np.stack([O.laplacian_abs(textured(rng, rows, cols)), rng.integers(0, 256, size=(rows, cols), dtype=np.uint8),
                      np.zeros((rows, cols), np.uint8)])
    pts, off = ctx.high_gradient_points(dev(batch), threshold)
    pts = pts.cpu().numpy()
    assert off[0] == 0 and len(pts) == off[-1]
    for f in range(3):
        want = O.high_gradient_points(batch[f], threshold)
        assert off[f + 1] - off[f] == len(want)
        assert np.array_equal(pts[off[f]:off[f + 1]], want)


@pytest.mark.parametrize("n_frames", [1, 30, 31, 70])
def test_high_gradient_points_frame_counts(ctx, n_frames):
    """The per-frame offsets reach the host through mapped memory for up to 30 frames and by a copy beyond:
    both against the oracle, twice in a row on one context (the second call with other counts)."""
    rng = np.random.default_rng(n_frames)
    for rows, cols in ((24, 40), (17, 33)):
        batch = rng.integers(0, 256, size=(n_frames, rows, cols), dtype=np.uint8)
        batch[n_frames // 2] = 0                             # a frame without any point
        pts, off = ctx.high_gradient_points(dev(batch), 128)
        pts = pts.cpu().numpy()
        assert off[0] == 0 and len(off) == n_frames + 1 and len(pts) == off[-1]
        for f in range(n_frames):
            want = O.high_gradient_points(batch[f], 128)
            assert off[f + 1] - off[f] == len(want)
            assert np.array_equal(pts[off[f]:off[f + 1]], want)


def test_high_gradient_points_capacity(ctx):
    rng = np.random.default_rng(9)
    g = rng.integers(0, 256, size=(40, 50), dtype=np.uint8)
    want = O.high_gradient_points(g, 100)
    pts, off = ctx.high_gradient_points(dev(g), 100, capacity=17)
    assert off[-1] == len(want) and len(pts) == 17
    assert np.array_equal(pts.cpu().numpy(), want[:17])
    pts, off = ctx.high_gradient_points(dev(g), 100, capacity=0)
    assert off[-1] == len(want)


@pytest.mark.parametrize("rows,cols", SHAPES)
@pytest.mark.parametrize("ksize", [1, 3, 5])
def test_gaussian_blur(ctx, rows, cols, ksize):
    rng = np.random.default_rng(rows + cols * 5 + ksize)
    batch = np.stack([textured(rng, rows, cols), rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)])
    got = ctx.gaussian_blur(dev(batch), ksize).cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], O.gaussian_blur(batch[f], ksize))


@pytest.mark.parametrize("rows,cols", [(368, 1232), (64, 64), (37, 129), (16, 16), (7, 7), (1, 1)])
@pytest.mark.parametrize("params", [(True, 10, True, 20, 2), (True, 1, False, 1, 1), (False, 3, True, 5, 9), (True, 7, True, 1, 400),
                                    (True, 400, True, 3, 1), (True, 2, False, 1, 1), (False, 1, False, 1, 1)])
def test_alter_rows(ctx, rows, cols, params):
    inv, nb_rows, trace, nb_lines, line_size = params
    if trace and rows // nb_lines == 0:
        import stereovision_amd as S
        with pytest.raises(S.SvxError):
            ctx.alter_rows(dev(np.zeros((rows, cols), np.uint8)), *params)
        return
    rng = np.random.default_rng(rows + cols)
    batch = rng.integers(0, 256, size=(2, rows, cols), dtype=np.uint8)
    got = ctx.alter_rows(dev(batch), *params).cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], O.alter_rows(batch[f], *params))


# ------------------------------------------------------------------ exFAST
@pytest.mark.parametrize("rows,cols", [(375, 1242), (368, 1232), (64, 64), (37, 129), (16, 70), (8, 8), (7, 7), (6, 40), (3, 3)])
@pytest.mark.parametrize("min_threshold,adaptivity", [(10, 1.0), (0, 0.5), (25, 2.5), (3, 0.25)])
def test_exfast_matches_oracle(ctx, rows, cols, min_threshold, adaptivity):
    rng = np.random.default_rng(rows * 13 + cols + min_threshold)
    frames = [textured(rng, rows, cols), textured(rng, rows, cols, blocks=False),
              rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)]
    batch = np.stack(frames)
    for nonmax in (True, False):
        xy, resp, off = ctx.exfast_detect(dev(batch), min_threshold, adaptivity, nonmax=nonmax)
        xy, resp = xy.cpu().numpy(), resp.cpu().numpy()
        assert off[0] == 0 and off[-1] == len(xy)
        for f in range(len(frames)):
            wxy, wresp = O.exfast_detect(frames[f], min_threshold, adaptivity, nonmax)
            assert off[f + 1] - off[f] == len(wxy), (f, nonmax)
            assert np.array_equal(xy[off[f]:off[f + 1]], wxy)
            assert np.array_equal(resp[off[f]:off[f + 1]], wresp)


def test_exfast_finds_corners_and_threshold_readings(ctx):
    import stereovision_amd as S
    rng = np.random.default_rng(77)
    img = textured(rng, 120, 200)
    wxy, wresp = O.exfast_detect(img, 10, 1.0, True)
    assert len(wxy) > 50                                    # the test is not vacuous
    wxy2, wresp2 = O.exfast_detect(img, 10, 1.0, True, round_sat=True)
    xy, resp, off = ctx.exfast_detect(dev(img), 10, 1.0, nonmax=True, flags=S.EXFAST_THRESHOLD_ROUND_SAT)
    assert np.array_equal(xy.cpu().numpy(), wxy2) and np.array_equal(resp.cpu().numpy(), wresp2)
    xy, resp, off = ctx.exfast_detect(dev(img), 10, 1.0, nonmax=True)
    assert np.array_equal(xy.cpu().numpy(), wxy) and np.array_equal(resp.cpu().numpy(), wresp)


def test_exfast_capacity_and_errors(ctx):
    import stereovision_amd as S
    rng = np.random.default_rng(5)
    img = textured(rng, 60, 90)
    wxy, wresp = O.exfast_detect(img, 10, 1.0, True)
    xy, resp, off = ctx.exfast_detect(dev(img), 10, 1.0, capacity=5)
    assert off[-1] == len(wxy) and np.array_equal(xy.cpu().numpy(), wxy[:5]) and np.array_equal(resp.cpu().numpy(), wresp[:5])
    with pytest.raises(S.SvxError):
        ctx.exfast_detect(dev(img), 10, 1.0, flags=64)


def test_detector_feeds_sparse_matcher(ctx):
    """exFAST key points (left with, right without suppression: highPerfStereoLib.cpp:358-359) through the
    sparse matcher give the same correspondences on the GPU path and in the oracle."""
    from stereovision_amd import synth
    import stereovision_amd as S
    left, right = synth.stereo_pair(96, 320, 32, seed=11)[:2]
    lxy, _, _ = ctx.exfast_detect(dev(left), 10, 1.0, nonmax=True)
    rxy, _, _ = ctx.exfast_detect(dev(right), 10, 1.0, nonmax=False)
    wl, _ = O.exfast_detect(left, 10, 1.0, True)
    wr, _ = O.exfast_detect(right, 10, 1.0, False)
    assert np.array_equal(lxy.cpu().numpy(), wl) and np.array_equal(rxy.cpu().numpy(), wr)
    cl = ctx.census5x5(dev(left), S.CENSUS_SSE24 | S.CENSUS_ZERO_REST)
    cr = ctx.census5x5(dev(right), S.CENSUS_SSE24 | S.CENSUS_ZERO_REST)
    got = ctx.sparse_match(cl, cr, lxy.cpu().numpy().astype(np.float32), rxy.cpu().numpy().astype(np.float32), 32)
    ocl = O.census5x5(O.u8_to_s8(left)); ocr = O.census5x5(O.u8_to_s8(right))
    want = O.sparse_match(ocl, ocr, wl.astype(np.float32), wr.astype(np.float32), 32)
    assert len(want[0]) > 10
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


# ------------------------------------------------------------------ support-point mesh
@pytest.mark.parametrize("rows,cols,n", [(64, 96, 12), (375, 1242, 900), (368, 1232, 5000), (40, 40, 3)])
def test_mesh_plane_prior(ctx, rows, cols, n):
    """Triangulation + plane table + interpolation loop in one call against the oracle's plane fit and
    rasteriser run over the same triangle list (the triangulation itself is checked in test_mesh_cpu.py)."""
    import stereovision_amd as S
    rng = np.random.default_rng(rows + n)
    pts = np.unique(np.stack([rng.integers(0, cols, n), rng.integers(0, rows, n)], 1).astype(np.int32), axis=0)
    pts = pts[rng.permutation(len(pts))]
    disp = np.zeros((rows, cols), np.float32)
    disp[pts[:, 1], pts[:, 0]] = rng.integers(0, 64, len(pts)).astype(np.float32)
    tri = S.delaunay(pts)
    assert O.delaunay_check(pts, tri) == 0
    tri_xy = pts[tri].reshape(-1, 6)
    abc = np.stack([O.plane_fit(np.concatenate([np.concatenate([pts[v], [disp[pts[v][1], pts[v][0]]]]) for v in t])) for t in tri]) \
        if len(tri) else np.zeros((0, 3), np.float32)
    want = O.plane_prior_raster(tri_xy, abc, rows, cols, disp.copy())
    prior = dev(disp.copy())
    got, nt = ctx.mesh_plane_prior(pts, dev(disp), prior)
    assert nt == len(tri)
    assert np.array_equal(got.cpu().numpy(), want)
    # in place on the disparity map, as the reference's loop does (highPerfStereoLib.cpp:534-553): the
    # vertex disparities are read by the plane fit before the rasteriser overwrites anything
    d = dev(disp.copy())
    ctx.mesh_plane_prior(pts, d, d)
    assert np.array_equal(d.cpu().numpy(), want)


# ------------------------------------------------------------------ point cloud
Q_KITTI = np.array([[1, 0, 0, -609.5593], [0, 1, 0, -172.854], [0, 0, 0, 721.5377], [0, 0, 1.0 / 0.54, 0]], np.float64)


@pytest.mark.parametrize("rows,cols,n", [(375, 1242, 100000), (64, 80, 1), (64, 80, 1023), (64, 80, 1025), (64, 80, 5000)])
def test_point_cloud(ctx, rows, cols, n):
    rng = np.random.default_rng(rows + n)
    roi = (7, 3)
    pts = np.stack([rng.integers(0, cols, n), rng.integers(0, rows, n)], 1).astype(np.int32)
    disp = rng.integers(-2, 90, size=(rows, cols)).astype(np.float32) * 0.25
    bgr = rng.integers(0, 256, size=(rows + 9, cols + 11, 3), dtype=np.uint8)
    for q, max_range in ((Q_KITTI, 8000.0), (Q_KITTI * 1000.0, 8000.0), (rng.normal(size=(4, 4)), 3.0)):
        wxyz, wrgb = O.point_cloud(pts, disp, q, roi, bgr, max_range)
        xyz, rgb, k = ctx.point_cloud(dev(pts), dev(disp), q, roi, dev(bgr), max_range)
        assert k == len(wxyz)
        # double precision, same expression order on both sides, no contraction: bit-exact
        assert np.array_equal(xyz.cpu().numpy(), wxyz)
        assert np.array_equal(rgb.cpu().numpy(), wrgb)
    wxyz, _ = O.point_cloud(pts, disp, Q_KITTI, (0, 0), None)
    xyz, rgb, k = ctx.point_cloud(dev(pts), dev(disp), Q_KITTI, capacity=min(10, n))
    assert rgb is None and k == len(wxyz) and np.array_equal(xyz.cpu().numpy(), wxyz[:min(10, n)])


def test_point_cloud_rejects_points_outside(ctx):
    import stereovision_amd as S
    disp = dev(np.zeros((20, 30), np.float32))
    with pytest.raises(S.SvxError):
        ctx.point_cloud(dev(np.array([[30, 0]], np.int32)), disp, Q_KITTI)
    with pytest.raises(S.SvxError):
        ctx.point_cloud(dev(np.array([[5, 5]], np.int32)), disp, Q_KITTI, roi=(28, 0), bgr=dev(np.zeros((20, 30, 3), np.uint8)))
    xyz, rgb, k = ctx.point_cloud(dev(np.zeros((0, 2), np.int32)), disp, Q_KITTI)
    assert k == 0


@pytest.mark.parametrize("frames,rows,cols", [(1, 375, 1242), (3, 64, 80), (2, 5, 7)])
def test_dense_point_cloud(ctx, frames, rows, cols):
    """Every pixel above the disparity floor, frames concatenated in raster order == the point-list routine on that list."""
    rng = np.random.default_rng(frames * 1000 + rows)
    disp = (rng.integers(-4, 200, size=(frames, rows, cols)).astype(np.float32) * 0.25)
    disp[rng.random(disp.shape) < 0.3] = -1.0
    roi = (5, 2)
    for min_d, max_range in ((0.0, 8000.0), (10.0, 60.0)):
        xyz, pix, off = ctx.dense_point_cloud(dev(disp), Q_KITTI, roi, min_d, max_range, with_pixels=True)
        xyz, pix = xyz.cpu().numpy(), pix.cpu().numpy()
        assert off[0] == 0 and off[-1] == len(xyz)
        for f in range(frames):
            ys, xs = np.nonzero(disp[f] > min_d)
            pts = np.stack([xs, ys], 1).astype(np.int32)
            wxyz, _ = O.point_cloud(pts, disp[f], Q_KITTI, roi, None, max_range)
            assert off[f + 1] - off[f] == len(wxyz)
            assert np.array_equal(xyz[off[f]:off[f + 1]], wxyz)
            # the pixels come back in raster order and are the ones the oracle kept
            p = pix[off[f]:off[f + 1]]
            key = p[:, 1].astype(np.int64) * cols + p[:, 0]
            assert np.all(np.diff(key) > 0)
    xyz, pix, off = ctx.dense_point_cloud(dev(disp), Q_KITTI, capacity=3)
    assert len(xyz) == min(3, int(off[-1])) and pix is None


# ------------------------------------------------------------------ stereo against Kinect depth
@pytest.mark.parametrize("rows,cols,n,depth_shape,seed", [(368, 1232, 120000, (480, 640), 1), (200, 300, 1025, (500, 700), 2),
                                                           (64, 80, 1, (480, 640), 3), (64, 80, 4000, (480, 640), 4)])
def test_kinect_evaluate(ctx, rows, cols, n, depth_shape, seed):
    """Correspondences, per-bin tables and totals identical to the oracle's: the per-point arithmetic is the same
    expression sequence in double/float on both sides (no contraction), the statistics the same sequential float sums."""
    import kinect_scene as K
    raw, disp, pts, q, cam, r, t = K.random_scene(rows, cols, n, seed, depth_shape)
    for roi in ((0, 0), (11, 4)):
        want = O.kinect_evaluate(raw, disp, pts, q, cam, r, t, roi)
        got = ctx.kinect_evaluate(dev(raw), dev(disp), dev(pts), q, cam, r, t, roi)
        assert got["n_corr"] == want["n_corr"]
        assert np.array_equal(got["bin_count"], want["bin_count"])
        for key in ("bin_mean", "bin_sigma", "mean", "sigma"):
            assert np.array_equal(got[key], want[key], equal_nan=True), key
        assert np.array_equal(np.float32([got["mean_rel"], got["sigma_rel"]]), np.float32([want["mean_rel"], want["sigma_rel"]]), equal_nan=True)
    if n > 1000:
        assert want["n_corr"] > 100


def test_kinect_evaluate_consistent_scene_and_errors(ctx):
    import stereovision_amd as S
    import kinect_scene as K
    raw, disp, pts, q, cam, r, t = K.consistent_scene()
    want = O.kinect_evaluate(raw, disp, pts, q, cam, r, t)
    got = ctx.kinect_evaluate(dev(raw), dev(disp), dev(pts), q, cam, r, t)
    assert got["n_corr"] == want["n_corr"] == len(pts)
    assert np.array_equal(got["mean"], want["mean"]) and np.array_equal(got["sigma"], want["sigma"])
    # no points: every statistic is NaN, as the accumulators give for an empty set
    none = ctx.kinect_evaluate(dev(raw), dev(disp), dev(pts[:0]), q, cam, r, t)
    assert none["n_corr"] == 0 and np.isnan(none["mean"]).all() and np.isnan(none["bin_mean"]).all()
    bad = raw.copy(); bad[5, 5] = 2048
    for broken_raw, broken_pts in ((bad, pts), (raw[:400], pts), (raw, np.array([[640, 0]], np.int32))):
        with pytest.raises(S.SvxError):
            ctx.kinect_evaluate(dev(broken_raw), dev(disp), dev(broken_pts), q, cam, r, t)


@pytest.mark.parametrize("n", [0, 1, 255, 5000])
def test_project_matches(ctx, n):
    """SparseMatch::projectMatches: float Q, float products summed left to right, division by w without a zero test."""
    rng = np.random.default_rng(n)
    xyd = np.stack([rng.integers(0, 1242, n), rng.integers(0, 375, n), rng.integers(0, 128, n)], 1).astype(np.float32)
    for q in (Q_KITTI, rng.normal(size=(4, 4))):
        want = O.project_matches(xyd, q)
        got = ctx.project_matches(dev(xyd) if n else torch.empty((0, 3), dtype=torch.float32, device="cuda"), q).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True)
    if n:
        assert not np.isfinite(O.project_matches(xyd[xyd[:, 2] == 0][:1], Q_KITTI)).all() or not (xyd[:, 2] == 0).any()
