"""CPU tests of the Kinect evaluation restatement (PerformanceEvaluator.cpp) and of the point cloud text writer.  The
reference holds no data for either (its Kinect captures are not in the repository): parity unpinned; the checks here are
a physically consistent scene, an independent float32 recomputation of the statistics, and the file format."""
import numpy as np

import oracle_lib as O
import kinect_scene as K


def f32_stats(values):
    """Boost's lazy variance on float samples, one float32 operation at a time."""
    s = np.float32(0); ss = np.float32(0)
    for v in values:
        v = np.float32(v)
        s = np.float32(s + v); ss = np.float32(ss + np.float32(v * v))
    with np.errstate(all="ignore"):
        n = np.float32(len(values))
        m = np.float32(s / n)
        var = np.float32(np.float32(ss / n) - np.float32(m * m))
        return m, np.float32(np.sqrt(var))


def test_consistent_scene_has_small_errors():
    raw, disp, pts, q, cam, r, t = K.consistent_scene()
    res = O.kinect_evaluate(raw, disp, pts, q, cam, r, t)
    # every point projects back onto its own pixel and finds its own depth
    assert res["n_corr"] == len(pts)
    # z (millimetres): the partner is the Kinect point of the TRUNCATED pixel, so a reprojection that lands a hair below
    # its integer pixel meets the neighbour's depth (8 mm per pixel of slope here)
    assert abs(res["mean"][2]) < 1.0 and res["sigma"][2] < 10.0
    assert abs(res["mean"][0]) < 1.0
    assert res["mean"][3] < 15.0                                          # distance: dominated by fx != fy on the y axis
    assert res["bin_count"].sum() == res["n_corr"]
    empty = res["bin_count"] == 0
    assert np.all(np.isnan(res["bin_mean"][:, empty])) and np.all(np.isnan(res["bin_sigma"][:, empty]))
    assert not np.any(np.isnan(res["bin_mean"][:, ~empty]))


def test_statistics_match_a_float32_recomputation():
    raw, disp, pts, q, cam, r, t = K.random_scene(200, 300, 3000, 5)
    res = O.kinect_evaluate(raw, disp, pts, q, cam, r, t, roi=(11, 4))
    assert 0 < res["n_corr"] < len(pts)
    bins, err = res["corr_bin"], res["corr_err"]
    assert np.array_equal(np.bincount(bins, minlength=2048), res["bin_count"])
    order = np.argsort(bins, kind="stable")
    for k in range(4):
        m, s = f32_stats(err[order, k])
        assert np.array_equal(np.array([m, s]), np.array([res["mean"][k], res["sigma"][k]]), equal_nan=True)
    rel = []
    for b in np.nonzero(res["bin_count"])[0]:
        for k in range(4):
            m, s = f32_stats(err[bins == b, k])
            assert np.array_equal(np.array([m, s]), np.array([res["bin_mean"][k, b], res["bin_sigma"][k, b]]), equal_nan=True)
        if b != 2047:
            mm = np.float32(1000 * (1.0 / (float(b) * -0.0030711016 + 3.3309495161)))
            rel.append(np.float32(res["bin_mean"][3, b] / mm))
    m, s = f32_stats(rel)
    assert np.array_equal(np.array([m, s]), np.array([res["mean_rel"], res["sigma_rel"]], np.float32), equal_nan=True)
    # distance error is the norm of the three axis errors
    d = np.sqrt((err[:, :3].astype(np.float64) ** 2).sum(1)).astype(np.float32)
    assert np.array_equal(d, err[:, 3])


def test_no_points_and_bad_depth():
    raw, disp, pts, q, cam, r, t = K.random_scene(50, 60, 10, 1)
    res = O.kinect_evaluate(raw, disp, pts[:0], q, cam, r, t)
    assert res["n_corr"] == 0 and np.isnan(res["mean"]).all() and np.isnan(res["mean_rel"])
    bad = raw.copy(); bad[3, 3] = 2048
    for broken in (bad, raw[:400], raw[:, :600]):
        try:
            O.kinect_evaluate(broken, disp, pts, q, cam, r, t)
            assert False, "accepted"
        except ValueError:
            pass


def test_point_cloud_text_file(tmp_path):
    """"x y z r g b" lines with the stream's default number formatting (highPerfStereo.cpp:721): six significant digits."""
    import stereovision_amd as S
    from stereovision_amd._lib import write_point_cloud_txt
    rng = np.random.default_rng(3)
    xyz = np.concatenate([rng.normal(size=(50, 3)) * 10.0 ** rng.integers(-8, 9, size=(50, 1)),
                          [[0.0, -0.0, 1.0], [1e21, 123456.5, 1234567.0], [0.1, 100000.0, 999999.5]]])
    rgb = rng.integers(0, 256, size=(len(xyz), 3), dtype=np.uint8)
    path = tmp_path / "cloud.txt"
    write_point_cloud_txt(path, xyz, rgb)
    want = "".join("%g %g %g %d %d %d\n" % (p[0], p[1], p[2], c[0], c[1], c[2]) for p, c in zip(xyz, rgb))
    assert path.read_text() == want
    write_point_cloud_txt(path, xyz[:2], None, const_rgb=(255, 0, 0))
    assert path.read_text() == "".join("%g %g %g 255 0 0\n" % tuple(p) for p in xyz[:2])
    write_point_cloud_txt(path, xyz[:0], None)
    assert path.read_text() == ""
    try:
        write_point_cloud_txt(tmp_path / "no" / "such" / "dir.txt", xyz, rgb)
        assert False, "accepted"
    except S.SvxError:
        pass
