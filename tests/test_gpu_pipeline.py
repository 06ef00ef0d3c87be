"""GPU parity of the full highPerfStereo() pipeline through the C ABI against the
oracle composition of tests/oracle_pipeline.py, bit-exact: disparity map, computed points, and the
intermediate counts.  Parity status: the pipeline's OpenCV steps and the mesh are unpinned on reference
outputs (OpenCV and Fade2D are not available); the census / matching / refinement steps are the pinned
ones.  Both sides use the same triangulation (svx_delaunay, itself checked against the Delaunay
definition in tests/test_mesh_cpu.py)."""
import numpy as np
import pytest

import oracle_pipeline as OP

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


def run_both(ctx, left, right, params, init=0.0, band_radius=0):
    import stereovision_amd as S
    dm = np.full(left.shape, init, np.float32)
    want_pts, info = OP.high_perf_stereo(left, right, params, dm, S.delaunay, band_radius)
    got_map = dev(np.full(left.shape, init, np.float32))
    ctx.set_option(S.OPT_PRIOR_BAND_RADIUS, band_radius)
    try:
        got_pts = ctx.high_perf_stereo(dev(left), dev(right), params, got_map)
    finally:
        ctx.set_option(S.OPT_PRIOR_BAND_RADIUS, 0)
    return got_pts, got_map.cpu().numpy(), want_pts, dm, info


@pytest.mark.parametrize("rows,cols,seed,radius,over", [
    (375, 1242, 11, 2, dict(maxDisp=70, nIters=2)),
    (120, 250, 12, 5, dict(maxDisp=32, nIters=3, gradThreshold=30)),
    (64, 64, 13, 1, dict(maxDisp=16, nIters=2, gradThreshold=20, costAggrWindowSize=5)),
])
def test_pipeline_with_banded_prior_search(ctx, rows, cols, seed, radius, over):
    """The pipeline with the search around the plane prior widened to d0 +- radius (a build extension, off by default):
    same composition on both sides, bit for bit; and the band changes the result (so the option is really taken)."""
    import stereovision_amd as S
    from stereovision_amd import synth
    p = S.StereoParameters.reference_defaults(**over)
    left, right, _ = synth.stereo_pair(rows, cols, min(p.maxDisp, 64), seed)
    got_pts, got_map, want_pts, want_map, info = run_both(ctx, left, right, p, band_radius=radius)
    assert np.array_equal(got_pts, want_pts), info
    assert np.array_equal(got_map, want_map)
    _, plain_map, _, _, _ = run_both(ctx, left, right, p)
    assert not np.array_equal(plain_map, got_map)


@pytest.mark.parametrize("rows,cols,seed,over", [
    (375, 1242, 0, dict(maxDisp=70, nIters=1)),
    (375, 1242, 1, dict(maxDisp=70, nIters=2, applyHistEqualization=1)),
    (375, 1242, 2, dict(maxDisp=100, nIters=3, gradThreshold=40, invertRows=1, nbRows=20, traceLines=1, nbLines=20, lineSize=2)),
    (120, 250, 3, dict(maxDisp=32, nIters=2, gradThreshold=30)),
    (64, 64, 4, dict(maxDisp=16, nIters=4, gradThreshold=20, costAggrWindowSize=5)),
    (100, 200, 5, dict(maxDisp=32, nIters=2, recordFullDisp=0)),
    (48, 80, 6, dict(maxDisp=16, nIters=8, gradThreshold=10)),
    (96, 160, 7, dict(maxDisp=32, nIters=0)),                          # no iteration: only the support points reach the map
    (64, 48, 8, dict(maxDisp=100, nIters=2, gradThreshold=20)),        # disparity range wider than the image
    (33, 47, 9, dict(maxDisp=16, nIters=2, gradThreshold=20)),         # crops to 32 x 32
])
def test_pipeline_matches_oracle(ctx, rows, cols, seed, over):
    import stereovision_amd as S
    from stereovision_amd import synth
    p = S.StereoParameters.reference_defaults(**over)
    left, right, _ = synth.stereo_pair(rows, cols, min(p.maxDisp, 64), seed)
    got_pts, got_map, want_pts, want_map, info = run_both(ctx, left, right, p)
    assert np.array_equal(got_pts, want_pts), info
    assert np.array_equal(got_map, want_map)
    # the run exercised the stages it is meant to cover
    assert info["n_high_grad"] > 0
    if rows >= 100:
        assert info["n_matches"] >= 3 and info["n_support"] >= info["n_matches"]
    # the part of the caller's map outside the 16-pixel crop is untouched
    assert np.all(got_map[16 * (rows // 16):] == 0) and np.all(got_map[:, 16 * (cols // 16):] == 0)


@pytest.mark.parametrize("rows,cols,seed,over", [
    (375, 1242, 21, dict(maxDisp=70, nIters=3)),
    (64, 64, 22, dict(maxDisp=16, nIters=4, gradThreshold=20, costAggrWindowSize=5)),
])
def test_pipeline_with_the_mesh_on_the_host(ctx, rows, cols, seed, over):
    """The growing host mesh (option SVX_OPT_MESH_ON_DEVICE = 0) and the triangulation on the device give the same frame."""
    import stereovision_amd as S
    from stereovision_amd import synth
    p = S.StereoParameters.reference_defaults(**over)
    left, right, _ = synth.stereo_pair(rows, cols, min(p.maxDisp, 64), seed)
    dev_pts, dev_map, want_pts, want_map, info = run_both(ctx, left, right, p)
    ctx.set_option(S.OPT_MESH_ON_DEVICE, 0)
    try:
        host_pts, host_map, _, _, _ = run_both(ctx, left, right, p)
    finally:
        ctx.set_option(S.OPT_MESH_ON_DEVICE, 1)
    assert np.array_equal(host_pts, want_pts) and np.array_equal(host_map, want_map), info
    assert np.array_equal(dev_pts, host_pts) and np.array_equal(dev_map, host_map)


def test_pipeline_point_buffer_smaller_than_the_points(ctx):
    """The caller's point buffer may hold fewer points than the frame computes: the first `capacity` of them come back (the
    high-gradient points, then the mesh vertices), the count is the whole number, the map does not depend on the buffer."""
    import stereovision_amd as S
    from stereovision_amd import synth
    p = S.StereoParameters.reference_defaults(maxDisp=32, nIters=2, gradThreshold=30)
    left, right, _ = synth.stereo_pair(120, 250, 32, 31)
    full_map = dev(np.zeros(left.shape, np.float32))
    full = ctx.high_perf_stereo(dev(left), dev(right), p, full_map)
    total = ctx.last_point_count
    assert total == len(full) > 200
    key = full[:, 1].astype(np.int64) * 4096 + full[:, 0]
    n_hg = int(np.flatnonzero(key[1:] < key[:-1])[0]) + 1          # the mesh vertices start where the raster order breaks
    for cap in (0, 1, 10, n_hg - 5, n_hg, n_hg + 7, total - 1, total, total + 100):
        m = dev(np.zeros(left.shape, np.float32))
        got = ctx.high_perf_stereo(dev(left), dev(right), p, m, capacity=cap)
        assert ctx.last_point_count == total
        assert len(got) == min(cap, total)
        assert np.array_equal(got, full[:min(cap, total)]), cap
        assert torch.equal(m, full_map)


def test_pipeline_recovers_the_scene(ctx):
    """On a pair generated from a known piecewise-planar field, most accepted high-gradient pixels carry that field."""
    import stereovision_amd as S
    from stereovision_amd import synth
    rows, cols = 375, 1242
    left, right, field = synth.stereo_pair(rows, cols, 64, seed=7)
    p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=3, gradThreshold=50)
    got_map = dev(np.zeros((rows, cols), np.float32))
    pts = ctx.high_perf_stereo(dev(left), dev(right), p, got_map)
    m = got_map.cpu().numpy()
    xs, ys = pts[:, 0], pts[:, 1]
    written = m[ys, xs] != 0
    assert written.sum() > 1000
    err = np.abs(m[ys, xs] - field[ys, xs])[written]
    assert (err <= 1.0).mean() > 0.8, (err <= 1.0).mean()


def test_pipeline_keeps_caller_values_and_errors(ctx):
    import stereovision_amd as S
    from stereovision_amd import synth
    left, right, _ = synth.stereo_pair(64, 96, 16, seed=8)
    p = S.StereoParameters.reference_defaults(maxDisp=16, nIters=1, gradThreshold=255)   # no pixel exceeds 255
    got_pts, got_map, want_pts, want_map, info = run_both(ctx, left, right, p, init=7.5)
    assert info["n_high_grad"] == 0
    assert np.array_equal(got_pts, want_pts) and np.array_equal(got_map, want_map)
    with pytest.raises(S.SvxError, match="same dimensions"):
        ctx.high_perf_stereo(dev(left), dev(right[:, :80]), p, dev(np.zeros(left.shape, np.float32)))
    with pytest.raises(S.SvxError, match="disparity map"):
        ctx.high_perf_stereo(dev(left), dev(right), p, dev(np.zeros((64, 80), np.float32)))
    with pytest.raises(S.SvxError, match="showImages"):
        ctx.high_perf_stereo(dev(left), dev(right), S.StereoParameters.reference_defaults(showImages=1), dev(np.zeros(left.shape, np.float32)))
    tiny = dev(np.zeros((8, 8), np.uint8))
    with pytest.raises(S.SvxError):
        ctx.high_perf_stereo(tiny, tiny, p, dev(np.zeros((8, 8), np.float32)))


def test_pipeline_batch_equals_single_calls(ctx):
    """svx_high_perf_stereo_batch (host threads with a context and stream each) against one call per frame: same maps, same
    points, for more frames than threads and for a second call that reuses the worker contexts."""
    import stereovision_amd as S
    from stereovision_amd import synth
    rows, cols, frames = 120, 250, 7
    p = S.StereoParameters.reference_defaults(maxDisp=32, nIters=2, gradThreshold=30)
    ls, rs = synth.stereo_batch(rows, cols, 32, frames, seed0=50)
    left, right = dev(ls), dev(rs)
    want_maps = torch.zeros((frames, rows, cols), dtype=torch.float32, device="cuda")
    want_pts = [ctx.high_perf_stereo(left[f], right[f], p, want_maps[f]) for f in range(frames)]
    for threads in (3, 2, 16):
        maps = torch.zeros((frames, rows, cols), dtype=torch.float32, device="cuda")
        pts = ctx.high_perf_stereo_batch(left, right, p, maps, n_threads=threads)
        assert torch.equal(maps, want_maps)
        assert len(pts) == frames and all(np.array_equal(a, b) for a, b in zip(pts, want_pts))
