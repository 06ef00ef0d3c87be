"""The support-point triangulation on the device (svx_delaunay_device) against the host triangulation (svx_delaunay,
itself checked against the Delaunay definition, Qhull and the reference's libelas triangulator in tests/test_mesh_cpu.py):
the two must return the same list of index triples, element for element, on random sets at several densities, on sets with
large empty areas, on lattices (every cell cocircular), collinear sets, repeated positions and tiny sets.
Parity status of the mesh itself: unpinned on reference outputs (Fade2D is a closed library)."""
import numpy as np
import pytest

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


def both(ctx, pts, rows, cols):
    import stereovision_amd as S
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
    want = S.delaunay(pts)
    got = ctx.delaunay_device(dev(pts), rows, cols).cpu().numpy()
    return got, want


def unique_random(rng, rows, cols, n):
    idx = rng.choice(rows * cols, size=min(n, rows * cols), replace=False)
    return np.stack([idx % cols, idx // cols], 1).astype(np.int32)


@pytest.mark.parametrize("rows,cols,n", [(368, 1232, 27000), (368, 1232, 3000), (368, 1232, 200), (375, 1242, 60000), (40, 50, 300),
                                         (16, 16, 40), (100, 100, 10000), (1072, 1920, 100000), (33, 2000, 900), (2000, 33, 900)])
def test_random_sets(ctx, rows, cols, n):
    rng = np.random.default_rng(rows * 7 + n)
    pts = unique_random(rng, rows, cols, n)
    got, want = both(ctx, pts, rows, cols)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_raster_order_like_the_pipeline(ctx):
    rng = np.random.default_rng(5)
    pts = unique_random(rng, 368, 1232, 27000)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
    got, want = both(ctx, pts, 368, 1232)
    assert np.array_equal(got, want)


def test_empty_areas_and_clusters(ctx):
    """Points in a few dense clusters with nothing between them: most triangles between the clusters reach far beyond a
    point's neighbourhood, so their points go through the pass over all points."""
    rng = np.random.default_rng(11)
    rows, cols = 368, 1232
    parts = []
    for cx, cy in [(100, 60), (600, 300), (1100, 50), (900, 200), (300, 330)]:
        p = np.stack([rng.integers(cx - 40, cx + 40, 1500), rng.integers(cy - 30, cy + 30, 1500)], 1)
        parts.append(p)
    pts = np.concatenate(parts)
    pts = pts[(pts[:, 0] >= 0) & (pts[:, 0] < cols) & (pts[:, 1] >= 0) & (pts[:, 1] < rows)]
    pts = np.unique(pts, axis=0)
    rng.shuffle(pts)
    got, want = both(ctx, pts, rows, cols)
    assert np.array_equal(got, want)
    assert ctx.mesh_device_stats()[0] > 0


@pytest.mark.parametrize("step", [1, 3, 4, 16, 17])
def test_lattice(ctx, step):
    """Every cell of a lattice is cocircular: the diagonals follow the rule on the indices, in any order of the list."""
    rows, cols = 90, 130
    ys, xs = np.mgrid[0:rows:step, 0:cols:step]
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    for seed in (None, 1):
        q = pts.copy()
        if seed is not None:
            np.random.default_rng(seed).shuffle(q)
        got, want = both(ctx, q, rows, cols)
        assert np.array_equal(got, want)


def test_collinear_tiny_and_repeated(ctx):
    rows, cols = 64, 200
    line = np.stack([np.arange(0, 200, 7), np.full(29, 20)], 1)
    got, want = both(ctx, line, rows, cols)
    assert len(want) == 0 and len(got) == 0
    for n in (0, 1, 2, 3, 4):
        pts = np.array([[5, 5], [100, 9], [50, 60], [180, 30]], np.int32)[:n]
        got, want = both(ctx, pts, rows, cols)
        assert np.array_equal(got, want)
    rng = np.random.default_rng(3)
    pts = unique_random(rng, rows, cols, 500)
    rep = np.concatenate([pts, pts[::3], pts[::5]])
    rng.shuffle(rep)
    got, want = both(ctx, rep, rows, cols)
    assert np.array_equal(got, want)
    # the line plus one point off it: a fan
    fan = np.concatenate([line, [[90, 50]]])
    got, want = both(ctx, fan, rows, cols)
    assert np.array_equal(got, want)


def test_many_small_configurations(ctx):
    """Sixty seeded point sets of random size, frame and shape (uniform, clustered, on a coarse lattice, with repeats)."""
    for seed in range(60):
        rng = np.random.default_rng(1000 + seed)
        rows, cols = int(rng.integers(8, 400)), int(rng.integers(8, 400))
        n = int(rng.integers(3, 4000))
        kind = seed % 4
        if kind == 0:
            pts = unique_random(rng, rows, cols, n)
        elif kind == 1:                                        # a few blobs
            c = rng.integers(0, [cols, rows], size=(4, 2))
            pts = c[rng.integers(0, 4, n)] + rng.normal(0, 6, size=(n, 2)).astype(np.int64)
            pts = pts[(pts[:, 0] >= 0) & (pts[:, 0] < cols) & (pts[:, 1] >= 0) & (pts[:, 1] < rows)]
        elif kind == 2:                                        # lattice of step 2..5 with holes
            step = int(rng.integers(2, 6))
            ys, xs = np.mgrid[0:rows:step, 0:cols:step]
            pts = np.stack([xs.ravel(), ys.ravel()], 1)
            pts = pts[rng.random(len(pts)) < 0.7]
        else:                                                  # repeats
            base = unique_random(rng, rows, cols, max(3, n // 3))
            pts = base[rng.integers(0, len(base), n)]
        pts = np.ascontiguousarray(pts, np.int32)
        rng.shuffle(pts)
        if len(pts) == 0:
            continue
        got, want = both(ctx, pts, rows, cols)
        assert np.array_equal(got, want), (seed, rows, cols, len(pts))


def test_fans(ctx):
    """A point next to a long line of collinear points is joined to every one of them: a ring of a hundred triangles on the
    device (the detector's border margin produces such lines along the frame), and a point that leads more triangles than
    the device form holds (the apex comes first in the list) is handed to the host triangulation by the prior entry point."""
    import stereovision_amd as S
    rows, cols = 300, 40
    line = np.stack([np.full(100, 20), np.arange(10, 210, 2)], 1)
    apex_last = np.concatenate([line, [[19, 111]]])
    got, want = both(ctx, apex_last, rows, cols)
    assert np.array_equal(got, want) and len(want) == 99
    apex_first = np.concatenate([[[19, 111]], line])
    with pytest.raises(S.SvxError):
        ctx.delaunay_device(dev(apex_first.astype(np.int32)), rows, cols)
    disp = np.random.default_rng(0).random((rows, cols)).astype(np.float32) * 9
    want_p, nt_h = ctx.mesh_plane_prior(apex_first, dev(disp), dev(np.zeros((rows, cols), np.float32)))
    got_p, nt_d = ctx.mesh_plane_prior_device(dev(apex_first.astype(np.int32)), dev(disp), dev(np.zeros((rows, cols), np.float32)))
    assert nt_h == nt_d == 99 and np.array_equal(got_p.cpu().numpy(), want_p.cpu().numpy())


def test_points_on_the_frame(ctx):
    rows, cols = 48, 80
    edge = [[x, 0] for x in range(0, cols, 3)] + [[x, rows - 1] for x in range(0, cols, 5)] + [[0, y] for y in range(1, rows - 1, 4)] + \
           [[cols - 1, y] for y in range(1, rows - 1, 2)]
    pts = np.array(edge + [[40, 24], [41, 24], [13, 30]], np.int32)
    got, want = both(ctx, pts, rows, cols)
    assert np.array_equal(got, want)


def test_point_outside_the_image_is_refused(ctx):
    import stereovision_amd as S
    pts = np.array([[1, 1], [10, 3], [5, 50], [7, 7]], np.int32)
    with pytest.raises(S.SvxError):
        ctx.delaunay_device(dev(pts), 40, 40)


def test_prior_from_device_mesh_equals_host_mesh(ctx):
    rng = np.random.default_rng(9)
    rows, cols = 368, 1232
    pts = unique_random(rng, rows, cols, 20000)
    disp = (rng.random((rows, cols)) * 60).astype(np.float32)
    want, nt_h = ctx.mesh_plane_prior(pts, dev(disp), dev(np.zeros((rows, cols), np.float32)))
    got, nt_d = ctx.mesh_plane_prior_device(dev(pts), dev(disp), dev(np.zeros((rows, cols), np.float32)))
    assert nt_h == nt_d
    assert np.array_equal(got.cpu().numpy(), want.cpu().numpy())


@pytest.mark.parametrize("rows,cols,n,kind", [(368, 1232, 5000, "random"), (120, 160, 2500, "random"), (64, 64, 1200, "lattice"),
                                              (368, 1232, 3000, "clustered"), (200, 300, 700, "border")])
def test_device_mesh_against_the_oracle_definition(ctx, rows, cols, n, kind):
    """The device triangulation checked by the ORACLE's definition checker (orc_delaunay_check: every triangle counter-clockwise
    with an empty circumcircle by exact brute force, the triangles cover the hull exactly once), not only against the host
    product's list: random sets, a lattice (every cell cocircular), clusters with empty areas, and points on the frame's
    border lines (long fans around the hull points)."""
    import oracle_lib as O
    rng = np.random.default_rng(n + rows)
    if kind == "lattice":
        xs, ys = np.meshgrid(np.arange(2, cols - 2, 2), np.arange(2, rows - 2, 2))
        pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
        pts = pts[rng.permutation(len(pts))[:n]]
    elif kind == "clustered":
        m = np.zeros((rows, cols), bool)
        for _ in range(12):
            cx, cy, r = rng.integers(0, cols), rng.integers(0, rows), rng.integers(15, 60)
            m[max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = True
        cand = np.flatnonzero(m)
        idx = rng.choice(cand, size=min(n, len(cand)), replace=False)
        pts = np.stack([idx % cols, idx // cols], 1).astype(np.int32)
    elif kind == "border":
        inner = unique_random(rng, rows, cols, n - 200)
        edge = np.concatenate([np.stack([rng.integers(0, cols, 50), np.zeros(50, int)], 1), np.stack([rng.integers(0, cols, 50), np.full(50, rows - 1)], 1),
                               np.stack([np.zeros(50, int), rng.integers(0, rows, 50)], 1), np.stack([np.full(50, cols - 1), rng.integers(0, rows, 50)], 1)])
        pts = np.unique(np.concatenate([inner, edge]).astype(np.int32), axis=0)
        pts = pts[rng.permutation(len(pts))]
    else:
        pts = unique_random(rng, rows, cols, n)
    got = ctx.delaunay_device(dev(pts), rows, cols).cpu().numpy()
    assert len(got) > 0
    assert O.delaunay_check(pts, got) == 0
