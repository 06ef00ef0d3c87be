"""Host-side Delaunay triangulation of libsvx (no GPU needed): checked against the definition by
the oracle's brute-force checker, and against scipy's Qhull triangulation where the triangulation
is unique (no four cocircular points)."""
import numpy as np
import pytest

import oracle_lib as O
import stereovision_amd as S


def canon(tri):
    """Orientation-free canonical form: sorted vertex triples, sorted list."""
    return sorted(tuple(sorted(map(int, t))) for t in tri)


def random_points(rng, n, w, h):
    return np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.int32)


@pytest.mark.parametrize("n,w,h", [(3, 10, 10), (4, 50, 50), (10, 100, 100), (60, 1242, 375), (300, 1242, 375), (1200, 1242, 375), (400, 40, 30)])
def test_delaunay_satisfies_definition(n, w, h):
    rng = np.random.default_rng(n * 7 + w)
    for trial in range(3):
        pts = random_points(rng, n, w, h)
        tri = S.delaunay(pts)
        assert O.delaunay_check(pts, tri) == 0


def test_delaunay_degenerate_inputs():
    assert len(S.delaunay(np.zeros((0, 2), np.int32))) == 0
    assert len(S.delaunay([[1, 1]])) == 0
    assert len(S.delaunay([[1, 1], [5, 9]])) == 0
    assert len(S.delaunay([[0, 0], [1, 1], [2, 2], [7, 7]])) == 0              # collinear
    assert len(S.delaunay([[3, 3]] * 5)) == 0                                   # one position
    # vertical and horizontal leading runs, both sides
    for pts in ([[0, 0], [0, 1], [0, 2], [0, 3], [2, 1]], [[0, 0], [0, 1], [0, 2], [2, -5]], [[0, 0], [1, 0], [2, 0], [3, 0], [3, 4]],
                [[0, 5], [0, 6], [1, 0]], [[0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [1, 2], [2, 0], [2, 1], [2, 2]]):
        pts = np.array(pts, np.int32)
        tri = S.delaunay(pts)
        assert len(tri) > 0 and O.delaunay_check(pts, tri) == 0, pts.tolist()
    # full integer lattice (every quad cocircular) and repeated positions
    y, x = np.mgrid[0:12, 0:17]
    grid = np.stack([x.ravel(), y.ravel()], 1).astype(np.int32)
    tri = S.delaunay(grid)
    assert len(tri) == 2 * 11 * 16 and O.delaunay_check(grid, tri) == 0
    rng = np.random.default_rng(0)
    dup = np.concatenate([grid[rng.permutation(len(grid))], grid[:40]])
    tri = S.delaunay(dup)
    assert O.delaunay_check(dup, tri) == 0
    assert tri.max() < len(grid)                                               # first occurrence represents a position


def test_delaunay_output_is_canonical():
    rng = np.random.default_rng(5)
    pts = random_points(rng, 200, 300, 200)
    tri = S.delaunay(pts)
    assert np.all(tri[:, 0] < tri[:, 1]) and np.all(tri[:, 0] < tri[:, 2])
    keys = [tuple(t) for t in tri.tolist()]
    assert keys == sorted(keys)
    # same orientation for all triangles
    a, b, c = pts[tri[:, 0]].astype(np.int64), pts[tri[:, 1]].astype(np.int64), pts[tri[:, 2]].astype(np.int64)
    o = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    assert np.all(o > 0)


def test_delaunay_matches_qhull_in_general_position():
    scipy_spatial = pytest.importorskip("scipy.spatial")
    rng = np.random.default_rng(11)
    for n in (20, 150, 800):
        # a large coordinate range makes cocircular quadruples and collinear hull triples very unlikely;
        # the checker would also flag a non-Delaunay result independently
        pts = np.unique(random_points(rng, n, 1 << 20, 1 << 20), axis=0)
        tri = S.delaunay(pts)
        assert O.delaunay_check(pts, tri) == 0
        ref = scipy_spatial.Delaunay(pts.astype(np.float64)).simplices
        assert canon(tri) == canon(ref)


def _reference_triangle():
    """The Delaunay triangulator inside the reference's libelas copy (elas/triangle.cpp), compiled from where it lies into
    oracle/_ref by oracle/Makefile and called as elas/elas.cpp:445-494 calls it."""
    import ctypes as C
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libref_triangle.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    lib = C.CDLL(so)
    lib.ref_triangle_delaunay.restype = C.c_int
    lib.ref_triangle_delaunay.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]

    def run(pts):
        xy = np.ascontiguousarray(pts, dtype=np.float32)
        out = np.zeros((2 * len(pts) + 8, 3), np.int32)
        n = lib.ref_triangle_delaunay(xy.ctypes.data, len(xy), out.ctypes.data, len(out))
        assert 0 <= n <= len(out)
        return out[:n]
    return run


def test_delaunay_against_the_reference_triangulator():
    """In general position the triangulation is unique: the same triangles as the reference's own triangulator.  With
    cocircular points (image-sized integer coordinates) the diagonals may differ, but the reference's result passes the
    same checker and has the same number of triangles."""
    ref = _reference_triangle()
    rng = np.random.default_rng(21)
    for n in (10, 300, 2000):
        pts = np.unique(random_points(rng, n, 1 << 20, 1 << 20), axis=0)
        assert canon(S.delaunay(pts)) == canon(ref(pts))
    for n, w, h in ((500, 1242, 375), (5000, 1242, 375), (300, 40, 30)):
        pts = np.unique(random_points(rng, n, w, h), axis=0)
        mine, theirs = S.delaunay(pts), ref(pts)
        assert O.delaunay_check(pts, theirs) == 0
        assert len(mine) == len(theirs)
        shared = len(set(canon(mine)) & set(canon(theirs)))
        assert shared >= 0.75 * len(mine)                                      # the rest are cocircular quadruples cut the other way


def test_checker_rejects_wrong_meshes():
    pts = np.array([[0, 0], [10, 0], [10, 10], [0, 10], [3, 4]], np.int32)
    good = S.delaunay(pts)
    assert O.delaunay_check(pts, good) == 0
    assert O.delaunay_check(pts, good[:-1]) != 0                                # hole
    assert O.delaunay_check(pts, np.array([[0, 1, 2], [0, 2, 3]], np.int32)) != 0   # point 4 unused
    quad = np.array([[0, 0], [10, 0], [11, 5], [0, 4]], np.int32)
    a = np.array([[0, 1, 2], [0, 2, 3]], np.int32); b = np.array([[0, 1, 3], [1, 2, 3]], np.int32)
    assert sorted([O.delaunay_check(quad, a) == 0, O.delaunay_check(quad, b) == 0]) == [False, True]


def test_delaunay_rejects_out_of_range():
    with pytest.raises(S.SvxError):
        S.delaunay([[0, 0], [1 << 25, 3], [5, 5]])


def _grown(pts, first, chunk):
    m = S.Mesh()
    m.insert(pts[:first])
    for i in range(first, len(pts), chunk):
        m.insert(pts[i:i + chunk])
    tri = m.triangles()
    assert m.counts()[0] == len(pts)
    m.close()
    return tri


@pytest.mark.parametrize("n,w,h,first,chunk", [(400, 1242, 375, 300, 7), (400, 40, 30, 10, 1), (2000, 1242, 375, 1500, 100),
                                                (300, 12, 12, 3, 5), (500, 1 << 20, 1 << 20, 50, 50), (60, 9, 2, 4, 1),
                                                (200, 200, 1, 150, 10), (200, 30, 30, 0, 1), (100, 50, 50, 2, 1)])
def test_mesh_grown_equals_triangulation_from_scratch(n, w, h, first, chunk):
    """The mesh is a function of the point list alone: however the list is split over insertions (points inside
    triangles, on edges, on and outside the hull, repeats, cocircular grids, collinear starts), the triangles are those of
    delaunay() on the whole list."""
    rng = np.random.default_rng(n + w + first)
    for rep in range(3):
        pts = random_points(rng, n, w, h)              # with repeats on the small grids
        want = S.delaunay(pts)
        assert O.delaunay_check(pts, want) == 0 or h == 1
        got = _grown(pts, first, chunk)
        assert np.array_equal(got, want)


def test_mesh_grown_on_a_grid_and_outwards():
    # a full grid (every square cocircular), inserted inside-out so that every later point lies outside the hull
    ys, xs = np.mgrid[0:12, 0:17]
    grid = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    order = np.argsort(((grid - grid.mean(0)) ** 2).sum(1), kind="stable")
    pts = grid[order]
    want = S.delaunay(pts)
    assert len(want) == 2 * 11 * 16 and O.delaunay_check(pts, want) == 0
    for first, chunk in ((3, 1), (20, 13), (len(pts), 1)):
        assert np.array_equal(_grown(pts, first, chunk), want)
    # and in reverse: points inside the hull of the first ones
    rev = pts[::-1].copy()
    assert np.array_equal(_grown(rev, 60, 9), S.delaunay(rev))


def test_delaunay_is_independent_of_the_order_of_construction():
    """Same point set, ids permuted: the triangle SETS agree when the tie rule sees the same ids, i.e. the canonical
    triangulation depends on the list, and relabelling the list relabels the result."""
    rng = np.random.default_rng(77)
    pts = np.unique(random_points(rng, 300, 25, 25), axis=0)
    tri = S.delaunay(pts)
    # rotate the picture by 90 degrees (an isometry keeps cocircularity and the ids): the same index triples
    rot = np.stack([-pts[:, 1], pts[:, 0]], 1).astype(np.int32)
    assert canon(S.delaunay(rot)) == canon(tri)
