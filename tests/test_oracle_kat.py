"""Pins the CPU oracle against the reference's known-answer values (tests/golden/survey_kat.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_kat.json")))


@pytest.fixture(scope="module")
def census_pair():
    left, right = O.kat_images()
    cl = O.census5x5(O.u8_to_s8(left))
    cr = O.census5x5(O.u8_to_s8(right))
    return cl, cr


def test_u8_to_s8_is_xor_0x80():
    a = np.arange(256, dtype=np.uint8)
    assert np.array_equal(O.u8_to_s8(a).view(np.uint8), a ^ 0x80)


def test_census_samples(census_pair):
    _, cr = census_pair
    for s in KAT["census_sse_right_samples"]:
        assert int(cr[s["y"], s["x"]]) == int(s["value"], 16), s


def test_census_fnv(census_pair):
    cl, cr = census_pair
    want = KAT["census_sse_fnv_rows2_29_cols2_61"]
    assert O.fnv(cr[2:30, 2:62].astype("<u4")) == want["right"]
    assert O.fnv(cl[2:30, 2:62].astype("<u4")) == want["left"]


def test_dense_wta(census_pair):
    cl, cr = census_pair
    k = KAT["dense_wta5x5"]
    d = O.dense_wta5x5(cl, cr, k["maxDisp"])
    for s in k["samples"]:
        assert d[s["y"], s["x"]] == s["d"], s
    assert O.fnv(d[4:28, 4:60].astype("<f4")) == k["fnv_rows4_27_cols4_59_f32"]


def test_window_match(census_pair):
    cl, cr = census_pair
    k = KAT["census_window_match"]
    rx, ry = k["ref_left_xy"]
    for px, py, want in k["vs_right_xy"]:
        assert O.window_cost(cl, cr, rx, ry, px, py) == want
    # extent probe: one word 0xFF at column x+dx of an otherwise-zero image
    e = k["extent_probe"]
    zero = np.zeros((16, 32), np.uint32)
    for dx in e["dx_inside"] + e["dx_outside"]:
        img = np.zeros((16, 32), np.uint32)
        img[8, 16 + dx] = e["word"]
        want = e["cost_inside"] if dx in e["dx_inside"] else 0
        assert O.window_cost(zero, img, 16, 8, 16, 8) == want, dx


def test_plane_prior_path(census_pair):
    cl, cr = census_pair
    k = KAT["plane_prior_path"]
    R, Cc = cl.shape
    yy, xx = np.mgrid[0:R, 0:Cc]
    prior = (np.float32(3.2) + np.float32(0.11) * yy.astype(np.float32)) + np.float32(0.004) * xx.astype(np.float32)
    prior = prior.astype(np.float32)
    pts = [(x, y) for y in range(2, R - 2) for x in range(2, Cc - 2) if (x + 2 * y) % 3 == 0]
    costs = np.full((R, Cc), k["costs_prefill"], np.int8)
    O.cost_eval(cl, cr, pts, prior, costs)
    assert O.fnv(costs) == k["fnv_costs_i8"]
    fd = np.zeros((R, Cc), np.float32)
    fc = np.full((R, Cc), k["finalCosts_init"], np.int8)
    g = O.refine(pts, prior, costs, k["tLow"], k["tHigh"], k["occGridSize"], fd, fc)
    assert O.fnv(fd) == k["fnv_finalDisp_f32"]
    assert O.fnv(fc) == k["fnv_finalCosts_i8"]
    assert (g["gw"], g["gh"]) == (k["grid_w"], k["grid_h"])
    touched = set()
    for c in k["cells"]:
        idx = c["j"] * g["gw"] + c["i"]
        touched.add(idx)
        assert [g["conf_x"][idx], g["conf_y"][idx], g["conf_c"][idx]] == [c["conf"][0], c["conf"][1], c["conf"][3]]
        assert abs(float(g["conf_d"][idx]) - c["conf"][2]) < 5e-5
        assert [g["inv_x"][idx], g["inv_y"][idx], g["inv_c"][idx]] == c["inv"]
    for e in k["epipolar"]:
        d, c = O.epipolar_match(cl, cr, e["censusSize"], e["aggr"], e["x"], e["y"], e["maxDisp"])
        assert (d, c) == (e["d"], e["cost"]), e


def test_banded_prior_search_radius_0_is_the_pinned_cost_evaluation(census_pair):
    """orc_cost_eval_band (a build extension) at radius 0 reproduces the reference's known-answer hash of costEvaluation and
    leaves the prior untouched; at radius 2 no point gets worse and the winners obey the definition (brute force here)."""
    cl, cr = census_pair
    k = KAT["plane_prior_path"]
    R, Cc = cl.shape
    yy, xx = np.mgrid[0:R, 0:Cc]
    prior = ((np.float32(3.2) + np.float32(0.11) * yy.astype(np.float32)) + np.float32(0.004) * xx.astype(np.float32)).astype(np.float32)
    pts = [(x, y) for y in range(2, R - 2) for x in range(2, Cc - 2) if (x + 2 * y) % 3 == 0]
    costs0, prior0 = np.full((R, Cc), k["costs_prefill"], np.int8), prior.copy()
    O.cost_eval_band(cl, cr, pts, prior0, costs0, 0, 16)
    assert O.fnv(costs0) == k["fnv_costs_i8"] and np.array_equal(prior0, prior)
    costs2, prior2 = np.full((R, Cc), k["costs_prefill"], np.int8), prior.copy()
    O.cost_eval_band(cl, cr, pts, prior2, costs2, 2, 16)
    d0_all = np.floor(prior.astype(np.float64) + 0.5).astype(np.int64)
    centre_ok = (xx - d0_all) >= 2                      # where costEvaluation itself evaluates the point
    assert (costs2 <= costs0)[centre_ok].all() and (costs2 < costs0).any()
    for x, y in pts[::7]:
        d0 = int(np.floor(np.float64(prior[y, x]) + 0.5))
        cands = [d for d in range(d0 - 2, d0 + 3) if x - d >= 2 and (d == d0 or 0 <= d < 16)]
        cost = {d: bin(int(cl[y, x]) ^ int(cr[y, x - d])).count("1") for d in cands}
        if not cands:
            assert costs2[y, x] == k["costs_prefill"]
            continue
        best = min(cands, key=lambda d: (cost[d], d))
        assert costs2[y, x] == cost[best]
        assert prior2[y, x] == (prior[y, x] if best == d0 else np.float32(best))


def test_sse_vs_generic_mapping():
    rng = np.random.default_rng(7)
    img = rng.integers(-128, 128, size=(40, 48), dtype=np.int8)
    sse = O.census5x5(img, "sse24")
    gen = O.census5x5(img, "generic25")
    g = gen[2:-2, 2:-2]
    assert np.array_equal(sse[2:-2, 2:-2], (g & 0xFFF) | ((g >> 13) << 12))
    assert int(sse.max()) < (1 << 24) and int(gen.max()) < (1 << 25)
    a, b = sse[2:-2, 2:-3], sse[2:-2, 3:-2]
    ga, gb = gen[2:-2, 2:-3], gen[2:-2, 3:-2]
    pc = np.vectorize(lambda v: bin(int(v)).count("1"))
    assert np.array_equal(pc(a ^ b), pc(ga ^ gb))


def test_popcount_against_reference_build():
    """The reference's own hammingdistance.cpp (compiled from where it lies into oracle/_ref)."""
    so = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libref_hamming.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = C.CDLL(so)
    ref.ref_hamming32.restype = C.c_uint
    ref.ref_hamming32.argtypes = [C.c_uint, C.c_uint]
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 2 ** 32, size=(4000, 2), dtype=np.uint64)
    for a, b in vals:
        assert ref.ref_hamming32(int(a), int(b)) == O.lib.orc_hamming32(int(a), int(b))
    for a, b in [(0, 0), (0, 0xFFFFFFFF), (0xFFFFFF, 0), (0x80000000, 1)]:
        assert ref.ref_hamming32(a, b) == O.lib.orc_hamming32(a, b) == bin(a ^ b).count("1")


def test_sparse_match_constant_disparity():
    """Property recorded from the reference: constant shift 7 => every lattice feature matched at d=7."""
    rng = np.random.default_rng(11)
    R, Cc, d = 64, 128, 7
    right = rng.integers(0, 256, size=(R, Cc), dtype=np.uint8)
    left = np.empty_like(right)
    left[:, d:] = right[:, :-d]
    left[:, :d] = rng.integers(0, 256, size=(R, d), dtype=np.uint8)
    cl = O.census5x5(O.u8_to_s8(left))
    cr = O.census5x5(O.u8_to_s8(right))
    rows = [10, 20, 30, 40]
    lf = [(x, y) for y in rows for x in (40, 55, 70, 85, 100)]
    rf = [(x, y) for y in rows for x in range(8, Cc - 8)]
    ol, orr, oc = O.sparse_match(cl, cr, lf, rf, 32, 1.0, 0.7, 1)
    assert len(ol) == len(lf)
    for i in range(len(ol)):
        assert orr[i] >= 0
        lx, ly = lf[ol[i]]
        rx, ry = rf[orr[i]]
        assert ly == ry and lx - rx == d and oc[i] == 0


def test_plane_fit_against_reference_build():
    """The reference's own CramerTheorem.cpp + Matrix.cpp (compiled from where they lie into oracle/_ref)."""
    so = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libref_cramer.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = C.CDLL(so)
    ref.ref_plane_fit.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(21)
    cases = []
    for _ in range(3000):
        xy = rng.integers(0, 1300, size=(3, 2)).astype(np.float64)
        d = rng.uniform(0, 128, size=(3, 1)).astype(np.float32).astype(np.float64)
        cases.append(np.concatenate([xy, d], 1))
    cases.append(np.array([[5, 5, 1.0], [10, 10, 2.0], [20, 20, 3.0]]))      # collinear -> singular
    cases.append(np.array([[3, 3, 1.0], [3, 3, 1.0], [9, 1, 4.0]]))          # repeated vertex
    cases.append(np.array([[0, 0, 7.0], [1, 0, 7.0], [0, 1, 7.0]]))          # constant plane
    for xyd in cases:
        xyd = np.ascontiguousarray(xyd)
        want = np.zeros(3, np.float32)
        ref.ref_plane_fit(xyd.ctypes.data, want.ctypes.data)
        got = O.plane_fit(xyd)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (xyd, got, want)
