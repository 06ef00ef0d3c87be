"""CPU checks of the oracle's restatements of the steps either side of the census path, against
hand-computed values and against independent numpy formulations of the same definitions."""
import numpy as np

import oracle_lib as O


def test_equalize_hist_hand_example():
    # hist: value 0 x2, 1 x1, 2 x1 -> i0 = 0, scale = 255/2: lut[1] = round_half_even(127.5) = 128, lut[2] = 255
    img = np.array([[0, 0, 1, 2]], np.uint8)
    assert O.equalize_hist(img).tolist() == [[0, 0, 128, 255]]
    assert O.equalize_hist(np.full((3, 3), 9, np.uint8)).tolist() == [[9] * 3] * 3


def test_laplacian_hand_example():
    img = np.array([[10, 20, 30], [40, 50, 60], [70, 80, 200]], np.uint8)
    # ksize 1 at the centre: 20 + 40 + 60 + 80 - 4*50 = 0; ksize 3: 2*(10+30+70+200) - 8*50 = 220
    assert O.laplacian_abs(img, 1)[1, 1] == 0
    assert O.laplacian_abs(img, 3)[1, 1] == 220
    # corner (0,0), reflect-101: up = row 1, left = col 1 -> ksize 1: 40 + 20 + 20 + 40 - 40 = 80
    assert O.laplacian_abs(img, 1)[0, 0] == 80
    # saturation of |v| to 255 and of v to int16
    assert O.laplacian_abs(img, 3, 2, 0)[1, 1] == 255
    big = np.zeros((3, 3), np.uint8); big[1, 1] = 255
    assert O.laplacian_abs(big, 3, 100, 0)[1, 1] == 255


def test_laplacian_matches_numpy_convolution():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(23, 31), dtype=np.uint8)
    pad = np.pad(img.astype(np.int64), 1, mode="reflect")
    k = np.array([[2, 0, 2], [0, -8, 0], [2, 0, 2]])
    want = sum(k[i, j] * pad[i:i + 23, j:j + 31] for i in range(3) for j in range(3))
    assert np.array_equal(O.laplacian_abs(img, 3), np.minimum(np.abs(want), 255).astype(np.uint8))


def test_high_gradient_points_order_and_border():
    g = np.zeros((8, 9), np.uint8)
    g[:] = 60
    pts = O.high_gradient_points(g, 50)
    want = [(x, y) for y in range(2, 6) for x in range(2, 7)]
    assert [tuple(p) for p in pts] == want
    assert len(O.high_gradient_points(g, 60)) == 0          # strictly greater


def test_gaussian_blur_matches_separable_definition():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(17, 19), dtype=np.uint8)
    assert np.array_equal(O.gaussian_blur(img, 1), img)
    for k, w in ((3, [1, 2, 1]), (5, [1, 4, 6, 4, 1])):
        h = k // 2
        pad = np.pad(img.astype(np.int64), h, mode="reflect")
        acc = sum(w[i] * w[j] * pad[i:i + 17, j:j + 19] for i in range(k) for j in range(k))
        norm = sum(w) ** 2
        assert np.array_equal(O.gaussian_blur(img, k), ((acc + norm // 2) // norm).astype(np.uint8))


def test_alter_rows_hand_example():
    img = np.full((10, 2), 10, np.uint8)
    # nb_rows 3 -> row_size 4, 2 bands: band 0 (rows 0-3) inverted, last band 1 (rows 4-9) not
    out = O.alter_rows(img, True, 3, False, 1, 1)
    assert out[:4].tolist() == [[245, 245]] * 4 and out[4:].tolist() == [[10, 10]] * 6
    # nb_lines 2 -> step 5, lines start at rows 2 and 7, 2 rows each
    out = O.alter_rows(img, False, 1, True, 2, 2)
    assert [int(r[0]) for r in out] == [10, 10, 0, 0, 10, 10, 10, 0, 0, 10]


def test_arc_rule_matches_brute_force():
    for m in list(range(0, 1 << 16, 7)) + [0xFFFF, 0x01FF, 0xFF80, 0xF01F, 0x00FF, 0xFE01]:
        bits = [(m >> i) & 1 for i in range(16)]
        brute = any(all(bits[(s + j) % 16] for j in range(9)) for s in range(16))
        assert bool(O.lib.orc_arc9(m)) == brute, hex(m)


def test_sparse_nonmax_is_the_8_neighbour_rule():
    """The list walk of fast9-inl.h:76-173 keeps exactly the corners with no listed 8-neighbour of
    greater or equal score, which is the dense form the GPU kernel evaluates."""
    rng = np.random.default_rng(3)
    for trial in range(40):
        rows, cols = int(rng.integers(4, 30)), int(rng.integers(4, 40))
        density = rng.choice([0.05, 0.3, 0.7, 1.0])
        mask = rng.random((rows, cols)) < density
        score = rng.integers(0, 6 if trial % 2 else 256, size=(rows, cols)).astype(np.uint8)
        ys, xs = np.nonzero(mask)
        xy = np.stack([xs, ys], 1).astype(np.int32)          # np.nonzero is row-major = raster order
        keep = O.sparse_nonmax(xy, score[ys, xs])
        smap = np.where(mask, score.astype(np.int32) + 1, 0)
        pad = np.pad(smap, 1)
        nb = np.max([pad[1 + dy:1 + dy + rows, 1 + dx:1 + dx + cols] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dy or dx], axis=0)
        want = (nb < smap)[ys, xs]
        assert np.array_equal(keep, want), trial


def test_exfast_properties():
    rng = np.random.default_rng(4)
    y, x = np.mgrid[0:60, 0:80]
    img = np.where((x % 20 < 8) & (y % 15 < 6), 200, 40).astype(np.uint8)      # isolated bright rectangles
    img = np.clip(img + rng.integers(-5, 6, size=img.shape), 0, 255).astype(np.uint8)
    all_xy, all_resp = O.exfast_detect(img, 10, 1.0, nonmax=False)
    nm_xy, nm_resp = O.exfast_detect(img, 10, 1.0, nonmax=True)
    assert len(all_xy) > len(nm_xy) > 0
    assert np.all(all_resp == -1000.0)
    assert set(map(tuple, nm_xy)) <= set(map(tuple, all_xy))
    for a in (all_xy, nm_xy):                                 # raster order, 3-pixel border
        keys = a[:, 1] * 80 + a[:, 0]
        assert np.all(np.diff(keys) > 0)
        assert a[:, 0].min() >= 3 and a[:, 0].max() < 77 and a[:, 1].min() >= 3 and a[:, 1].max() < 57
    assert len(O.exfast_detect(np.full((40, 40), 128, np.uint8), 10, 1.0)[0]) == 0


def test_right_view_is_the_mirrored_left_view():
    """Definition check on the oracle itself: right-view map == mirror(left-view map of the mirrored, swapped pair)."""
    rng = np.random.default_rng(4)
    cl = rng.integers(0, 1 << 32, size=(20, 70, 2), dtype=np.uint64).astype(np.uint32) & np.uint32(0x11111111)
    cr = np.roll(cl, -4, axis=1)
    right = O.dense_wta_words(cl, cr, 24, view=1)
    mirrored = O.dense_wta_words(cr[:, ::-1].copy(), cl[:, ::-1].copy(), 24, view=0)[:, ::-1]
    assert np.array_equal(right, mirrored)


def test_census9x7_hand_example():
    img = np.full((9, 11), 100, np.uint8)
    img[1, 1] = 5                      # neighbour (dy,dx) = (-3,-4) of pixel (4,5): index 0 -> bit 62
    img[4, 6] = 7                      # neighbour (0,+1): index 3*9+5 = 32 -> bit 30
    c = O.census9x7(img)
    assert c[4, 5, 1] == 1 << (62 - 32) and c[4, 5, 0] == 1 << 30
    assert c[0].sum() == 0 and c[:, :4].sum() == 0 and c[:, -4:].sum() == 0 and c[-3:].sum() == 0


def test_lr_check_hand_example():
    dl = np.zeros((5, 12), np.float32); dr = np.zeros((5, 12), np.float32)
    dl[2, 8] = 3; dr[2, 5] = 3          # consistent
    dl[2, 9] = 4; dr[2, 5] = 3          # off by one: kept at threshold 1, dropped at 0
    dl[2, 3] = 2                         # x_r = 1 < 2: cannot be confirmed
    out1 = O.lr_check(dl, dr, 1.0)
    assert out1[2, 8] == 3 and out1[2, 9] == 4 and out1[2, 3] == -1
    out0 = O.lr_check(dl, dr, 0.0)
    assert out0[2, 9] == -1 and out0[0, 0] == 0


def test_lr_check_treats_what_is_no_disparity_as_invalid():
    """Negative values (an earlier pass's markers), NaN, infinities and values beyond the row are invalidated without a read;
    a second pass over a checked map therefore leaves it as it is."""
    rows, cols = 7, 20
    dl = np.full((rows, cols), 2.0, np.float32); dr = np.full((rows, cols), 2.0, np.float32)
    for x, v in zip(range(4, 12), [-1.0, -10.0, np.nan, 1e9, np.inf, -np.inf, float(cols), -0.5]):
        dl[3, x] = v
    out = O.lr_check(dl, dr, 1.0, -1.0)
    assert np.all(out[3, 4:12] == -1.0)
    assert out[3, 12] == 2.0 and out[3, 3] == -1.0          # x_r = 1 < 2 cannot be confirmed
    assert np.array_equal(O.lr_check(out, dr, 1.0, -1.0), out)
    d = np.array([[1.0, np.nan, 1e9, -3.0]], np.float32).repeat(6, 0).repeat(4, 1)
    c = np.zeros((6, 16, 2), np.uint32)
    assert np.array_equal(O.subpixel_refine(c, c, 8, d), d, equal_nan=True)
