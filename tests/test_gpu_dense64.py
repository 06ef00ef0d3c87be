"""GPU parity tests of the 64-bit (9x7) census, the dense search in either view on one- and two-word
descriptors, and the left-right check, through the C ABI, bit-exact against the CPU oracle.  These are
build-defined generalisations (the reference has neither a 9x7 census nor a dense left-right check), so
the oracle is the definition; the one-word left view is additionally tied to the reference-pinned a8."""
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


def dev_u32(a):
    return dev(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32))


@pytest.mark.parametrize("rows,cols", [(32, 64), (7, 9), (6, 8), (37, 129), (48, 256), (50, 261), (1, 1), (16, 70), (375, 1242), (19, 1023),
                                       (40, 514), (33, 10), (8, 11), (368, 1232), (7, 12), (23, 260), (17, 1028)])
@pytest.mark.parametrize("form", [2, 1, 0])
def test_census9x7(ctx, rows, cols, form):
    """Every form of the tile kernel (three comparisons per subtraction, two per subtraction, one per instruction; widths that
    are no multiple of 4 run form 1 in place of form 2) against the oracle, on noise, on an image of few levels (many ties)
    and on one that holds the extreme values 0 and 255 side by side."""
    import stereovision_amd as S
    rng = np.random.default_rng(rows * 100 + cols)
    imgs = rng.integers(0, 256, size=(3, rows, cols), dtype=np.uint8)
    imgs[1] = (imgs[1] // 64) * 64                                        # many ties
    imgs[2] = np.where(rng.random((rows, cols)) < 0.5, 0, 255).astype(np.uint8)
    imgs[2, ::3, ::5] = 254; imgs[2, 1::4, 2::3] = 1
    ctx.set_option(S.OPT_CENSUS97_PACKED, form)
    try:
        got = ctx.census9x7(dev(imgs)).cpu().numpy().view(np.uint32)
    finally:
        ctx.set_option(S.OPT_CENSUS97_PACKED, 2)
    for f in range(3):
        assert np.array_equal(got[f], O.census9x7(imgs[f])), (f, np.argwhere(got[f] != O.census9x7(imgs[f]))[:5])


@pytest.mark.parametrize("rows,cols,max_disp", [(24, 64, 16), (12, 40, 64), (30, 150, 128), (9, 21, 5), (5, 20, 8), (16, 300, 200),
                                                 (110, 90, 70), (8, 19, 4), (4, 30, 4), (20, 70, 1100)])
@pytest.mark.parametrize("words", [1, 2])
@pytest.mark.parametrize("view,matrix_cores", [(0, -1), (0, 0), (1, -1)])     # left view: matrix-core search and vector sweep; right view: sweep only
def test_dense_wta_matches_oracle(ctx, rows, cols, max_disp, words, view, matrix_cores):
    import stereovision_amd as S
    rng = np.random.default_rng(rows * 1000 + cols + max_disp + words)
    shape = (2, rows, cols) + ((2,) if words == 2 else ())
    # few distinct values give many cost ties, which exercise the first-minimum rule in both views
    cl = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x01010101 if cols % 2 else 0xffffffff)
    cr = np.roll(cl, -3, axis=2) ^ (rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x00010000))
    ctx.set_option(S.OPT_MATRIX_CORES, matrix_cores)
    try:
        got = ctx.dense_wta(dev_u32(cl), dev_u32(cr), max_disp, view=view, words=words).cpu().numpy()
    finally:
        ctx.set_option(S.OPT_MATRIX_CORES, -1)
    for f in range(2):
        want = O.dense_wta_words(cl[f], cr[f], max_disp, view)
        assert np.array_equal(got[f], want), (f, np.argwhere(got[f] != want)[:5])


@pytest.mark.parametrize("rows,cols,max_disp", [(24, 64, 16), (12, 40, 64), (30, 150, 128), (9, 21, 5), (5, 20, 8), (16, 300, 200),
                                                 (110, 90, 70), (8, 19, 4), (4, 30, 4), (20, 70, 1100), (40, 333, 128), (7, 1232, 128)])
@pytest.mark.parametrize("words", [1, 2])
@pytest.mark.parametrize("matrix_cores", [-1, 0])
def test_dense_wta_both_views_from_one_sweep(ctx, rows, cols, max_disp, words, matrix_cores):
    """The right view taken from the left sweep's cost values equals the oracle's right-view search (ties included), with
    the matrix-core search and with the vector sweep."""
    import stereovision_amd as S
    rng = np.random.default_rng(rows * 1000 + cols + max_disp + words + 5)
    shape = (2, rows, cols) + ((2,) if words == 2 else ())
    cl = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x01010101 if cols % 2 else 0xffffffff)
    cr = np.roll(cl, -3, axis=2) ^ (rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x00010000))
    ctx.set_option(S.OPT_MATRIX_CORES, matrix_cores)
    try:
        dl, dr = ctx.dense_wta_both(dev_u32(cl), dev_u32(cr), max_disp, words=words)
    finally:
        ctx.set_option(S.OPT_MATRIX_CORES, -1)
    dl, dr = dl.cpu().numpy(), dr.cpu().numpy()
    for f in range(2):
        wl, wr = O.dense_wta_words(cl[f], cr[f], max_disp, 0), O.dense_wta_words(cl[f], cr[f], max_disp, 1)
        assert np.array_equal(dl[f], wl), (f, np.argwhere(dl[f] != wl)[:5])
        assert np.array_equal(dr[f], wr), (f, np.argwhere(dr[f] != wr)[:5], dr[f][dr[f] != wr][:5], wr[dr[f] != wr][:5])


@pytest.mark.parametrize("rows,cols,max_disp", [(24, 64, 16), (12, 40, 64), (30, 150, 128), (5, 36, 1), (110, 90, 70), (9, 300, 100),
                                                 (101, 37, 33), (7, 200, 128), (40, 133, 65), (20, 300, 200), (12, 400, 256), (9, 290, 129),
                                                 (8, 700, 512), (6, 600, 300), (7, 1300, 1024)])
def test_dense_wta_matrix_matches_oracle(ctx, rows, cols, max_disp):
    """The matrix-core form: both views bit for bit, on descriptors with few distinct values (many cost ties)."""
    rng = np.random.default_rng(rows * 1000 + cols + max_disp)
    shape = (2, rows, cols, 2)
    cl = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x01010101 if cols % 2 else 0xffffffff)
    cr = np.roll(cl, -3, axis=2) ^ (rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x00010000))
    dl, dr = ctx.dense_wta_matrix(dev_u32(cl), dev_u32(cr), max_disp)
    dl0, _ = ctx.dense_wta_matrix(dev_u32(cl), dev_u32(cr), max_disp, both=False)
    dl, dr, dl0 = dl.cpu().numpy(), dr.cpu().numpy(), dl0.cpu().numpy()
    for f in range(2):
        want_l, want_r = O.dense_wta_words(cl[f], cr[f], max_disp, 0), O.dense_wta_words(cl[f], cr[f], max_disp, 1)
        assert np.array_equal(dl[f], want_l), (f, np.argwhere(dl[f] != want_l)[:5])
        assert np.array_equal(dl0[f], want_l), (f, np.argwhere(dl0[f] != want_l)[:5])
        assert np.array_equal(dr[f], want_r), (f, np.argwhere(dr[f] != want_r)[:5])


@pytest.mark.parametrize("rows,cols,max_disp", [(40, 420, 64), (23, 300, 128), (30, 330, 256), (12, 900, 400)])
def test_dense_wta_matrix_close_costs(ctx, rows, cols, max_disp):
    """Descriptors of real image texture: neighbouring disparities cost almost the same, so one wrong operand anywhere in
    a window moves the winner (random descriptors with one planted match would not notice)."""
    from stereovision_amd import synth
    left, right, _ = synth.stereo_pair(rows, cols, min(max_disp, cols // 3), seed=rows)
    cl, cr = O.census9x7(left), O.census9x7(right)
    dl, dr = ctx.dense_wta_matrix(dev_u32(cl), dev_u32(cr), max_disp)
    want_l, want_r = O.dense_wta_words(cl, cr, max_disp, 0), O.dense_wta_words(cl, cr, max_disp, 1)
    assert np.array_equal(dl.cpu().numpy(), want_l), np.argwhere(dl.cpu().numpy() != want_l)[:5]
    assert np.array_equal(dr.cpu().numpy(), want_r), np.argwhere(dr.cpu().numpy() != want_r)[:5]


@pytest.mark.parametrize("rows,cols,max_disp,frames", [(24, 64, 16, 2), (40, 420, 64, 1), (23, 300, 128, 2), (12, 900, 400, 1), (9, 40, 5, 1),
                                                        (368, 1232, 128, 2), (480, 640, 64, 1)])
def test_one_word_search_on_the_matrix_cores(ctx, rows, cols, max_disp, frames):
    """One-word (the reference's 24-bit 5x5) descriptors through the matrix-core search (two window pixels per instruction):
    left view, and both views from one pass, against the oracle (small frames) or the vector sweep (large ones), and the
    fused u8 entry point of the reference-pinned path."""
    import stereovision_amd as S
    from stereovision_amd import synth
    ls, rs = synth.stereo_batch(rows, cols, min(max_disp, max(2, cols // 3)), frames, seed0=rows + max_disp)
    cl = np.stack([O.census5x5(O.u8_to_s8(a)) for a in ls]); cr = np.stack([O.census5x5(O.u8_to_s8(a)) for a in rs])
    dcl, dcr = dev_u32(cl), dev_u32(cr)
    ctx.set_option(S.OPT_MATRIX_CORES, 0)
    try:
        assert S.dense_kernel_name(rows, cols, 1, max_disp, False, ctx) != "dense_wta_mx_kernel"
        wl, wr = ctx.dense_wta(dcl, dcr, max_disp, S.VIEW_LEFT, 1), ctx.dense_wta(dcl, dcr, max_disp, S.VIEW_RIGHT, 1)
    finally:
        ctx.set_option(S.OPT_MATRIX_CORES, -1)
    if cols >= 36 and rows >= 5:
        assert S.dense_kernel_name(rows, cols, 1, max_disp, False, ctx) == "dense_wta_mx_kernel"
    ml = ctx.dense_wta(dcl, dcr, max_disp, S.VIEW_LEFT, 1)
    bl, br = ctx.dense_wta_both(dcl, dcr, max_disp, words=1)
    fused = ctx.dense_census_wta(dev(ls), dev(rs), max_disp)
    assert torch.equal(ml, wl), (ml != wl).nonzero()[:5].tolist()
    assert torch.equal(bl, wl) and torch.equal(br, wr), ((bl != wl).nonzero()[:3].tolist(), (br != wr).nonzero()[:3].tolist())
    assert torch.equal(fused, wl)
    if rows * cols <= 40000:
        for f in range(frames):
            assert np.array_equal(ml[f].cpu().numpy(), O.dense_wta_words(cl[f], cr[f], max_disp, 0))
            assert np.array_equal(br[f].cpu().numpy(), O.dense_wta_words(cl[f], cr[f], max_disp, 1))
    else:
        y0 = rows // 2
        assert np.array_equal(ml[0].cpu().numpy()[y0:y0 + 2], O.dense_wta_words(cl[0], cr[0], max_disp, 0, y0, y0 + 2)[y0:y0 + 2])


def test_dense_wta_matrix_refuses_what_it_does_not_cover(ctx):
    import stereovision_amd as S
    cl = dev_u32(np.zeros((1, 8, 30, 2), np.uint32))
    with pytest.raises(S.SvxError):
        ctx.dense_wta_matrix(cl, cl, 16)
    cl = dev_u32(np.zeros((1, 8, 64, 2), np.uint32))
    with pytest.raises(S.SvxError):
        ctx.dense_wta_matrix(cl, cl, 1025)


def test_left_view_one_word_is_the_reference_search(ctx):
    """svx_dense_wta(words=1, left view) == the reference-pinned a8 restatement on its interior."""
    rng = np.random.default_rng(3)
    cl = rng.integers(0, 1 << 24, size=(40, 120), dtype=np.uint32)
    cr = np.roll(cl, -5, axis=1)
    got = ctx.dense_wta(dev_u32(cl), dev_u32(cr), 32).cpu().numpy()
    want = O.dense_wta5x5(cl, cr, 32)
    assert np.array_equal(got[2:-2, 2:-2], want[2:-2, 2:-2])
    assert np.array_equal(O.dense_wta_words(cl, cr, 32)[2:-2, 2:-2], want[2:-2, 2:-2])


@pytest.mark.parametrize("rows,cols", [(30, 100), (5, 5), (64, 333)])
@pytest.mark.parametrize("threshold", [0.0, 1.0, 2.5])
def test_lr_check(ctx, rows, cols, threshold):
    rng = np.random.default_rng(rows + cols)
    dl = rng.integers(0, 20, size=(2, rows, cols)).astype(np.float32)
    dr = rng.integers(0, 20, size=(2, rows, cols)).astype(np.float32)
    got = ctx.lr_check(dev(dl), dev(dr), threshold, invalid=-7.0).cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], O.lr_check(dl[f], dr[f], threshold, -7.0))


def test_lr_check_values_that_are_no_disparity(ctx):
    """Left maps holding an earlier pass's invalid markers, NaN or huge values: such pixels become invalid and nothing is
    read for them (a negative value used to address the next row, a huge one memory past the frame)."""
    rng = np.random.default_rng(11)
    rows, cols = 24, 70
    dl = rng.integers(0, 20, size=(3, rows, cols)).astype(np.float32)
    dr = rng.integers(0, 20, size=(3, rows, cols)).astype(np.float32)
    bad = np.array([-1.0, -10.0, np.nan, 1e9, -0.5, np.inf, -np.inf, float(cols), float(cols) - 1.0, 3e38], np.float32)
    idx = rng.integers(0, dl.size, size=400)
    dl.reshape(-1)[idx] = bad[rng.integers(0, len(bad), size=400)]
    dl[2, -3:, -6:] = 1e9                       # the last pixels of the last frame
    dl[2, -3:, :4] = -10.0
    got = ctx.lr_check(dev(dl), dev(dr), 1.0, invalid=-7.0).cpu().numpy()
    for f in range(3):
        want = O.lr_check(dl[f], dr[f], 1.0, -7.0)
        assert np.array_equal(got[f], want, equal_nan=True)
        inner = got[f][2:-2, 2:-2]
        src = dl[f][2:-2, 2:-2]
        assert np.all(inner[~((src >= 0) & (src < cols))] == -7.0)


@pytest.mark.parametrize("rows,cols,D,frames", [(375, 1242, 64, 2), (368, 1232, 128, 2), (1072, 1920, 200, 1), (540, 960, 256, 1),
                                                 (300, 2000, 33, 1), (300, 2000, 65, 1), (300, 2000, 129, 1), (270, 1920, 512, 1), (200, 1600, 300, 2),
                                                 (120, 2400, 700, 1), (90, 2600, 1024, 1)])
def test_matrix_search_equals_vector_sweep_on_whole_frames(ctx, rows, cols, D, frames):
    """Every tile count of the matrix-core search (3, 5 and 9 tiles: up to 64, 128 and 256 disparities), both views, on
    whole large frames against the vector sweep (itself checked against the oracle on bands of such frames): a build of
    the matrix-core kernel that spills, or whose hand-placed instructions no longer fit the compiler's, shows up here."""
    import stereovision_amd as S
    from stereovision_amd import synth
    ls, rs = synth.stereo_batch(rows, cols, D, frames, seed0=7 + D)
    cl, cr = ctx.census9x7(dev(ls)), ctx.census9x7(dev(rs))
    ctx.set_option(S.OPT_MATRIX_CORES, 0)
    try:
        wl = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, 2)
        wr = ctx.dense_wta(cl, cr, D, S.VIEW_RIGHT, 2)
    finally:
        ctx.set_option(S.OPT_MATRIX_CORES, -1)
    ml, mr = ctx.dense_wta_matrix(cl, cr, D)
    ml0, _ = ctx.dense_wta_matrix(cl, cr, D, both=False)
    if D > 128:                             # passes of nine tiles and passes of five, whichever is the default at D
        for wide in (1, 0):
            ctx.set_option(S.OPT_MX_WIDE_TILES, wide)
            try:
                nl, nr = ctx.dense_wta_matrix(cl, cr, D)
            finally:
                ctx.set_option(S.OPT_MX_WIDE_TILES, -1)
            assert torch.equal(nl, wl) and torch.equal(nr, wr), wide
    assert torch.equal(ml, wl), (ml != wl).nonzero()[:5].tolist()
    assert torch.equal(ml0, wl), (ml0 != wl).nonzero()[:5].tolist()
    assert torch.equal(mr, wr), (mr != wr).nonzero()[:5].tolist()
    # a band of rows against the oracle, so that the two device kernels cannot be wrong together
    hl, hr = cl.cpu().numpy().view(np.uint32), cr.cpu().numpy().view(np.uint32)
    y0 = rows // 2
    for view, got in ((0, ml), (1, mr)):
        want = O.dense_wta_words(hl[0], hr[0], D, view, y0, y0 + 2)
        assert np.array_equal(got[0].cpu().numpy()[y0:y0 + 2], want[y0:y0 + 2])


@pytest.mark.parametrize("rows,cols,max_disp,thr", [(64, 200, 48, 1.0), (375, 1242, 128, 1.0), (40, 90, 16, -1.0), (368, 1232, 128, 0.0)])
def test_fused_census9x7_lr(ctx, rows, cols, max_disp, thr):
    from stereovision_amd import synth
    left, right, field = synth.stereo_pair(rows, cols, max_disp, seed=rows)
    got = ctx.dense_census9x7_lr(dev(left), dev(right), max_disp, lr_threshold=thr).cpu().numpy()
    cl, cr = O.census9x7(left), O.census9x7(right)
    want = O.dense_wta_words(cl, cr, max_disp, 0)
    if thr >= 0:
        want = O.lr_check(want, O.dense_wta_words(cl, cr, max_disp, 1), thr, -1.0)
    assert np.array_equal(got, want)
    valid = want >= 0
    interior = np.zeros_like(valid); interior[8:-8, max_disp + 8:-8] = True
    # sanity: the synthetic disparity field is recovered on most checked pixels
    assert (np.abs(want - field)[valid & interior] <= 1).mean() > 0.9


@pytest.mark.parametrize("strip_rows", [1, 2, 6, 7, 8, 9])
@pytest.mark.parametrize("D,words", [(40, 2), (100, 2), (200, 2), (100, 1), (40, 1)])
def test_matrix_search_strip_lengths(ctx, strip_rows, D, words):
    """The steady state of the matrix-core search runs three rows per trip with a remainder of 0, 1 or 2 rows, and strips
    of fewer than two output rows have no steady row at all: every remainder, with strips forced short so that a frame
    has many strip ends, both views, against the vector sweep."""
    import stereovision_amd as S
    from stereovision_amd import synth
    rows, cols = 61, 300
    ls, rs = synth.stereo_batch(rows, cols, D, 2, seed0=31 + D + strip_rows)
    if words == 2:
        cl, cr = ctx.census9x7(dev(ls)), ctx.census9x7(dev(rs))
    else:
        cl, cr = ctx.census5x5(dev(ls)), ctx.census5x5(dev(rs))
    ctx.set_option(S.OPT_MATRIX_CORES, 0)
    try:
        wl = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, words)
        wr = ctx.dense_wta(cl, cr, D, S.VIEW_RIGHT, words)
    finally:
        ctx.set_option(S.OPT_MATRIX_CORES, -1)
    ctx.set_option(S.OPT_MX_STRIP_ROWS, strip_rows)
    try:
        ml, mr = ctx.dense_wta_both(cl, cr, D, words)
        ml0 = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, words)
    finally:
        ctx.set_option(S.OPT_MX_STRIP_ROWS, 0)
    assert torch.equal(ml, wl), (ml != wl).nonzero()[:5].tolist()
    assert torch.equal(ml0, wl), (ml0 != wl).nonzero()[:5].tolist()
    assert torch.equal(mr, wr), (mr != wr).nonzero()[:5].tolist()


@pytest.mark.parametrize("D", [48, 160, 240])
@pytest.mark.parametrize("words", [1, 2])
def test_matrix_search_narrowest_frame_on_a_fresh_context(D, words):
    """cols == 36, the narrowest frame the matrix-core search takes: its one workgroup leaves more partial keys per row
    than the vector sweep's layout has room for (192 against 160 words at up to 64 disparities, 512 against 480 at 129..192),
    so the workspace must be sized for the keys.  Two contexts whose workspace has never been larger than this request needs
    (one for the both-views search, one for the fused entry); the grown workspace's slack is overrun after 12 frame-rows."""
    import stereovision_amd as S
    from stereovision_amd import synth
    rows, cols, frames = 64, 36, 3
    ls, rs = synth.stereo_batch(rows, cols, min(D, cols // 2), frames, seed0=900 + D)
    a = S.Context(0)
    try:
        dl, dr = dev(ls), dev(rs)
        if words == 2:
            hl, hr = np.stack([O.census9x7(x) for x in ls]), np.stack([O.census9x7(x) for x in rs])
        else:
            hl, hr = np.stack([O.census5x5(O.u8_to_s8(x)) for x in ls]), np.stack([O.census5x5(O.u8_to_s8(x)) for x in rs])
        assert S.dense_kernel_name(rows, cols, words, D, True, a) == "dense_wta_mx_kernel"
        ml, mr = a.dense_wta_both(dev_u32(hl), dev_u32(hr), D, words)
        ml, mr = ml.cpu().numpy(), mr.cpu().numpy()
        for f in range(frames):
            assert np.array_equal(ml[f], O.dense_wta_words(hl[f], hr[f], D, 0)), f
            assert np.array_equal(mr[f], O.dense_wta_words(hl[f], hr[f], D, 1)), f
    finally:
        a.close()
    if words == 2:
        b = S.Context(0)
        try:
            fused = b.dense_census9x7_lr(dl, dr, D, lr_threshold=1.0).cpu().numpy()
            for f in range(frames):
                want = O.lr_check(O.dense_wta_words(hl[f], hr[f], D, 0), O.dense_wta_words(hl[f], hr[f], D, 1), 1.0, -1.0)
                assert np.array_equal(fused[f], want), f
        finally:
            b.close()


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_matrix_search_random_sizes(ctx, seed):
    """Fifteen random sizes per seed (widths that leave idle wavefronts or a shifted last x-block, forced short strips, every
    tile count, both descriptor widths, one to three frames): left view, both views and the fused left-right entry against
    the vector sweep."""
    import stereovision_amd as S
    from stereovision_amd import synth
    rng = np.random.default_rng(seed)
    for case in range(15):
        cols = int(rng.choice([rng.integers(36, 80), rng.integers(80, 400), rng.integers(400, 1400)]))
        rows = int(rng.integers(5, 90))
        D = int(rng.choice([rng.integers(1, 65), rng.integers(65, 129), rng.integers(129, 400)]))
        words = int(rng.integers(1, 3))
        frames = int(rng.integers(1, 4))
        strip = int(rng.choice([0, 0, rng.integers(1, 12)]))
        what = "rows=%d cols=%d D=%d words=%d frames=%d strip=%d" % (rows, cols, D, words, frames, strip)
        ls, rs = synth.stereo_batch(rows, cols, min(D, max(cols // 2, 1)), frames, seed0=100 * seed + case)
        dl, dr = dev(ls), dev(rs)
        cl, cr = (ctx.census9x7(dl), ctx.census9x7(dr)) if words == 2 else (ctx.census5x5(dl), ctx.census5x5(dr))
        ctx.set_option(S.OPT_MATRIX_CORES, 0)
        try:
            wl = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, words)
            wr = ctx.dense_wta(cl, cr, D, S.VIEW_RIGHT, words)
        finally:
            ctx.set_option(S.OPT_MATRIX_CORES, -1)
        ctx.set_option(S.OPT_MX_STRIP_ROWS, strip)
        try:
            ml, mr = ctx.dense_wta_both(cl, cr, D, words)
            ml0 = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, words)
            fused = ctx.dense_census9x7_lr(dl, dr, D, lr_threshold=1.0) if words == 2 else None
        finally:
            ctx.set_option(S.OPT_MX_STRIP_ROWS, 0)
        assert torch.equal(ml, wl) and torch.equal(ml0, wl), what
        assert torch.equal(mr, wr), what
        if fused is not None:
            assert torch.equal(fused, ctx.lr_check(wl.clone(), wr, 1.0)), what


# ------------------------------------------------------------------ sub-pixel refinement (build-defined)
SUBPIXEL_TOL = 1e-3   # px, the tolerance required of sub-pixel disparities; the results are in fact identical


@pytest.mark.parametrize("rows,cols,max_disp", [(24, 64, 16), (30, 150, 128), (9, 21, 5), (5, 20, 8), (64, 300, 200), (4, 30, 4), (21, 1300, 1000), (13, 700, 340)])
@pytest.mark.parametrize("words", [1, 2])
@pytest.mark.parametrize("tiles", [2, 1, 0])
def test_subpixel_refine_matches_oracle(ctx, rows, cols, max_disp, words, tiles):
    """The three forms of the step: both neighbourhoods of a block staged in LDS (the default up to 252 disparities), the left one
    alone, and every descriptor from memory."""
    import stereovision_amd as S
    rng = np.random.default_rng(rows * 7 + cols + words)
    shape = (2, rows, cols) + ((2,) if words == 2 else ())
    cl = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32)
    cr = np.roll(cl, -3, axis=2) ^ (rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x00f10300))
    # start from the true integer winners, then plant invalidated, fractional, zero and out-of-range values
    disp = np.stack([O.dense_wta_words(cl[f], cr[f], max_disp, 0) for f in range(2)])
    mask = rng.random(disp.shape)
    disp[mask < 0.05] = -1.0
    disp[(mask >= 0.05) & (mask < 0.10)] += 0.25
    disp[(mask >= 0.10) & (mask < 0.15)] = float(max_disp - 1)
    disp[(mask >= 0.15) & (mask < 0.20)] = rng.integers(0, max_disp, size=int(((mask >= 0.15) & (mask < 0.20)).sum()))
    ctx.set_option(S.OPT_SUBPIXEL_TILES, tiles)
    try:
        got = ctx.subpixel_refine(dev_u32(cl), dev_u32(cr), max_disp, dev(disp), words=words).cpu().numpy()
    finally:
        ctx.set_option(S.OPT_SUBPIXEL_TILES, 2)
    changed = 0
    for f in range(2):
        want = O.subpixel_refine(cl[f], cr[f], max_disp, disp[f])
        assert np.max(np.abs(got[f] - want), initial=0.0) <= SUBPIXEL_TOL
        assert np.array_equal(got[f], want)
        changed += int((want != disp[f]).sum())
    if rows >= 9 and cols > 20:
        assert changed > 0


@pytest.mark.parametrize("rows,cols,max_disp", [(375, 1242, 128), (64, 200, 48), (70, 900, 300), (33, 260, 252), (33, 260, 253)])
def test_fused_census9x7_subpixel(ctx, rows, cols, max_disp):
    from stereovision_amd import synth
    left, right, field = synth.stereo_pair(rows, cols, max_disp, seed=rows + 1)
    got = ctx.dense_census9x7_subpixel(dev(left), dev(right), max_disp, lr_threshold=1.0).cpu().numpy()
    cl, cr = O.census9x7(left), O.census9x7(right)
    want = O.lr_check(O.dense_wta_words(cl, cr, max_disp, 0), O.dense_wta_words(cl, cr, max_disp, 1), 1.0, -1.0)
    want = O.subpixel_refine(cl, cr, max_disp, want)
    assert np.max(np.abs(got - want)) <= SUBPIXEL_TOL and np.array_equal(got, want)
    valid = want >= 0
    # a winner is a strict minimum on its left and a minimum on its right: the vertex of the parabola lies within half a pixel
    assert np.all(np.abs(want[valid] - np.rint(want[valid])) <= 0.5)
    interior = np.zeros_like(valid); interior[8:-8, max_disp + 8:-8] = True
    if (valid & interior).sum() > 100:
        assert (np.abs(want - field)[valid & interior] <= 1).mean() > 0.9


@pytest.mark.parametrize("tiles", [2, 1, 0])
@pytest.mark.parametrize("matrix", [-1, 0])
def test_fused_subpixel_equals_its_steps(ctx, tiles, matrix):
    """The fused entry hands the sub-pixel step the winners' keys where the search ran on the matrix cores (the winner's cost is
    then read from its key); every combination of the options must equal census, search of both views, check and sub-pixel step
    called one by one, on random sizes and ranges, with invalidated pixels, zero disparities and pixels too close to the border."""
    import stereovision_amd as S
    from stereovision_amd import synth
    rng = np.random.default_rng(5 + tiles)
    for case in range(10):
        rows, cols = int(rng.integers(5, 60)), int(rng.integers(36, 700))
        D = int(rng.choice([rng.integers(1, 40), rng.integers(40, 253), rng.integers(253, 600)]))
        frames = int(rng.integers(1, 4))
        ls, rs = synth.stereo_batch(rows, cols, min(D, max(cols // 2, 1)), frames, seed0=300 + case)
        dl, dr = dev(ls), dev(rs)
        cl, cr = ctx.census9x7(dl), ctx.census9x7(dr)
        wl, wr = ctx.dense_wta_both(cl, cr, D, 2)
        want = ctx.subpixel_refine(cl, cr, D, ctx.lr_check(wl, wr, 1.0), words=2)
        ctx.set_option(S.OPT_SUBPIXEL_TILES, tiles)
        ctx.set_option(S.OPT_MATRIX_CORES, matrix)
        try:
            got = ctx.dense_census9x7_subpixel(dl, dr, D, lr_threshold=1.0)
        finally:
            ctx.set_option(S.OPT_SUBPIXEL_TILES, 2)
            ctx.set_option(S.OPT_MATRIX_CORES, -1)
        assert torch.equal(got, want), (rows, cols, D, frames, (got != want).nonzero()[:5].tolist())


def test_subpixel_4k_512_disparities(ctx):
    """The 3840x2160, D=512 configuration: the fused run against the oracle on a band of rows, and the sub-pixel step
    against the oracle on the whole frame starting from the device's own integer map."""
    from stereovision_amd import synth
    import stereovision_amd as S
    rows, cols, D = 2160, 3840, 512
    left, right, _ = synth.stereo_pair(rows, cols, D, seed=99)
    dl, dr = dev(left), dev(right)
    cl, cr = ctx.census9x7(dl), ctx.census9x7(dr)
    integer = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, 2)
    refined = ctx.subpixel_refine(cl, cr, D, integer.clone(), words=2).cpu().numpy()
    hcl, hcr = cl.cpu().numpy().view(np.uint32), cr.cpu().numpy().view(np.uint32)
    want = O.subpixel_refine(hcl, hcr, D, integer.cpu().numpy())
    assert np.array_equal(refined, want)
    y0, y1 = 1000, 1002
    band = O.dense_wta_words(hcl, hcr, D, 0, y0, y1)
    assert np.array_equal(integer.cpu().numpy()[y0:y1], band[y0:y1])
    fused = ctx.dense_census9x7_subpixel(dl, dr, D, lr_threshold=-1.0).cpu().numpy()
    assert np.array_equal(fused, refined)
    # with the check (the map then reaches the sub-pixel step as keys) against the same steps one by one
    wl, wr = ctx.dense_wta_both(cl, cr, D, 2)
    want = ctx.subpixel_refine(cl, cr, D, ctx.lr_check(wl, wr, 1.0), words=2)
    assert torch.equal(ctx.dense_census9x7_subpixel(dl, dr, D, lr_threshold=1.0), want)


@pytest.mark.parametrize("frames,rows,cols,D,band", [(2, 1072, 1920, 256, (500, 503)), (1, 2160, 3840, 512, (1000, 1002)),
                                                      (4, 2160, 3840, 512, (1500, 1501))])
def test_left_right_full_size_configs(ctx, frames, rows, cols, D, band):
    """The 1920x1080 (cropped to /16, D=256) and 3840x2160 (D=512) configurations through the fused left-right entry
    point: against the oracle on a band of rows of every frame; the pixels the check rejects are exactly those where the two
    separately computed views disagree (whole frame, device results of the one-view searches).  Four 4K frames need more
    than the 4 GiB the partial minima of one sweep may take, so that batch is swept in two groups of frames."""
    import stereovision_amd as S
    from stereovision_amd import synth
    ls, rs = synth.stereo_batch(rows, cols, D, frames, seed0=40 + frames)
    dl, dr = dev(ls), dev(rs)
    got = ctx.dense_census9x7_lr(dl, dr, D, lr_threshold=1.0).cpu().numpy()
    cl, cr = ctx.census9x7(dl), ctx.census9x7(dr)
    left_map = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, 2)
    right_map = ctx.dense_wta(cl, cr, D, S.VIEW_RIGHT, 2)
    y0, y1 = band
    for f in range(frames):
        hcl, hcr = O.census9x7(ls[f]), O.census9x7(rs[f])
        wl = O.dense_wta_words(hcl, hcr, D, 0, y0, y1)
        wr = O.dense_wta_words(hcl, hcr, D, 1, y0, y1)
        assert np.array_equal(left_map[f].cpu().numpy()[y0:y1], wl[y0:y1])
        assert np.array_equal(right_map[f].cpu().numpy()[y0:y1], wr[y0:y1])
        want_band = O.lr_check(wl, wr, 1.0, -1.0)
        assert np.array_equal(got[f][y0:y1], want_band[y0:y1])
    # whole frames: the one-sweep path equals the check applied to the two one-view searches
    two = ctx.lr_check(left_map.clone(), right_map, 1.0).cpu().numpy()
    assert np.array_equal(got, two)
    both_l, both_r = ctx.dense_wta_both(cl, cr, D, words=2)
    assert torch.equal(both_l, left_map) and torch.equal(both_r, right_map)
