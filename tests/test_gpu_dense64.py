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


@pytest.mark.parametrize("rows,cols", [(32, 64), (7, 9), (6, 8), (37, 129), (48, 256), (50, 261), (1, 1), (16, 70)])
def test_census9x7(ctx, rows, cols):
    rng = np.random.default_rng(rows * 100 + cols)
    imgs = rng.integers(0, 256, size=(3, rows, cols), dtype=np.uint8)
    imgs[1] = (imgs[1] // 64) * 64                                        # many ties
    got = ctx.census9x7(dev(imgs)).cpu().numpy().view(np.uint32)
    for f in range(3):
        assert np.array_equal(got[f], O.census9x7(imgs[f]))


@pytest.mark.parametrize("rows,cols,max_disp", [(24, 64, 16), (12, 40, 64), (30, 150, 128), (9, 21, 5), (5, 20, 8), (16, 300, 200),
                                                 (110, 90, 70), (8, 19, 4), (4, 30, 4), (20, 70, 1100)])
@pytest.mark.parametrize("words", [1, 2])
@pytest.mark.parametrize("view", [0, 1])
def test_dense_wta_matches_oracle(ctx, rows, cols, max_disp, words, view):
    rng = np.random.default_rng(rows * 1000 + cols + max_disp + words)
    shape = (2, rows, cols) + ((2,) if words == 2 else ())
    # few distinct values give many cost ties, which exercise the first-minimum rule in both views
    cl = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x01010101 if cols % 2 else 0xffffffff)
    cr = np.roll(cl, -3, axis=2) ^ (rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32) & np.uint32(0x00010000))
    got = ctx.dense_wta(dev_u32(cl), dev_u32(cr), max_disp, view=view, words=words).cpu().numpy()
    for f in range(2):
        want = O.dense_wta_words(cl[f], cr[f], max_disp, view)
        assert np.array_equal(got[f], want), (f, np.argwhere(got[f] != want)[:5])


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
