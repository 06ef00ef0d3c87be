"""The SIMD + OpenMP CPU baseline (oracle/svx_cpu_simd.c, what bench.py times on the host beside the GPU) against the scalar
oracle: identical census words, disparity maps of both views and left-right check, one- and two-word descriptors, odd
sizes, several threads."""
import numpy as np
import pytest

import oracle_lib as O


@pytest.mark.parametrize("rows,cols", [(7, 9), (20, 37), (64, 130), (6, 20)])
def test_simd_census9x7(rows, cols):
    img = np.random.default_rng(rows * cols).integers(0, 256, size=(rows, cols), dtype=np.uint8)
    assert np.array_equal(O.simd_census9x7(img, threads=3), O.census9x7(img))


@pytest.mark.parametrize("rows,cols,max_disp,words", [(24, 64, 16, 2), (12, 40, 64, 2), (40, 150, 128, 2), (9, 21, 5, 1), (5, 20, 8, 2),
                                                       (70, 97, 33, 1), (37, 300, 200, 2)])
def test_simd_dense_wta_both_views(rows, cols, max_disp, words):
    from stereovision_amd import synth
    left, right, _ = synth.stereo_pair(rows, cols, min(max_disp, max(2, cols // 3)), seed=rows + cols)
    if words == 2:
        cl, cr = O.census9x7(left), O.census9x7(right)
    else:
        cl, cr = O.census5x5(O.u8_to_s8(left)), O.census5x5(O.u8_to_s8(right))
    dl, dr = O.simd_dense_wta_words(cl, cr, max_disp, threads=4)
    want_l, want_r = O.dense_wta_words(cl, cr, max_disp, 0), O.dense_wta_words(cl, cr, max_disp, 1)
    assert np.array_equal(dl, want_l), np.argwhere(dl != want_l)[:5]
    assert np.array_equal(dr, want_r), np.argwhere(dr != want_r)[:5]
    assert np.array_equal(O.simd_lr_check(dl, dr, 1.0, -1.0, threads=2), O.lr_check(want_l, want_r, 1.0, -1.0))
    # a band of rows only (what the bounded benchmark sample uses)
    bl, _ = O.simd_dense_wta_words(cl, cr, max_disp, threads=2, right=False, y0=3, y1=min(rows, 11))
    assert np.array_equal(bl[3:11], want_l[3:11]) and not bl[11:].any()


def test_simd_dense_wta_ties():
    rng = np.random.default_rng(5)
    cl = (rng.integers(0, 3, size=(30, 96, 2)).astype(np.uint32) * 0x111)
    cr = (rng.integers(0, 3, size=(30, 96, 2)).astype(np.uint32) * 0x111)
    dl, dr = O.simd_dense_wta_words(cl, cr, 40, threads=3)
    assert np.array_equal(dl, O.dense_wta_words(cl, cr, 40, 0)) and np.array_equal(dr, O.dense_wta_words(cl, cr, 40, 1))
