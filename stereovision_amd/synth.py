"""Deterministic synthetic rectified stereo pairs (no datasets are available offline).

Right image: uint8 noise from numpy PCG64(seed), box-blurred 3x3 once (census texture without flat
regions).  Left image: the right image shifted by a piecewise-planar disparity field
d(x,y) = clip(round(d0 + gx*x + gy*y), 0, D-1) made of 3-5 planar patches (vertical bands); occluded
and left-edge pixels are filled from an independent noise stream.
"""
import numpy as np


def _box3(img):
    p = np.pad(img.astype(np.uint16), 1, mode="edge")
    s = sum(p[dy:dy + img.shape[0], dx:dx + img.shape[1]] for dy in range(3) for dx in range(3))
    return ((s + 4) // 9).astype(np.uint8)


def disparity_field(rows, cols, max_disp, rng):
    n_patch = int(rng.integers(3, 6))
    edges = np.sort(rng.choice(np.arange(cols // 8, cols - cols // 8), size=n_patch - 1, replace=False))
    bounds = [0] + [int(e) for e in edges] + [cols]
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    d = np.zeros((rows, cols), np.float64)
    for i in range(n_patch):
        d0 = rng.uniform(0.05, 0.8) * (max_disp - 1)
        gx = rng.uniform(-0.02, 0.02)
        gy = rng.uniform(-0.03, 0.03)
        sl = slice(bounds[i], bounds[i + 1])
        d[:, sl] = d0 + gx * (xx[:, sl] - bounds[i]) + gy * yy[:, sl]
    return np.clip(np.rint(d), 0, max_disp - 1).astype(np.int32)


def stereo_pair(rows, cols, max_disp, seed):
    """Returns (left u8, right u8, generating integer disparity field)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    right = _box3(rng.integers(0, 256, size=(rows, cols), dtype=np.uint8))
    d = disparity_field(rows, cols, max_disp, rng)
    fill = np.random.Generator(np.random.PCG64(seed + 1_000_003)).integers(0, 256, size=(rows, cols), dtype=np.uint8)
    xx = np.arange(cols)[None, :].repeat(rows, 0)
    src = xx - d
    ok = src >= 0
    left = np.where(ok, np.take_along_axis(right, np.clip(src, 0, cols - 1), axis=1), fill).astype(np.uint8)
    return left, right, d


def stereo_batch(rows, cols, max_disp, n_frames, seed0=0):
    ls, rs = [], []
    for f in range(n_frames):
        l, r, _ = stereo_pair(rows, cols, max_disp, seed0 + f)
        ls.append(l)
        rs.append(r)
    return np.stack(ls), np.stack(rs)
