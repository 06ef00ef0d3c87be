"""Times the chain of the 4K configuration on one GPU (3840x2160, 512 disparities, 8 frames): the fused sub-pixel entry (census,
search in passes, left-right check, sub-pixel step) and the point cloud of every valid pixel.  Measurement aid (run on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402
from stereovision_amd._lib import _dptr  # noqa: E402

rows, cols, D, F = 2160, 3840, 512, 8
ctx = S.Context(0)
ls, rs = synth.stereo_batch(rows, cols, D, 2, seed0=3)
left = torch.from_numpy(ls).cuda().repeat(F // 2, 1, 1)
right = torch.from_numpy(rs).cuda().repeat(F // 2, 1, 1)
out = torch.empty((F, rows, cols), dtype=torch.float32, device="cuda")
q = np.array([[1, 0, 0, -cols / 2], [0, 1, 0, -rows / 2], [0, 0, 0, 2000.0], [0, 0, 1 / 0.5, 0]], np.float64)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(2):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


print("lr entry        %.2f ms" % timed(lambda: ctx._call("svx_dense_census9x7_lr_u8", _dptr(left), _dptr(right), rows, cols, F, D, 1.0, _dptr(out))))
print("subpixel entry  %.2f ms" % timed(lambda: ctx._call("svx_dense_census9x7_subpixel_u8", _dptr(left), _dptr(right), rows, cols, F, D, 1.0, _dptr(out))))
xyz, _, off = ctx.dense_point_cloud(out, q, min_disparity=0.0, max_range=1e9)
print("points          %d of %d pixels" % (int(off[-1]), F * rows * cols))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): ctx.dense_point_cloud(out, q, min_disparity=0.0, max_range=1e9)
torch.cuda.synchronize()
print("point cloud     %.2f ms (host-synchronous call, incl. the 1.6 GB output allocation)" % ((time.perf_counter() - t0) / 3 * 1e3))
