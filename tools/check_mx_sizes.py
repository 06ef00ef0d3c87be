"""Compares the matrix-core search with the vector sweep (both on the device) on whole large frames; prints where they differ.
Measurement / debugging aid (run on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402

ctx = S.Context(0)
for rows, cols, D, frames in [(2160, 3840, 128, 1), (2160, 3840, 256, 1), (1072, 1920, 200, 2), (375, 1242, 64, 4)]:
    ls, rs = synth.stereo_batch(rows, cols, D, frames, seed0=7)
    cl, cr = ctx.census9x7(torch.from_numpy(ls).cuda()), ctx.census9x7(torch.from_numpy(rs).cuda())
    wl = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, 2).cpu().numpy()
    wr = ctx.dense_wta(cl, cr, D, S.VIEW_RIGHT, 2).cpu().numpy()
    ml, mr = ctx.dense_wta_matrix(cl, cr, D)
    ml, mr = ml.cpu().numpy(), mr.cpu().numpy()
    for name, a, b in (("left", ml, wl), ("right", mr, wr)):
        bad = np.argwhere(a != b)
        print(rows, cols, D, name, "differences:", len(bad), "first:", bad[:4].tolist(),
              "cols:", (int(bad[:, 2].min()), int(bad[:, 2].max())) if len(bad) else None,
              "rows:", (int(bad[:, 1].min()), int(bad[:, 1].max())) if len(bad) else None)
        if len(bad):
            f, y, x = bad[0]
            print("   got", a[f, y, x], "want", b[f, y, x])
