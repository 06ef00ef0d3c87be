"""Where the matrix-core search differs from the vector sweep: mismatch counts per x-block and per row band (debugging aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stereovision_amd as S
from stereovision_amd import synth
ctx = S.Context(0)
for rows, cols, D, frames in [(375, 1242, 64, 1), (120, 1242, 64, 1), (375, 600, 64, 1), (368, 1232, 128, 2)]:
    ls, rs = synth.stereo_batch(rows, cols, D, frames, seed0=7 + D)
    cl, cr = ctx.census9x7(torch.from_numpy(ls).cuda()), ctx.census9x7(torch.from_numpy(rs).cuda())
    ctx.set_option(S.OPT_MATRIX_CORES, 0)
    wl = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, 2).cpu().numpy()
    ctx.set_option(S.OPT_MATRIX_CORES, -1)
    for rep in range(2):
        ml, _ = ctx.dense_wta_matrix(cl, cr, D, both=False)
        ml = ml.cpu().numpy()
        bad = ml != wl
        print(rows, cols, D, "rep", rep, "differ:", int(bad.sum()))
        if bad.any():
            xb = (np.arange(cols) - 2) // 32
            per_block = [int(bad[..., (xb == b)].sum()) for b in range((cols - 4 + 31) // 32)]
            print("  per x-block:", per_block)
            per_rows = [int(bad[:, y:y + 16].sum()) for y in range(0, rows, 16)]
            print("  per 16 rows:", per_rows)
            f, y, x = np.argwhere(bad)[0]
            print("  first:", (f, y, x), "got", ml[f, y, x], "want", wl[f, y, x])
