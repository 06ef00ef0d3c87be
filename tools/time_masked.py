"""Separates the cost of masked (image-edge) and unmasked x-blocks of the sweep kernels by timing two widths (measurement aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stereovision_amd as S
rows, D = 368, 128
ctx = S.Context(0)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
res = {}
for cols, F in ((1232, 64), (2464, 32)):
    rng = np.random.default_rng(cols)
    cl = torch.from_numpy(rng.integers(0, 2**31, size=(F, rows, cols, 2), dtype=np.int64).astype(np.int32)).cuda()
    cr = torch.from_numpy(rng.integers(0, 2**31, size=(F, rows, cols, 2), dtype=np.int64).astype(np.int32)).cuda()
    disp = torch.empty((F, rows, cols), dtype=torch.float32, device="cuda")
    f, r, c = F, rows, cols
    both = timed(lambda: ctx._call("svx_probe_both_views_sweep", S._lib._dptr(cl), S._lib._dptr(cr), r, c, f, 2, D, S._lib._dptr(disp)))
    left = timed(lambda: ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, 2, out=disp))
    nb = (cols - 4 + 15) // 16
    res[cols] = (both, left, nb, F)
    print(cols, "frames", F, "x-blocks", nb, "both %.3f ms" % both, "left %.3f ms" % left, flush=True)
# t = F * (U * unmasked + M * masked); masked blocks: xb - 5 < 127 -> 9 blocks (left), plus 1 (right edge, both only)
for name, idx, extra in (("both", 0, 1), ("left", 1, 0)):
    (t1, nb1, F1), (t2, nb2, F2) = [(res[c][idx], res[c][2], res[c][3]) for c in (1232, 2464)]
    m = 9 + extra
    # t1/F1 = U*(nb1-m) + M*m ; t2/F2 = U*(nb2-m) + M*m
    U = (t2 / F2 - t1 / F1) / (nb2 - nb1)
    M = (t1 / F1 - U * (nb1 - m)) / m
    print(name, "per-frame per-block: unmasked %.4f us, masked %.4f us, ratio %.2f" % (U * 1e3, M * 1e3, M / U))
