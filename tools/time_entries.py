"""Times the fused entry points of the 9x7 configuration on the benchmark workload (measurement aid, run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stereovision_amd as S
from stereovision_amd import synth
from stereovision_amd._lib import _dptr

rows, cols, D, F = 368, 1232, 128, 64
ctx = S.Context(0)
ls, rs = synth.stereo_batch(rows, cols, D, 8, seed0=0)
left = torch.from_numpy(ls).cuda().repeat(F // 8, 1, 1); right = torch.from_numpy(rs).cuda().repeat(F // 8, 1, 1)
out = torch.empty((F, rows, cols), dtype=torch.float32, device="cuda")


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


cl, cr = ctx.census9x7(left), ctx.census9x7(right)
print("census pair        %.3f ms" % timed(lambda: (ctx.census9x7(left), ctx.census9x7(right))))
print("lr entry           %.3f ms" % timed(lambda: ctx._call("svx_dense_census9x7_lr_u8", _dptr(left), _dptr(right), rows, cols, F, D, 1.0, _dptr(out))))
print("subpixel entry     %.3f ms" % timed(lambda: ctx._call("svx_dense_census9x7_subpixel_u8", _dptr(left), _dptr(right), rows, cols, F, D, 1.0, _dptr(out))))
d = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, 2)
print("subpixel step only %.3f ms" % timed(lambda: ctx.subpixel_refine(cl, cr, D, d.clone(), words=2)))
print("clone only         %.3f ms" % timed(lambda: d.clone()))
