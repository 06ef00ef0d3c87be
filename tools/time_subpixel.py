"""Times the sub-pixel step alone in its three forms (option SVX_OPT_SUBPIXEL_TILES 0, 1, 2) on checked disparity maps of three frame
sizes, and compares their results (measurement aid, run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stereovision_amd as S
from stereovision_amd import synth


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


ctx = S.Context(0)
for rows, cols, D, F in ((368, 1232, 128, 64), (1072, 1920, 256, 16), (2160, 3840, 512, 4), (480, 640, 64, 64)):
    ls, rs = synth.stereo_batch(rows, cols, D, min(F, 4), seed0=0)
    rep = F // min(F, 4)
    left = torch.from_numpy(ls).cuda().repeat(rep, 1, 1); right = torch.from_numpy(rs).cuda().repeat(rep, 1, 1)
    cl, cr = ctx.census9x7(left), ctx.census9x7(right)
    d = ctx.dense_census9x7_lr(left, right, D, lr_threshold=1.0)
    work = torch.empty_like(d)
    t_copy = timed(lambda: work.copy_(d))
    res = {}
    for tiles in (1, 2, 0):
        ctx.set_option(S.OPT_SUBPIXEL_TILES, tiles)
        t = timed(lambda: ctx.subpixel_refine(cl, cr, D, work.copy_(d), words=2)) - t_copy
        res[tiles] = ctx.subpixel_refine(cl, cr, D, d.clone(), words=2)
        print("%dx%d D=%d frames=%d tiles=%d: %.4f ms, equal to tiles=1: %s" % (cols, rows, D, F, tiles, t, bool(torch.equal(res[tiles], res[1]))), flush=True)
    ctx.set_option(S.OPT_SUBPIXEL_TILES, 2)
    del left, right, cl, cr, d, work, res
    torch.cuda.empty_cache()
