"""Times the three forms of the 9x7 census tile kernel on the benchmark's batch (64 KITTI-size frames, one image per call and
both images in the fused entry's pair launch).  Measurement aid, run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stereovision_amd as S
from stereovision_amd import synth

rows, cols, D, F = 368, 1232, 128, 64
ctx = S.Context(0)
ls, rs = synth.stereo_batch(rows, cols, D, 8, seed0=0)
left = torch.from_numpy(ls).cuda().repeat(F // 8, 1, 1)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


ref = None
for form in (2, 1, 0):
    ctx.set_option(S.OPT_CENSUS97_PACKED, form)
    out = ctx.census9x7(left)
    if ref is None:
        ref = out
    ms = timed(lambda: ctx.census9x7(left))
    px = F * rows * cols
    print("form %d: %.4f ms per 64-frame image batch  %.2f TB/s of 1 B in + 8 B out per pixel  equal to form 2: %s"
          % (form, ms, px * 9 / ms / 1e9, bool(torch.equal(out, ref))))
