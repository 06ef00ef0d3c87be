import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
import stereovision_amd as S
from stereovision_amd import synth
rows, cols = 375, 1242
ctx = S.Context(0)
l, r = [torch.from_numpy(a).cuda() for a in synth.stereo_pair(rows, cols, 64, 0)[:2]]
for it in (2, 1, 2, 3, 2):
    p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=it)
    out = torch.zeros((rows, cols), dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.high_perf_stereo(l, r, p, out, copy=False)      # (the library's call alone: no copy of the points on the Python side)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print("nIters", it, "ms per frame: min %.2f median %.2f mean %.2f max %.2f" % (ts.min(), np.median(ts), ts.mean(), ts.max()), "first five", np.round(ts[:5], 2))
