"""Stress of svx_high_perf_stereo_batch: many KITTI-size frames on 2 .. 16 host threads, repeatedly, every frame's map and points
against the single call's.  Not collected by pytest (run by hand on the GPU box):  python3 tests/stress_pipeline_batch.py <rounds>"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402

rounds = int(sys.argv[1])
rows, cols, frames = 375, 1242, 48
ctx = S.Context(0)
p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=3)
ls, rs = synth.stereo_batch(rows, cols, 64, frames, seed0=100)
left, right = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
want_maps = torch.zeros((frames, rows, cols), dtype=torch.float32, device="cuda")
want_pts = [ctx.high_perf_stereo(left[f], right[f], p, want_maps[f]) for f in range(frames)]
bad = 0
t0 = time.time()
for r in range(rounds):
    for threads in (2, 5, 16, 3, 8):
        maps = torch.zeros((frames, rows, cols), dtype=torch.float32, device="cuda")
        pts = ctx.high_perf_stereo_batch(left, right, p, maps, n_threads=threads)
        ok = torch.equal(maps, want_maps) and all(np.array_equal(a, b) for a, b in zip(pts, want_pts))
        if not ok:
            bad += 1
            print("MISMATCH round %d threads %d" % (r, threads), flush=True)
    print("round %d done, %d mismatches, %.0f s" % (r + 1, bad, time.time() - t0), flush=True)
print("stress_pipeline_batch: %d rounds x 5 thread counts x %d frames: %d mismatches" % (rounds, frames, bad))
sys.exit(1 if bad else 0)
