"""Soak of svx_high_perf_stereo against the oracle composition (tests/oracle_pipeline.py) on random frame sizes and
parameters, bit for bit: computed points and disparity map.  Not collected by pytest (run by hand on the GPU box):
    python3 tests/soak_pipeline.py <cases> <seed>"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_pipeline as OP  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = S.Context(0)
bad = 0
t0 = time.time()
for case in range(cases):
    rows, cols = int(rng.integers(33, 260)), int(rng.integers(40, 420))
    over = dict(maxDisp=int(rng.choice([8, 16, 32, 70, 100])), nIters=int(rng.integers(0, 5)), gradThreshold=int(rng.choice([10, 20, 30, 60])),
                applyHistEqualization=int(rng.integers(0, 2)), costAggrWindowSize=int(rng.choice([5, 11, 11, 11])),
                recordFullDisp=int(rng.random() < 0.9))
    if rng.random() < 0.2:
        over.update(invertRows=1, nbRows=int(rng.integers(2, 20)), traceLines=int(rng.integers(0, 2)), nbLines=int(rng.integers(2, 20)), lineSize=2)
    p = S.StereoParameters.reference_defaults(**over)
    left, right, _ = synth.stereo_pair(rows, cols, min(p.maxDisp, 64), int(rng.integers(0, 1 << 30)))
    init = float(rng.choice([0.0, 0.0, 3.0]))
    want_map = np.full(left.shape, init, np.float32)
    want_pts, info = OP.high_perf_stereo(left, right, p, want_map, S.delaunay, 0)
    got_map = torch.full(left.shape, init, dtype=torch.float32, device="cuda")
    got_pts = ctx.high_perf_stereo(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), p, got_map)
    ok = np.array_equal(got_pts, want_pts) and np.array_equal(got_map.cpu().numpy(), want_map)
    if not ok:
        bad += 1
        print("MISMATCH case %d rows %d cols %d %s %s" % (case, rows, cols, over, info), flush=True)
    if case % 20 == 19:
        print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("soak_pipeline: %d cases (seed %d): %d mismatches" % (cases, seed, bad))
sys.exit(1 if bad else 0)
