"""Times the device triangulation (svx_delaunay_device) against the host one on random point sets (run on the GPU box)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import stereovision_amd as S

ctx = S.Context(0)
for rows, cols, n, kind in [(368, 1232, 27000, "uniform"), (368, 1232, 27000, "clustered"), (368, 1232, 5000, "uniform"), (1072, 1920, 100000, "uniform")]:
    rng = np.random.default_rng(1)
    if kind == "uniform":
        idx = rng.choice(rows * cols, size=n, replace=False)
        pts = np.stack([idx % cols, idx // cols], 1).astype(np.int32)
    else:      # half of the frame empty: blobs of texture with nothing between them
        m = np.zeros((rows, cols), bool)
        for _ in range(40):
            cx, cy, r = rng.integers(0, cols), rng.integers(0, rows), rng.integers(20, 80)
            m[max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = True
        cand = np.flatnonzero(m)
        idx = rng.choice(cand, size=min(n, len(cand)), replace=False)
        pts = np.stack([idx % cols, idx // cols], 1).astype(np.int32)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
    d = torch.from_numpy(pts).cuda()
    t0 = time.perf_counter(); want = S.delaunay(pts); th = time.perf_counter() - t0
    got = ctx.delaunay_device(d, rows, cols)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        got = ctx.delaunay_device(d, rows, cols)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / 20
    print("%dx%d %6d points %-9s host %.2f ms  device %.3f ms  left over %s  equal %s" % (cols, rows, len(pts), kind, th * 1e3, td * 1e3, ctx.mesh_device_stats(),
          np.array_equal(got.cpu().numpy(), want)), flush=True)
