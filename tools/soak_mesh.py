"""Soak of the device triangulation: random point sets of many kinds (uniform at several densities, clusters with empty areas,
jittered lattices, points on a few lines and on the sides of a box, repeated positions, tiny sets) in frames of random size,
svx_delaunay_device against the host's svx_delaunay, element for element.  usage: soak_mesh.py <cases> <seed>  (run on the GPU box)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import stereovision_amd as S

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = S.Context(0)
bad = 0
kinds = {}
t0 = time.time()
for case in range(cases):
    rows, cols = int(rng.integers(8, 1100)), int(rng.integers(8, 2000))
    kind = rng.choice(["uniform", "sparse", "dense", "clusters", "lattice", "lines", "box", "repeats", "tiny", "mix"])
    area = rows * cols
    def uniform(n):
        idx = rng.choice(area, size=int(min(max(n, 1), area)), replace=False)
        return np.stack([idx % cols, idx // cols], 1)
    if kind == "uniform": pts = uniform(area * rng.uniform(0.005, 0.08))
    elif kind == "sparse": pts = uniform(rng.integers(3, 200))
    elif kind == "dense": pts = uniform(min(area * rng.uniform(0.2, 0.6), 60000))
    elif kind == "tiny": pts = uniform(rng.integers(1, 8))
    elif kind == "clusters":
        parts = []
        for _ in range(int(rng.integers(1, 8))):
            cx, cy, r = rng.integers(0, cols), rng.integers(0, rows), int(rng.integers(3, 120))
            m = int(rng.integers(5, 3000))
            parts.append(np.stack([rng.integers(cx - r, cx + r + 1, m), rng.integers(cy - r, cy + r + 1, m)], 1))
        pts = np.concatenate(parts)
    elif kind == "lattice":
        step = int(rng.integers(1, 24))
        ys, xs = np.mgrid[int(rng.integers(0, step)):rows:step, int(rng.integers(0, step)):cols:step]
        pts = np.stack([xs.ravel(), ys.ravel()], 1)
        if rng.random() < 0.5:
            pts = pts + rng.integers(-1, 2, size=pts.shape) * (rng.random(size=pts.shape) < 0.2)
        if len(pts) > 60000: pts = pts[rng.choice(len(pts), 60000, replace=False)]
    elif kind == "lines":
        parts = []
        for _ in range(int(rng.integers(1, 6))):
            if rng.random() < 0.5:
                y = int(rng.integers(0, rows)); xs = rng.choice(cols, size=int(rng.integers(2, cols)), replace=False)
                parts.append(np.stack([xs, np.full_like(xs, y)], 1))
            else:
                x = int(rng.integers(0, cols)); ys = rng.choice(rows, size=int(rng.integers(2, rows)), replace=False)
                parts.append(np.stack([np.full_like(ys, x), ys], 1))
        parts.append(uniform(rng.integers(0, 300)))
        pts = np.concatenate(parts)
    elif kind == "box":       # what the pipeline's support points look like: many points ON the sides of their bounding box
        x0, x1 = sorted(rng.integers(0, cols, 2)); y0, y1 = sorted(rng.integers(0, rows, 2))
        inner = uniform(area * rng.uniform(0.002, 0.05))
        inner = inner[(inner[:, 0] >= x0) & (inner[:, 0] <= x1) & (inner[:, 1] >= y0) & (inner[:, 1] <= y1)]
        xs = rng.integers(x0, x1 + 1, int(rng.integers(0, 400))); ys = rng.integers(y0, y1 + 1, int(rng.integers(0, 200)))
        pts = np.concatenate([inner, np.stack([xs, np.full_like(xs, y1)], 1), np.stack([xs[::2], np.full_like(xs[::2], y0)], 1),
                              np.stack([np.full_like(ys, x0), ys], 1), np.stack([np.full_like(ys[::3], x1), ys[::3]], 1)])
    elif kind == "repeats":
        pts = uniform(rng.integers(3, 5000))
        pts = np.concatenate([pts, pts[rng.integers(0, len(pts), int(rng.integers(1, 2000)))]])
    else:
        pts = np.concatenate([uniform(area * rng.uniform(0.001, 0.03)), uniform(rng.integers(1, 50)) // 2 * 2])
    pts = pts[(pts[:, 0] >= 0) & (pts[:, 0] < cols) & (pts[:, 1] >= 0) & (pts[:, 1] < rows)]
    if kind not in ("repeats",) and rng.random() < 0.8: pts = np.unique(pts, axis=0)
    if rng.random() < 0.5: pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
    else: rng.shuffle(pts)
    pts = np.ascontiguousarray(pts, np.int32)
    if len(pts) == 0: continue
    want = S.delaunay(pts)
    try:
        got = ctx.delaunay_device(torch.from_numpy(pts).cuda(), rows, cols).cpu().numpy()
        ok = got.shape == want.shape and np.array_equal(got, want)
        why = "differs"
    except S.SvxError as e:       # a ring beyond the device form's limits is a documented refusal, not a mismatch
        ok = "256 triangles" in str(e) or "more than" in str(e)
        why = str(e)
    kinds[str(kind)] = kinds.get(str(kind), 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH case %d kind %s rows %d cols %d n %d: %s" % (case, kind, rows, cols, len(pts), why), flush=True)
    if case % 100 == 99: print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("soak_mesh: %d cases (seed %d) %s: %d mismatches" % (cases, seed, dict(sorted(kinds.items())), bad))
sys.exit(1 if bad else 0)
