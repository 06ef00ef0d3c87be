"""Which support points does the first pass of the device triangulation leave to the second one?  Takes the mesh vertices of a
pipeline frame, triangulates them in the real frame and in a frame cut to the points' bounding box, and prints the counts and
times of both (measurement aid, run on the GPU box)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import stereovision_amd as S
from stereovision_amd import synth

ctx = S.Context(0)
rows, cols = 375, 1242
p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=1)
l, r = [torch.from_numpy(a).cuda() for a in synth.stereo_pair(rows, cols, 64, 0)[:2]]
out = torch.zeros((rows, cols), dtype=torch.float32, device="cuda")
pts = ctx.high_perf_stereo(l, r, p, out)
key = pts[:, 1].astype(np.int64) * 4096 + pts[:, 0]
brk = np.flatnonzero(key[1:] < key[:-1])
sup = pts[brk[0] + 1:] if len(brk) else pts
print("support points", len(sup), "x range", sup[:, 0].min(), sup[:, 0].max(), "y range", sup[:, 1].min(), sup[:, 1].max())
R, C = 368, 1232


def run(points, rr, cc, what):
    d = torch.from_numpy(np.ascontiguousarray(points, np.int32)).cuda()
    ctx.delaunay_device(d, rr, cc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.delaunay_device(d, rr, cc)
    torch.cuda.synchronize()
    print("%-40s %dx%d  %.3f ms per triangulation, left to the second pass: %d" % (what, cc, rr, (time.perf_counter() - t0) / 20 * 1e3, ctx.mesh_device_stats()[0]))


run(sup, R, C, "the frame as it is")
x0, y0 = sup[:, 0].min(), sup[:, 1].min()
run(sup - [x0, y0], sup[:, 1].max() - y0 + 1, sup[:, 0].max() - x0 + 1, "frame cut to the points' bounding box")
# histogram of the points' columns at the left end: where does the set begin in each band of rows?
for y in range(0, R, 46):
    band = sup[(sup[:, 1] >= y) & (sup[:, 1] < y + 46)]
    print("rows %3d..%3d: first columns" % (y, y + 45), np.sort(band[:, 0])[:6], "last", np.sort(band[:, 0])[-3:])
