"""What the second pass of the device triangulation spends on each point it takes: ring steps, searches of all points, time
(measurement aid; needs a library built with -DSVX_MESH_PROBE, named by SVX_LIBRARY; run on the GPU box)."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
import stereovision_amd as S
from stereovision_amd import synth, _lib

ctx = S.Context(0)
rows, cols = 375, 1242
R, C = 368, 1232
lib = ctypes.CDLL(_lib.lib_path())
dump = lib.svx_mesh_probe_dump
dump.restype = ctypes.c_int
l, r = [torch.from_numpy(a).cuda() for a in synth.stereo_pair(rows, cols, 64, 0)[:2]]
for iters in (1, 2, 3):
    p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=iters)
    out = torch.zeros((rows, cols), dtype=torch.float32, device="cuda")
    pts = ctx.high_perf_stereo(l, r, p, out)
    key = pts[:, 1].astype(np.int64) * 4096 + pts[:, 0]
    brk = np.flatnonzero(key[1:] < key[:-1])
    sup = pts[brk[0] + 1:] if len(brk) else pts
    d = torch.from_numpy(np.ascontiguousarray(sup, np.int32)).cuda()
    buf = (ctypes.c_int * (8192 * 6))()
    dump(buf, 8192)
    ctx.delaunay_device(d, R, C)
    n = dump(buf, 8192)
    a = np.frombuffer(buf, dtype=np.int32)[:n * 6].reshape(n, 6)
    print("support points after %d iteration(s): %d; second pass takes %d points" % (iters, len(sup), n))
    if n == 0:
        continue
    t = a[:, 3] / 100.0
    print("  time per point us: median %.1f  90%% %.1f  max %.1f ; steps median %d max %d ; searches of all points median %d max %d ; held candidates median %d max %d"
          % (np.median(t), np.percentile(t, 90), t.max(), np.median(a[:, 1]), a[:, 1].max(), np.median(a[:, 2]), a[:, 2].max(), np.median(a[:, 4]), a[:, 4].max()))
    for row in a[np.argsort(-a[:, 3])[:8]]:
        print("    x %4d y %3d  steps %3d  searches %3d  held %4d  %.1f us  (%.2f us a step)" % (row[5] & 0xffff, row[5] >> 16, row[1], row[2], row[4], row[3] / 100.0, row[3] / 100.0 / max(row[1], 1)))
    fit = np.polyfit(a[:, 1], t, 1)
    print("  time ~ %.2f us x steps + %.1f us; correlation with searches: %.2f" % (fit[0], fit[1], np.corrcoef(a[:, 2], t)[0, 1]))
