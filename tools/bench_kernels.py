"""Times every libsvx entry point at KITTI size (1232x368, D=128) with stream events and prints one JSON
object per kernel with its algorithmic bytes / ops and the achieved rate.  Measurement aid (run on the GPU box)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    rows, cols, D, F = 368, 1232, 128, 64
    ctx = S.Context(0)
    ls, rs = synth.stereo_batch(rows, cols, D, 4, seed0=0)
    left = torch.from_numpy(np.tile(ls, (F // 4, 1, 1))).cuda()
    right = torch.from_numpy(np.tile(rs, (F // 4, 1, 1))).cuda()
    npx = rows * cols * F
    out = []

    def rec(name, ms, bytes_=None, ops=None, note=""):
        r = {"kernel": name, "ms": round(ms, 4), "note": note}
        if bytes_ is not None:
            r["algorithmic_GB"] = round(bytes_ / 1e9, 4)
            r["GBps"] = round(bytes_ / ms / 1e6, 1)
            r["hbm_frac_of_8TBs"] = round(bytes_ / ms / 1e6 / 8000.0, 4)
        if ops is not None:
            r["Tlane_ops_per_s"] = round(ops / ms / 1e9, 3)
            r["valu_frac_of_39.3T"] = round(ops / ms / 1e9 / 39.32, 4)
        out.append(r)
        print(json.dumps(r))

    s8 = torch.empty_like(left, dtype=torch.int8)
    rec("u8_to_s8 (a1)", timed(lambda: ctx.u8_to_s8(left, out=s8)), bytes_=2 * npx, note="64 frames")
    cl = torch.zeros(left.shape, dtype=torch.int32, device="cuda")
    cr = torch.zeros(left.shape, dtype=torch.int32, device="cuda")
    rec("census5x5_u8 (a1+a2)", timed(lambda: ctx.census5x5(left, S.CENSUS_SSE24 | S.CENSUS_ZERO_REST, out=cl)), bytes_=5 * npx,
        ops=48.0 * npx, note="64 frames; ops = 24 compare + 24 carry-add per pixel")
    ctx.census5x5(right, S.CENSUS_SSE24 | S.CENSUS_ZERO_REST, out=cr)
    disp = torch.empty(left.shape, dtype=torch.float32, device="cuda")
    rec("dense_wta5x5 sweep (a8)", timed(lambda: ctx.dense_wta5x5(cl, cr, D, out=disp)), bytes_=12 * npx, ops=2.0 * npx * D,
        note="64 frames; ops = 2 per pix*disp (v_xor + v_bcnt floor)")
    rec("dense_census_wta_u8 fused (a1+a2+a8)", timed(lambda: ctx.dense_census_wta(left, right, D, out=disp)), bytes_=6 * npx,
        ops=2.0 * npx * D, note="64 frames, three launches")

    # plane-prior path on one frame: ~25 % of the pixels are high-gradient points
    rng = np.random.default_rng(0)
    l0, r0 = cl[0].contiguous(), cr[0].contiguous()
    mask = rng.random((rows, cols)) < 0.25
    mask[:2] = mask[-2:] = False
    mask[:, :2] = mask[:, -2:] = False
    ys, xs = np.nonzero(mask)
    pts = torch.from_numpy(np.stack([xs, ys], 1).astype(np.int32)).cuda()
    npts = pts.shape[0]
    prior = torch.from_numpy(rng.uniform(0, 100, size=(rows, cols)).astype(np.float32)).cuda()
    costs = torch.full((rows, cols), 10, dtype=torch.int8, device="cuda")
    rec("cost_eval (a10)", timed(lambda: ctx.cost_eval(l0, r0, pts, prior, costs)), bytes_=npts * (8 + 4 + 4 + 4 + 1),
        note="1 frame, %d points; bytes = point + prior + 2 census words + cost" % npts)
    fd = torch.zeros((rows, cols), dtype=torch.float32, device="cuda")
    fc = torch.full((rows, cols), 25, dtype=torch.int8, device="cuda")
    g = ctx.refine(pts, prior, costs, 2, 10, 64, fd, fc)
    rec("refine (a11, 3 launches)", timed(lambda: ctx.refine(pts, prior, costs, 2, 10, 64, fd, fc)), bytes_=npts * (8 + 4 + 1 + 1 + 4 + 1),
        note="1 frame, %d points, 64-px cells" % npts)
    prior2 = prior.clone()
    rec("support_resampling (a13 with a9, 11x11)", timed(lambda: ctx.support_resampling(l0, r0, g, 5, 11, 2, 10, D, prior2), n=5),
        note="1 frame, %d cells, includes the host read-back of the result list" % (g["gw"] * g["gh"]))
    ep = pts[:: max(1, npts // 4096)][:4096].contiguous()
    rec("epipolar_match_batch (a9, 11x11)", timed(lambda: ctx.epipolar_match_batch(l0, r0, ep, 5, 11, D)),
        ops=2.0 * ep.shape[0] * D * 121, note="%d points; ops = 2 per popcount pair of the 11x11 aggregate" % ep.shape[0])
    # sparse matcher: ~4000 features per image
    nl = 4000
    lf = np.stack([rng.integers(8, cols - 8, nl), rng.integers(4, rows - 4, nl)], 1).astype(np.float32)
    rf = np.stack([rng.integers(8, cols - 8, nl), rng.integers(4, rows - 4, nl)], 1).astype(np.float32)
    rec("sparse_match (a5+a6, host in/out)", timed(lambda: ctx.sparse_match(l0, r0, lf, rf, 70, 1.0, 0.5, 1), n=5),
        note="%d x %d features, maxDisp 70; includes host sort, upload and read-back" % (nl, nl))
    # plane prior: ~2N triangles over N = 2000 points (fan-free random triangulation substitute: grid split)
    gx, gy = np.meshgrid(np.arange(0, cols - 1, 28), np.arange(0, rows - 1, 23))
    tris = []
    for j in range(gx.shape[0] - 1):
        for i in range(gx.shape[1] - 1):
            a = (gx[j, i], gy[j, i]); b = (gx[j, i + 1], gy[j, i + 1]); c = (gx[j + 1, i], gy[j + 1, i]); d = (gx[j + 1, i + 1], gy[j + 1, i + 1])
            tris.append([*a, *b, *c]); tris.append([*b, *d, *c])
    tri = torch.from_numpy(np.array(tris, np.int32)).cuda()
    abc = ctx.plane_fit(tri, prior)
    rec("plane_fit (a12)", timed(lambda: ctx.plane_fit(tri, prior)), note="%d triangles" % tri.shape[0])
    rec("plane_prior_raster (a12, 3 launches)", timed(lambda: ctx.plane_prior_raster(tri, abc, prior2)), bytes_=rows * cols * (4 + 4 + 4),
        note="%d triangles; bytes = owner map write+read and prior write per pixel" % tri.shape[0])
    ctx.close()


if __name__ == "__main__":
    main()
