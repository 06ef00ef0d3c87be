"""Seeded cases of every routine on the path, run through one of two back ends (the CPU oracle, or libsvx on the GPU) and
reduced to FNV-1a hashes.  tests/golden/make_golden.py stores the oracle's hashes in tests/golden/oracle_hashes.json;
tests/test_oracle_golden.py checks that the oracle still produces them, tests/test_gpu_golden.py that libsvx does, at
sizes up to the full 368x1232 / 128 disparities, without running the oracle on the GPU box."""
import numpy as np

import oracle_lib as O

SIZES = [(32, 64, 16), (64, 128, 32), (480, 640, 64), (368, 1232, 128)]          # rows, cols, disparities (480x640/64 = configs[0])


def inputs(rows, cols, D):
    from stereovision_amd import synth
    left, right, _ = synth.stereo_pair(rows, cols, D, seed=1000 + rows)
    rng = np.random.default_rng(rows * 7 + cols)
    n = min(500, rows * cols // 8)
    ref_xy = np.stack([rng.integers(2, cols - 3, n), rng.integers(2, rows - 2, n)], 1).astype(np.int32)
    cmp_xy = np.stack([np.maximum(2, ref_xy[:, 0] - rng.integers(0, D, n)), np.clip(ref_xy[:, 1] + rng.integers(-1, 2, n), 2, rows - 3)], 1).astype(np.int32)
    y, x = np.mgrid[0:rows, 0:cols]
    prior = (np.float32(0.2) * np.float32(D) + np.float32(0.01) * x.astype(np.float32) + np.float32(0.03) * y.astype(np.float32)).astype(np.float32)
    keep = ((x + 2 * y) % 3 == 0) & (y >= 2) & (y < rows - 2) & (x >= 2) & (x < cols - 2)
    pts = np.stack([x[keep], y[keep]], 1).astype(np.int32)                       # raster order
    feat_l = np.stack([x[(x % 7 == 3) & (y % 5 == 2) & (y >= 4) & (y < rows - 4) & (x >= 4) & (x < cols - 4)],
                       y[(x % 7 == 3) & (y % 5 == 2) & (y >= 4) & (y < rows - 4) & (x >= 4) & (x < cols - 4)]], 1).astype(np.float32)
    feat_r = np.stack([x[(x % 2 == 1) & (y % 5 == 2) & (y >= 4) & (y < rows - 4) & (x >= 4) & (x < cols - 4)],
                       y[(x % 2 == 1) & (y % 5 == 2) & (y >= 4) & (y < rows - 4) & (x >= 4) & (x < cols - 4)]], 1).astype(np.float32)
    return dict(left=left, right=right, ref_xy=ref_xy, cmp_xy=cmp_xy, prior=prior, pts=pts, feat_l=feat_l, feat_r=feat_r)


class OracleBackend:
    name = "oracle"

    def run(self, rows, cols, D, inp):
        out = {}
        sl, sr = O.u8_to_s8(inp["left"]), O.u8_to_s8(inp["right"])
        out["a1_left_s8"] = sl
        cl, cr = O.census5x5(sl), O.census5x5(sr)
        out["a2_census_left"], out["a2_census_right"] = cl, cr
        out["a2g_census_left_generic25"] = O.census5x5(sl, "generic25")
        for mode in ("5x6", "5x5"):
            out["a4_window_" + mode] = np.array([O.window_cost(cl, cr, int(r[0]), int(r[1]), int(c[0]), int(c[1]), mode)
                                                  for r, c in zip(inp["ref_xy"], inp["cmp_xy"])], np.int16)
        ml, mr, mc = O.sparse_match(cl, cr, inp["feat_l"], inp["feat_r"], D, 1.0, 0.7, 1)
        out["a5_sparse_left"], out["a5_sparse_right"], out["a5_sparse_cost"] = ml, mr, mc
        out["a8_dense_wta5x5"] = O.dense_wta5x5(cl, cr, D)
        ep = [O.epipolar_match(cl, cr, 5, 11, int(p[0]), int(p[1]), D) for p in inp["ref_xy"][:200]]
        out["a9_epipolar_d"] = np.array([e[0] for e in ep], np.float32)
        out["a9_epipolar_c"] = np.array([e[1] for e in ep], np.int8)
        costs = np.full((rows, cols), 10, np.int8)
        O.cost_eval(cl, cr, inp["pts"], inp["prior"], costs)
        out["a10_costs"] = costs
        fd, fc = np.zeros((rows, cols), np.float32), np.full((rows, cols), 25, np.int8)
        g = O.refine(inp["pts"], inp["prior"], costs, 2, 10, 16, fd, fc)
        out["a11_final_disp"], out["a11_final_costs"] = fd, fc
        for k in ("conf_x", "conf_y", "conf_d", "conf_c", "inv_x", "inv_y", "inv_c"):
            out["a11_grid_" + k] = g[k]
        c9l, c9r = O.census9x7(inp["left"]), O.census9x7(inp["right"])
        out["w2_census9x7_left"], out["w2_census9x7_right"] = c9l, c9r
        dl, dr = O.dense_wta_words(c9l, c9r, D, 0), O.dense_wta_words(c9l, c9r, D, 1)
        out["w2_dense_left"], out["w2_dense_right"] = dl, dr
        out["w2_lr_checked"] = O.lr_check(dl.copy(), dr, 1.0, -1.0)
        out["w2_subpixel"] = O.subpixel_refine(c9l, c9r, D, dl.copy())
        return out


class GpuBackend:
    name = "libsvx"

    def __init__(self, ctx):
        import torch
        self.ctx, self.t = ctx, torch

    def dev(self, a):
        return self.t.from_numpy(np.ascontiguousarray(a)).cuda()

    def run(self, rows, cols, D, inp):
        import stereovision_amd as S
        ctx, dev, t = self.ctx, self.dev, self.t
        host = lambda v: v.cpu().numpy()
        out = {}
        left, right = dev(inp["left"]), dev(inp["right"])
        sl, sr = ctx.u8_to_s8(left), ctx.u8_to_s8(right)
        out["a1_left_s8"] = host(sl)
        cl, cr = ctx.census5x5(sl), ctx.census5x5(sr)
        out["a2_census_left"], out["a2_census_right"] = host(cl).view(np.uint32), host(cr).view(np.uint32)
        out["a2g_census_left_generic25"] = host(ctx.census5x5(sl, S.CENSUS_GENERIC25)).view(np.uint32)
        for mode, flag in (("5x6", S.WINDOW_5X6), ("5x5", S.WINDOW_5X5)):
            out["a4_window_" + mode] = host(ctx.census_window_cost(cl, cr, dev(inp["ref_xy"]), dev(inp["cmp_xy"]), flag))
        ml, mr, mc = ctx.sparse_match(cl, cr, inp["feat_l"], inp["feat_r"], D, 1.0, 0.7, 1)
        out["a5_sparse_left"], out["a5_sparse_right"], out["a5_sparse_cost"] = ml, mr, mc
        out["a8_dense_wta5x5"] = host(ctx.dense_wta5x5(cl, cr, D))
        d, c = ctx.epipolar_match_batch(cl, cr, dev(inp["ref_xy"][:200]), 5, 11, D)
        out["a9_epipolar_d"], out["a9_epipolar_c"] = host(d), host(c)
        costs = t.full((rows, cols), 10, dtype=t.int8, device="cuda")
        prior, pts = dev(inp["prior"]), dev(inp["pts"])
        ctx.cost_eval(cl, cr, pts, prior, costs)
        out["a10_costs"] = host(costs)
        fd = t.zeros((rows, cols), dtype=t.float32, device="cuda")
        fc = t.full((rows, cols), 25, dtype=t.int8, device="cuda")
        g = ctx.refine(pts, prior, costs, 2, 10, 16, fd, fc)
        out["a11_final_disp"], out["a11_final_costs"] = host(fd), host(fc)
        for k in ("conf_x", "conf_y", "conf_d", "conf_c", "inv_x", "inv_y", "inv_c"):
            out["a11_grid_" + k] = host(g[k])
        c9l, c9r = ctx.census9x7(left), ctx.census9x7(right)
        out["w2_census9x7_left"], out["w2_census9x7_right"] = host(c9l).view(np.uint32), host(c9r).view(np.uint32)
        dl, dr = ctx.dense_wta(c9l, c9r, D, S.VIEW_LEFT, 2), ctx.dense_wta(c9l, c9r, D, S.VIEW_RIGHT, 2)
        out["w2_dense_left"], out["w2_dense_right"] = host(dl), host(dr)
        out["w2_lr_checked"] = host(ctx.lr_check(dl.clone(), dr, 1.0))
        out["w2_subpixel"] = host(ctx.subpixel_refine(c9l, c9r, D, dl.clone(), words=2))
        # the fused entry points must reproduce the composed results
        out["a8_dense_wta5x5_fused_equal"] = np.array([int(t.equal(ctx.dense_census_wta(left, right, D), ctx.dense_wta5x5(
            ctx.census5x5(left, S.CENSUS_SSE24 | S.CENSUS_ZERO_REST), ctx.census5x5(right, S.CENSUS_SSE24 | S.CENSUS_ZERO_REST), D)))], np.int32)
        out["w2_lr_fused_equal"] = np.array([int(np.array_equal(host(ctx.dense_census9x7_lr(left, right, D, 1.0)), out["w2_lr_checked"]))], np.int32)
        return out


def hashes(backend, sizes=SIZES):
    res = {}
    for rows, cols, D in sizes:
        got = backend.run(rows, cols, D, inputs(rows, cols, D))
        res["%dx%d_d%d" % (rows, cols, D)] = {k: O.fnv(np.ascontiguousarray(v)) for k, v in sorted(got.items()) if not k.endswith("_equal")}
        for k, v in got.items():
            if k.endswith("_equal"):
                assert int(v[0]) == 1, k
    return res
