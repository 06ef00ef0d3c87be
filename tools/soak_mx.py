"""Random sizes through the matrix-core search (left view, both views, fused left-right entry) against the vector sweep:
widths that leave idle wavefronts or a shifted last x-block, short strips, every tile count and pass width, one- and two-word descriptors.
Measurement aid (run on the GPU box): tools/soak_mx.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = S.Context(0)
bad = 0
for case in range(n_cases):
    cols = int(rng.choice([rng.integers(36, 80), rng.integers(80, 400), rng.integers(400, 1400)]))
    rows = int(rng.integers(5, 90))
    D = int(rng.choice([rng.integers(1, 65), rng.integers(65, 129), rng.integers(129, 400), rng.integers(400, 1025)]))
    words = int(rng.integers(1, 3))
    frames = int(rng.integers(1, 4))
    strip = int(rng.choice([0, 0, rng.integers(1, 12)]))
    wide = int(rng.choice([-1, -1, 0, 1]))
    ls, rs = synth.stereo_batch(rows, cols, min(D, max(cols // 2, 1)), frames, seed0=1000 + case)
    dl, dr = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
    cl, cr = (ctx.census9x7(dl), ctx.census9x7(dr)) if words == 2 else (ctx.census5x5(dl), ctx.census5x5(dr))
    ctx.set_option(S.OPT_MATRIX_CORES, 0)
    wl = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, words)
    wr = ctx.dense_wta(cl, cr, D, S.VIEW_RIGHT, words)
    ctx.set_option(S.OPT_MATRIX_CORES, -1)
    ctx.set_option(S.OPT_MX_STRIP_ROWS, strip)
    ctx.set_option(S.OPT_MX_WIDE_TILES, wide)
    ml, mr = ctx.dense_wta_both(cl, cr, D, words)
    ml0 = ctx.dense_wta(cl, cr, D, S.VIEW_LEFT, words)
    ok = torch.equal(ml, wl) and torch.equal(mr, wr) and torch.equal(ml0, wl)
    if words == 2:
        fused = ctx.dense_census9x7_lr(dl, dr, D, lr_threshold=1.0)
        want = ctx.lr_check(wl.clone(), wr, 1.0)
        ok = ok and torch.equal(fused, want)
    ctx.set_option(S.OPT_MX_STRIP_ROWS, 0)
    ctx.set_option(S.OPT_MX_WIDE_TILES, -1)
    if not ok:
        bad += 1
        print("MISMATCH rows=%d cols=%d D=%d words=%d frames=%d strip=%d wide=%d: left %d, left-only %d, right %d" % (
            rows, cols, D, words, frames, strip, wide, int((ml != wl).sum()), int((ml0 != wl).sum()), int((mr != wr).sum())), flush=True)
    if case % 20 == 19:
        print("%d cases, %d mismatches" % (case + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
