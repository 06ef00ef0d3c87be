"""Right view alone (svx_dense_wta, view = right): the both-views matrix-core search against the mirrored vector sweep (measurement aid, run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import stereovision_amd as S
from stereovision_amd import synth
ctx = S.Context(0)
rows, cols, nf, D = 368, 1232, 64, 128
rng = np.random.default_rng(0)
cl = torch.from_numpy(rng.integers(0, 2**32, size=(nf, rows, cols, 2), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
cr = torch.from_numpy(rng.integers(0, 2**32, size=(nf, rows, cols, 2), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
for mx in (1, 0):
    ctx.set_option(S.OPT_MATRIX_CORES, -1 if mx else 0)
    out = ctx.dense_wta(cl, cr, D, words=2, view=S.VIEW_RIGHT)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = ctx.dense_wta(cl, cr, D, words=2, view=S.VIEW_RIGHT)
    torch.cuda.synchronize()
    print("right view alone, 64 frames 1232x368 / 128, two-word descriptors, matrix cores %s: %.3f ms" % ("on" if mx else "off", (time.perf_counter() - t0) * 100))
    if mx: keep = out.clone()
print("equal:", bool(torch.equal(keep, out)))
ctx.set_option(S.OPT_MATRIX_CORES, -1)
for name, call in (("left view alone", lambda: ctx.dense_wta(cl, cr, D, words=2, view=S.VIEW_LEFT)), ("both views", lambda: ctx.dense_wta_both(cl, cr, D, words=2))):
    call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): call()
    torch.cuda.synchronize()
    print("%s, matrix cores on: %.3f ms" % (name, (time.perf_counter() - t0) * 100))
