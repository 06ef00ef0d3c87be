"""Times the both-views matrix-core search (and the left-view one) of the library named by SVX_LIBRARY on the benchmark
workload, and compares both maps with the vector sweep.  Measurement aid (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402
from stereovision_amd._lib import _dptr  # noqa: E402

rows, cols, D, F = 368, 1232, int(os.environ.get("MX_D", "128")), 64
check = os.environ.get("MX_CHECK", "1") == "1"
ctx = S.Context(0)
ls, rs = synth.stereo_batch(rows, cols, D, 8, seed0=0)
left = torch.from_numpy(ls).cuda().repeat(F // 8, 1, 1)
right = torch.from_numpy(rs).cuda().repeat(F // 8, 1, 1)
cl, cr = ctx.census9x7(left), ctx.census9x7(right)
if os.environ.get("MX_ZERO", "0") == "1":       # all-zero descriptors: the same instructions on operands that toggle nothing (what the clock does under them)
    cl.zero_(); cr.zero_(); check = False
disp = torch.empty((F, rows, cols), dtype=torch.float32, device="cuda")


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


both = timed(lambda: ctx._call("svx_probe_both_views_sweep", _dptr(cl), _dptr(cr), rows, cols, F, 2, D, _dptr(disp)))
ok = "unchecked"
if check:
    ml, mr = ctx.dense_wta_matrix(cl[:8], cr[:8], D)
    ctx.set_option(S.OPT_MATRIX_CORES, 0)
    wl, wr = ctx.dense_wta(cl[:8], cr[:8], D, S.VIEW_LEFT, 2), ctx.dense_wta(cl[:8], cr[:8], D, S.VIEW_RIGHT, 2)
    ok = "exact" if torch.equal(ml, wl) and torch.equal(mr, wr) else "WRONG (left %d, right %d pixels differ)" % (int((ml != wl).sum()), int((mr != wr).sum()))
print("%-28s both-views sweep %.4f ms   results %s" % (os.path.basename(os.environ.get("SVX_LIBRARY", "libsvx.so")), both, ok), flush=True)
