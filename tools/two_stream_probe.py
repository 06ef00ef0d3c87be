"""Does a batch split over two streams run faster than one launch sequence?  The census and the check of one half can share
the chip with the search of the other half.  Measurement aid (run on the GPU box): tools/two_stream_probe.py [frames] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows, cols, D = 368, 1232, 128
ls, rs = synth.stereo_batch(rows, cols, D, frames, seed0=0)
left, right = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = S.Context(0)
out = torch.empty((frames, rows, cols), dtype=torch.float32, device="cuda")
ref = one.dense_census9x7_lr(left, right, D, out=out).clone()
print("one stream, %d frames: %.4f ms per step" % (frames, timed(lambda: one.dense_census9x7_lr(left, right, D, out=out), steps)), flush=True)

for parts in (2, 4):
    ctxs, streams = [], []
    for _ in range(parts):
        s = torch.cuda.Stream()
        c = S.Context(0)
        c.bind_stream(s)
        ctxs.append(c); streams.append(s)
    cut = [frames * i // parts for i in range(parts + 1)]
    outs = torch.empty_like(out)
    torch.cuda.synchronize()

    def step():
        for i, c in enumerate(ctxs):
            a, b = cut[i], cut[i + 1]
            c.dense_census9x7_lr(left[a:b], right[a:b], D, out=outs[a:b])

    ms = timed(step, steps)
    print("%d streams, %d frames each: %.4f ms per step, equal %s" % (parts, frames // parts, ms, bool(torch.equal(outs, ref))), flush=True)
