"""What a plain streaming kernel sustains on this box: a fill (write only), a copy (read + write) and a reduction (read only) over
buffers of the census output's size and larger.  Measurement aid (run on the GPU box)."""
import torch
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for mb in (232, 464, 1856):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.int32, device="cuda"); b = torch.empty_like(a)
    tf = timed(lambda: a.fill_(7)); tc = timed(lambda: b.copy_(a)); tr = timed(lambda: a.sum())
    print("%5d MB: fill %.4f ms = %.2f TB/s written   copy %.4f ms = %.2f TB/s read + written   sum %.4f ms = %.2f TB/s read"
          % (mb, tf, mb / tf / 1e3, tc, 2 * mb / tc / 1e3, tr, mb / tr / 1e3))
