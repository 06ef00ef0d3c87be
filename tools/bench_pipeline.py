"""Times svx_high_perf_stereo (the full support-point / plane-prior pipeline of the reference) on KITTI-size
synthetic pairs and, beside it, the CPU oracle composition of the same pipeline.  Measurement aid (run on the GPU box)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import stereovision_amd as S  # noqa: E402
from stereovision_amd import synth  # noqa: E402


def single(n_iters, frames=20):
    """Only a stream of single-threaded frames (for a kernel trace): prints frames and wall time."""
    rows, cols = 375, 1242
    ctx = S.Context(0)
    p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=n_iters)
    l, r = [torch.from_numpy(a).cuda() for a in synth.stereo_pair(rows, cols, 64, 0)[:2]]
    out = torch.zeros((rows, cols), dtype=torch.float32, device="cuda")
    ctx.high_perf_stereo(l, r, p, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.high_perf_stereo(l, r, p, out)
    torch.cuda.synchronize()
    print(json.dumps({"frames": frames + 1, "nIters": n_iters, "ms_per_frame": round((time.perf_counter() - t0) * 1e3 / frames, 3),
                      "mesh_points_left_by_pass_1_and_2": ctx.mesh_device_stats()}))


def main():
    if "--single" in sys.argv:
        return single(int(sys.argv[sys.argv.index("--single") + 1]))
    rows, cols = 375, 1242
    ctx = S.Context(0)
    for n_iters in (1, 2, 3):
        p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=n_iters)
        pairs = [synth.stereo_pair(rows, cols, 64, seed)[:2] for seed in range(4)]
        dev = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in pairs]
        out = torch.zeros((rows, cols), dtype=torch.float32, device="cuda")
        ctx.high_perf_stereo(dev[0][0], dev[0][1], p, out)          # warm-up (workspace growth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_pts = 0
        for l, r in dev:
            out.zero_()
            n_pts += len(ctx.high_perf_stereo(l, r, p, out))
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) * 1e3 / len(dev)
        ctx.set_option(S.OPT_STAGE_TIMES, 1)                        # one more frame with a synchronisation at every stage mark
        out.zero_()
        ctx.high_perf_stereo(dev[-1][0], dev[-1][1], p, out)
        stages = ctx.pipeline_stage_times()
        ctx.set_option(S.OPT_STAGE_TIMES, 0)
        rec = {"pipeline": "svx_high_perf_stereo", "rows": rows, "cols": cols, "nIters": n_iters, "ms_per_frame": round(gpu_ms, 2),
               "frames_per_s": round(1e3 / gpu_ms, 1), "points_per_frame": n_pts // len(dev), "stage_ms_last_frame": stages}
        if "--cpu" in sys.argv:
            import oracle_pipeline as OP
            dm = np.zeros((rows, cols), np.float32)
            t0 = time.perf_counter()
            OP.high_perf_stereo(pairs[0][0], pairs[0][1], p, dm, S.delaunay)
            rec["cpu_oracle_ms_per_frame"] = round((time.perf_counter() - t0) * 1e3, 1)
        print(json.dumps(rec), flush=True)
    # the batch entry point: host threads inside the library, a context and a stream each
    p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=2)
    ls, rs = synth.stereo_batch(rows, cols, 64, 32, seed0=0)
    left, right = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
    maps = torch.zeros((32, rows, cols), dtype=torch.float32, device="cuda")
    for T in (1, 2, 4, 8, 16):
        ctx.high_perf_stereo_batch(left, right, p, maps, n_threads=T, copy=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.high_perf_stereo_batch(left, right, p, maps, n_threads=T, copy=False)      # (the library's call alone: no copies of the points on the Python side)
        dt = time.perf_counter() - t0
        print(json.dumps({"pipeline": "svx_high_perf_stereo_batch", "nIters": 2, "host_threads": T, "frames": 32,
                          "ms_per_frame": round(dt * 1e3 / 32, 2), "frames_per_s": round(32 / dt, 1)}), flush=True)
    ctx.close()
    # Frames are independent: host threads with a context each (the host share of a frame, the triangulation, is what
    # limits one thread) keep one GPU busy.  ctypes releases the interpreter lock during a call.
    if "--threads" in sys.argv:
        import threading
        p = S.StereoParameters.reference_defaults(maxDisp=70, nIters=2)
        pairs = [synth.stereo_pair(rows, cols, 64, seed)[:2] for seed in range(4)]
        for T in (1, 2, 4, 8, 16):
            ctxs = [S.Context(0) for _ in range(T)]
            streams = [torch.cuda.Stream() for _ in range(T)]
            for c, st in zip(ctxs, streams):
                c.bind_stream(st)                                    # a stream per context: no thread waits for another's kernels
            devs = [[(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in pairs] for _ in range(T)]
            outs = [torch.zeros((rows, cols), dtype=torch.float32, device="cuda") for _ in range(T)]
            per_thread = 8
            def work(i, n):
                for k in range(n):
                    l, r = devs[i][k % len(pairs)]
                    ctxs[i].high_perf_stereo(l, r, p, outs[i])
            for i in range(T):
                work(i, 1)                                           # warm-up (workspace growth)
            torch.cuda.synchronize()
            th = [threading.Thread(target=work, args=(i, per_thread)) for i in range(T)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            secs = time.perf_counter() - t0
            print(json.dumps({"pipeline": "svx_high_perf_stereo", "nIters": 2, "host_threads": T, "frames": T * per_thread,
                              "frames_per_s": round(T * per_thread / secs, 1), "ms_per_frame_per_thread": round(secs / per_thread * 1e3, 2)}), flush=True)
            for c in ctxs:
                c.close()


if __name__ == "__main__":
    main()
