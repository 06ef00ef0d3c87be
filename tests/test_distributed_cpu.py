"""Multi-process (gloo, world_size 2 and 4, CPU) test of the frame-sharding logic bench.py uses on N GPUs:
every rank owns its own batch (weak scaling), the timed region is bracketed by barriers, the step time
is the MAX over ranks, the throughput is the units of all ranks over that time, and the evaluation
gather brings every rank's disparity maps to rank 0.  The compute itself is replaced by the CPU oracle
here (this is a test of the host logic, not of the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from stereovision_amd import sharding, synth

    rows, cols, D, frames_per_rank = 24, 64, 16, 2
    seeds = sharding.frame_seeds(rank, frames_per_rank)
    assert seeds == list(range(rank * frames_per_rank, (rank + 1) * frames_per_rank))
    ls, rs = synth.stereo_batch(rows, cols, D, frames_per_rank, seed0=seeds[0])
    disp = np.stack([O.dense_wta5x5(O.census5x5(O.u8_to_s8(l)), O.census5x5(O.u8_to_s8(r)), D) for l, r in zip(ls, rs)])
    # each rank reports a different local time; the job time is the max
    local_t = 0.5 + 0.25 * rank
    job_t = sharding.max_over_ranks(local_t, torch.device("cpu"))
    assert abs(job_t - (0.5 + 0.25 * (world - 1))) < 1e-12
    value = sharding.throughput_mpixdisp(rows, cols, D, frames_per_rank, world, steps=3, seconds=job_t)
    assert abs(value - rows * cols * D * frames_per_rank * world * 3 / job_t / 1e6) < 1e-9
    gathered = sharding.gather_to_rank0(torch.from_numpy(disp))
    if rank == 0:
        assert len(gathered) == world
        for r in range(world):
            l2, r2 = synth.stereo_batch(rows, cols, D, frames_per_rank, seed0=r * frames_per_rank)
            want = np.stack([O.dense_wta5x5(O.census5x5(O.u8_to_s8(a)), O.census5x5(O.u8_to_s8(b)), D) for a, b in zip(l2, r2)])
            assert np.array_equal(gathered[r].numpy(), want)
    else:
        assert gathered is None
    # variable-length point clouds (one per rank, different sizes, an empty one included) gathered to rank 0
    sizes = [0, 37] if world == 2 else [5, 0, 61, 23]
    n_mine = sizes[rank]
    Q = np.array([[1, 0, 0, -30.0], [0, 1, 0, -12.0], [0, 0, 0, 700.0], [0, 0, 1.0 / 0.5, 0]], np.float64)
    ys, xs = np.nonzero(disp[0] > 0)
    pts = np.stack([xs, ys], 1).astype(np.int32)[:n_mine]
    xyz, _ = O.point_cloud(pts, disp[0], Q)
    clouds, counts = sharding.gather_point_clouds_to_rank0(torch.from_numpy(xyz))
    assert counts[rank] == len(xyz) and len(counts) == world
    if rank == 0:
        for r in range(world):
            l2, r2 = synth.stereo_batch(rows, cols, D, 1, seed0=r * frames_per_rank)
            d2 = O.dense_wta5x5(O.census5x5(O.u8_to_s8(l2[0])), O.census5x5(O.u8_to_s8(r2[0])), D)
            y2, x2 = np.nonzero(d2 > 0)
            want, _ = O.point_cloud(np.stack([x2, y2], 1).astype(np.int32)[:sizes[r]], d2, Q)
            assert clouds[r].shape == (counts[r], 3) and np.array_equal(clouds[r].numpy(), want)
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert clouds is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharding_and_gathers(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): the parent starts the two ranks itself, before
    anything touches a GPU, and rank 0 prints one JSON line with n_gpus 2 (--dry-run: the rank plumbing only, no GPU)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["dry_run"] is True


def test_bench_launch_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(4, 29512, ["--gpus", "4", "--steps", "7"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert os.path.basename(cmd[-5]) == "bench.py"
