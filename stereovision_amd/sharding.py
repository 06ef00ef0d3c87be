"""Frame-parallel sharding helpers (host logic of the multi-GPU path).

Rectified stereo pairs are independent units, so frame f of a job goes to rank f // frames_per_rank and
no collective is needed while matching.  torch.distributed (RCCL on GPUs, gloo on CPU) is used only for
the barrier / max-over-ranks around the timed region and to gather disparity maps to rank 0 for
evaluation."""
import torch
import torch.distributed as dist


def frame_seeds(rank, frames_per_rank):
    """Seeds (= global frame indices) of the synthetic frames owned by `rank`."""
    return list(range(rank * frames_per_rank, (rank + 1) * frames_per_rank))


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def barrier():
    if world_size() > 1:
        dist.barrier()


def max_over_ranks(seconds, device):
    """Job time of a step region = the slowest rank's time."""
    if world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def throughput_mpixdisp(rows, cols, max_disp, frames_per_rank, world, steps, seconds):
    """Whole-job Mpix*disp/s: every rank processed frames_per_rank frames per step."""
    return rows * cols * max_disp * frames_per_rank * world * steps / seconds / 1e6


def gather_to_rank0(disp):
    """Gathers every rank's [frames, rows, cols] disparity tensor to rank 0 (a list indexed by rank); None elsewhere."""
    if world_size() == 1:
        return [disp]
    rank = dist.get_rank()
    bufs = [torch.empty_like(disp) for _ in range(world_size())] if rank == 0 else None
    dist.gather(disp, bufs, dst=0)
    return bufs
