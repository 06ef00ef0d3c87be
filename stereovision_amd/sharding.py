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


def gather_point_clouds_to_rank0(xyz):
    """Gathers every rank's [n_r, 3] point tensor (n_r differs per rank) to rank 0: first the counts (one all_gather of
    an int64 per rank), then one grouped batch of point-to-point transfers in which every peer with points sends exactly
    its n_r points and rank 0 receives each into a buffer of that size (no padding to the largest cloud: over xGMI every
    peer has its own link to GPU 0, so the transfers run side by side).  Returns (list of [n_r, 3] tensors indexed by
    rank, counts) on rank 0 and (None, counts) elsewhere."""
    if world_size() == 1:
        return [xyz], [int(xyz.shape[0])]
    rank, world = dist.get_rank(), world_size()
    xyz = xyz.contiguous()
    n = torch.tensor([xyz.shape[0]], dtype=torch.int64, device=xyz.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    if rank == 0:
        bufs = [xyz] + [torch.empty((counts[r], 3), dtype=xyz.dtype, device=xyz.device) for r in range(1, world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(1, world) if counts[r] > 0]
    else:
        bufs = None
        ops = [dist.P2POp(dist.isend, xyz, 0)] if counts[rank] > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return bufs, counts
