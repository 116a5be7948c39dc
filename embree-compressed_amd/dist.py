"""Multi-GPU plumbing: one process per GPU, ray batches sharded by contiguous index range, BVH replicated,
no data-path collective.  torch.distributed is used only for the barrier and for the
MAX-over-ranks of the timed region; backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """[begin, end) of the rays owned by `rank`: contiguous, disjoint, covering [0,total)."""
    return (rank * total) // world, ((rank + 1) * total) // world


def batch_seed(rank, step):
    """Seed of the synthetic batch `step` of `rank`: distinct per (rank, step) for up to 100003 steps."""
    return rank * 100003 + step


def barrier(world, device=None):
    if world <= 1:
        return
    import torch.distributed as dist
    dist.barrier()


def max_over_ranks(value, world, device="cpu"):
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, world, device="cpu"):
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_rate(rays_per_rank, elapsed_local, world, device="cpu"):
    """Whole-job throughput: all rays of all ranks / MAX over ranks of the timed region (bench.py contract)."""
    total = sum_over_ranks(rays_per_rank, world, device)
    worst = max_over_ranks(elapsed_local, world, device)
    return total / worst, worst
