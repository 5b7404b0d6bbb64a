"""Env-index sharding across the GPUs of one node.

Every env is independent (no cross-env reads anywhere in game.py / kinematics.py / rewards.py),
so the step path needs NO collective: rank r owns the contiguous block
[offset, offset + count) of global env indices, resident on its GPU for the whole run, and the
counter-based reset RNG is keyed on the GLOBAL index, so results do not depend on the sharding.
The only optional exchange is gathering per-rank rollout statistics to rank 0, off the step path.
"""
import os

import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
    """Contiguous block of `total_envs` owned by `rank` (the first `total % world` ranks get one
    extra env).  Returns (offset, count)."""
    base, rem = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_process_group(backend=None):
    """One process per GPU; backend "nccl" is RCCL on ROCm, "gloo" for CPU tests."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over ranks (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_episode_stats(returns, lengths, dst=0):
    """Optional, off the step path: gather variable-length per-rank episode statistics
    (1-D tensors) to rank `dst`.  Returns (returns, lengths) lists on dst, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [returns], [lengths]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([returns.numel()], dtype=torch.int64, device=returns.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    pad_r = torch.zeros(m, dtype=returns.dtype, device=returns.device)
    pad_l = torch.zeros(m, dtype=lengths.dtype, device=lengths.device)
    pad_r[:returns.numel()] = returns
    pad_l[:lengths.numel()] = lengths
    out_r = [torch.zeros_like(pad_r) for _ in range(world)]
    out_l = [torch.zeros_like(pad_l) for _ in range(world)]
    dist.all_gather(out_r, pad_r)
    dist.all_gather(out_l, pad_l)
    if rank != dst:
        return None, None
    return ([r[:int(s.item())] for r, s in zip(out_r, sizes)],
            [l[:int(s.item())] for l, s in zip(out_l, sizes)])
