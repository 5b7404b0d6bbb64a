"""Env-index sharding across the GPUs of one node.

Every env is independent (no cross-env reads anywhere in game.py / kinematics.py / rewards.py),
so the step path needs NO collective: rank r owns the contiguous block
[offset, offset + count) of global env indices, resident on its GPU for the whole run, and the
counter-based reset RNG is keyed on the GLOBAL index, so results do not depend on the sharding.
The only optional exchanges are off the step path: gathering per-rank episode statistics, and gathering the
rollout buffers of a collection ([T, E / world, ...] per rank) to a learner rank (`gather_rollout`).
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


def gather_rollout(buffers, dst=0, total_envs=None):
    """Optional, off the step path (BASELINE.json north star: "optional RCCL gather of rollout buffers only"):
    gather the rollout buffers of one collection to the learner rank `dst`.

    buffers      dict name -> tensor [T, E_rank, ...] -- what `ACAS2DVecEnv.rollout()` / `collect()` return (obs,
                 actions, reward, done, values, ...); same T, trailing shape and dtype on every rank, E_rank = this
                 rank's `shard_range` count (ranks may differ by one env).  Keys that start with "_" are skipped.
    Returns      on `dst`: dict name -> [T, E_total, ...] with the envs in GLOBAL index order (rank r's block at
                 `shard_range(total, r, world)`); None on the other ranks.

    One `torch.distributed.gather` per buffer (bool buffers travel as uint8).  With backend "nccl" (RCCL) a gather
    is grouped point-to-point sends into `dst`: on MI355X every sender uses its own xGMI link to the learner
    (7 links x ~153 GB/s in parallel), where a ring all-gather would be bound by ONE link per hop and would deliver
    the data to seven ranks that do not need it.  At BASELINE configs[3] (131 072 envs per GPU, N = 8, float32:
    29 observation values + action + reward + done per env and step) one step of the whole shard is 16.8 MB per
    GPU: ~0.11 ms on its link, against 10 us for the step itself -- which is why this is never on the step path
    (gather a T-step collection once per PPO iteration, overlapped with the next collection)."""
    names = [k for k in sorted(buffers) if not k.startswith("_") and torch.is_tensor(buffers[k])]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: buffers[k] for k in names}
    world, rank = dist.get_world_size(), dist.get_rank()
    e_rank = int(buffers[names[0]].shape[1])
    counts = [torch.zeros(1, dtype=torch.int64, device=buffers[names[0]].device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([e_rank], dtype=torch.int64, device=buffers[names[0]].device))
    counts = [int(c.item()) for c in counts]
    if total_envs is not None and sum(counts) != int(total_envs):
        raise ValueError("gather_rollout: the shards hold %d envs, expected %d" % (sum(counts), total_envs))
    e_max = max(counts)
    out = {}
    for k in names:
        t = buffers[k]
        if t.shape[1] != e_rank:
            raise ValueError("gather_rollout: buffer %r has %d envs, %r has %d" % (k, t.shape[1], names[0], e_rank))
        as_bool = t.dtype == torch.bool
        src = t.view(torch.uint8) if as_bool else t
        if e_rank != e_max:                      # ranks differ by at most one env: pad the short ones
            pad = torch.zeros((src.shape[0], e_max) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
            pad[:, :e_rank] = src
            src = pad
        src = src.contiguous()
        parts = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
        dist.gather(src, parts, dst=dst)
        if rank == dst:
            full = torch.cat([p[:, :c] for p, c in zip(parts, counts)], dim=1)
            out[k] = full.view(torch.bool) if as_bool else full
    return out if rank == dst else None
