"""World-size-2 `gloo` test of the N > 1 path (CPU, no GPU): env-index sharding with no
collective on the step path.  Each rank steps its own block [offset, offset + count) -- here
with the CPU oracle standing in for the kernel -- and the only communication is the bench's
barrier / MAX-over-ranks timing and the optional off-path statistics gather.  The union of the
shards must equal the unsharded run bit for bit (episodes are keyed on the GLOBAL env index)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, n_traffic, steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import gym_acas2d_amd as g
    from oracle import oracle as O
    r, _, w = g.sharding.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    off, cnt = g.shard_range(total, rank, world)
    env = O.OracleEnvs(cnt, n_traffic, seed=13, env_offset=off, auto_reset=True)
    obs = [env.reset().copy()]
    rng = np.random.default_rng(0)
    done_total = 0
    for _ in range(steps):
        a = rng.uniform(-1, 1, total)[off:off + cnt]          # same global action stream
        o, _, d, _, n = env.step(a)
        obs.append(o.copy())
        done_total += n
    dist.barrier()
    t_max = g.sharding.max_over_ranks(1.0 + rank)
    n_sum = g.sharding.sum_over_ranks(done_total)
    rets, lens = g.sharding.gather_episode_stats(torch.arange(rank + 2, dtype=torch.float64),
                                                 torch.arange(rank + 2, dtype=torch.int64))
    if rank == 0:
        assert [len(x) for x in rets] == [k + 2 for k in range(world)]
        assert torch.equal(lens[1], torch.arange(3))
    else:
        assert rets is None
    # the optional gather of a collection's rollout buffers to the learner rank (off the step path)
    T = 4
    gidx = torch.arange(off, off + cnt, dtype=torch.float64)
    roll = {"obs": torch.as_tensor(np.stack(obs[1:T + 1])),                          # [T, cnt, D]
            "reward": gidx[None, :] + 1000.0 * torch.arange(T, dtype=torch.float64)[:, None],
            "done": (gidx[None, :].long() + torch.arange(T)[:, None]) % 3 == 0, "_scratch": object()}
    full = g.sharding.gather_rollout(roll, dst=1, total_envs=total)
    if rank == 1:
        assert sorted(full) == ["done", "obs", "reward"] and full["obs"].shape == (T, total, 5 + 3 * n_traffic)
        assert full["done"].dtype == torch.bool
        e = torch.arange(total, dtype=torch.float64)
        assert torch.equal(full["reward"], e[None, :] + 1000.0 * torch.arange(T, dtype=torch.float64)[:, None])
        assert torch.equal(full["done"], (e[None, :].long() + torch.arange(T)[:, None]) % 3 == 0)
        np.save(os.path.join(out_dir, "gathered_obs.npy"), full["obs"].numpy())
    else:
        assert full is None
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), obs=np.stack(obs), t_max=t_max, n_sum=n_sum,
             done=done_total, own_psi=env.own_psi, episode=env.episode)
    dist.destroy_process_group()


def test_two_rank_sharding_equals_unsharded(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    O.build()
    total, N, steps, world = 301, 8, 25, 2          # odd total: ranks get 151 + 150
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, total, N, steps, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    ref = O.OracleEnvs(total, N, seed=13, auto_reset=True)
    obs = [ref.reset().copy()]
    rng = np.random.default_rng(0)
    done = 0
    for _ in range(steps):
        o, _, _, _, n = ref.step(rng.uniform(-1, 1, total))
        obs.append(o.copy())
        done += n
    assert np.array_equal(np.concatenate([p["obs"] for p in parts], axis=1), np.stack(obs))
    # the gathered rollout on the learner rank is the unsharded rollout, envs in global order
    assert np.array_equal(np.load(tmp_path / "gathered_obs.npy"), np.stack(obs)[1:5])
    assert np.array_equal(np.concatenate([p["own_psi"] for p in parts]), ref.own_psi)
    assert np.array_equal(np.concatenate([p["episode"] for p in parts]), ref.episode)
    assert all(float(p["t_max"]) == 2.0 for p in parts)                  # MAX over ranks
    assert all(int(p["n_sum"]) == done for p in parts) and done > 0      # SUM over ranks
