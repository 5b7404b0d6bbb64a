#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched ACAS2D step engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ONE VecEnv step() = one launch of the step
kernel over all envs of the rank (action -> integrate -> observe -> evaluate -> is_done ->
auto-reset), API-faithful: state, actions, observations, rewards and done flags all go through
HBM-resident buffers every step.  Workload at every N: BASELINE.json configs[2] per GPU
(65 536 envs x 8 traffic, float32), synthetic: episodes from the device reset distribution
(seed 13), actions ~ U(-1, 1) pre-generated on the device, inputs resident before the timed
region.  Multi-GPU = independent env shards (global env index = rank * envs + e), no collective
on the step path (weak scaling); the only communication is the timing barrier / MAX.

One JSON line on rank 0 with `roofline` (HBM bound; algorithmic bytes B(N,s) = s(16+9N)+9 per
env-step / average launch duration measured with HIP events on the launch stream) and
`cpu_baseline` (the CPU oracle = a scalar float64 port of the reference step, timed on this
box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--traffic", type=int, default=8)
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32")
    ap.add_argument("--launch", choices=("graph", "eager"), default="graph",
                    help="replay the step launches from a captured hipGraph (default) or launch eagerly")
    ap.add_argument("--chunk", type=int, default=100, help="steps per captured graph / action rows")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="rehearsal only: force this GPU index on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the secondary fused-rollout measurement")
    ap.add_argument("--rollout-steps", type=int, default=200)
    ap.add_argument("--no-auto-reset", action="store_true", help="diagnostic: latch outcomes instead of resetting")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-collisions", action="store_true",
                    help="diagnostic (SURVEY.md 8d, C4): collision distance 0, so episodes end at the goal / by "
                         "timeout only and resets are rare -- isolates the step itself at large N_TRAFFIC")
    return ap.parse_args()


def cpu_baseline(envs, traffic, seconds):
    """The oracle (scalar C float64 port of the reference step, 1 thread) on a bounded sample of
    the same workload: the same env count and traffic, as many steps as fit in ~`seconds`."""
    from oracle import oracle as O
    env = O.OracleEnvs(envs, traffic, seed=13, auto_reset=True)
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (8, envs))
    env.step(acts[0])
    t0 = time.perf_counter()
    n = 0
    while True:
        env.step(acts[n % 8])
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 400:
            break
    dt = time.perf_counter() - t0
    out = {"value": envs * n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": "%d envs x %d traffic x %d steps, float64 scalar C port of the reference step "
                     "(oracle/acas2d_oracle.c), 1 thread of %d host cpus, %.1f s" %
                     (envs, traffic, n, os.cpu_count() or 0, dt)}
    # the same port spread over host cores (OpenMP over envs), a few seconds more
    # (16 = the CPU share of a one-GPU box; asking for all 256 logical CPUs there is slower than one)
    cores = O.set_threads(min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    try:
        if cores > 1:
            env.step(acts[0])
            t0, m = time.perf_counter(), 0
            while time.perf_counter() - t0 < min(4.0, seconds) and m < 2000:
                env.step(acts[m % 8])
                m += 1
            dt = time.perf_counter() - t0
            out["multi_core"] = {"value": envs * m / dt, "cores": cores, "steps": m}
    finally:
        O.set_threads(1)
    return out


def load_traffic(envs, traffic, dtype):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), if one exists
    for exactly this workload; else null."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        for rec in json.load(open(p)):
            if (rec["envs"], rec["traffic"], rec["dtype"]) == (envs, traffic, dtype):
                return rec["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        pass
    return None


def main():
    args = parse()
    import gym_acas2d_amd as g
    if os.environ.get("ACAS2D_BENCH_LIB"):        # diagnostic builds (tools/): ablation / stamps
        g.native.LIB_PATH = os.path.join(ROOT, "gym-acas2d_amd", "csrc", os.environ["ACAS2D_BENCH_LIB"])
    rank, local_rank, world = g.sharding.dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    dev_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    g.sharding.init_process_group(args.backend)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    E, N, K, W = args.envs, args.traffic, args.steps, args.warmup

    env = g.ACAS2DVecEnv(E, N, device=dev, dtype=dtype, seed=13, env_offset=rank * E,
                         auto_reset=not args.no_auto_reset)
    if args.no_collisions:
        env._ccfg.collision_dist = 0.0
    env.reset()
    chunk = max(1, min(args.chunk, K))
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    actions = torch.rand(chunk, E, generator=gen, device=dev, dtype=dtype) * 2 - 1
    rows = [actions[t] for t in range(chunk)]

    def run_eager(n, start=0):
        for t in range(n):
            env.step_from(rows[(start + t) % chunk])

    graph = None
    if args.launch == "graph":
        run_eager(3)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run_eager(chunk)

    def run(n):
        if graph is None:
            run_eager(n)
            return
        full, rest = divmod(n, chunk)
        for _ in range(full):
            graph.replay()
        run_eager(rest)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    run(W)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                       # same stream as the kernel launches (torch's current stream)
    run(K)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    wall = g.sharding.max_over_ranks(wall, device=dev)
    dev_ms = g.sharding.max_over_ranks(dev_ms, device=dev)
    episodes = g.sharding.sum_over_ranks(float(env.episode.to(torch.int64).sum().item()), device=dev)

    # ---- secondary: the same workload through acas2d_rollout_* (T steps fused per launch) ----
    fused = None
    if not args.no_rollout and not args.no_auto_reset:
        try:
            T = args.rollout_steps
            act = torch.rand(T, E, generator=gen, device=dev, dtype=dtype) * 2 - 1
            out = env.rollout(act)
            torch.cuda.synchronize()
            barrier()
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            r0.record()
            for _ in range(reps):
                out = env.rollout(act, out=out)
            r1.record()
            torch.cuda.synchronize()
            ms = g.sharding.max_over_ranks(r0.elapsed_time(r1), device=dev)
            sz = 4 if args.dtype == "f32" else 8
            per_step = sz * (7 + 3 * N) + 2                                   # action, obs, reward, done, outcome
            state = sz * (6 + 4 * N) + 12 + sz + sz * (3 + 2 * N) + 4 + sz    # state read + written once per launch
            b = E * (per_step * T + state)
            fused = {"value": E * world * T * reps / (ms * 1e-3), "unit": "env-steps/s", "steps_per_launch": T,
                     "launch_ms": ms / reps, "algorithmic_bytes_per_env_step": per_step + state / T,
                     "achieved_GBps": b * reps / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": b * reps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "acas2d_rollout: state in registers across steps, actions[t] in, obs/reward/done[t] out; "
                             "bit-identical to per-step launches"}
            del out, act
        except Exception as e:  # noqa: BLE001
            fused = {"error": str(e)}

    if rank == 0:
        s = 4 if args.dtype == "f32" else 8
        bytes_per_launch = E * g.ACAS2DConfig.algorithmic_bytes_per_env_step(N, s)
        launch_us = dev_ms * 1e3 / K
        achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
        geo = g.native.launch_geometry(E, N, s)
        out = {
            "metric": "env-steps/sec at 65536 envs x N_TRAFFIC=8; achieved HBM GB/s vs peak",
            "value": E * world * K / wall,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "%d envs x N_TRAFFIC=%d per GPU, %s, one step-kernel launch per step(), "
                                   "auto-reset %s, random actions U(-1,1)%s"
                                   % (E, N, args.dtype, "off" if args.no_auto_reset else "on",
                                      ", collisions disabled (diagnostic)" if args.no_collisions else ""),
                       "envs_per_gpu": E, "n_traffic": N, "launch": args.launch,
                       "lanes_per_env": geo["lanes_per_env"], "traffic_per_lane": geo["traffic_per_lane"],
                       "grid_blocks": geo["grid_blocks"],
                       "parallelism": "env-index shards x%d, no collective on the step path" % world,
                       "episodes_finished": int(episodes)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(E, N, args.dtype),
                         "kernel": "acas2d::step_kernel<%s, C=%d, G=%d>" % ("float" if s == 4 else "double",
                                                                            geo["traffic_per_lane"], geo["lanes_per_env"]),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launch_us": launch_us},
        }
        if fused is not None:
            out["fused_rollout"] = fused
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(E, N, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
