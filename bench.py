#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched ACAS2D step engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torchrun environment, this process only starts the ranks (a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` of this
same file) and exits with their return code; launched under torchrun it is one of the ranks.

A "step" is one pass of the hot path over one batch: ONE VecEnv step() = one launch of the step
kernel over all envs of the rank (action -> integrate -> observe -> evaluate -> is_done ->
auto-reset), API-faithful: state, actions, observations, rewards and done flags all go through
HBM-resident buffers every step.  Workload at every N: BASELINE.json configs[2] per GPU
(65 536 envs x 8 traffic, float32), synthetic: episodes from the device reset distribution
(seed 13), actions ~ U(-1, 1) pre-generated on the device (a ring of <= 32 rows the steps cycle
through: ring_rows()), inputs resident before the timed region.  Multi-GPU = independent env shards (global env index = rank * envs + e), no collective
on the step path (weak scaling); the only communication is the timing barrier / MAX.

Timing.  The step launches are captured once into a hipGraph and the graph is replayed: after the
W warm-up steps it is replayed for >= 0.3 s (clocks up, the graph uploaded -- a first replay is
several times slower than a steady one), then the timed region runs the K steps R times back to
back between barrier + synchronize on both sides; R ("repeats") is 1 for K >= 2000 and
ceil(2000 / K) below that, so that one graph-launch latency (~15-40 us) is not smeared over a
handful of 5 us steps.  A graph holds up to --chunk (200) consecutive steps -- a divisor of K, or,
for a short K, several repetitions of the K steps (`steps_per_graph` in the line).  `ms_per_step`,
`value` and `roofline` are means over the K * R timed steps (`timed_steps`).

One JSON line on rank 0 with `roofline` (HBM bound; algorithmic bytes B(N,s) = s(16+9N)+9 per
env-step / average launch duration measured with HIP events on the launch stream) and
`cpu_baseline` (the CPU oracle = a scalar float64 C port of the reference step, timed on this
box's host cores on a bounded sample: the like-for-like single-env loop of BASELINE.json
configs[0] and the batch of the headline workload).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy)
METRIC = "env-steps/sec at 65536 envs x N_TRAFFIC=8; achieved HBM GB/s vs peak"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--traffic", type=int, default=8)
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32")
    ap.add_argument("--fast-math", action="store_true",
                    help="float64 only: the FAST formulation in float64 arithmetic (ACAS2DConfig.fast_math)")
    ap.add_argument("--launch", choices=("graph", "eager"), default="graph",
                    help="replay the step launches from captured hipGraphs (default) or launch eagerly")
    ap.add_argument("--chunk", type=int, default=200, help="largest number of steps per captured graph")
    ap.add_argument("--action-rows", type=int, default=0,
                    help="distinct pre-generated action rows the steps cycle through (0 = the largest divisor <= 32 of the "
                         "steps of one graph; see ring_rows)")
    ap.add_argument("--repeats", type=int, default=0, help="timed repetitions of the K steps (0 = auto: ceil(2000 / K))")
    ap.add_argument("--spin-seconds", type=float, default=0.3, help="graph replays before the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="rehearsal only: force this GPU index on every rank")
    ap.add_argument("--rehearsal", action="store_true",
                    help="no engine, no GPU: a stand-in step on the CPU, to exercise the launcher / rendezvous / "
                         "barrier / MAX-over-ranks / JSON path (tests); the line says so and carries no roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the secondary fused-rollout measurement")
    ap.add_argument("--no-extra", action="store_true", help="skip the other single-GPU BASELINE configs")
    ap.add_argument("--rollout-steps", type=int, default=200)
    ap.add_argument("--no-auto-reset", action="store_true", help="diagnostic: latch outcomes instead of resetting")
    ap.add_argument("--in-place", action="store_true", help="A/B: step the state in place instead of double-buffered")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-collisions", action="store_true",
                    help="diagnostic (SURVEY.md 8d, C4): collision distance 0, so episodes end at the goal / by "
                         "timeout only and resets are rare -- isolates the step itself at large N_TRAFFIC")
    ap.add_argument("--no-terminations", action="store_true",
                    help="diagnostic: no collisions, no goal, no timeout -- nothing ever finishes (the reset-free floor)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """--gpus N > 1 from a plain `python bench.py`: start the N ranks as a CHILD process (never exec: this
    process may not be replaced once anything touched the GPU, and nothing has yet -- only `import`s ran)
    and hand its return code back."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


# ---- CPU baselines (the oracle as the thing timed beside the GPU, never as the product) --------------
def cpu_baseline(envs, traffic, seconds):
    """The oracle (scalar C float64 port of the reference step) on bounded samples.
    `value`: the headline workload's batch (same env count and traffic), 1 thread, as many steps as fit
    in ~`seconds`.  `single_env`: BASELINE.json configs[0] / BASELINE.md section 3's like-for-like line --
    ONE env, N_TRAFFIC = 3, random actions, auto-reset, one core, stepped in sequence."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    # -- like-for-like: 1 env x 3 traffic, sequential step() calls on one core
    one = O.OracleEnvs(1, 3, seed=13, auto_reset=True)
    one.reset()
    one.single_env_loop(rng.uniform(-1, 1, 1000))
    n1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        one.single_env_loop(rng.uniform(-1, 1, 100000))
        n1 += 100000
    dt1 = time.perf_counter() - t0
    acts1 = rng.uniform(-1, 1, 20000)
    t0 = time.perf_counter()
    for a in acts1:
        one.step([a])
    dt1py = time.perf_counter() - t0
    single = {"value": n1 / dt1, "unit": "env-steps/s", "cores": 1, "kind": "port",
              "sample": "1 env x 3 traffic, %d sequential step() calls in C (oracle/acas2d_oracle.c "
                        "acas2d_oracle_single_env_loop), auto-reset, %.1f s" % (n1, dt1),
              "through_python_binding": {"value": len(acts1) / dt1py, "unit": "env-steps/s",
                                         "sample": "the same env, %d step() calls one ctypes call each" % len(acts1)},
              "reference_python_step": {"value": 5950.0, "unit": "env-steps/s",
                                        "note": "the real reference step(), clock throttle stubbed, timed in the build "
                                                "container only (it cannot travel): BASELINE.md section 2, Xeon 2.1 GHz, "
                                                "1 thread; as shipped it sleeps to <= 100 steps/s (environment.py:31)"}}
    # -- the headline batch on one thread
    env = O.OracleEnvs(envs, traffic, seed=13, auto_reset=True)
    env.reset()
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
                     (envs, traffic, n, os.cpu_count() or 0, dt),
           "single_env": single}
    # the same port spread over host cores (OpenMP over envs), a few seconds more
    # (16 = the CPU share of a one-GPU box; asking for all 256 logical CPUs there is slower than one)
    cores = O.set_threads(min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    try:
        if cores > 1:
            env.step(acts[0])
            t0, m = time.perf_counter(), 0
            while time.perf_counter() - t0 < min(3.0, seconds) and m < 2000:
                env.step(acts[m % 8])
                m += 1
            dt = time.perf_counter() - t0
            out["multi_core"] = {"value": envs * m / dt, "cores": cores, "steps": m}
    finally:
        O.set_threads(1)
    return out


def kernel_sources_sha256():
    """Digest of the files the device code is built from (gym-acas2d_amd/csrc: *.hpp, *.inl, *.hip, Makefile) -- stored
    in profiles/traffic.json by tools/summarize_profile.py, compared here: a counter pass taken on other kernel
    sources than the ones being timed shows as `kernel_sources_match: false` in the line."""
    import hashlib
    d = os.path.join(ROOT, "gym-acas2d_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        if n.endswith((".hpp", ".inl", ".hip")) or n == "Makefile":
            h.update(n.encode() + b"\0" + open(os.path.join(d, n), "rb").read())
    return h.hexdigest()


def load_traffic(envs, traffic, dtype, detail=False):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json, written by
    tools/summarize_profile.py from separate FETCH_SIZE / WRITE_SIZE passes of THIS command), if one exists for exactly
    this workload; else null.  `detail`: the whole record -- the read side under both corrections (the guide's x 2 for
    wide coalesced reads, and the factor fitted at 4 M envs where every input byte is fetched exactly once), the
    build the passes were taken on."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        for rec in json.load(open(p)):
            if (rec["envs"], rec["traffic"], rec["dtype"]) == (envs, traffic, dtype):
                if detail:
                    rec["kernel_sources_match"] = rec.get("kernel_sources_sha256") == kernel_sources_sha256()
                return rec if detail else rec["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        pass
    return None


def pick_chunk(k, cap):
    """Steps per captured graph: the largest divisor of K that is <= cap, so that the K timed steps are
    whole replays; a K with no useful divisor keeps `cap` and runs the remainder eagerly."""
    cap = max(1, min(cap, k))
    best = max(d for d in range(1, cap + 1) if k % d == 0)
    return best if best * 10 >= cap else cap


def ring_rows(chunk, asked=0, cap=32):
    """Distinct pre-generated action rows the steps cycle through: a divisor of the graph's step count.  Default: the
    largest one <= 32.  A policy writes its action tensor immediately before the step that reads it, so in use the
    row is cache-resident; 200 distinct rows (52 MB at the headline size, each re-read after 4.8 GB of other traffic)
    come from HBM instead and cost 0.1 us per launch -- reported beside the headline as its own line."""
    want = asked if asked > 0 else cap
    return max(d for d in range(1, min(want, chunk) + 1) if chunk % d == 0)


class StepRunner:
    """`chunk` consecutive step() launches (row t of a resident [chunk, E] action buffer each) captured
    into one hipGraph; run(n) = n // chunk replays + the remainder eagerly."""

    def __init__(self, env, actions, use_graph, graph_steps=None):
        import torch
        self.env, self.rows, self.chunk = env, [actions[t] for t in range(actions.shape[0])], actions.shape[0]
        self.graph = None
        self.graph_steps = graph_steps or self.chunk      # a multiple of the action rows: the rows repeat inside the graph
        assert self.graph_steps % self.chunk == 0
        if use_graph:
            if self.graph_steps % 2 and getattr(env, "double_buffer", False):
                env.set_double_buffer(False)      # a replayed graph must hold an even number of double-buffered steps
            self.eager(3)
            torch.cuda.synchronize()
            self.gen0 = getattr(env, "generation", 0)     # the state generation the captured launches start from
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.eager(self.graph_steps)

    def eager(self, n):
        for t in range(n):
            self.env.step_from(self.rows[t % self.chunk])

    def run(self, n):
        if self.graph is None:
            self.eager(n)
            return
        full, rest = divmod(n, self.graph_steps)
        if full and hasattr(self.env, "align_generation"):
            self.env.align_generation(self.gen0)          # no-op unless an odd number of eager steps ran in between
        for _ in range(full):
            self.graph.replay()
        self.eager(rest)

    def spin(self, seconds):
        """Replay for `seconds` of wall time (at least once): the graph is uploaded and the clocks are up
        before anything is timed."""
        import torch
        n, t0 = 0, time.perf_counter()
        while True:
            self.run(self.graph_steps)
            n += 1
            if n % 8 == 0 or self.graph is None:
                torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                break
        torch.cuda.synchronize()
        return n


def make_env(g, E, N, dtype, dev, rank, args, fast_math=False):
    env = g.ACAS2DVecEnv(E, N, device=dev, dtype=dtype, seed=13, env_offset=rank * E,
                         auto_reset=not args.no_auto_reset, config=g.ACAS2DConfig(n_traffic=N, fast_math=fast_math),
                         double_buffer=False if (args.in_place or args.no_auto_reset) else None)
    if args.no_collisions or args.no_terminations:
        env._ccfg.collision_dist = 0.0
    if args.no_terminations:
        env._ccfg.goal_radius = 0.0
        env._ccfg.max_steps = 2 ** 30
    env.reset()
    return env


def time_config(g, E, N, dtype_name, dev, args, steps=1000, chunk=100, rank=0, sync_ranks=None, action_rows=0, note=""):
    """One secondary configuration: per-launch time by HIP events over `steps` graph-replayed steps after a short
    spin.  Returns the numbers of the roofline line for that workload.  `sync_ranks` (multi-GPU): a barrier in front
    of the timed steps; the caller takes the MAX of launch_us over the ranks."""
    import torch
    dtype = torch.float32 if dtype_name == "f32" else torch.float64
    env = make_env(g, E, N, dtype, dev, rank, args, fast_math=dtype_name == "f64-fast")
    gen = torch.Generator(device=dev).manual_seed(1000)
    rows = ring_rows(chunk, action_rows)
    actions = torch.rand(rows, E, generator=gen, device=dev, dtype=dtype) * 2 - 1
    runner = StepRunner(env, actions, True, chunk)
    runner.spin(0.1)
    if sync_ranks is not None:
        torch.cuda.synchronize()
        sync_ranks()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    runner.run(steps)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / steps
    s = 4 if dtype_name == "f32" else 8
    b = E * g.ACAS2DConfig.algorithmic_bytes_per_env_step(N, s)
    geo = g.native.launch_geometry(E, N, s)
    out = {"workload": "%d envs x N_TRAFFIC=%d, %s%s" % (E, N, dtype_name, note), "launch_us": us, "action_rows": rows,
           "env_steps_per_s": E / (us * 1e-6), "algorithmic_bytes_per_launch": b,
           "achieved_GBps": b / (us * 1e-6) / 1e9, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "lanes_per_env": geo["lanes_per_env"], "traffic_per_lane": geo["traffic_per_lane"], "steps": steps}
    tr = load_traffic(E, N, dtype_name, detail=True)
    if tr is not None:
        out["traffic"] = tr["hbm_bytes_per_launch"]
        out["traffic_over_algorithmic"] = tr["hbm_bytes_per_launch"] / b
    del runner, env, actions
    return out


class RehearsalEnv:
    """--rehearsal: NOT the engine.  A stand-in with the one method the timing loop calls, on the CPU, so
    that the launcher / rendezvous / barrier / MAX / JSON path can be exercised where no GPU exists."""

    def __init__(self, E):
        import torch
        self.acc = torch.zeros(E)
        self.episode = torch.zeros(E, dtype=torch.int32)

    def step_from(self, row):
        self.acc.add_(row)


def main():
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world_env == 0 and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    import torch
    import gym_acas2d_amd as g
    if os.environ.get("ACAS2D_BENCH_LIB"):        # diagnostic builds (tools/): ablation / stamps
        g.native.LIB_PATH = os.path.join(ROOT, "gym-acas2d_amd", "csrc", os.environ["ACAS2D_BENCH_LIB"])
    rank, local_rank, world = g.sharding.dist_env()
    args.gpus = world
    fail_rank = os.environ.get("ACAS2D_BENCH_FAIL_RANK")        # tests only: this rank dies after the rendezvous
    E, N, K, W = args.envs, args.traffic, args.steps, args.warmup
    dtype = torch.float32 if args.dtype == "f32" else torch.float64

    if args.rehearsal:
        dev = torch.device("cpu")
        g.sharding.init_process_group("gloo")
        env = RehearsalEnv(E)
        sync = lambda: None  # noqa: E731
        if fail_rank is not None and int(fail_rank) == rank:
            sys.exit(3)
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
        dev_index = local_rank if args.device is None else args.device
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        g.sharding.init_process_group(args.backend)
        env = make_env(g, E, N, dtype, dev, rank, args, fast_math=args.fast_math and args.dtype == "f64")
        sync = torch.cuda.synchronize

    chunk = pick_chunk(K, args.chunk)
    chunk = ring_rows(chunk, args.action_rows)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    actions = torch.rand(chunk, E, generator=gen, device=dev, dtype=dtype) * 2 - 1
    use_graph = args.launch == "graph" and not args.rehearsal
    repeats = args.repeats if args.repeats > 0 else (1 if K >= 2000 else min(400, math.ceil(2000 / K)))
    # the timed region is `repeats` x K consecutive steps; a short K is captured several times over per graph (the
    # actions repeat every `chunk` steps either way), so that a small --steps does not time graph-launch gaps instead
    m = max(d for d in range(1, max(1, args.chunk // chunk) + 1) if (K * repeats) % (chunk * d) == 0)
    graph_steps = chunk * m
    runner = StepRunner(env, actions, use_graph, graph_steps)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # ---- warm-up: the W steps asked for, then the captured graphs replayed for spin-seconds ----
    runner.run(W)
    sync()
    spin_replays = runner.spin(args.spin_seconds) if not args.rehearsal else 0
    # ---- timed region: the K steps, R times, between barrier + synchronize on both sides ----
    sync()
    barrier()
    sync()
    if not args.rehearsal:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if not args.rehearsal:
        ev0.record()                   # same stream as the kernel launches (torch's current stream)
    runner.run(K * repeats)
    if not args.rehearsal:
        ev1.record()
    sync()
    barrier()
    sync()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) if not args.rehearsal else wall * 1e3
    red_dev = dev if not args.rehearsal and args.backend == "nccl" else "cpu"
    wall = g.sharding.max_over_ranks(wall, device=red_dev)
    dev_ms = g.sharding.max_over_ranks(dev_ms, device=red_dev)
    episodes = g.sharding.sum_over_ranks(float(env.episode.to(torch.int64).sum().item()), device=red_dev)
    timed_steps = K * repeats

    # ---- secondary: the same workload through acas2d_rollout_* (T steps fused per launch) ----
    fused = None
    if not args.no_rollout and not args.no_auto_reset and not args.rehearsal:
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
            ms = g.sharding.max_over_ranks(r0.elapsed_time(r1), device=red_dev)
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

    # ---- multi-GPU: BASELINE.json configs[3] as well -- 1 048 576 envs over 8 GPUs = 131 072 per GPU;
    #      every rank times its shard behind a barrier, the line carries the MAX launch time and the whole-job rate
    c3 = None
    if world > 1 and not args.rehearsal and not args.no_extra and (E, N, args.dtype) == (65536, 8, "f32"):
        try:
            mine = time_config(g, 131072, 8, "f32", dev, args, steps=1000, chunk=100, rank=rank, sync_ranks=barrier)
            us = g.sharding.max_over_ranks(mine["launch_us"], device=red_dev)
            b = mine["algorithmic_bytes_per_launch"]
            c3 = {"workload": "%d envs x N_TRAFFIC=8 over %d GPUs (131072 per GPU), f32" % (131072 * world, world),
                  "launch_us_max_over_ranks": us, "value": 131072 * world / (us * 1e-6), "unit": "env-steps/s",
                  "per_gpu_achieved_GBps": b / (us * 1e-6) / 1e9, "per_gpu_frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                  "steps": mine["steps"]}
        except Exception as e:  # noqa: BLE001
            c3 = {"error": str(e)}
    if rank == 0:
        s = 4 if args.dtype == "f32" else 8
        bytes_per_launch = E * g.ACAS2DConfig.algorithmic_bytes_per_env_step(N, s)
        launch_us = dev_ms * 1e3 / timed_steps
        achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
        diag = ("" if not args.no_collisions else ", collisions disabled (diagnostic)") + \
               ("" if not args.no_terminations else ", nothing terminates (diagnostic)")
        out = {
            "metric": METRIC,
            "value": E * world * timed_steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "repeats": repeats, "timed_steps": timed_steps,
            "ms_per_step": wall * 1e3 / timed_steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
        }
        if args.rehearsal:
            out["rehearsal"] = True
            out["value"] = None                    # nothing of the engine was measured
            out["data"] = "rehearsal: NO engine ran (stand-in CPU step); launcher / barrier / MAX-over-ranks path only"
            out["config"] = {"workload": "rehearsal, %d stand-in values per rank" % E, "envs_per_gpu": E,
                             "parallelism": "env-index shards x%d, no collective on the step path" % world,
                             "backend": "gloo", "chunk": chunk}
            out["roofline"] = None
        else:
            geo = g.native.launch_geometry(E, N, s)
            out["config"] = {"workload": "%d envs x N_TRAFFIC=%d per GPU, %s%s, one step-kernel launch per step(), "
                                         "auto-reset %s, random actions U(-1,1)%s"
                                         % (E, N, args.dtype, " (FAST formulation)" if args.fast_math and args.dtype == "f64" else "",
                                            "off" if args.no_auto_reset else "on", diag),
                             "envs_per_gpu": E, "n_traffic": N, "launch": args.launch, "steps_per_graph": graph_steps,
                             "action_rows": chunk,
                             "state_buffers": "double (read generation g, write 1 - g)" if getattr(env, "double_buffer", False)
                                              else "in place",
                             "state_layout": "consecutive rows (all loads through preloaded base pointers)"
                                             if getattr(env, "consecutive_layout", False) else "separate arrays",
                             "warmup_graph_replays": spin_replays,
                             "lanes_per_env": geo["lanes_per_env"], "traffic_per_lane": geo["traffic_per_lane"],
                             "grid_blocks": geo["grid_blocks"],
                             "parallelism": "env-index shards x%d, no collective on the step path" % world,
                             "episodes_finished": int(episodes)}
            tr = load_traffic(E, N, args.dtype, detail=True)
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                               "traffic_detail": tr,
                               "kernel": "acas2d::step_kernel<%s, C=%d, G=%d>" % ("float" if s == 4 else "double",
                                                                                  geo["traffic_per_lane"], geo["lanes_per_env"]),
                               "algorithmic_bytes_per_launch": bytes_per_launch,
                               "launch_us": launch_us}
        if fused is not None:
            out["fused_rollout"] = fused
        if world == 1 and not args.rehearsal and not args.no_extra:
            # the other single-GPU configurations of BASELINE.json (parity-test cases; reported, not `value`)
            extra = []
            # (the last one: 1.5 GB of working set >> the 256 MB Infinity Cache -- the true-HBM data point)
            for (e2, n2, d2, st2) in ((4096, 3, "f32", 1000), (65536, 64, "f32", 1000), (65536, 8, "f64", 1000),
                                      (65536, 8, "f64-fast", 1000), (131072, 8, "f32", 1000), (4194304, 8, "f32", 100)):
                if (e2, n2, d2) == (E, N, args.dtype):
                    continue
                try:
                    extra.append(time_config(g, e2, n2, d2, dev, args, steps=st2, chunk=min(100, st2)))
                except Exception as e:  # noqa: BLE001
                    extra.append({"workload": "%d x %d %s" % (e2, n2, d2), "error": str(e)})
            if (E, N, args.dtype) == (65536, 8, "f32"):
                try:        # the headline workload with as many DISTINCT action rows as a graph holds steps: read cold
                    extra.append(time_config(g, E, N, "f32", dev, args, steps=2000, chunk=200, action_rows=200,
                                             note=", 200 distinct action rows (each read cold from HBM)"))
                except Exception as e:  # noqa: BLE001
                    extra.append({"workload": "cold actions", "error": str(e)})
            out["other_configs"] = extra
        if c3 is not None:
            out["configs3_shard"] = c3
        if world == 1 and not args.no_cpu_baseline and not args.rehearsal:
            out["cpu_baseline"] = cpu_baseline(E, N, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
