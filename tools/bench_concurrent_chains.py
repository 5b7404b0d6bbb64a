#!/usr/bin/env python3
"""K independent VecEnvs of E/K envs each, every one replaying its own hipGraph of step launches on
its own stream -- the "several actors per GPU" deployment.  The chains are independent (no join per
step), so one chain's load / store phases overlap another's arithmetic.  DIAGNOSTIC: the headline
number of bench.py is ONE VecEnv, one launch per step().  usage: bench_concurrent_chains.py [E] [N]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gym_acas2d_amd as g  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
CHUNK, REPS = 100, 20
res = []
for K in (1, 2, 4, 8):
    envs, graphs, streams = [], [], []
    for k in range(K):
        v = g.ACAS2DVecEnv(E // K, N, device=dev, dtype=torch.float32, seed=13, env_offset=k * (E // K))
        v.reset()
        gen = torch.Generator(device=dev).manual_seed(1000 + k)
        acts = torch.rand(CHUNK, E // K, generator=gen, device=dev) * 2 - 1
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for t in range(3):
                v.step_from(acts[t])
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for t in range(CHUNK):
                v.step_from(acts[t])
        envs.append((v, acts)); graphs.append(gr); streams.append(s)
    def run(n):
        for _ in range(n):
            for gr, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    gr.replay()
    run(2); torch.cuda.synchronize()
    t0 = time.time(); run(REPS); torch.cuda.synchronize(); dt = time.time() - t0
    res.append({"chains": K, "envs_per_chain": E // K, "env_steps_per_s": E * CHUNK * REPS / dt,
                "us_per_step_of_all_envs": dt / (CHUNK * REPS) * 1e6})
    print(json.dumps(res[-1]), flush=True)
