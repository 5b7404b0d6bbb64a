#!/usr/bin/env python3
"""Seed robustness of the large-batch PPO defaults (tools/train_ppo.py: 1 024 envs x 256 steps, fused collector +
fused update): for each named hyper-parameter set and each seed, train for --timesteps and evaluate deterministically
on the reference's 100 test episodes (testing_main.py; the reference's own policy scores 100 / 100 goals, mean return
1210.07).  One JSON line per run, one summary line per set.

    python tools/ppo_seed_sweep.py --sets default lr1e-4 --seeds 13 14 15 --timesteps 3e7
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gym_acas2d_amd as g  # noqa: E402
import helpers as H  # noqa: E402

SETS = {
    "default": dict(),                                              # train_ppo.py: n_steps 256, batch 4096, lr 3e-4
    "lr1e-4": dict(learning_rate=1e-4),
    "lr1.5e-4": dict(learning_rate=1.5e-4),
    "batch16k": dict(batch_size=16384),
    "batch1k": dict(batch_size=1024),
    "steps512": dict(n_steps=512),
    "steps1024": dict(n_steps=1024, batch_size=8192),
    "ent1e-3": dict(ent_coef=1e-3),
    "epochs5": dict(n_epochs=5),
    "clip0.1": dict(clip_range=0.1),
    "lr1e-4_steps512": dict(learning_rate=1e-4, n_steps=512),
    "gamma0.995": dict(gamma=0.995),
}
ap = argparse.ArgumentParser()
ap.add_argument("--sets", nargs="+", default=["default"])
ap.add_argument("--seeds", type=int, nargs="+", default=[13, 14, 15])
ap.add_argument("--envs", type=int, default=1024)
ap.add_argument("--timesteps", type=float, default=3.0e7)
ap.add_argument("--out", default=None)
args = ap.parse_args()
sink = open(args.out, "a") if args.out else None


def emit(rec):
    line = json.dumps(rec)
    print(line, flush=True)
    if sink:
        sink.write(line + "\n")
        sink.flush()


own, trf, goal = H.parity_reset_states(g.ACAS2DConfig(), 13, 0, 100)
for name in args.sets:
    kw = {**dict(n_steps=256, batch_size=4096), **SETS[name]}
    goals = []
    for seed in args.seeds:
        t0 = time.time()
        venv = g.ACAS2DVecEnv(args.envs, 1, device="cuda:0", dtype=torch.float32, seed=13)
        tr = g.PPOTrainer(venv, g.PPOConfig(seed=seed, **kw), collector="fused", updater="fused")
        hist = tr.learn(int(args.timesteps), log=None)
        out = g.evaluate_policy_fused(tr.policy, own, trf, goal)
        rec = {"set": name, "config": kw, "seed": seed, "timesteps": int(args.timesteps), "wall_s": time.time() - t0,
               "train_ep_rew_mean_last": hist[-1].get("ep_rew_mean"), "std": hist[-1].get("std"),
               "eval_mean_return": float(out["total_reward"].mean()), "eval_mean_steps": float(out["steps"].mean()),
               "goal": int((out["outcome"] == 1).sum()), "collision": int((out["outcome"] == 2).sum()),
               "timeout": int((out["outcome"] == 3).sum())}
        goals.append(rec["goal"])
        emit(rec)
        del tr, venv
    emit({"set": name, "summary": True, "goals": goals, "min_goals": int(np.min(goals)), "mean_goals": float(np.mean(goals))})
