#!/usr/bin/env python3
"""training_main.py equivalent on the MI355X engine: PPO on E parallel ACAS2D envs, everything on
the GPU.  Prints one JSON line per iteration; evaluates the final policy deterministically on the
reference's 100 test episodes (testing_main.py)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gym_acas2d_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1024)
ap.add_argument("--traffic", type=int, default=1)
ap.add_argument("--timesteps", type=float, default=6.0e7)
ap.add_argument("--n-steps", type=int, default=512)   # (512 x 4096: 100 / 100 goals on three seeds, ppo_seed_sweep.py)
ap.add_argument("--batch-size", type=int, default=4096)
ap.add_argument("--collector", choices=("graphs", "fused", "eager"), default="fused",
                help="fused: the whole collection of an iteration in one hand-written launch (ACAS2DVecEnv.collect)")
ap.add_argument("--updater", choices=("graphs", "fused"), default="fused",
                help="fused: every minibatch update as two hand-written launches (acas2d_ppo_update_f32)")
ap.add_argument("--seed", type=int, default=13)
ap.add_argument("--out", default=None)
args = ap.parse_args()

venv = g.ACAS2DVecEnv(args.envs, args.traffic, device="cuda:0", dtype=torch.float32, seed=13)
trainer = g.PPOTrainer(venv, g.PPOConfig(n_steps=args.n_steps, batch_size=args.batch_size, seed=args.seed), collector=args.collector,
                       use_graphs=args.collector != "eager", updater=args.updater if args.collector != "eager" else "graphs")
hist = trainer.learn(int(args.timesteps), log=lambda r: print(json.dumps(r), flush=True))

if args.traffic == 1:
    import helpers as H
    own, trf, goal = H.parity_reset_states(g.ACAS2DConfig(), 13, 0, 100)
    out = g.evaluate_policy_fused(trainer.policy, own, trf, goal)     # predict + step x 1001 in one launch
    print(json.dumps({"eval_100_reference_episodes": {"mean_return": float(out["total_reward"].mean()),
                                                      "mean_steps": float(out["steps"].mean()),
                                                      "goal": int((out["outcome"] == 1).sum()),
                                                      "collision": int((out["outcome"] == 2).sum()),
                                                      "timeout": int((out["outcome"] == 3).sum())},
                      "reference_trained_policy": {"mean_return": 1210.07, "mean_steps": 704.35, "goal": 100}}))
if args.out:
    torch.save(trainer.policy.state_dict(), args.out)
