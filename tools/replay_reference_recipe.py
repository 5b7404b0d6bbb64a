#!/usr/bin/env python3
"""The reference's own training recipe, replayed on the MI355X engine (gym_ACAS2D/training_main.py:28-52):

    PPO('MlpPolicy', env, seed=13).learn(1 048 576)  with  EvalCallback(eval_env, eval_freq=32 768, n_eval_episodes=10)

ONE env (n_envs = 1, N_TRAFFIC = 1), SB3 1.1.0's PPO defaults as recorded in the reference's model zips
(`PPOConfig.sb3()`: n_steps 2048, batch 64, 10 epochs, lr 3e-4, ...), a deterministic evaluation on 10 fresh episodes
every 32 768 steps -- 32 evaluations -- printed next to the reference's own curve
(models/best_model_1048576_11/results/evaluations.npz as the fixture tests/golden/ref_training_evaluations.npz; final
evaluation 1198.22 +/- 85.34, models/logs/training_ACAS2D_PPO_1048576_11.txt:13001-13002).

What can and cannot agree.  SB3 is absent from the reference tree and from this image, so its PPO is RESTATED here
(ppo.py; arithmetic pinned against autograd / Adam / a float64 restatement by tests/test_ppo.py) and *parity with
SB3's own stream of random numbers is unpinned*: initial weights, action noise and minibatch permutations come from
other generators, and the evaluation episodes are drawn from the reference's reset distribution (reset_parity.py)
by a stream of their own.  PPO on this task is seed-sensitive (the reference's own curve swings between 50 and 1200
from one evaluation to the next), so what is compared is the SHAPE of the learning curve and the level it reaches:
--seeds runs several seeds.  One JSON line per evaluation, one summary line per seed.

    python tools/replay_reference_recipe.py [--seeds 13 14 15] [--envs 1] [--out profiles/r03_recipe.jsonl]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gym_acas2d_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, nargs="+", default=[13])
ap.add_argument("--envs", type=int, default=1, help="the reference trains on ONE env (DummyVecEnv, n_envs = 1)")
ap.add_argument("--timesteps", type=int, default=1048576)         # settings.py:10-11  N_STEPS * 512
ap.add_argument("--eval-every", type=int, default=32768)          # settings.py:12     TOTAL_STEPS / 32
ap.add_argument("--eval-episodes", type=int, default=10)          # settings.py:8      EVAL_EPISODES
ap.add_argument("--collector", default="fused")
ap.add_argument("--updater", default="fused")
ap.add_argument("--out", default=None)
args = ap.parse_args()

ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_training_evaluations.npz"))
ref_at = {int(t): (float(r.mean()), float(r.std()), float(l.mean())) for t, r, l in zip(ref["timesteps"], ref["results"], ref["ep_lengths"])}
sink = open(args.out, "w") if args.out else None


def emit(rec):
    line = json.dumps(rec)
    print(line, flush=True)
    if sink:
        sink.write(line + "\n")
        sink.flush()


def evaluate(policy, rng, n):
    """EvalCallback: n deterministic episodes on fresh draws of the reference's reset distribution (game.py:80-116)."""
    own, trf, goal = g.reset_parity.draw_episodes(g.ACAS2DConfig(), n, rng)
    out = g.evaluate_policy_fused(policy, own, trf, goal, dtype=torch.float32)
    return out["total_reward"], out["steps"] - 1, out["outcome"]


emit({"recipe": "training_main.py:28-52", "config": "PPOConfig.sb3(): n_steps 2048, batch 64, 10 epochs, gamma 0.99, lambda 0.95, "
      "clip 0.2, lr 3e-4, ent 0, vf 0.5, max_grad_norm 0.5", "envs": args.envs, "timesteps": args.timesteps,
      "eval_every": args.eval_every, "eval_episodes": args.eval_episodes, "collector": args.collector, "updater": args.updater,
      "parity": "unpinned (SB3 1.1.0 is not in the reference tree nor in this image: its PPO is restated, its random streams are not)",
      "reference_final_eval": {"mean_reward": 1198.22, "std": 85.34, "mean_ep_length": 771.6,
                               "source": "models/logs/training_ACAS2D_PPO_1048576_11.txt:13001-13002"}})
for seed in args.seeds:
    venv = g.ACAS2DVecEnv(args.envs, 1, device="cuda:0", dtype=torch.float32, seed=seed)
    trainer = g.PPOTrainer(venv, g.PPOConfig.sb3(seed=seed), collector=args.collector, updater=args.updater)
    eval_rng = random.Random(1000 + seed)
    t0, next_eval, evals = time.time(), args.eval_every, []

    def log(rec):
        global next_eval
        while rec["timesteps"] >= next_eval and next_eval <= args.timesteps:
            r, l, oc = evaluate(trainer.policy, eval_rng, args.eval_episodes)
            rr = ref_at.get(next_eval)
            ev = {"seed": seed, "eval_at": next_eval, "mean_reward": float(r.mean()), "std_reward": float(r.std()),
                  "mean_ep_length": float(l.mean()), "goal": int((oc == 1).sum()), "collision": int((oc == 2).sum()),
                  "timeout": int((oc == 3).sum()), "train_ep_rew_mean": rec.get("ep_rew_mean"), "std": rec.get("std"),
                  "reference_mean_reward": rr[0] if rr else None, "reference_mean_ep_length": rr[2] if rr else None,
                  "wall_s": time.time() - t0}
            evals.append(ev)
            emit(ev)
            next_eval += args.eval_every

    trainer.learn(args.timesteps, log=log)
    last4 = evals[-4:]
    emit({"seed": seed, "summary": True, "wall_s": time.time() - t0, "steps_per_s": args.timesteps / (time.time() - t0),
          "final_eval_mean_reward": evals[-1]["mean_reward"], "final_eval_goals": evals[-1]["goal"],
          "mean_of_last_4_evals": float(np.mean([e["mean_reward"] for e in last4])),
          "best_eval": max(e["mean_reward"] for e in evals),
          "reference": {"final_eval": 1198.22, "mean_of_last_4_evals": float(ref["results"][-4:].mean()),
                        "best_eval": float(ref["results"].mean(1).max()), "wall_s": 14688.0, "steps_per_s": 71.0}})
