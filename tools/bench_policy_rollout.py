#!/usr/bin/env python3
"""Fused policy rollout (acas2d_rollout_policy_*) beside torch policy.predict() + step():
(1) the reference's recorded policy evaluation from the fused float64 launch, (2) env-steps/s of both
ways at E envs x N_TRAFFIC=1, float32.  usage: bench_policy_rollout.py [E]"""
import os, sys, time, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gym_acas2d_amd as g, helpers as H
if os.environ.get("ACAS2D_BENCH_LIB"):            # diagnostic builds
    g.native.LIB_PATH = os.path.join(ROOT, "gym-acas2d_amd", "csrc", os.environ["ACAS2D_BENCH_LIB"])
dev = "cuda:0"
pol = g.load_sb3_policy(os.path.join(H.GOLDEN, "ref_policy_best_model.npz"), device=dev)
res = {}
# 1. tightness of the reference table with the fused f64 evaluation
own, trf, goal = H.parity_reset_states(g.ACAS2DConfig(), 13, 0, 100)
v = g.ACAS2DVecEnv(100, 1, device=dev, dtype=torch.float64, auto_reset=True)
v.set_state(own, trf, goal, np.zeros(100, np.int32), observe=True)
out = v.rollout_policy(pol, 1001)
done = out["done"].cpu().numpy(); t0 = done.argmax(0); e = np.arange(100)
ret = out["episode_return"].cpu().numpy()[t0, e]; st = out["episode_steps"].cpu().numpy()[t0, e]
d = H.describe(ret); res["f64_fused_table"] = {k: round(float(x), 6) for k, x in d.items()}; res["f64_steps_mean"] = float(st.mean())
# 2. throughput: fused policy rollout vs torch policy + step per step, E envs x N=1, f32
E, T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 200
a = g.ACAS2DVecEnv(E, 1, device=dev, dtype=torch.float32, seed=13); a.reset()
o = a.rollout_policy(pol, T)
torch.cuda.synchronize()
s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): o = a.rollout_policy(pol, T, out=o)
f.record(); torch.cuda.synchronize()
ms = s.elapsed_time(f) / 5
res["fused_policy_rollout"] = {"envs": E, "steps_per_launch": T, "launch_ms": ms, "env_steps_per_s": E * T / ms * 1e3}
b = g.ACAS2DVecEnv(E, 1, device=dev, dtype=torch.float32, seed=13); obs = b.reset()
for _ in range(20): obs, _, _, _ = b.step(pol.predict(obs))
torch.cuda.synchronize(); t = time.time()
for _ in range(200): obs, _, _, _ = b.step(pol.predict(obs))
torch.cuda.synchronize(); dt = time.time() - t
res["torch_policy_plus_step"] = {"envs": E, "ms_per_step": dt / 200 * 1e3, "env_steps_per_s": E * 200 / dt}
print(json.dumps(res, indent=1))
