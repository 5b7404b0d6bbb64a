#!/usr/bin/env python3
"""Wave start / end times of one step launch (DIAGNOSTIC build, see diag_stamps.py): when does the dispatcher
start each wave, when does it end, and which waves end last?  Realtime stamps (100 MHz): 10 ns resolution."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gym_acas2d_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--traffic", type=int, default=8)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--no-terminations", action="store_true")
ap.add_argument("--no-collisions", action="store_true", help="fewer finished envs per step (goal arrivals and time-outs only)")
ap.add_argument("--warm-steps", type=int, default=300)
args = ap.parse_args()

g.native.LIB_PATH = os.path.join(ROOT, "gym-acas2d_amd", "csrc", os.environ.get("ACAS2D_DIAG_LIB", "libacas2d_hip_diag.so"))
g.native._lib = None
L = g.native.lib()
env = g.ACAS2DVecEnv(args.envs, args.traffic, device="cuda:0", dtype=torch.float32, seed=13)
if args.no_terminations:
    env._ccfg.collision_dist = 0.0; env._ccfg.goal_radius = 0.0; env._ccfg.max_steps = 2 ** 30
if args.no_collisions:
    env._ccfg.collision_dist = 0.0
geo = g.native.launch_geometry(args.envs, args.traffic, 4)
n_waves = geo["grid_blocks"] * 4
buf = torch.zeros(n_waves, 16, dtype=torch.int64, device="cuda:0")
L.acas2d_debug_set_stamps_f32.argtypes = [C.c_void_p]
assert L.acas2d_debug_set_stamps_f32(buf.data_ptr()) == 0
env.reset()
gen = torch.Generator(device="cuda:0").manual_seed(0)
acts = torch.rand(64, args.envs, generator=gen, device="cuda:0") * 2 - 1
for t in range(args.warm_steps):
    env.step_from(acts[t % 64])
S, E_, D = [], [], []
for t in range(args.steps):
    buf.zero_()
    for k in range(3):                     # back-to-back launches like the bench; the last one is read
        env.step_from(acts[(t + k) % 64])
    torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.float64)
    t0 = st[:, 0].min()
    S.append((st[:, 0] - t0) / 100.0); E_.append((st[:, 7] - t0) / 100.0)
    D.append(env.outputs["done"].cpu().numpy().reshape(-1, 64 // geo["lanes_per_env"]).any(1))
S, E_, D = np.array(S), np.array(E_), np.array(D)
q = [0, 10, 25, 50, 75, 90, 99, 100]
print("start  (us after the first wave's start), percentiles %s: %s" % (q, np.round(np.percentile(S, q, axis=1).mean(1), 2)))
print("end    percentiles: %s" % np.round(np.percentile(E_, q, axis=1).mean(1), 2))
print("length percentiles: %s" % np.round(np.percentile(E_ - S, q, axis=1).mean(1), 2))
# start time by dispatch order: logical wave w <- block remap; sort waves by start and show block index ranks
order = np.argsort(S.mean(0))
print("mean start of waves by logical index decile: %s" % np.round([S.mean(0)[k::10].mean() for k in range(10)], 2))
blk = np.arange(n_waves) // 4
print("mean start by logical-block decile (blocks 0..%d): %s" % (blk.max(), np.round([S.mean(0)[(blk * 10 // (blk.max() + 1)) == k].mean() for k in range(10)], 2)))
last = E_.argmax(1)
print("last wave to end: its start percentile %s, had a finished env %s" %
      (np.round([(S[i] < S[i, last[i]]).mean() for i in range(len(S))], 2)[:12], D[np.arange(len(S)), last][:12]))
fin = D
print("waves with a finished env per launch: %.1f; their end percentiles [50, 90, 100]: %s; lifetime percentiles: %s" %
      (fin.sum(1).mean(), np.round(np.percentile(E_[fin], [50, 90, 100]), 2) if fin.any() else None,
       np.round(np.percentile((E_ - S)[fin], [50, 90, 100]), 2) if fin.any() else None))
# waves that share a SIMD: a block's wave w runs on SIMD w of its CU, the CU holds two blocks -- which two is the
# dispatcher's choice, so pairs are not identified here; the count of blocks with >= 2 resetting waves is
blk_fin = fin.reshape(fin.shape[0], -1, 4)
print("blocks with a resetting wave per launch: %.1f, with two or more: %.1f" % (blk_fin.any(2).sum(1).mean(), (blk_fin.sum(2) >= 2).sum(1).mean()))
print("waves with a finished env: length median %.2f us, others %.2f us; end median %.2f vs %.2f" %
      (np.median((E_ - S)[fin]), np.median((E_ - S)[~fin]), np.median(E_[fin]), np.median(E_[~fin])))
late = S > np.percentile(S, 90, axis=1, keepdims=True)
print("the 10 %% last-started waves: length median %.2f us, end median %.2f; the 10 %% first-started: length %.2f, end %.2f" %
      (np.median((E_ - S)[late]), np.median(E_[late]), np.median((E_ - S)[S < np.percentile(S, 10, axis=1, keepdims=True)]),
       np.median(E_[S < np.percentile(S, 10, axis=1, keepdims=True)])))
