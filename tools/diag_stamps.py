#!/usr/bin/env python3
"""Where does a step-kernel launch spend its time?  DIAGNOSTIC ONLY.

Loads libacas2d_hip_diag.so (make -C gym-acas2d_amd/csrc diag: same kernels + in-kernel
s_memtime / s_memrealtime stamps per wave), runs a few steps and prints, per phase, the
distribution over waves.  Never quote this build's run time: every stamp is an s_memtime round trip
(~500 cycles, measured with two back-to-back stamps), so sections that hold several stamps -- the
reset section has four -- read ~2000 cycles long before they do anything.  Read the SHARES, and use
product-build experiments (tools/README.md) for absolute costs.  Stamps per wave: 0 realtime@start,
1 clk@start, 2 clk after all loads landed, 3 clk after observe/evaluate (reward/done/outcome stores
issued), 4 clk after [packed shapes: the reset of finished envs, then] the state stores and the tile
flush are issued, 5 clk after the generic walk's late reset (packed shapes: == 4), 6 clk after all
stores are acknowledged, 7 realtime@end; 8..11 inside the reset of a finished env.
"""
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
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--no-auto-reset", action="store_true")
args = ap.parse_args()

g.native.LIB_PATH = os.path.join(ROOT, "gym-acas2d_amd", "csrc", os.environ.get("ACAS2D_DIAG_LIB", "libacas2d_hip_diag.so"))
g.native._lib = None
L = g.native.lib()
env = g.ACAS2DVecEnv(args.envs, args.traffic, device="cuda:0", dtype=torch.float32, seed=13,
                     auto_reset=not args.no_auto_reset)
geo = g.native.launch_geometry(args.envs, args.traffic, 4)
n_waves = geo["grid_blocks"] * 4
buf = torch.zeros(n_waves, 16, dtype=torch.int64, device="cuda:0")
L.acas2d_debug_set_stamps_f32.argtypes = [C.c_void_p]
assert L.acas2d_debug_set_stamps_f32(buf.data_ptr()) == 0
env.reset()
gen = torch.Generator(device="cuda:0").manual_seed(0)
for _ in range(200):                       # get past the start-up transient / into steady resets
    env.step(torch.rand(args.envs, generator=gen, device="cuda:0") * 2 - 1)
rows = []
inner = []
for _ in range(args.steps):
    buf.zero_()
    _, _, done, _ = env.step(torch.rand(args.envs, generator=gen, device="cuda:0") * 2 - 1)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.float64)
    st = st[st[:, 1] > 0]
    epw = 64 // geo["lanes_per_env"]
    dn = done.cpu().numpy()
    pad = (-len(dn)) % epw
    wave_done = np.pad(dn, (0, pad)).reshape(-1, epw).any(1)
    # waves are remapped across XCDs; map stamp rows back: stamp index == logical wave index
    wave_done = wave_done[:len(st)]
    t0 = st[:, 0].min()
    span_us = (st[:, 7].max() - t0) / 100.0           # s_memrealtime ticks at 100 MHz
    clk = np.median((st[:, 6] - st[:, 1]) / np.maximum((st[:, 7] - st[:, 0]) / 100.0, 1e-9)) / 1e3   # GHz
    ph = {"start skew us": (st[:, 0] - t0) / 100.0,
          "loads": st[:, 2] - st[:, 1], "compute": st[:, 3] - st[:, 2], "reset+stores+flush": st[:, 4] - st[:, 3],
          "late reset": st[:, 5] - st[:, 4], "store drain": st[:, 6] - st[:, 5],
          "wave total": st[:, 6] - st[:, 1]}
    rs = st[st[:, 8] > 0]
    if len(rs):
        inner.append([np.median(rs[:, 8] - rs[:, 3]), np.median(rs[:, 9] - rs[:, 8]), np.median(rs[:, 10] - rs[:, 9]),
                      np.median(rs[:, 11] - rs[:, 10]), np.median(rs[:, 4] - rs[:, 11])])
    rows.append((span_us, clk, {k: (np.median(v), np.percentile(v, 99), v.max()) for k, v in ph.items()},
                 wave_done.mean(), (np.median((st[:, 5] - st[:, 3])[wave_done]) - np.median((st[:, 5] - st[:, 3])[~wave_done]))
                 if wave_done.any() and (~wave_done).any() else 0,
                 (st[:, 7] - t0).argmax(), wave_done[(st[:, 7] - t0).argmax()]))
print("config: %d envs x %d traffic, shape %s, %d waves" % (args.envs, args.traffic, geo, n_waves))
span = np.array([r[0] for r in rows])
print("kernel span (first wave start -> last wave end), us: median %.2f  min %.2f  max %.2f" %
      (np.median(span), span.min(), span.max()))
print("shader clock while running: %.2f GHz" % np.median([r[1] for r in rows]))
print("phase (cycles unless noted): median over waves | p99 | max   [medians over %d launches]" % len(rows))
for k in rows[0][2]:
    a = np.array([r[2][k] for r in rows])
    print("  %-14s %10.1f | %10.1f | %10.1f" % (k, *np.median(a, axis=0)))
print("waves with a finished env: %.1f %%; they spend %.0f cycles more than the others between compute and the last store issue (median)" %
      (100 * np.mean([r[3] for r in rows]), np.median([r[4] for r in rows])))
print("last-finishing wave had a finished env in %d of %d launches" % (sum(bool(r[6]) for r in rows), len(rows)))
if inner:
    m = np.median(np.array(inner), axis=0)
    print("waves with a finished env (last reset in the wave), cycles incl. ~500 per stamp: entry %.0f | term_obs + Philox + "
          "hand-off %.0f | own_context %.0f | traffic obs %.0f | pick-up + state stores + flush %.0f" % tuple(m))
