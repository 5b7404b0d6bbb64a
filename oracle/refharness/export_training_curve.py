#!/usr/bin/env python3
"""The reference's own training evaluation curve as a data fixture: gym_ACAS2D/models/best_model_1048576_11/results/
evaluations.npz (written by SB3's EvalCallback during the reference's one committed training run: 32 evaluations x 10
deterministic episodes, training_main.py:31-35) -> tests/golden/ref_training_evaluations.npz.  Pure numbers (timesteps,
episode returns, episode lengths), loaded with allow_pickle=False; nothing of the reference's code is involved.
TEST INFRASTRUCTURE: tools/replay_reference_recipe.py prints them next to its own curve.

Usage: python oracle/refharness/export_training_curve.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("ACAS2D_REFERENCE", "/root/reference")
src = os.path.join(REF, "gym_ACAS2D", "models", "best_model_1048576_11", "results", "evaluations.npz")
d = np.load(src, allow_pickle=False)
out = os.path.join(ROOT, "tests", "golden", "ref_training_evaluations.npz")
np.savez_compressed(out, timesteps=d["timesteps"].astype(np.int64), results=d["results"].astype(np.float64),
                    ep_lengths=d["ep_lengths"].astype(np.int64))
print("wrote", out, d["results"].shape, "final eval %.2f +/- %.2f" % (d["results"][-1].mean(), d["results"][-1].std()))
