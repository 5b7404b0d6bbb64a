#!/usr/bin/env python3
"""Export the reference's committed trained policy as a data fixture.

Reads gym_ACAS2D/models/best_model_1048576_11/best_model.zip (an SB3 1.1.0 PPO MlpPolicy zip)
with a loader that executes nothing from the file (zipfile + torch.load(weights_only=True)) and
writes its 13 float32 tensors to tests/golden/ref_policy_best_model.npz.  Data only -- no code of
the reference or of SB3 is copied.  The expected evaluation aggregates that pin it are the ones
printed in notebooks/simulation_ACAS2D_PPO_1048576_11_100.ipynb (cell 4, `simulation.describe()`).
"""
import io
import os
import zipfile

import numpy as np
import torch

REF = os.environ.get("ACAS2D_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(REF, "gym_ACAS2D", "models", "best_model_1048576_11", "best_model.zip")
with zipfile.ZipFile(src) as z:
    sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    version = z.read("_stable_baselines3_version").decode()
out = os.path.join(ROOT, "tests", "golden", "ref_policy_best_model.npz")
np.savez_compressed(out, sb3_version=np.array(version), **{k: v.numpy() for k, v in sd.items()})
print("wrote", out, {k: tuple(v.shape) for k, v in sd.items()})
