"""Stable-Baselines3 `MlpPolicy` actor as a plain torch module + a `testing_main.py`-style
evaluation loop.

The reference ships trained PPO policies as SB3 1.1.0 zips
(gym_ACAS2D/models/best_model_1048576_11/best_model.zip, models/checkpoints_*/model_*_steps.zip)
and evaluates them with `model.predict(state, deterministic=True)` (testing_main.py:74).  SB3 is
not required here: the zip's `policy.pth` is a state dict of 13 float32 tensors, read with
`torch.load(weights_only=True)` (nothing in the file is executed).

`predict(obs, deterministic=True)` follows SB3 1.1.0's `BasePolicy.predict` for a Box action
space without squashing: observation -> float32, two tanh layers of 64, linear action head, the
mean action clipped to [-1, 1].
"""
import io
import zipfile

import numpy as np
import torch

_ACTOR_KEYS = ("mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias",
               "mlp_extractor.policy_net.2.weight", "mlp_extractor.policy_net.2.bias",
               "action_net.weight", "action_net.bias")


class SB3ActorPolicy(torch.nn.Module):
    def __init__(self, state_dict):
        super().__init__()
        w1, b1, w2, b2, wa, ba = (torch.as_tensor(np.asarray(state_dict[k]), dtype=torch.float32)
                                  for k in _ACTOR_KEYS)
        self.l1 = torch.nn.Linear(w1.shape[1], w1.shape[0])
        self.l2 = torch.nn.Linear(w2.shape[1], w2.shape[0])
        self.head = torch.nn.Linear(wa.shape[1], wa.shape[0])
        with torch.no_grad():
            for lin, w, b in ((self.l1, w1, b1), (self.l2, w2, b2), (self.head, wa, ba)):
                lin.weight.copy_(w)
                lin.bias.copy_(b)
        self.obs_dim = w1.shape[1]

    @torch.no_grad()
    def forward(self, obs):
        x = obs.to(torch.float32)
        return self.head(torch.tanh(self.l2(torch.tanh(self.l1(x)))))

    def actor_weights(self):
        """(w1 [64,D], b1, w2 [64,64], b2, w3 [1,64], b3) for ACAS2DVecEnv.rollout_policy()."""
        return tuple(t.detach() for t in (self.l1.weight, self.l1.bias, self.l2.weight, self.l2.bias,
                                          self.head.weight, self.head.bias))

    @torch.no_grad()
    def predict(self, obs, deterministic=True):
        """[E, obs_dim] -> [E, 1] float32 actions in [-1, 1] (deterministic = the mean action)."""
        if not deterministic:
            raise NotImplementedError("only the deterministic evaluation path of testing_main.py:74")
        return self.forward(obs).clamp_(-1.0, 1.0)


def load_sb3_policy(path, device="cpu"):
    """Load the actor of an SB3 PPO zip (or of an .npz export of its policy.pth)."""
    if str(path).endswith(".npz"):
        sd = dict(np.load(path, allow_pickle=False))
    else:
        with zipfile.ZipFile(path) as z:
            sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    return SB3ActorPolicy(sd).to(device)


def evaluate_policy(venv, policy, max_steps=None):
    """testing_main.simulate() (testing_main.py:62-105) on a batch: every env of `venv` (which must
    NOT auto-reset and must already hold its episode, e.g. via set_state) is stepped with the
    deterministic policy until done.  Returns numpy arrays outcome, steps (game.steps at done),
    total_reward, path_length (2 px per step() call at the default airspeed; game.d_path)."""
    assert not venv.auto_reset, "evaluate_policy wants auto_reset=False (one episode per env)"
    E = venv.num_envs
    max_steps = max_steps or venv.config.max_steps
    obs = venv.outputs["obs"]
    outcome = torch.zeros(E, dtype=torch.uint8, device=venv.device)
    steps = torch.zeros(E, dtype=torch.int32, device=venv.device)
    ret = torch.zeros(E, dtype=venv.dtype, device=venv.device)
    active = torch.ones(E, dtype=torch.bool, device=venv.device)
    for _ in range(max_steps):
        obs, _, done, infos = venv.step(policy.predict(obs))
        fin = active & done
        outcome = torch.where(fin, infos.outcome, outcome)
        steps = torch.where(fin, venv.steps, steps)
        ret = torch.where(fin, venv.total_reward, ret)
        active &= ~done
        if not bool(active.any()):
            break
    steps_np = steps.cpu().numpy()
    step_len = venv.config.airspeed * venv.config.dt
    return {"outcome": outcome.cpu().numpy(), "steps": steps_np,
            "total_reward": ret.cpu().numpy().astype(np.float64),
            "path_length": step_len * (steps_np - 1), "unfinished": int(active.sum().item())}


def evaluate_policy_fused(policy, own, traffic, goal=None, dtype=torch.float64, device="cuda:0", config=None,
                          max_steps=None):
    """evaluate_policy() as ONE kernel launch (ACAS2DVecEnv.rollout_policy): the episodes given by
    `own` [E,4] / `traffic` [E,N,4] / `goal` are run with the deterministic SB3 actor evaluated inside
    the rollout kernel; results are those of each env's FIRST episode (the launch has VecEnv
    auto-reset semantics and keeps stepping the envs that finish early).  Thread-per-env shapes only
    (N in {1,2,3,4,8} float32, {1,2,3} float64)."""
    from .vec_env import ACAS2DVecEnv
    own, traffic = np.asarray(own), np.asarray(traffic)
    E, N = own.shape[0], traffic.shape[1]
    v = ACAS2DVecEnv(E, N, device=device, dtype=dtype, auto_reset=True, config=config)
    v.set_state(own, traffic, goal, np.zeros(E, np.int32), observe=True)
    T = (max_steps or v.config.max_steps) + 1
    out = v.rollout_policy(policy, T)
    done = out["done"].cpu().numpy()
    fin = done.any(0)
    t0, e = done.argmax(0), np.arange(E)
    steps = np.where(fin, out["episode_steps"].cpu().numpy()[t0, e], 0)
    step_len = v.config.airspeed * v.config.dt
    return {"outcome": np.where(fin, out["outcome"].cpu().numpy()[t0, e], 0).astype(np.uint8), "steps": steps,
            "total_reward": np.where(fin, out["episode_return"].cpu().numpy()[t0, e], 0.0).astype(np.float64),
            "path_length": step_len * (steps - 1), "unfinished": int((~fin).sum())}
