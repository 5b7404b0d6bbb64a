"""PPO on the device-resident ACAS2DVecEnv.

The reference trains with Stable-Baselines3 (`training_main.py:44-52`:
`PPO('MlpPolicy', env, seed=13).learn(1_048_576)`), one env, CPU.  SB3 is not available here and
its one-env-at-a-time loop is exactly what the batched engine replaces, so this module restates
the algorithm SB3 1.1.0 runs with the hyper-parameters recorded in the reference's model zips
(n_steps 2048, batch 64, epochs 10, gamma 0.99, lambda 0.95, clip 0.2, lr 3e-4, ent 0, vf 0.5,
max-grad-norm 0.5, Adam eps 1e-5, orthogonal init, state-independent log-std, advantages
normalised per minibatch) on E parallel envs: rollouts, GAE and the updates all stay on the GPU.

The network uses SB3's `MlpPolicy` parameter names (separate 2x64 tanh actor / critic), so the
reference's trained zips load as initial weights (`ActorCritic.load_sb3_state_dict`) and a policy
trained here can be evaluated with `policy.SB3ActorPolicy` / `evaluate_policy`.
"""
import dataclasses
import math
import time

import numpy as np
import torch
from torch import nn


@dataclasses.dataclass
class PPOConfig:
    n_steps: int = 128            # per env per iteration (SB3 default 2048 with ONE env; E envs here)
    batch_size: int = 16384       # minibatch (SB3 default 64 is sized for a 2048-sample buffer)
    n_epochs: int = 10
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    learning_rate: float = 3e-4
    ent_coef: float = 0.0
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    seed: int = 13                # settings.py:28


def _ortho(layer, gain):
    nn.init.orthogonal_(layer.weight, gain=gain)
    nn.init.zeros_(layer.bias)
    return layer


class _Mlp(nn.Module):
    def __init__(self, obs_dim):
        super().__init__()
        self.policy_net = nn.Sequential(_ortho(nn.Linear(obs_dim, 64), math.sqrt(2)), nn.Tanh(),
                                        _ortho(nn.Linear(64, 64), math.sqrt(2)), nn.Tanh())
        self.value_net = nn.Sequential(_ortho(nn.Linear(obs_dim, 64), math.sqrt(2)), nn.Tanh(),
                                       _ortho(nn.Linear(64, 64), math.sqrt(2)), nn.Tanh())


class ActorCritic(nn.Module):
    """SB3 `ActorCriticPolicy` for a Box(1) action: same parameter names as its state dict."""

    def __init__(self, obs_dim, log_std_init=0.0):
        super().__init__()
        self.mlp_extractor = _Mlp(obs_dim)
        self.action_net = _ortho(nn.Linear(64, 1), 0.01)
        self.value_net = _ortho(nn.Linear(64, 1), 1.0)
        self.log_std = nn.Parameter(torch.full((1,), float(log_std_init)))

    def load_sb3_state_dict(self, sd):
        self.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()
                              if k in self.state_dict()}, strict=True)

    def forward(self, obs):
        x = obs.to(torch.float32)
        mean = self.action_net(self.mlp_extractor.policy_net(x))
        value = self.value_net(self.mlp_extractor.value_net(x)).squeeze(-1)
        return mean, value

    def distribution(self, obs):
        mean, value = self.forward(obs)
        return torch.distributions.Normal(mean, self.log_std.exp().expand_as(mean)), value

    @torch.no_grad()
    def predict(self, obs, deterministic=True):
        mean, _ = self.forward(obs)
        a = mean if deterministic else torch.normal(mean, self.log_std.exp().expand_as(mean))
        return a.clamp(-1.0, 1.0)


@torch.no_grad()
def compute_gae(rewards, values, dones, last_value, gamma, lam):
    """SB3 RolloutBuffer.compute_returns_and_advantage.  rewards/values/dones: [T, E] where
    dones[t] says the episode ended AT step t (the value after it is not bootstrapped)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_value)
    for t in reversed(range(T)):
        next_value = last_value if t == T - 1 else values[t + 1]
        nonterminal = 1.0 - dones[t].to(rewards.dtype)
        delta = rewards[t] + gamma * next_value * nonterminal - values[t]
        last = delta + gamma * lam * nonterminal * last
        adv[t] = last
    return adv, adv + values


class PPOTrainer:
    def __init__(self, venv, config=None, policy=None):
        self.venv = venv
        self.cfg = config or PPOConfig()
        torch.manual_seed(self.cfg.seed)
        self.device = venv.device
        self.policy = (policy or ActorCritic(venv.obs_dim)).to(self.device)
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=self.cfg.learning_rate, eps=1e-5)
        self.obs = venv.reset().clone()
        self.num_timesteps = 0
        self.ep_returns, self.ep_lengths, self.ep_outcomes = [], [], []

    def collect(self):
        cfg, E, T = self.cfg, self.venv.num_envs, self.cfg.n_steps
        dev = self.device
        b_obs = torch.empty(T, E, self.venv.obs_dim, dtype=torch.float32, device=dev)
        b_act = torch.empty(T, E, 1, dtype=torch.float32, device=dev)
        b_logp = torch.empty(T, E, dtype=torch.float32, device=dev)
        b_val = torch.empty(T, E, dtype=torch.float32, device=dev)
        b_rew = torch.empty(T, E, dtype=torch.float32, device=dev)
        b_done = torch.empty(T, E, dtype=torch.bool, device=dev)
        with torch.no_grad():
            for t in range(T):
                dist, value = self.policy.distribution(self.obs)
                action = dist.sample()
                b_obs[t], b_act[t], b_val[t] = self.obs, action, value
                b_logp[t] = dist.log_prob(action).sum(-1)
                # the env sees the clipped action, the buffer keeps the raw one (SB3 collect_rollouts)
                obs, rew, done, infos = self.venv.step(action.clamp(-1.0, 1.0).to(self.venv.dtype))
                b_rew[t], b_done[t] = rew.to(torch.float32), done
                if bool(done.any()):
                    self.ep_returns.append(infos.episode_return[done].float().cpu())
                    self.ep_lengths.append((infos.episode_steps[done] - 1).cpu())
                    self.ep_outcomes.append(infos.outcome[done].cpu())
                self.obs = obs.clone()
            _, last_value = self.policy.forward(self.obs)
            adv, ret = compute_gae(b_rew, b_val, b_done, last_value, cfg.gamma, cfg.gae_lambda)
        self.num_timesteps += T * E
        flat = lambda x: x.reshape(T * E, *x.shape[2:])  # noqa: E731
        return flat(b_obs), flat(b_act), flat(b_logp), flat(adv), flat(ret), flat(b_val)

    def update(self, obs, act, old_logp, adv, ret, old_val):
        cfg = self.cfg
        n = obs.shape[0]
        stats = {}
        for _ in range(cfg.n_epochs):
            perm = torch.randperm(n, device=self.device)
            for i in range(0, n, cfg.batch_size):
                idx = perm[i:i + cfg.batch_size]
                dist, value = self.policy.distribution(obs[idx])
                logp = dist.log_prob(act[idx]).sum(-1)
                a = adv[idx]
                a = (a - a.mean()) / (a.std() + 1e-8)
                ratio = (logp - old_logp[idx]).exp()
                pg = -torch.min(a * ratio, a * ratio.clamp(1 - cfg.clip_range, 1 + cfg.clip_range)).mean()
                vf = torch.nn.functional.mse_loss(value, ret[idx])
                ent = -dist.entropy().sum(-1).mean()
                loss = pg + cfg.ent_coef * ent + cfg.vf_coef * vf
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(self.policy.parameters(), cfg.max_grad_norm)
                self.opt.step()
            stats = {"pg_loss": pg.item(), "value_loss": vf.item(), "std": self.policy.log_std.detach().exp().item()}
        return stats

    def recent_episodes(self, clear=True):
        if not self.ep_returns:
            return None
        r, l, o = torch.cat(self.ep_returns), torch.cat(self.ep_lengths), torch.cat(self.ep_outcomes)
        if clear:
            self.ep_returns, self.ep_lengths, self.ep_outcomes = [], [], []
        return {"episodes": int(r.numel()), "ep_rew_mean": float(r.mean()), "ep_len_mean": float(l.float().mean()),
                "goal": float((o == 1).float().mean()), "collision": float((o == 2).float().mean()),
                "timeout": float((o == 3).float().mean())}

    def learn(self, total_timesteps, log=print):
        t0 = time.time()
        it = 0
        history = []
        while self.num_timesteps < total_timesteps:
            batch = self.collect()
            stats = self.update(*batch)
            it += 1
            ep = self.recent_episodes() or {}
            rec = {"iteration": it, "timesteps": self.num_timesteps,
                   "fps": self.num_timesteps / max(time.time() - t0, 1e-9), **ep, **stats}
            history.append(rec)
            if log:
                log(rec)
        return history
