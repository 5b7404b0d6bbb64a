"""PPO on the device-resident ACAS2DVecEnv.

The reference trains with Stable-Baselines3 (`training_main.py:44-52`:
`PPO('MlpPolicy', env, seed=13).learn(1_048_576)`), one env, CPU.  SB3 is not available here and
its one-env-at-a-time loop is exactly what the batched engine replaces, so this module restates
the algorithm SB3 1.1.0 runs (clipped surrogate, advantages normalised per minibatch, value loss
without clipping, state-independent log-std, orthogonal init, Adam eps 1e-5, gradient-norm clipping)
on E parallel envs: rollouts, GAE and the updates all stay on the GPU.  `PPOConfig.sb3()` carries the
hyper-parameters recorded in the reference's model zips (n_steps 2048, batch 64, epochs 10, gamma
0.99, lambda 0.95, clip 0.2, lr 3e-4, ent 0, vf 0.5, max-grad-norm 0.5); the DEFAULTS differ in the two
sizes that only make sense for one env (n_steps 128 per env, minibatch 16 384: a choice for E >= 1024,
stated here so that it is not mistaken for SB3's).  SB3's own semantics stay *parity unpinned*: no SB3
fixture exists in the reference; what is pinned is the arithmetic of one update (tests/test_ppo.py:
a float64 NumPy restatement with finite-difference gradients) and graph path == eager path.

The network uses SB3's `MlpPolicy` parameter names (separate 2x64 tanh actor / critic), so the
reference's trained zips load as initial weights (`ActorCritic.load_sb3_state_dict`) and a policy
trained here can be evaluated with `policy.SB3ActorPolicy` / `evaluate_policy`.

The loop is launch-bound (a 2x64 MLP on a few thousand rows: ~25 small kernels per env step, ~60 per
minibatch update), so on the GPU it runs from three hipGraphs (`use_graphs`, default on CUDA
devices): ONE env step of the collector (policy forward, sample, log-prob, the step kernel, the
buffer writes -- the time index is a device counter, so the same graph is replayed n_steps times),
the GAE recursion, and ONE minibatch update (gather by a static index buffer, losses, backward,
gradient clipping, capturable Adam).  No host synchronisation inside an iteration; episode
statistics are read once per iteration from the [T, E] side-channel buffers.
"""
import dataclasses
import math
import time

import numpy as np
import torch
from torch import nn


@dataclasses.dataclass
class PPOConfig:
    # The large-batch defaults (E >= 1 024 envs): 512 steps per env and iteration, minibatches of 4 096.  Measured for
    # seed robustness (tools/ppo_seed_sweep.py, 1 024 envs, fused collector + fused update, 60 M steps, seeds 13 / 14 /
    # 15, deterministic evaluation on the reference's 100 test episodes): 100 / 100 / 100 goals (mean return 1 257 /
    # 1 247 / 1 210; the reference's own policy: 100 goals, 1 210.07) -- against 24 / 42 / 44 goals at 30 M steps with
    # 256 steps per iteration and 91 - 99 with 512 (profiles/r03_ppo_seed_sweep_*.jsonl).  SB3's own values: sb3().
    n_steps: int = 512            # per env per iteration (SB3 default 2048 with ONE env; E envs here)
    batch_size: int = 4096        # minibatch (SB3 default 64 is sized for a 2048-sample buffer)
    n_epochs: int = 10
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    learning_rate: float = 3e-4
    ent_coef: float = 0.0
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    seed: int = 13                # settings.py:28

    @classmethod
    def sb3(cls, **overrides):
        """The values SB3 1.1.0's PPO ran the reference's training with (training_main.py:44-52 passes none, so
        these are SB3's defaults as recorded in models/**/*.zip): sized for ONE env -- with E envs the buffer
        is E x 2048 samples cut into minibatches of 64."""
        return cls(**{**dict(n_steps=2048, batch_size=64, n_epochs=10, gamma=0.99, gae_lambda=0.95, clip_range=0.2,
                             learning_rate=3e-4, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5), **overrides})


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

    def actor_weights(self):
        """(w1 [64,D], b1, w2 [64,64], b2, w3 [1,64], b3) for ACAS2DVecEnv.rollout_policy()."""
        pn = self.mlp_extractor.policy_net
        return tuple(t.detach() for t in (pn[0].weight, pn[0].bias, pn[2].weight, pn[2].bias,
                                          self.action_net.weight, self.action_net.bias))

    def forward(self, obs):
        x = obs.to(self.log_std.dtype)                     # float32 (SB3 casts observations the same way)
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


LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def _normal_logp(mean, log_std, x):
    """log N(x; mean, exp(log_std)) summed over the action dimension (torch.distributions.Normal.log_prob)."""
    return (-((x - mean) ** 2) / (2.0 * (2.0 * log_std).exp()) - log_std - LOG_SQRT_2PI).sum(-1)


def ppo_loss(policy, cfg, obs, act, old_logp, adv, ret):
    """SB3 1.1.0 PPO.train() for one minibatch of a Box(1) action space: advantages normalised over the
    minibatch, clipped surrogate, plain MSE value loss (clip_range_vf = None), entropy of the state-independent
    Gaussian.  The ONE loss both the captured and the op-by-op update run.  Returns (loss, pg, vf)."""
    mean, value = policy.forward(obs)
    log_std = policy.log_std
    logp = _normal_logp(mean, log_std, act)
    a = (adv - adv.mean()) / (adv.std() + 1e-8)
    ratio = (logp - old_logp).exp()
    pg = -torch.min(a * ratio, a * ratio.clamp(1 - cfg.clip_range, 1 + cfg.clip_range)).mean()
    vf = torch.nn.functional.mse_loss(value, ret)
    ent = -(0.5 + LOG_SQRT_2PI + log_std).sum()           # -entropy of N(., exp(log_std)), the same for every state
    return pg + cfg.ent_coef * ent + cfg.vf_coef * vf, pg, vf


class FusedUpdate:
    """One PPO minibatch update as two hand-written launches (acas2d_ppo_update_f32, csrc/acas2d_ppo.hip): forward,
    ppo_loss(), backward, clip_grad_norm_ and Adam for the SB3 MlpPolicy actor-critic, on the parameter tensors in
    place.  `obs` [n, D], `act` / `old_logp` / `adv` / `ret` [n] are the flat float32 rollout buffers (their storage
    must stay put), `idx` an int64 device tensor naming the minibatch's rows (rewritten by the caller between
    calls).  Keeps its own Adam moments (torch.optim.Adam's arithmetic, eps 1e-5 as SB3 sets it)."""

    def __init__(self, policy, cfg, obs, act, old_logp, adv, ret, beta1=0.9, beta2=0.999, adam_eps=1e-5):
        import ctypes as C
        from . import native
        self._C, self._native, self._lib = C, native, native.lib()
        dev = obs.device
        D = obs.shape[-1]
        n = int(self._lib.acas2d_ppo_workspace_floats(D))
        z = lambda k, dt=torch.float32: torch.zeros(k, dtype=dt, device=dev)  # noqa: E731
        self.grad, self.m, self.v, self.step_count, self.stats = z(n), z(n), z(n), z(1, torch.int32), z(8)
        pn, vn = policy.mlp_extractor.policy_net, policy.mlp_extractor.value_net
        self._params = [pn[0].weight, pn[0].bias, pn[2].weight, pn[2].bias, policy.action_net.weight, policy.action_net.bias,
                        vn[0].weight, vn[0].bias, vn[2].weight, vn[2].bias, policy.value_net.weight, policy.value_net.bias,
                        policy.log_std]
        assert all(p.dtype == torch.float32 and p.is_contiguous() and p.device == dev for p in self._params)
        self._bufs = [t.reshape(-1) if i else t.reshape(-1, D) for i, t in enumerate((obs, act, old_logp, adv, ret))]
        assert all(t.dtype == torch.float32 and t.is_contiguous() for t in self._bufs)
        self.cfg, self.D, self.betas, self.adam_eps, self.device = cfg, D, (beta1, beta2), adam_eps, dev

    def _struct(self, idx):
        assert idx.dtype == torch.int64 and idx.is_contiguous()
        p = lambda t: t.data_ptr()  # noqa: E731
        cfg = self.cfg
        return self._native.CPpoUpdate(*[p(t) for t in self._params], *[p(t) for t in self._bufs], p(idx), idx.numel(), self.D,
                                       cfg.clip_range, cfg.vf_coef, cfg.ent_coef, cfg.max_grad_norm, cfg.learning_rate,
                                       self.betas[0], self.betas[1], self.adam_eps, p(self.grad), p(self.m), p(self.v),
                                       p(self.step_count), p(self.stats))

    def step(self, idx):
        u = self._struct(idx)
        self._native.check(self._lib.acas2d_ppo_update_f32(
            self._C.byref(u), self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def last_losses(self):
        s = self.stats.cpu().tolist()
        return {"pg_loss": s[4], "value_loss": s[5], "grad_norm": s[2]}


class PPOTrainer:
    """collector: "graphs" (default on a GPU: one captured env step replayed n_steps times), "fused" (the whole
    collection of an iteration in ONE hand-written launch, ACAS2DVecEnv.collect(): actor, critic, Gaussian sampling
    and the env step inside the kernel; its noise comes from the kernel's own Philox stream instead of torch's
    generator) or "eager" (op by op).  updater: "graphs" (default with `use_graphs`: one captured minibatch update of
    torch ops) or "fused" (FusedUpdate: the minibatch update as two hand-written launches, its own Adam state)."""

    def __init__(self, venv, config=None, policy=None, use_graphs=None, collector=None, updater=None):
        self.venv = venv
        self.cfg = config or PPOConfig()
        torch.manual_seed(self.cfg.seed)
        self.device = venv.device
        self.policy = (policy or ActorCritic(venv.obs_dim)).to(self.device)
        self.use_graphs = (torch.device(self.device).type == "cuda") if use_graphs is None else bool(use_graphs)
        self.collector = collector or ("graphs" if self.use_graphs else "eager")
        if self.collector not in ("graphs", "fused", "eager") or (self.collector != "eager" and not self.use_graphs):
            raise ValueError("collector %r needs use_graphs" % (self.collector,))
        self._fused_out = None
        if self.collector == "graphs" and getattr(venv, "double_buffer", False):
            # ONE captured env step is replayed n_steps times: every replay must read what the previous one wrote,
            # which the double-buffered step (read generation g, write 1 - g) does not give a single-step graph
            venv.set_double_buffer(False)
        self.updater = updater or "graphs"
        if self.updater not in ("graphs", "fused") or (self.updater == "fused" and not self.use_graphs):
            raise ValueError("updater %r needs use_graphs" % (self.updater,))
        self._fused_update = None
        # the hand-written launches exist for the observation widths / traffic counts below: say so HERE, not at the
        # first collect() / update() of a run
        f32 = getattr(venv, "dtype", torch.float32) == torch.float32
        if self.collector == "fused" and venv.n_traffic not in ((1, 2, 3, 4, 8) if f32 else (1, 2, 3)):
            raise ValueError("collector='fused' needs a thread-per-env work shape: n_traffic in {1, 2, 3, 4, 8} (float32) / "
                             "{1, 2, 3} (float64), got %d -- use collector='graphs'" % venv.n_traffic)
        if self.updater == "fused" and venv.obs_dim not in (8, 11, 14, 17, 29):
            raise ValueError("updater='fused' is built for obs_dim in {8, 11, 14, 17, 29} (n_traffic 1, 2, 3, 4, 8), got %d "
                             "-- use updater='graphs'" % venv.obs_dim)
        # (fused: one multi-tensor kernel for the 13 parameter tensors instead of ~10 foreach launches)
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=self.cfg.learning_rate, eps=1e-5,
                                    capturable=self.use_graphs, **({"fused": True} if self.use_graphs else {}))
        self.obs = venv.reset().to(torch.float32).clone()
        # Exact parallel flight makes the reference's d_cpa 0/0 = NaN (kinematics.py:48), and its reward
        # with it when the aircraft is traffic[0]; the engine reproduces that.  In float64 it all but
        # never happens; in float32 headings coincide bit for bit about once per 2e7 env steps, and one
        # NaN poisons PPO for good -- so the collector replaces non-finite observations / rewards by 0
        # (what SB3 users wrap such envs in VecCheckNan for) and counts the events.
        self.nan_events = torch.zeros((), dtype=torch.int64, device=self.device)
        self.num_timesteps = 0
        self.ep_returns, self.ep_lengths, self.ep_outcomes = [], [], []
        self._graphs = None

    # ---- rollout buffers (static: the graphs write into them) ------------------------------------
    def _alloc(self):
        E, T, D, dev = self.venv.num_envs, self.cfg.n_steps, self.venv.obs_dim, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        # zeros, not empty: the capture warm-up below runs GAE and two updates over the whole buffer
        self.b_obs = torch.zeros(T, E, D, **f32)
        self.b_act = torch.zeros(T, E, 1, **f32)
        self.b_logp, self.b_val, self.b_rew = (torch.zeros(T, E, **f32) for _ in range(3))
        self.b_adv, self.b_ret, self.b_epret = (torch.zeros(T, E, **f32) for _ in range(3))
        self.b_done = torch.zeros(T, E, dtype=torch.bool, device=dev)
        self.b_eplen = torch.zeros(T, E, dtype=torch.int32, device=dev)
        self.b_outcome = torch.zeros(T, E, dtype=torch.uint8, device=dev)
        self.t_idx = torch.zeros(1, dtype=torch.int64, device=dev)

    def _collect_step(self):
        """One env step of SB3's collect_rollouts into row t_idx of the buffers (graph-capturable:
        no host reads, the row index lives on the device)."""
        v, t = self.venv, self.t_idx
        mean, value = self.policy.forward(self.obs)
        log_std = self.policy.log_std
        action = mean + log_std.exp() * torch.randn_like(mean)
        logp = _normal_logp(mean, log_std, action)
        self.b_obs.index_copy_(0, t, self.obs.unsqueeze(0))
        self.b_act.index_copy_(0, t, action.unsqueeze(0))
        self.b_val.index_copy_(0, t, value.unsqueeze(0))
        self.b_logp.index_copy_(0, t, logp.unsqueeze(0))
        # the env sees the clipped action, the buffer keeps the raw one (SB3 collect_rollouts)
        v.actions_buffer.copy_(action.clamp(-1.0, 1.0).reshape(-1))
        v.step_inplace()
        out = v.outputs
        rew, nxt = out["reward"].to(torch.float32), out["obs"].to(torch.float32)
        self.nan_events.add_(torch.isnan(rew).sum() + torch.isnan(nxt).any(-1).sum())
        self.b_rew.index_copy_(0, t, torch.nan_to_num(rew, nan=0.0).unsqueeze(0))
        self.b_done.index_copy_(0, t, out["done"].view(torch.bool).unsqueeze(0))
        self.b_epret.index_copy_(0, t, out["episode_return"].to(torch.float32).unsqueeze(0))
        self.b_eplen.index_copy_(0, t, out["episode_steps"].unsqueeze(0))
        self.b_outcome.index_copy_(0, t, out["outcome"].unsqueeze(0))
        self.obs.copy_(torch.nan_to_num(nxt, nan=0.0))
        t.add_(1)

    def _gae(self):
        _, last_value = self.policy.forward(self.obs)
        adv, ret = compute_gae(self.b_rew, self.b_val, self.b_done, last_value, self.cfg.gamma, self.cfg.gae_lambda)
        self.b_adv.copy_(adv)
        self.b_ret.copy_(ret)

    def _minibatch(self, idx):
        """One PPO minibatch update on the rows named by a static index buffer."""
        T, E = self.cfg.n_steps, self.venv.num_envs
        flat = lambda x: x.reshape(T * E, *x.shape[2:])  # noqa: E731
        # (torch.distributions validates its arguments with a host read: not capturable -- ppo_loss() does not use it)
        loss, pg, vf = ppo_loss(self.policy, self.cfg, flat(self.b_obs)[idx], flat(self.b_act)[idx],
                                flat(self.b_logp)[idx], flat(self.b_adv)[idx], flat(self.b_ret)[idx])
        loss.backward()
        nn.utils.clip_grad_norm_(self.policy.parameters(), self.cfg.max_grad_norm)
        self.opt.step()
        return pg.detach(), vf.detach()

    def _capture(self):
        """Warm the three bodies up on a side stream, then capture them (PyTorch's whole-network
        capture recipe: gradients are None at capture time, so backward assigns static buffers).
        The warm-up steps the env and takes real optimizer steps on an all-zero buffer: the env, the
        observation, the parameters and the Adam state are put back IN PLACE afterwards (the graphs
        hold their addresses), so training starts from exactly the state it was constructed in."""
        self._alloc()
        n = self.cfg.n_steps * self.venv.num_envs
        B = min(self.cfg.batch_size, n)
        self.mb_idx = torch.zeros(B, dtype=torch.int64, device=self.device)
        # the partial last minibatch of an epoch, as SB3 takes it -- unless it is ONE row: the advantage normalisation
        # divides by the standard deviation of the minibatch, which one row does not have (NaN in SB3 as well)
        self.mb_tail = torch.zeros(n % B, dtype=torch.int64, device=self.device) if n % B > 1 else None
        env_before = self.venv.state_dict()
        obs_before, env_obs_before = self.obs.clone(), self.venv.outputs["obs"].clone()
        params_before = [p.detach().clone() for p in self.policy.parameters()]
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            with torch.no_grad():
                for _ in range(2):
                    self.t_idx.zero_()
                    self._collect_step()
                self._gae()
            for idx in (self.mb_idx, self.mb_idx, self.mb_tail):
                if idx is not None:
                    self.opt.zero_grad(set_to_none=True)
                    self._minibatch(idx)
        torch.cuda.current_stream(self.device).wait_stream(side)
        g_step, g_gae, g_upd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.t_idx.zero_()
        with torch.no_grad():
            with torch.cuda.graph(g_step):
                self._collect_step()
            with torch.cuda.graph(g_gae):
                self._gae()
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g_upd):
            self._pg, self._vf = self._minibatch(self.mb_idx)
        g_tail = None
        if self.mb_tail is not None:                      # the last, partial minibatch of an epoch: its own static shape
            g_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_tail):
                self._minibatch(self.mb_tail)
        self._graphs = (g_step, g_gae, g_upd, g_tail)
        with torch.no_grad():
            for p, q in zip(self.policy.parameters(), params_before):
                p.copy_(q)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()                         # exp_avg, exp_avg_sq, step: a fresh Adam
            self.venv.load_state_dict(env_before)
            self.venv.outputs["obs"].copy_(env_obs_before)
            self.obs.copy_(obs_before)
            self.nan_events.zero_()
            self.t_idx.zero_()

    def collect(self):
        cfg, E, T = self.cfg, self.venv.num_envs, self.cfg.n_steps
        if self.use_graphs:
            if self._graphs is None:
                self._capture()
            if self.collector == "fused":
                # one launch: rows t = 0 .. T-1 of the static buffers, then the captured GAE as usual
                out = self.venv.collect(self.policy, T, noise_seed=cfg.seed, noise_step=self.num_timesteps // E,
                                        out=self._fused_out)
                self._fused_out = out
                obs_all, rew = out["obs"].to(torch.float32), out["reward"].to(torch.float32)
                self.nan_events.add_(torch.isnan(rew).sum() + torch.isnan(obs_all[1:]).any(-1).sum())
                obs_all = torch.nan_to_num(obs_all, nan=0.0, posinf=0.0, neginf=0.0)    # what the kernel fed the networks
                self.b_obs.copy_(obs_all[:T])
                self.obs.copy_(obs_all[T])
                self.b_act.copy_(out["actions"].to(torch.float32).unsqueeze(-1))
                self.b_val.copy_(out["values"].to(torch.float32))
                self.b_logp.copy_(out["logp"].to(torch.float32))
                self.b_rew.copy_(torch.nan_to_num(rew, nan=0.0))
                self.b_done.copy_(out["done"])
                self.b_epret.copy_(out["episode_return"].to(torch.float32))
                self.b_eplen.copy_(out["episode_steps"])
                self.b_outcome.copy_(out["outcome"])
            else:
                self.t_idx.zero_()
                for _ in range(T):
                    self._graphs[0].replay()
            self._graphs[1].replay()
            done = self.b_done
            if bool(done.any()):                          # the iteration's one host synchronisation
                self.ep_returns.append(self.b_epret[done].cpu())
                self.ep_lengths.append((self.b_eplen[done] - 1).cpu())
                self.ep_outcomes.append(self.b_outcome[done].cpu())
            self.num_timesteps += T * E
            return None
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
                self.nan_events += torch.isnan(rew).sum() + torch.isnan(obs).any(-1).sum()
                b_rew[t], b_done[t] = torch.nan_to_num(rew.to(torch.float32), nan=0.0), done
                if bool(done.any()):
                    self.ep_returns.append(infos.episode_return[done].float().cpu())
                    self.ep_lengths.append((infos.episode_steps[done] - 1).cpu())
                    self.ep_outcomes.append(infos.outcome[done].cpu())
                self.obs = torch.nan_to_num(obs.to(torch.float32), nan=0.0)
            _, last_value = self.policy.forward(self.obs)
            adv, ret = compute_gae(b_rew, b_val, b_done, last_value, cfg.gamma, cfg.gae_lambda)
        self.num_timesteps += T * E
        flat = lambda x: x.reshape(T * E, *x.shape[2:])  # noqa: E731
        return flat(b_obs), flat(b_act), flat(b_logp), flat(adv), flat(ret), flat(b_val)

    def update(self, obs=None, act=None, old_logp=None, adv=None, ret=None, old_val=None):
        cfg = self.cfg
        if self.use_graphs and self.updater == "fused":
            n, B = cfg.n_steps * self.venv.num_envs, self.mb_idx.numel()
            if self._fused_update is None:
                self._fused_update = FusedUpdate(self.policy, cfg, self.b_obs, self.b_act, self.b_logp, self.b_adv, self.b_ret)
            fu = self._fused_update
            for _ in range(cfg.n_epochs):
                perm = torch.randperm(n, device=self.device)
                for i in range(0, n - B + 1, B):
                    self.mb_idx.copy_(perm[i:i + B])
                    fu.step(self.mb_idx)
                if self.mb_tail is not None:
                    self.mb_tail.copy_(perm[n - n % B:])
                    fu.step(self.mb_tail)
            st = fu.last_losses()
            return {"pg_loss": st["pg_loss"], "value_loss": st["value_loss"], "std": self.policy.log_std.detach().exp().item()}
        if self.use_graphs:
            n, B = cfg.n_steps * self.venv.num_envs, self.mb_idx.numel()
            for _ in range(cfg.n_epochs):
                perm = torch.randperm(n, device=self.device)
                for i in range(0, n - B + 1, B):          # whole minibatches: one static shape
                    self.mb_idx.copy_(perm[i:i + B])
                    self._graphs[2].replay()
                if self.mb_tail is not None:              # ... and the partial one SB3 also takes, its own graph
                    self.mb_tail.copy_(perm[n - n % B:])
                    self._graphs[3].replay()
            return {"pg_loss": self._pg.item(), "value_loss": self._vf.item(),
                    "std": self.policy.log_std.detach().exp().item()}
        n = obs.shape[0]
        stats = {}
        for _ in range(cfg.n_epochs):
            perm = torch.randperm(n, device=self.device)
            for i in range(0, n, cfg.batch_size):
                idx = perm[i:i + cfg.batch_size]
                if idx.numel() < 2:
                    continue                              # a one-row tail has no advantage standard deviation
                loss, pg, vf = ppo_loss(self.policy, cfg, obs[idx], act[idx], old_logp[idx], adv[idx], ret[idx])
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(self.policy.parameters(), cfg.max_grad_norm)
                self.opt.step()
            stats = {"pg_loss": pg.item(), "value_loss": vf.item(), "std": self.policy.log_std.detach().exp().item()}
        return stats

    def optimizer_state(self):
        """The Adam state that is actually being stepped: with updater='fused' the moments live in FusedUpdate's
        flat buffers (layout: actor w1 b1 w2 b2 w3 b3, critic likewise, log_std -- include/acas2d.h) and `self.opt`
        is never stepped; otherwise `self.opt.state_dict()`."""
        if self.updater == "fused" and self.use_graphs:
            fu = self._fused_update
            if fu is None:
                return {"updater": "fused", "step": 0, "exp_avg": None, "exp_avg_sq": None}
            return {"updater": "fused", "step": int(fu.step_count.item()), "exp_avg": fu.m, "exp_avg_sq": fu.v}
        return {"updater": self.updater, **self.opt.state_dict()}

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
            stats = self.update() if batch is None else self.update(*batch)
            it += 1
            ep = self.recent_episodes() or {}
            rec = {"iteration": it, "timesteps": self.num_timesteps,
                   "fps": self.num_timesteps / max(time.time() - t0, 1e-9), **ep, **stats,
                   "nan_events": int(self.nan_events)}
            history.append(rec)
            if log:
                log(rec)
        return history
