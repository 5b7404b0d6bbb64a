"""PPO loop on the VecEnv.  CPU: GAE against a scalar restatement of SB3's
RolloutBuffer recursion, MlpPolicy parameter layout (the reference's trained zip loads into it and
reproduces its actions).  GPU: a short training run improves the return."""
import os

import numpy as np
import pytest
import torch

import helpers as H


def bits_equal(x, y):
    """torch.equal on the bit patterns (a NaN d_cpa -- exact parallel flight, kinematics.py:48 -- equals itself)."""
    if x.is_floating_point():
        bits = torch.int32 if x.dtype == torch.float32 else torch.int64
        return x.shape == y.shape and torch.equal(x.contiguous().view(bits), y.contiguous().view(bits))
    return torch.equal(x, y)


def test_gae_matches_scalar_recursion():
    import gym_acas2d_amd as g
    rng = np.random.default_rng(0)
    T, E, gamma, lam = 17, 5, 0.99, 0.95
    r, v = rng.normal(size=(T, E)), rng.normal(size=(T, E))
    d = rng.random((T, E)) < 0.2
    lv = rng.normal(size=E)
    adv = np.zeros((T, E))
    for e in range(E):
        last = 0.0
        for t in reversed(range(T)):
            nv = lv[e] if t == T - 1 else v[t + 1, e]
            nt = 0.0 if d[t, e] else 1.0
            delta = r[t, e] + gamma * nv * nt - v[t, e]
            last = delta + gamma * lam * nt * last
            adv[t, e] = last
    a, ret = g.compute_gae(torch.tensor(r), torch.tensor(v), torch.tensor(d), torch.tensor(lv), gamma, lam)
    np.testing.assert_allclose(a.numpy(), adv, atol=1e-12)
    np.testing.assert_allclose(ret.numpy(), adv + v, atol=1e-12)


def test_actor_critic_has_sb3_layout_and_loads_the_reference_policy():
    import gym_acas2d_amd as g
    sd = dict(np.load(os.path.join(H.GOLDEN, "ref_policy_best_model.npz"), allow_pickle=False))
    sd.pop("sb3_version")
    ac = g.ActorCritic(8)
    assert set(ac.state_dict()) == set(sd)                      # SB3 1.1.0 MlpPolicy parameter names
    ac.load_sb3_state_dict(sd)
    obs = torch.as_tensor(np.random.default_rng(1).uniform(-1, 1, (64, 8)))
    ref = g.load_sb3_policy(os.path.join(H.GOLDEN, "ref_policy_best_model.npz"))
    assert torch.allclose(ac.predict(obs), ref.predict(obs), atol=1e-6)
    fresh = g.ActorCritic(8)
    assert fresh.log_std.detach().item() == 0.0 and fresh.action_net.weight.abs().max() < 0.05   # SB3 init gains


def _np_loss(theta, shapes, cfg, obs, act, old_logp, adv, ret):
    """ppo_loss() restated in float64 NumPy from SB3 1.1.0's PPO.train(): separate 2 x 64 tanh actor / critic,
    Gaussian log-prob, per-minibatch advantage normalisation (torch.std: Bessel-corrected), clipped surrogate,
    MSE value loss, entropy bonus."""
    p, k = {}, 0
    for name, shp in shapes:
        n = int(np.prod(shp))
        p[name] = theta[k:k + n].reshape(shp)
        k += n

    def mlp(prefix, x):
        h = np.tanh(x @ p[prefix + ".0.weight"].T + p[prefix + ".0.bias"])
        return np.tanh(h @ p[prefix + ".2.weight"].T + p[prefix + ".2.bias"])

    mean = mlp("mlp_extractor.policy_net", obs) @ p["action_net.weight"].T + p["action_net.bias"]
    value = (mlp("mlp_extractor.value_net", obs) @ p["value_net.weight"].T + p["value_net.bias"])[:, 0]
    log_std = p["log_std"]
    logp = (-((act - mean) ** 2) / (2.0 * np.exp(2.0 * log_std)) - log_std - 0.5 * np.log(2.0 * np.pi)).sum(-1)
    a = (adv - adv.mean()) / (adv.std(ddof=1) + 1e-8)
    ratio = np.exp(logp - old_logp)
    pg = -np.minimum(a * ratio, a * np.clip(ratio, 1 - cfg.clip_range, 1 + cfg.clip_range)).mean()
    vf = ((value - ret) ** 2).mean()
    ent = -(0.5 + 0.5 * np.log(2.0 * np.pi) + log_std).sum()
    return pg + cfg.ent_coef * ent + cfg.vf_coef * vf


def test_one_minibatch_update_against_a_float64_numpy_restatement():
    """The arithmetic of ONE update -- loss, gradient (central finite differences of the NumPy restatement: nothing
    of torch's autograd), global-norm clipping as torch.nn.utils.clip_grad_norm_ does it, the first Adam step
    (eps 1e-5) -- against the torch path the trainer runs.  SB3 itself stays parity unpinned (no fixture)."""
    import gym_acas2d_amd as g
    torch.manual_seed(3)
    rng = np.random.default_rng(5)
    D, B = 8, 40
    cfg = g.PPOConfig(ent_coef=0.01, max_grad_norm=0.5)
    pol = g.ActorCritic(D).double()
    with torch.no_grad():
        pol.action_net.weight.mul_(30.0)                        # away from the near-zero SB3 init: ratios spread, some clip
        pol.log_std.fill_(-0.3)
    obs, act = rng.uniform(-1, 1, (B, D)), rng.normal(0, 0.7, (B, 1))
    adv, ret = rng.normal(0, 2, B), rng.normal(0, 1, B)
    names = [(k, tuple(v.shape)) for k, v in pol.state_dict().items()]
    theta = np.concatenate([v.detach().numpy().ravel() for v in pol.state_dict().values()])
    mean0 = pol.forward(torch.as_tensor(obs))[0].detach().numpy()
    old_logp = (-((act - mean0) ** 2) / (2 * np.exp(-0.6)) + 0.3 - 0.5 * np.log(2 * np.pi)).sum(-1) + rng.normal(0, 0.25, B)
    args = (cfg, obs, act, old_logp, adv, ret)
    loss_t, pg_t, vf_t = g.ppo_loss(pol, cfg, *(torch.as_tensor(x) for x in args[1:]))
    assert abs(loss_t.item() - _np_loss(theta, names, *args)) < 1e-12
    ratio = np.exp((-((act - mean0) ** 2) / (2 * np.exp(-0.6)) + 0.3 - 0.5 * np.log(2 * np.pi)).sum(-1) - old_logp)
    assert ((ratio < 0.8) | (ratio > 1.2)).sum() >= 5 and ((ratio > 0.8) & (ratio < 1.2)).sum() >= 5     # both branches of the clip
    # gradient by central differences of the restatement
    grad, h = np.zeros_like(theta), 1e-6
    for i in range(theta.size):
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        grad[i] = (_np_loss(tp, names, *args) - _np_loss(tm, names, *args)) / (2 * h)
    opt = torch.optim.Adam(pol.parameters(), lr=cfg.learning_rate, eps=1e-5)
    loss_t.backward()
    g_t = np.concatenate([pol.get_parameter(k).grad.numpy().ravel() for k, _ in names])
    assert np.abs(g_t - grad).max() < 1e-7 * max(1.0, np.abs(grad).max())
    torch.nn.utils.clip_grad_norm_(pol.parameters(), cfg.max_grad_norm)
    opt.step()
    total = np.sqrt((grad ** 2).sum())
    assert total > cfg.max_grad_norm                            # the clip is active in this case
    gc = grad * min(1.0, cfg.max_grad_norm / (total + 1e-6))
    m_hat, v_hat = gc, gc ** 2                                  # first step: the bias corrections cancel the (1 - beta) factors
    expect = theta - cfg.learning_rate * m_hat / (np.sqrt(v_hat) + 1e-5)
    after = np.concatenate([pol.get_parameter(k).detach().numpy().ravel() for k, _ in names])
    assert np.abs(after - expect).max() < 2e-9, np.abs(after - expect).max()
    sb3 = g.PPOConfig.sb3()
    assert (sb3.n_steps, sb3.batch_size, sb3.n_epochs, sb3.max_grad_norm, sb3.learning_rate) == (2048, 64, 10, 0.5, 3e-4)


def _eval_on_reference_episodes(g, policy):
    own, trf, goal = H.parity_reset_states(g.ACAS2DConfig(), 13, 0, 100)
    ev = g.ACAS2DVecEnv(100, 1, device="cuda:0", dtype=torch.float64, auto_reset=False)
    ev.set_state(own, trf, goal, np.zeros(100, np.int32))
    return g.evaluate_policy(ev, policy)


@pytest.mark.gpu
def test_short_ppo_run_learns():
    """16 iterations (4.2 M env steps, ~8 s from hipGraphs) of the configuration that solves the task
    in 60 M (profiles/r01_ppo_train_graphs_1024x256.jsonl).  Judged by the deterministic evaluation on the
    reference's 100 test episodes: the untrained policy's mean action is ~0, i.e. the reference's
    constant-action baseline (58 collisions / 42 goals, mean return -70.8, notebook
    baseline_ACAS2D_PPO_11_100.ipynb:280); by then PPO has learned to stay clear of the traffic."""
    import gym_acas2d_amd as g
    venv = g.ACAS2DVecEnv(1024, 1, device="cuda:0", dtype=torch.float32, seed=13)
    tr = g.PPOTrainer(venv, g.PPOConfig(n_steps=256, batch_size=4096))
    before = _eval_on_reference_episodes(g, tr.policy)
    assert (before["outcome"] == 2).sum() >= 40 and before["total_reward"].mean() < 0      # ~ the baseline
    hist = tr.learn(16 * 256 * 1024, log=None)
    assert len(hist) == 16 and hist[-1]["timesteps"] == 16 * 256 * 1024
    after = _eval_on_reference_episodes(g, tr.policy)
    assert (after["outcome"] == 2).sum() <= 10, np.bincount(after["outcome"], minlength=4)
    assert after["total_reward"].mean() > before["total_reward"].mean() + 50


@pytest.mark.gpu
def test_captured_update_equals_the_op_by_op_update():
    """One iteration's update from the three hipGraphs == the same update launched op by op, minibatch for
    minibatch including the partial last one of every epoch (n = 5120 samples, minibatch 2048), from the same
    batch, the same permutations and the same initial weights; and capturing leaves the trainer as constructed
    (weights, Adam state, env)."""
    import gym_acas2d_amd as g
    cfg = g.PPOConfig(n_steps=20, batch_size=2048, n_epochs=3)
    mk = lambda: g.ACAS2DVecEnv(256, 1, device="cuda:0", dtype=torch.float32, seed=13)  # noqa: E731
    tg, te = g.PPOTrainer(mk(), cfg), g.PPOTrainer(mk(), cfg, use_graphs=False)
    init = [p.detach().clone() for p in tg.policy.parameters()]
    for p, q in zip(te.policy.parameters(), init):
        assert torch.equal(p, q)                               # same seed, same init
    env0 = {k: v.clone() for k, v in tg.venv.state_dict().items()}
    tg._capture()
    for p, q in zip(tg.policy.parameters(), init):
        assert torch.equal(p, q)                               # the warm-up's optimizer steps were undone in place
    assert all(float(v.abs().sum()) == 0 for st in tg.opt.state.values() for v in st.values() if torch.is_tensor(v))
    for k, v in tg.venv.state_dict().items():
        assert torch.equal(v, env0[k]), k
    assert tg.collect() is None and tg.mb_tail is not None and tg.mb_tail.numel() == 1024
    flat = lambda x: x.reshape(20 * 256, *x.shape[2:]).clone()  # noqa: E731
    batch = [flat(b) for b in (tg.b_obs, tg.b_act, tg.b_logp, tg.b_adv, tg.b_ret, tg.b_val)]
    torch.manual_seed(99)
    sg = tg.update()
    torch.manual_seed(99)
    se = te.update(*batch)
    with torch.no_grad():
        pg_, pe_ = (torch.cat([p.reshape(-1) for p in t.policy.parameters()]) for t in (tg, te))
        p0 = torch.cat([p.reshape(-1) for p in init])
    worst, moved = float((pg_ - pe_).abs().max()), float((pg_ - p0).abs().max())
    mean_diff, mean_moved = float((pg_ - pe_).abs().mean()), float((pg_ - p0).abs().mean())
    # 9 Adam steps of ~lr each.  The two paths differ by float32 summation order only (fused multi-tensor Adam,
    # atomics in the gather's backward), which Adam's g / (|g| + eps) amplifies for the few parameters whose
    # gradient is ~0; a dropped or different minibatch would move a third of the steps (>= 0.3 x moved).
    print("captured vs op-by-op update: worst %.3g of moved %.3g, mean %.3g of mean moved %.3g" % (worst, moved, mean_diff, mean_moved))
    assert moved > 1e-3 and worst < 0.1 * moved and mean_diff < 0.02 * mean_moved, (worst, moved, mean_diff, mean_moved)
    # (reported from different minibatches: the captured path's last FULL one, the op-by-op path's partial one)
    assert abs(sg["value_loss"] - se["value_loss"]) < 0.05 * abs(se["value_loss"])


def _random_actor_critic(g, D, seed):
    torch.manual_seed(seed)
    pol = g.ActorCritic(D).to("cuda:0")
    with torch.no_grad():
        pol.action_net.weight.mul_(40.0)            # away from SB3's near-zero init: the mean depends on the observation
        pol.log_std.fill_(-0.7)
    return pol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name,N,E,T", (("float32", 1, 1024, 420), ("float32", 8, 2048, 40), ("float64", 3, 1000, 40)))
def test_fused_collector_against_torch_and_a_twin_env(g_mod, dtype_name, N, E, T):
    """acas2d_collect_*: SB3's collect_rollouts in one launch.  (a) on the observations the kernel itself recorded,
    torch's forward gives the same values and -- for the actions the kernel drew -- the same log-probabilities;
    (b) the implied noise (action - mean) / std is standard normal and uncorrelated; (c) a twin env stepped with the
    clipped actions reproduces every observation, reward and mask bit for bit; (d) the noise depends on (seed,
    global env index, step) only: reproducible, different for another step, invariant to sharding."""
    g = g_mod
    dtype = getattr(torch, dtype_name)
    D = 5 + 3 * N
    pol = _random_actor_critic(g, D, 1)
    mk = lambda E_, off=0: g.ACAS2DVecEnv(E_, N, device="cuda:0", dtype=dtype, seed=21, env_offset=off)  # noqa: E731
    env, twin = mk(E), mk(E)
    env.reset(); twin.reset()
    out = env.collect(pol, T, noise_seed=7, noise_step=1000)
    torch.cuda.synchronize()
    obs = torch.nan_to_num(out["obs"].to(torch.float32), nan=0.0, posinf=0.0, neginf=0.0)
    with torch.no_grad():
        mean, value = pol.forward(obs[:T].reshape(T * E, D))
    mean, value = mean.reshape(T, E), value.reshape(T, E)
    std = float(pol.log_std.detach().exp())
    act, logp = out["actions"].to(torch.float32), out["logp"].to(torch.float32)
    assert float((out["values"].to(torch.float32) - value).abs().max()) < 5e-5 * max(1.0, float(value.abs().max()))
    eps = (act - mean) / std
    ref_logp = -0.5 * eps ** 2 - float(pol.log_std.detach()) - 0.5 * np.log(2 * np.pi)
    assert float((logp - ref_logp).abs().max()) < 2e-3 and float((logp - ref_logp).abs().mean()) < 2e-5
    e = eps.double().cpu().numpy()
    n = e.size
    assert abs(e.mean()) < 5 / np.sqrt(n) and abs(e.var() - 1) < 8 / np.sqrt(n) and abs((e ** 4).mean() - 3) < 0.15
    assert abs(np.corrcoef(e[:-1].ravel(), e[1:].ravel())[0, 1]) < 5 / np.sqrt(n)          # step to step
    assert abs(np.corrcoef(e[:, :-1].ravel(), e[:, 1:].ravel())[0, 1]) < 5 / np.sqrt(n)    # env to env
    dones = 0
    for t in range(T):
        o, r, d, _ = twin.step(act[t].clamp(-1, 1).to(dtype))
        assert bits_equal(o, out["obs"][t + 1]) and bits_equal(r, out["reward"][t]) and bits_equal(d, out["done"][t]), t
        dones += int(d.sum())
    assert dones > 0 and bits_equal(env.outputs["obs"], out["obs"][T])
    for name in ("own_x", "trf_x", "steps", "episode", "total_reward"):
        assert torch.equal(getattr(env, name), getattr(twin, name)), name
    # (d)
    again = mk(E); again.reset()
    o2 = again.collect(pol, T, noise_seed=7, noise_step=1000)
    assert bits_equal(o2["actions"], out["actions"]) and bits_equal(o2["obs"], out["obs"])
    other = mk(E); other.reset()
    o3 = other.collect(pol, 4, noise_seed=7, noise_step=1001)
    assert not torch.equal(o3["actions"][0], out["actions"][0]) and torch.equal(o3["actions"][0] - o3["values"][0] * 0, o3["actions"][0])
    half = mk(E // 2, off=E // 2); half.reset()
    o4 = half.collect(pol, T, noise_seed=7, noise_step=1000)
    assert bits_equal(o4["actions"], out["actions"][:, E // 2:]) and bits_equal(o4["obs"], out["obs"][:, :, :][:, E // 2:])


@pytest.mark.gpu
@pytest.mark.parametrize("D,n,B", ((8, 6000, 2048 + 37), (29, 3000, 1000), (14, 700, 64)))
def test_fused_update_against_torch_autograd_and_adam(g_mod, D, n, B):
    """acas2d_ppo_update_f32 (two hand-written launches) against the torch path the trainer otherwise runs --
    ppo_loss(), autograd, clip_grad_norm_, torch.optim.Adam(eps = 1e-5) -- on the same minibatch: the loss values, the
    gradient norm and the parameters after one and after three updates (the Adam moments carry over)."""
    g = g_mod
    torch.manual_seed(11)
    dev = "cuda:0"
    cfg = g.PPOConfig(ent_coef=0.01, max_grad_norm=0.5, learning_rate=3e-4)
    mine = _random_actor_critic(g, D, 5)
    ref = g.ActorCritic(D).to(dev)
    ref.load_state_dict(mine.state_dict())
    obs = torch.rand(n, D, device=dev) * 2 - 1
    act = torch.randn(n, device=dev) * 0.7
    adv, ret = torch.randn(n, device=dev) * 2, torch.randn(n, device=dev)
    with torch.no_grad():
        mean, _ = ref.forward(obs)
        old_logp = g.ppo._normal_logp(mean, ref.log_std, act.unsqueeze(-1)) + torch.randn(n, device=dev) * 0.25
    # the raw gradient first (max_grad_norm < 0: nothing applied), entry for entry against autograd
    import dataclasses
    probe = g.FusedUpdate(mine, dataclasses.replace(cfg, max_grad_norm=-1.0), obs, act, old_logp, adv, ret)
    idx0 = torch.randperm(n, device=dev)[:B].contiguous()
    probe.step(idx0)
    loss0, _, _ = g.ppo_loss(ref, cfg, obs[idx0], act[idx0].unsqueeze(-1), old_logp[idx0], adv[idx0], ret[idx0])
    loss0.backward()
    pn, vn = ref.mlp_extractor.policy_net, ref.mlp_extractor.value_net
    want = torch.cat([t.grad.reshape(-1) for t in (pn[0].weight, pn[0].bias, pn[2].weight, pn[2].bias, ref.action_net.weight,
                                                    ref.action_net.bias, vn[0].weight, vn[0].bias, vn[2].weight, vn[2].bias,
                                                    ref.value_net.weight, ref.value_net.bias, ref.log_std)])
    got = probe.grad.clone()
    got[-1] -= cfg.ent_coef                                    # (the entropy term is added by the apply launch)
    err = float((got - want).abs().max()) / float(want.abs().max())
    cos = float(torch.dot(got, want) / (got.norm() * want.norm()))
    print("fused PPO gradient vs autograd (D = %d, B = %d): max |diff| / max |g| = %.2e, cosine %.8f, norms %.6g %.6g"
          % (D, B, err, cos, float(got.norm()), float(want.norm())))
    assert err < 2e-4 and cos > 0.999999
    ref.zero_grad(set_to_none=True)
    fu = g.FusedUpdate(mine, cfg, obs, act, old_logp, adv, ret)
    opt = torch.optim.Adam(ref.parameters(), lr=cfg.learning_rate, eps=1e-5)
    for k in range(3):
        idx = torch.randperm(n, device=dev)[:B].contiguous()
        loss, pg, vf = g.ppo_loss(ref, cfg, obs[idx], act[idx].unsqueeze(-1), old_logp[idx], adv[idx], ret[idx])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        norm = float(torch.nn.utils.clip_grad_norm_(ref.parameters(), cfg.max_grad_norm))
        opt.step()
        fu.step(idx)
        torch.cuda.synchronize()
        st = fu.last_losses()
        assert abs(st["pg_loss"] - float(pg.detach())) < 2e-5 * max(1.0, abs(float(pg.detach()))) + 2e-6, (k, st, float(pg.detach()))
        assert abs(st["value_loss"] - float(vf.detach())) < 1e-4 * float(vf.detach()), (k, st, float(vf.detach()))
        assert abs(st["grad_norm"] - norm) < 2e-4 * norm, (k, st["grad_norm"], norm)
        for (name, p), q in zip(mine.named_parameters(), ref.parameters()):
            d = float((p - q).abs().max())
            assert d < 0.02 * cfg.learning_rate, (k, name, d)        # an Adam step moves a parameter by ~lr
    assert int(fu.step_count) == 3 and float(fu.grad.abs().max()) == 0.0


@pytest.mark.gpu
def test_ppo_learns_with_the_fused_update(g_mod):
    """The short run of test_short_ppo_run_learns with both halves of an iteration hand-written: one launch collects,
    two launches per minibatch update."""
    g = g_mod
    venv = g.ACAS2DVecEnv(1024, 1, device="cuda:0", dtype=torch.float32, seed=13)
    tr = g.PPOTrainer(venv, g.PPOConfig(n_steps=256, batch_size=4096), collector="fused", updater="fused")
    before = _eval_on_reference_episodes(g, tr.policy)
    hist = tr.learn(16 * 256 * 1024, log=None)
    assert len(hist) == 16 and np.isfinite(hist[-1]["value_loss"])
    after = _eval_on_reference_episodes(g, tr.policy)
    assert (after["outcome"] == 2).sum() <= 10, np.bincount(after["outcome"], minlength=4)
    assert after["total_reward"].mean() > before["total_reward"].mean() + 50


@pytest.fixture(scope="module")
def g_mod():
    import gym_acas2d_amd as g
    return g


@pytest.mark.gpu
def test_ppo_learns_with_the_fused_collector(g_mod):
    """The same short run as test_short_ppo_run_learns with the whole collection of every iteration in one launch."""
    g = g_mod
    venv = g.ACAS2DVecEnv(1024, 1, device="cuda:0", dtype=torch.float32, seed=13)
    tr = g.PPOTrainer(venv, g.PPOConfig(n_steps=256, batch_size=4096), collector="fused")
    before = _eval_on_reference_episodes(g, tr.policy)
    hist = tr.learn(16 * 256 * 1024, log=None)
    assert len(hist) == 16 and hist[-1]["timesteps"] == 16 * 256 * 1024 and hist[-1]["nan_events"] <= 2
    after = _eval_on_reference_episodes(g, tr.policy)
    assert (after["outcome"] == 2).sum() <= 10, np.bincount(after["outcome"], minlength=4)
    assert after["total_reward"].mean() > before["total_reward"].mean() + 50


@pytest.mark.gpu
def test_eager_ppo_path_still_runs_and_counts_nan_events():
    """use_graphs=False is the same algorithm launched op by op (the debugging path)."""
    import gym_acas2d_amd as g
    venv = g.ACAS2DVecEnv(256, 1, device="cuda:0", dtype=torch.float32, seed=13)
    tr = g.PPOTrainer(venv, g.PPOConfig(n_steps=32, batch_size=2048), use_graphs=False)
    hist = tr.learn(2 * 32 * 256, log=None)
    assert len(hist) == 2 and np.isfinite(hist[-1]["value_loss"]) and hist[-1]["nan_events"] == 0
    tg = g.PPOTrainer(g.ACAS2DVecEnv(256, 1, device="cuda:0", dtype=torch.float32, seed=13),
                      g.PPOConfig(n_steps=32, batch_size=2048))
    assert tg.use_graphs
    hg = tg.learn(2 * 32 * 256, log=None)
    assert len(hg) == 2 and np.isfinite(hg[-1]["value_loss"]) and hg[-1]["timesteps"] == hist[-1]["timesteps"]
