"""PPO loop on the VecEnv.  CPU: GAE against a scalar restatement of SB3's
RolloutBuffer recursion, MlpPolicy parameter layout (the reference's trained zip loads into it and
reproduces its actions).  GPU: a short training run improves the return."""
import os

import numpy as np
import pytest
import torch

import helpers as H


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
