"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes ->
libacas2d_hip.so), against the CPU oracle and the committed golden vectors.

Tolerances (north star: masks bit-exact, 1e-5 abs on float positions / rewards):
  float64 instantiation -- the parity gate.  Asserted far tighter than required: 1e-9 abs on
      positions, observations, rewards and returns over full episodes; done / outcome / step
      masks bit-exact (cases placed within 1e-9 of a threshold excepted and counted).
  float32 instantiation -- throughput mode.  float32 cannot REPRESENT 1600-px positions or
      +-1000 rewards to 1e-5 (ulp(1024..2048) = 1.2e-4, ulp(1000) = 6.1e-5), so single steps from
      identical (float32-representable) states are held to: observations 1e-5 abs (2e-5 for the
      signed d_cpa entry), non-terminal rewards 1e-5 abs, positions / terminal rewards 1 float32
      ulp (1.3e-4); masks exact outside a 1e-3 band around the thresholds.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def g():
    import gym_acas2d_amd as g
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    g.native.lib()          # fails loudly if the HIP extension is missing
    return g


@pytest.fixture(scope="module")
def O(oracle_mod):
    return oracle_mod


@pytest.fixture(params=("exact", "fast"))
def math(request):
    """The two formulations of the float64 build (include/acas2d.h, ACAS2D_MATH_*): the reference's operation
    order with libm, and the float32 build's algebraic formulation in float64 arithmetic.  Both are held to the
    same fixtures at the same 1e-9 (the contract asks 1e-5)."""
    return request.param


def bits_equal(x, y):
    """torch.equal on the BIT patterns: the engine reproduces the reference's NaN d_cpa in exact parallel flight
    (kinematics.py:48), which float32 headings hit about once per 4e6 env-steps at N = 8 -- and NaN != NaN."""
    if x.is_floating_point():
        bits = torch.int32 if x.dtype == torch.float32 else torch.int64
        return x.shape == y.shape and torch.equal(x.contiguous().view(bits), y.contiguous().view(bits))
    return torch.equal(x, y)


class GpuEngine:
    """The HIP path behind the OracleEnvs interface (numpy float64 views), see helpers.py."""

    def __init__(self, g, E, N, dtype=None, auto_reset=False, seed=13, env_offset=0, math="exact"):
        self.v = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype or torch.float64,
                                auto_reset=auto_reset, seed=seed, env_offset=env_offset,
                                config=g.ACAS2DConfig(n_traffic=N, fast_math=(math == "fast")))
        self.E, self.N = E, N

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy().astype(np.float64) if t.is_floating_point() else t.detach().cpu().numpy()

    def __getattr__(self, name):
        if name in ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y",
                    "trf_psi", "trf_v", "steps", "total_reward", "status"):
            return self._np(getattr(self.v, name))
        if name == "episode":
            return self.v.episode.cpu().numpy().view(np.uint32)
        if name in ("term_obs", "ep_return", "ep_steps"):
            key = {"term_obs": "terminal_observation", "ep_return": "episode_return",
                   "ep_steps": "episode_steps"}[name]
            return self._np(self.v.outputs[key])
        raise AttributeError(name)

    def set_state(self, own, trf, goal=None, steps=None):
        self.v.set_state(own, trf, goal, steps, observe=False)

    def observe(self):
        # observe() on the state as it stands: re-inject it with observe=True
        st = np.stack([self.own_x, self.own_y, self.own_psi, self.own_v], 1)
        tr = np.stack([self.trf_x, self.trf_y, self.trf_psi, self.trf_v], -1)
        go = np.stack([self.goal_x, self.goal_y], 1)
        return self._np(self.v.set_state(st, tr, go, self.steps, observe=True))

    def reset(self):
        return self._np(self.v.reset())

    def step(self, actions):
        obs, rew, done, _ = self.v.step(np.asarray(actions, np.float64))
        out = self.v.outputs
        d = done.cpu().numpy().astype(np.uint8)
        return self._np(obs), self._np(rew), d, out["outcome"].cpu().numpy(), int(d.sum())


def grazing(fx_obs, N, cfg, band):
    """Rows whose post-step geometry lies within `band` of a collision / goal threshold."""
    d_sep = fx_obs[:, 5::3][:, :N] * cfg.d_sep_max
    d_goal = fx_obs[:, 3] * cfg.d_goal_max
    with np.errstate(invalid="ignore"):
        return (np.abs(d_sep - cfg.collision_dist) < band).any(1) | (np.abs(d_goal - cfg.goal_radius) < band)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", (1, 3, 8, 64))
def test_f64_edge_vectors(g, O, N, math):
    fx = H.load("ref_edge_n%d.npz" % N)
    E = len(fx["action"])
    env = GpuEngine(g, E, N, math=math)
    env.set_state(fx["own"], fx["trf"], fx["goal"], fx["steps"])
    obs, reward, done, outcome, _ = env.step(fx["action"])
    assert np.array_equal(np.isnan(obs), np.isnan(fx["obs"]))
    want = fx["obs"].copy()
    if math == "fast":
        # kinematics.py:47 takes arctan(v12y / v12x): d_cpa changes SIGN with the sign of v12x.  The hand-placed
        # mirror-image headings make the reference's v12x an exact 0 (or +-1 ulp) -- a coin toss of libm's cos that
        # no other sincos reproduces (the float32 tests exclude |v12x| < 0.02 for the same reason); the magnitude
        # still has to match.
        rad = lambda d: d / 360.0 * 2 * np.pi  # noqa: E731
        op, ov = fx["own_out"][:, 2][:, None], fx["own_out"][:, 3][:, None]
        tp, tv = fx["trf_out"][..., 2], fx["trf_out"][..., 3]
        v12x = ov * np.cos(rad(op)) - tv * np.cos(rad(tp))
        coin = np.abs(v12x) < 1e-9
        # The exempt set is pinned by the fixture's own geometry, not by a bound: exactly the hand-placed entries with
        # equal airspeeds whose headings are parallel (psi_t == psi) or mirror images (psi_t + psi == 360) -- 9 entries
        # at N = 1, 16 at N = 3 / 8 / 64 -- and nothing else has |v12x| < 1e-9.
        assert np.array_equal(coin, (tv == ov) & ((tp == op) | (tp + op == 360.0)))
        assert int(coin.sum()) == {1: 9}.get(N, 16)
        flip = coin & (np.sign(obs[:, 6::3][:, :N]) != np.sign(want[:, 6::3][:, :N]))
        want[:, 6::3][:, :N][flip] *= -1.0
    np.testing.assert_allclose(obs, want, rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(reward, fx["reward"], rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(np.stack([env.own_x, env.own_y, env.own_psi, env.own_v], 1),
                               fx["own_out"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.stack([env.trf_x, env.trf_y, env.trf_psi, env.trf_v], -1),
                               fx["trf_out"], rtol=0, atol=1e-9)
    assert np.array_equal(env.steps, fx["steps_out"])
    cfgc = O.default_config()
    ok = ~grazing(fx["obs"], N, cfgc, 1e-9)
    assert ok.sum() >= E - 12
    assert np.array_equal(done[ok], fx["done"][ok]) and np.array_equal(outcome[ok], fx["outcome"][ok])
    assert np.array_equal(env.status[ok], fx["outcome"][ok])        # latched (auto_reset off)


@pytest.mark.parametrize("N", (1, 3, 8, 64))
def test_f64_reference_rollouts(g, N, math):
    fx = H.load("ref_rollout_n%d.npz" % N)
    n_ep = len(fx["ep_own"])
    env = GpuEngine(g, n_ep, N, math=math)
    env.set_state(fx["ep_own"], fx["ep_trf"], fx["ep_goal"], np.zeros(n_ep, np.int32))
    np.testing.assert_allclose(env.observe(), fx["ep_obs0"], rtol=0, atol=1e-9)
    res = H.replay_rollout(env, fx)
    assert res["n"] == len(fx["action"])
    assert res["done_mismatch"] == 0 and res["outcome_mismatch"] == 0 and res["steps_mismatch"] == 0
    assert res["pos"] < 1e-9 and res["psi"] < 1e-9 and res["obs"] < 1e-9
    assert res["reward"] < 1e-9 and res["total_reward"] < 1e-8


def test_f64_reference_csv_baseline(g, math):
    """The reference's own golden CSV replayed on the GPU (100 constant-action episodes)."""
    dg = H.load("csv_baseline_digest.npz")
    cfg = g.ACAS2DConfig(n_traffic=1)
    own, trf, goal = H.parity_reset_states(cfg, 13, 2, 100)
    env = GpuEngine(g, 100, 1, math=math)
    env.set_state(own, trf, goal, np.zeros(100, np.int32))
    env.observe()
    out = H.replay_baseline(env, dg, own, trf)
    assert out["unfinished"] == 0
    assert np.array_equal(out["outcome"], dg["outcome"]) and np.array_equal(out["steps"], dg["steps"])
    for k in ("own_sub", "trf_sub", "own_first2", "own_last"):
        np.testing.assert_allclose(out[k], dg[k], rtol=0, atol=1e-9, equal_nan=True)
    assert np.abs(out["total_reward"] - dg["total_reward"]).max() < 1e-8


def test_single_env_adapter_reference_surface(g):
    """ACAS2DEnv: random.seed(13) names the reference's episodes, old gym
    4-tuple API, numpy float64, env.game.* attributes, CSV episode 1 end to end."""
    import random
    random.seed(13)
    env = g.ACAS2DEnv()
    obs = env.reset()
    assert obs.dtype == np.float64 and obs.shape == (8,) and env.observation_space.shape == (8,)
    assert env.action_space.shape == (1,)
    p, t = env.game.player, env.game.traffic[0]
    assert (p.x, p.y, p.v_air) == (48, 500.0, 200) and p.psi == 358.1242450086868
    assert (t.x, t.y, t.v_air) == (1552, 48, 200.0) and t.psi == 136.41722591475224
    want = [0.001, 0.99478957, 0., 0.41314554, 0., 0.26677536, 0.08703283, -0.92934645]
    np.testing.assert_allclose(obs, want, atol=1e-8)
    env.reset()                                 # third game after the seed = CSV episode 1
    total, n = 0.0, 0
    for _ in range(1000):
        o, r, d, info = env.step(np.array([0]))
        assert isinstance(r, float) and isinstance(d, bool) and info == {}
        total += r
        n += 1
        if n == 1:
            assert abs(env.game.path[-1][0] - 49.998468716044925) < 1e-11
            assert abs(env.game.path[-1][1] - 499.92175173490904) < 1e-11
        if d:
            break
    assert env.game.outcome == 2 and g.OUTCOME_NAMES[env.game.outcome] == "Collision"
    assert env.game.steps == 390
    assert abs(env.game.total_reward - (-988.6418379569138)) < 1e-8 and abs(total - env.game.total_reward) < 1e-9
    assert len(env.game.path) == 390 and len(env.game.traffic_paths[0]) == 390
    assert env.game.traffic_paths[0][1] == env.game.traffic_paths[0][0]      # appendix A quirk
    # stepping a finished env: player moves, traffic frozen (game.py:243-245)
    t_before = (env.game.traffic[0].x, env.game.traffic[0].y)
    env.step(np.array([0.0]))
    assert (env.game.traffic[0].x, env.game.traffic[0].y) == t_before


def test_records_in_the_reference_csv_layout(g, tmp_path):
    """baseline_main.simulate() on the adapter: same columns as the reference's CSV, and the first
    episodes agree with it (outcome, steps, return, path)."""
    import csv
    import random
    dg = H.load("csv_baseline_digest.npz")
    random.seed(13)
    env = g.ACAS2DEnv()
    env.reset()                                  # the game check_env consumed (baseline_main.py:22)
    cols = g.records.simulate(env, episodes=3)
    assert tuple(cols) == g.records.BASELINE_COLUMNS and cols["Episode"] == [1, 2, 3]
    for i in range(3):
        assert cols["Outcome"][i] == {1: "Goal", 2: "Collision", 3: "Timeout"}[int(dg["outcome"][i])]
        assert cols["Time Steps"][i] == dg["steps"][i] and len(cols["Path"][i]) == dg["n_points"][i]
        assert abs(cols["Total Reward"][i] - dg["total_reward"][i]) < 1e-8
        np.testing.assert_allclose(cols["Path"][i][:2], dg["own_first2"][i], atol=1e-9)
        np.testing.assert_allclose(cols["Path"][i][-1], dg["own_last"][i], atol=1e-9)
        np.testing.assert_allclose(cols["Traffic Paths"][i][0][:3], dg["trf_first3"][i], atol=1e-9)
    out = tmp_path / "baseline.csv"
    g.records.to_csv(cols, out)
    rows = list(csv.DictReader(open(out)))
    assert list(rows[0]) == list(g.records.BASELINE_COLUMNS) and rows[0]["Outcome"] == "Collision"
    import ast
    assert rows[0]["Path"].startswith("[(48.0, 500.0), (49.99846871604")      # the reference's file: "[(48, 500.0), (49.998468716044925, ..."
    assert len(ast.literal_eval(rows[0]["Path"])) == 390 and len(ast.literal_eval(rows[0]["Traffic Paths"])[0]) == 390


@pytest.mark.parametrize("N", (1, 3))
def test_testing_main_record_columns_vs_reference(g, N, tmp_path):
    """every column testing_main.py:114-138 writes -- Path Length and the thirteen per-step
    record lists ACAS2DGame keeps (game.py:132-160, :231-241, :266-276) -- from the engine's trace rows
    (include/acas2d.h, Acas2dState.trace), against the lists captured from the unmodified reference
    (tests/golden/ref_records_n{N}.npz, oracle/refharness/capture_golden.py capture_records): random-action
    episodes replayed from the captured initial states.  d_sep is the separation AFTER the player moved and
    BEFORE the traffic did (:236-237 vs :243-245), r_step the step reward before the terminal bonuses."""
    fx = H.load("ref_records_n%d.npz" % N)
    env = g.ACAS2DEnv(n_traffic=N)
    n_ep = len(fx["outcome"])
    acts = iter(fx["actions"])
    states = [(fx["own0"][i], fx["trf0"][i], fx["goal0"][i]) for i in range(n_ep)]
    cols = g.records.simulate(env, episodes=n_ep, policy=lambda obs: np.array([next(acts)]), columns="testing",
                              initial_states=states)
    assert tuple(cols) == g.records.TESTING_COLUMNS
    off = fx["off_records"]
    worst = {}
    for i in range(n_ep):
        lo, hi = off[i], off[i + 1]
        assert cols["Outcome"][i] == {1: "Goal", 2: "Collision", 3: "Timeout"}[int(fx["outcome"][i])]
        assert cols["Time Steps"][i] == fx["steps"][i]
        assert abs(cols["Total Reward"][i] - fx["total_reward"][i]) < 1e-8
        assert abs(cols["Path Length"][i] - fx["d_path"][i]) < 1e-9
        np.testing.assert_allclose(np.array(cols["Path"][i]), fx["path"][lo:hi], atol=1e-9, rtol=0)
        np.testing.assert_allclose(np.array(cols["Traffic Paths"][i]).transpose(1, 0, 2), fx["traffic_paths"][lo:hi],
                                   atol=1e-9, rtol=0)
        for col, attr in g.records.TESTING_RECORDS:
            got, want = np.array(cols[col][i]), fx[attr][lo:hi]
            assert got.shape == want.shape, (col, got.shape, want.shape)
            assert np.array_equal(np.isnan(got), np.isnan(want)), col
            err = float(np.nanmax(np.abs(got - want)))
            worst[col] = max(worst.get(col, 0.0), err)
            assert err < 1e-9, (i, col, err)
    assert next(acts, None) is None                              # every recorded action was consumed
    assert worst["a_lat"] == 0.0 and worst["psi"] < 1e-11
    out = tmp_path / "testing.csv"
    g.records.to_csv(cols, out)
    import csv
    csv.field_size_limit(1 << 30)
    rows = list(csv.DictReader(open(out)))
    assert list(rows[0]) == list(g.records.TESTING_COLUMNS) and len(rows) == n_ep


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", (1, 3, 8, 64))
def test_f32_single_step_vs_f64_oracle(g, O, N):
    fx = H.load("ref_edge_n%d.npz" % N)
    sel = ~np.isnan(fx["obs"]).any(1)
    own = fx["own"][sel].astype(np.float32).astype(np.float64)        # float32-representable inputs
    trf = fx["trf"][sel].astype(np.float32).astype(np.float64)
    act = fx["action"][sel].astype(np.float32).astype(np.float64)
    E = len(act)
    ref = O.OracleEnvs(E, N)
    ref.set_state(own, trf, fx["goal"], fx["steps"][sel])
    o, r, d, oc, _ = ref.step(act)
    env = GpuEngine(g, E, N, dtype=torch.float32)
    env.set_state(own, trf, fx["goal"], fx["steps"][sel])
    obs, rew, done, outcome, _ = env.step(act)
    cfgc = O.default_config()
    ok = ~grazing(o, N, cfgc, 1e-3)
    assert ok.sum() >= E - 30
    assert np.array_equal(done[ok], d[ok]) and np.array_equal(outcome[ok], oc[ok])
    col = np.arange(o.shape[1])
    cpa = (col >= 5) & ((col - 5) % 3 == 1)
    assert np.abs(obs[:, ~cpa] - o[:, ~cpa]).max() < 1e-5
    # d_cpa = d * sin(a_rel - arctan(v12y / v12x)) (kinematics.py:48-49) is ill-conditioned in the
    # reference itself where the relative velocity v12 is tiny (near-parallel flight: float32
    # rounding of 200*cos(psi), ~1.2e-5, turns into an angle error 2.4e-5 / |v12|) and jumps by
    # +-2 d where v12x changes sign.  Entries with |v12| < 2 px/s or |v12x| < 0.02 px/s (of 200)
    # are excluded -- counted: a handful.
    rad = np.deg2rad
    v12x = (ref.own_v * np.cos(rad(ref.own_psi)))[:, None] - ref.trf_v * np.cos(rad(ref.trf_psi))
    v12y = (ref.own_v * np.sin(rad(ref.own_psi)))[:, None] - ref.trf_v * np.sin(rad(ref.trf_psi))
    well = (np.abs(v12x) > 0.02) & (np.hypot(v12x, v12y) > 2.0)
    assert (~well).mean() < 0.01 or (~well).sum() <= 4
    assert np.abs(obs[:, cpa] - o[:, cpa])[well].max() < 2e-5
    ok &= well[:, 0]                                  # the reward reads traffic[0]'s d_cpa
    term = d.astype(bool)
    assert np.abs(rew[ok & ~term] - r[ok & ~term]).max() < 1e-5
    assert np.abs(rew[ok & term] - r[ok & term]).max() <= 1.3e-4          # 1 ulp of float32(1000)
    pos_err = max(np.abs(env.own_x - ref.own_x).max(), np.abs(env.own_y - ref.own_y).max(),
                  np.abs(env.trf_x - ref.trf_x).max(), np.abs(env.trf_y - ref.trf_y).max())
    assert pos_err <= 1.3e-4                                             # 1 ulp of float32(1600)
    assert np.abs(env.own_psi - ref.own_psi).max() <= 3.1e-5             # 1 ulp of float32(360)


@pytest.mark.parametrize("N", (1, 3, 8, 64))
def test_f32_reproduces_the_reference_nan_pattern(g, N):
    """Parallel flight (identical heading and speed) makes the reference's relative velocity 0/0:
    d_cpa is NaN (kinematics.py:48) and so is the reward whenever it reads traffic[0]'s d_cpa.  The
    algebraic float32 formulation (0 * inf) must put NaN in exactly the same places."""
    fx = H.load("ref_edge_n%d.npz" % N)
    rows = np.isnan(fx["obs"]).any(1)
    assert rows.sum() >= 8
    f32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
    env = GpuEngine(g, int(rows.sum()), N, dtype=torch.float32)
    env.set_state(f32(fx["own"][rows]), f32(fx["trf"][rows]), fx["goal"], fx["steps"][rows])
    obs, rew, done, outcome, _ = env.step(f32(fx["action"][rows]))
    assert np.array_equal(np.isnan(obs), np.isnan(fx["obs"][rows]))
    assert np.array_equal(np.isnan(rew), np.isnan(fx["reward"][rows]))
    ok = ~np.isnan(fx["obs"][rows])
    assert np.abs(obs[ok] - fx["obs"][rows][ok]).max() < 2e-5
    # masks: the fixture's boundary rows sit within an ulp of float64 of the 96 px / 144 px
    # thresholds (game.py:291,299) -- float32 inputs cannot represent that, so they are compared
    # only where the reference's own distances clear the threshold by a float32 rounding margin
    clear = ~grazing(fx["obs"][rows], N, g.ACAS2DConfig(n_traffic=N).to_c(), 1e-3)
    assert clear.sum() >= 4
    assert np.array_equal(done[clear], fx["done"][rows][clear])
    assert np.array_equal(outcome[clear], fx["outcome"][rows][clear])


@pytest.mark.parametrize("N", (1, 8))
def test_f32_full_episodes_vs_f64_oracle(g, O, N):
    """Whole episodes in float32 against the float64 oracle on the same actions: rounding
    accumulates (<= ~0.1 px over 1000 steps), so outcomes / lengths may differ only for episodes
    that cross a threshold within that margin.  Reported, and bounded at 2 %."""
    E, T = 512, 1001
    cfg = g.ACAS2DConfig(n_traffic=N)
    own, trf, goal = H.parity_reset_states(cfg, 4242, 0, E)
    rng = np.random.default_rng(5)
    ref = O.OracleEnvs(E, N)
    ref.set_state(own, trf, goal, np.zeros(E, np.int32))
    ref.observe()
    env = GpuEngine(g, E, N, dtype=torch.float32)
    env.set_state(own, trf, goal, np.zeros(E, np.int32))
    env.observe()
    fin_r, fin_g = np.zeros(E, np.int32), np.zeros(E, np.int32)
    oc_r, oc_g = np.zeros(E, np.uint8), np.zeros(E, np.uint8)
    max_pos = 0.0
    for k in range(T):
        a = rng.uniform(-1, 1, E).astype(np.float32).astype(np.float64)
        _, _, d1, o1, _ = ref.step(a)
        _, _, d2, o2, _ = env.step(a)
        new = (fin_r == 0) & (d1 != 0)
        fin_r[new], oc_r[new] = k + 1, o1[new]
        new = (fin_g == 0) & (d2 != 0)
        fin_g[new], oc_g[new] = k + 1, o2[new]
        both = (fin_r == 0) & (fin_g == 0)
        if both.any():
            max_pos = max(max_pos, float(np.abs(env.own_x - ref.own_x)[both].max()),
                          float(np.abs(env.own_y - ref.own_y)[both].max()))
        if (fin_r > 0).all() and (fin_g > 0).all():
            break
    agree = (fin_r == fin_g) & (oc_r == oc_g)
    print("f32 vs f64 episodes N=%d: %d/%d agree, max |dpos| while both running %.3g px" %
          (N, agree.sum(), E, max_pos))
    assert (fin_r > 0).all() and (fin_g > 0).all()
    assert agree.mean() >= 0.98
    assert np.abs(fin_r - fin_g).max() <= 2 or agree.mean() >= 0.98
    assert max_pos < 0.25


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,E,T", ((3, 4096, 60), (8, 2048, 200), (64, 512, 40), (1, 256, 450), (100, 96, 12)))
def test_f64_auto_reset_vs_oracle(g, O, N, E, T, math):
    """VecEnv semantics and the device Philox reset against the oracle, env-for-env: same seed
    => same episodes, bit-exact reset states, terminal obs / returns / lengths, episode counters."""
    ref = O.OracleEnvs(E, N, seed=99, env_offset=1000, auto_reset=True)
    env = GpuEngine(g, E, N, auto_reset=True, seed=99, env_offset=1000, math=math)
    o_ref, o_gpu = ref.reset(), env.reset()
    for name in ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi", "trf_v"):
        assert np.array_equal(getattr(env, name), getattr(ref, name)), name     # reset: bit-exact
    np.testing.assert_allclose(o_gpu, o_ref, rtol=0, atol=1e-9)
    rng = np.random.default_rng(1)
    dones = 0
    for _ in range(T):
        a = rng.uniform(-1, 1, E)
        o1, r1, d1, oc1, n1 = ref.step(a)
        o2, r2, d2, oc2, n2 = env.step(a)
        assert np.array_equal(d1, d2) and np.array_equal(oc1, oc2)
        np.testing.assert_allclose(o2, o1, rtol=0, atol=1e-9)
        np.testing.assert_allclose(r2, r1, rtol=0, atol=1e-9)
        assert np.array_equal(env.steps, ref.steps) and np.array_equal(env.episode, ref.episode)
        d = d1.astype(bool)
        if d.any():
            dones += int(d.sum())
            np.testing.assert_allclose(env.term_obs[d], ref.term_obs[d], rtol=0, atol=1e-9)
            np.testing.assert_allclose(env.ep_return[d], ref.ep_return[d], rtol=0, atol=1e-8)
            assert np.array_equal(env.ep_steps[d], ref.ep_steps[d])
            assert np.array_equal(env.trf_psi[d], ref.trf_psi[d]) and np.array_equal(env.own_psi[d], ref.own_psi[d])
        np.testing.assert_allclose(env.total_reward, ref.total_reward, rtol=0, atol=1e-8)
    assert dones > 0


def _oracle_config_from(O, cfg):
    """OracleConfig carrying a NON-default product configuration (same field names)."""
    cc, oc = cfg.to_c(), O.OracleConfig()
    for name, _ in O.OracleConfig._fields_:
        if name != "_pad":                      # (the product's `math` selector: the oracle has one formulation)
            setattr(oc, name, getattr(cc, name))
    return oc


@pytest.mark.parametrize("N,E,T", ((2, 1500, 80), (5, 777, 90), (6, 1024, 60), (7, 333, 60), (12, 640, 50), (33, 200, 30)))
def test_f64_odd_traffic_counts_and_nondefault_config_vs_oracle(g, O, N, E, T):
    """Generic work shapes (N not a power-of-two multiple of the vector width) and a configuration
    in which every tunable differs from settings.py -- different airspace, frame rate, radii,
    reward constants, an airspeed-factor RANGE (so traffic and player speeds differ and the
    kinematics.py:74 quirk matters everywhere), short episodes (timeouts occur)."""
    cfg = g.ACAS2DConfig(n_traffic=N, max_steps=120, width=2000, height=1200, fps=50, aircraft_size=20,
                         airspeed=180, airspeed_factor_min=0.8, airspeed_factor_max=1.3,
                         acc_lat_limit=150.0, player_initial_heading_lim=10, traffic_initial_heading_lim=25,
                         reward_goal=500, reward_collision=-750)
    ref = O.OracleEnvs(E, N, seed=3, env_offset=17, auto_reset=True, config=_oracle_config_from(O, cfg))
    env = GpuEngine.__new__(GpuEngine)
    env.v = g.ACAS2DVecEnv(E, device="cuda:0", dtype=torch.float64, seed=3, env_offset=17, config=cfg)
    env.E, env.N = E, N
    o1, o2 = ref.reset(), env.reset()
    for name in ("own_psi", "trf_x", "trf_y", "trf_psi", "trf_v", "goal_x", "own_x"):
        assert np.array_equal(getattr(env, name), getattr(ref, name)), name
    assert len(np.unique(ref.trf_v)) > 10                      # speeds really vary
    np.testing.assert_allclose(o2, o1, rtol=0, atol=1e-9)
    rng = np.random.default_rng(9)
    seen = set()
    for _ in range(T):
        a = rng.uniform(-1, 1, E)
        o1, r1, d1, oc1, _ = ref.step(a)
        o2, r2, d2, oc2, _ = env.step(a)
        assert np.array_equal(d1, d2) and np.array_equal(oc1, oc2)
        np.testing.assert_allclose(o2, o1, rtol=0, atol=1e-9)
        np.testing.assert_allclose(r2, r1, rtol=0, atol=1e-9)
        seen |= set(np.unique(oc1))
    assert np.array_equal(env.steps, ref.steps) and np.array_equal(env.episode, ref.episode)
    assert {0, 2} <= seen


def test_f32_statistical_single_step_vs_f64_oracle(g, O):
    """200 000 mid-episode states (oracle rollouts with resets, N = 8), ONE float32 step each from
    the identical float32-representable state, against the float64 oracle: the distribution of the
    observation error, not just a maximum over a few fixtures."""
    E, N = 200_000, 8
    ref = O.OracleEnvs(E, N, seed=1234, auto_reset=True)
    ref.reset()
    rng = np.random.default_rng(4)
    for _ in range(int(rng.integers(20, 40))):
        ref.step(rng.uniform(-1, 1, E))
    f32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
    own = f32(np.stack([ref.own_x, ref.own_y, ref.own_psi, ref.own_v], 1))
    trf = f32(np.stack([ref.trf_x, ref.trf_y, ref.trf_psi, ref.trf_v], -1))
    steps = ref.steps.copy()
    act = f32(rng.uniform(-1, 1, E))
    chk = O.OracleEnvs(E, N)
    chk.set_state(own, trf, None, steps)
    o, r, d, oc, _ = chk.step(act)
    env = GpuEngine(g, E, N, dtype=torch.float32)
    env.set_state(own, trf, None, steps)
    obs, rew, done, outcome, _ = env.step(act)
    cfgc = O.default_config()
    ok = ~grazing(o, N, cfgc, 1e-3)
    assert ok.mean() > 0.999
    assert np.array_equal(done[ok], d[ok]) and np.array_equal(outcome[ok], oc[ok])
    rad = np.deg2rad
    v12x = (chk.own_v * np.cos(rad(chk.own_psi)))[:, None] - chk.trf_v * np.cos(rad(chk.trf_psi))
    v12y = (chk.own_v * np.sin(rad(chk.own_psi)))[:, None] - chk.trf_v * np.sin(rad(chk.trf_psi))
    well = (np.abs(v12x) > 0.02) & (np.hypot(v12x, v12y) > 2.0)
    err = np.abs(obs - o)
    err[:, [1, 4]] = np.minimum(err[:, [1, 4]], 1.0 - err[:, [1, 4]])      # headings live on a circle
    col = np.arange(o.shape[1])
    cpa = (col >= 5) & ((col - 5) % 3 == 1)
    vcl = (col >= 5) & ((col - 5) % 3 == 2)
    # closing speed = dot(dv, dp) / |dp| / dt (kinematics.py:77): for aircraft a few pixels apart
    # (random spawns on top of the player) float32 position rounding (1.2e-4 px) is a visible
    # fraction of |dp| -- ill-conditioned in the reference itself; entries with |dp| < 16 px are
    # bounded separately
    near = (o[:, 5::3] * cfgc.d_sep_max) < 16.0
    e_own, e_dist = err[:, :5], err[:, (col >= 5) & ((col - 5) % 3 == 0)]
    e_vc, e_vc_near, e_cpa = err[:, vcl][~near], err[:, vcl][near], err[:, cpa][well]
    print("f32 one-step |obs error| vs f64 oracle over %d states: player entries max %.2e; distance max %.2e; closing "
          "speed max %.2e p99.9 %.2e (|dp| >= 16 px; %d entries closer: max %.2e); d_cpa (well-conditioned, %.2f %%) "
          "max %.2e p99.9 %.2e" % (E, e_own.max(), e_dist.max(), e_vc.max(), np.quantile(e_vc, 0.999), near.sum(),
                                   e_vc_near.max() if near.any() else 0.0, 100 * well.mean(), e_cpa.max(),
                                   np.quantile(e_cpa, 0.999)))
    assert e_own.max() < 1e-5 and e_dist.max() < 1e-5 and e_vc.max() < 1e-5 and e_cpa.max() < 2e-5
    assert near.mean() < 1e-3 and (not near.any() or e_vc_near.max() < 1e-3)
    assert np.quantile(err[:, ~cpa], 0.999) < 2e-6
    # the shaped reward multiplies in (d_cpa / 192)^4 (rewards.py:12-16), i.e. it amplifies the
    # d_cpa entry's error by up to 4 * 1886 / 192 = 39x: 1e-5 holds for 99.99 % of the states,
    # the worst of 200 000 stays below 5e-5
    nt = ok & ~d.astype(bool) & well[:, 0]
    e_rew = np.abs(rew[nt] - r[nt])
    print("f32 one-step |reward error| (non-terminal): max %.2e p99.99 %.2e" % (e_rew.max(), np.quantile(e_rew, 0.9999)))
    assert np.quantile(e_rew, 0.9999) < 1e-5 and e_rew.max() < 5e-5
    assert max(np.abs(env.own_x - chk.own_x).max(), np.abs(env.trf_x - chk.trf_x).max(),
               np.abs(env.trf_y - chk.trf_y).max()) <= 1.3e-4


def test_f64_fast_statistical_single_step_vs_oracle(g, O):
    """The float64 FAST formulation beyond the fixtures: 200 000 mid-episode states (oracle rollouts with resets,
    N = 8), ONE step each from the identical state, against the oracle.  Everything within 1e-9 (the contract asks
    1e-5) except where the reference itself is ill-conditioned: d_cpa divides by |v12| and takes the sign of v12x
    (kinematics.py:40-49), so its error scales with 1 / |v12| and its sign is a coin toss at v12x = +-1e-13."""
    E, N = 200_000, 8
    ref = O.OracleEnvs(E, N, seed=4321, auto_reset=True)
    ref.reset()
    rng = np.random.default_rng(9)
    for _ in range(int(rng.integers(20, 40))):
        ref.step(rng.uniform(-1, 1, E))
    own = np.stack([ref.own_x, ref.own_y, ref.own_psi, ref.own_v], 1)
    trf = np.stack([ref.trf_x, ref.trf_y, ref.trf_psi, ref.trf_v], -1)
    steps, act = ref.steps.copy(), rng.uniform(-1, 1, E)
    chk = O.OracleEnvs(E, N)
    chk.set_state(own, trf, None, steps)
    o, r, d, oc, _ = chk.step(act)
    env = GpuEngine(g, E, N, math="fast")
    env.set_state(own, trf, None, steps)
    obs, rew, done, outcome, _ = env.step(act)
    ok = ~grazing(o, N, O.default_config(), 1e-9)
    assert ok.mean() > 0.99999
    assert np.array_equal(done[ok], d[ok]) and np.array_equal(outcome[ok], oc[ok])
    rad = np.deg2rad
    v12x = (chk.own_v * np.cos(rad(chk.own_psi)))[:, None] - chk.trf_v * np.cos(rad(chk.trf_psi))
    v12y = (chk.own_v * np.sin(rad(chk.own_psi)))[:, None] - chk.trf_v * np.sin(rad(chk.trf_psi))
    well = (np.abs(v12x) > 1e-6) & (np.hypot(v12x, v12y) > 1e-3)
    err = np.abs(obs - o)
    err[:, [1, 4]] = np.minimum(err[:, [1, 4]], 1.0 - err[:, [1, 4]])      # headings live on a circle
    col = np.arange(o.shape[1])
    cpa = (col >= 5) & ((col - 5) % 3 == 1)
    e_cpa = err[:, cpa]
    print("f64 FAST one-step |obs error| vs oracle over %d states: all but d_cpa max %.2e; d_cpa max %.2e (%.4f %% of the "
          "entries ill-conditioned and set aside); reward max %.2e; positions max %.2e"
          % (E, np.nanmax(err[:, ~cpa]), np.nanmax(e_cpa[well]), 100 * (~well).mean(),
             np.nanmax(np.abs(rew - r)[ok & well[:, 0]]), np.abs(env.trf_x - chk.trf_x).max()))
    # measured: 1.1e-15 / 8.4e-13 / 1.1e-13 (reward) / 2.3e-13 (positions)
    assert np.nanmax(err[:, ~cpa]) < 1e-12 and np.nanmax(e_cpa[well]) < 1e-10 and (~well).mean() < 1e-4
    assert np.array_equal(np.isnan(obs), np.isnan(o))
    assert np.nanmax(np.abs(rew - r)[ok & well[:, 0]]) < 1e-11
    for name in ("own_x", "own_y", "own_psi", "trf_x", "trf_y"):
        assert np.nanmax(np.abs(getattr(env, name) - getattr(chk, name))) < 1e-11, name
    assert np.array_equal(env.steps, chk.steps)


def test_f32_reset_names_the_same_episodes(g, O):
    """(seed, global env index, episode counter) names ONE episode per element type, whichever path draws
    it: reset() / reset_masked() (reset_kernel) and the auto-reset inside a step (both of its walks) call
    the same per-entity function.  The float32 build evaluates the draws in float32 (24 random bits):
    equal to the float64 oracle up to float32 rounding of the 1600-px / 360-degree ranges (positions
    2.5e-4, headings 6e-5); the float64 build is bit-equal to the oracle (test_f64_auto_reset_vs_oracle)."""
    E, N = 4096, 8
    ref = O.OracleEnvs(E, N, seed=5, auto_reset=True)

    def close_to_oracle(env, sel):
        assert np.abs(env.trf_x[sel] - ref.trf_x[sel]).max() < 2.5e-4
        assert np.abs(env.trf_y[sel] - ref.trf_y[sel]).max() < 2.5e-4
        for name in ("trf_psi", "own_psi"):
            dpsi = np.abs(getattr(env, name)[sel] - getattr(ref, name)[sel])
            assert np.minimum(dpsi, 360 - dpsi).max() < 6e-5, name
        assert np.array_equal(env.trf_v[sel], ref.trf_v[sel])

    ref.reset()
    env = GpuEngine(g, E, N, dtype=torch.float32, auto_reset=True, seed=5)
    env.reset()
    close_to_oracle(env, np.ones(E, bool))
    # step both with the same actions until plenty of envs have been reset inside the step
    rng = np.random.default_rng(3)
    checked = 0
    twin = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=torch.float32, auto_reset=True, seed=5)
    for _ in range(40):
        a = rng.uniform(-1, 1, E).astype(np.float32).astype(np.float64)
        _, _, d1, _, _ = ref.step(a)
        _, _, d2, _, _ = env.step(a)
        both = (d1 != 0) & (d2 != 0) & (env.episode == ref.episode)
        if both.any():
            checked += int(both.sum())
            close_to_oracle(env, both)
        fresh = torch.as_tensor(d2 != 0, device="cuda:0")
        if fresh.any():
            # the same episodes drawn by reset_kernel on a twin env: bit for bit what the step produced
            twin.episode.copy_(env.v.episode)
            twin._launch_reset(fresh.to(torch.uint8), do_init=1)
            for name in ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi", "trf_v"):
                assert torch.equal(getattr(twin, name)[fresh], getattr(env.v, name)[fresh]), name
            assert bits_equal(twin.outputs["obs"][fresh], env.v.outputs["obs"][fresh])
    assert checked > 50


def _dtype_and_config(g, dtype_name, N):
    """"float64fast" = the float64 build's FAST formulation (ACAS2DConfig.fast_math)."""
    fast = dtype_name == "float64fast"
    return getattr(torch, "float64" if fast else dtype_name), g.ACAS2DConfig(n_traffic=N, fast_math=fast)


@pytest.mark.parametrize("dtype_name,N,E,T", (("float32", 8, 4096, 160), ("float64", 8, 1024, 120), ("float32", 64, 512, 40),
                                               ("float32", 3, 2048, 60), ("float32", 1, 640, 450), ("float64", 3, 333, 50),
                                               ("float64fast", 8, 1024, 120), ("float64fast", 3, 333, 50)))
def test_rollout_equals_sequential_steps(g, dtype_name, N, E, T):
    """acas2d_rollout_* (T steps fused in one launch, state in registers) == T x acas2d_step_*,
    bit for bit: observations, rewards, masks, side channels and the final state."""
    dtype, cfg = _dtype_and_config(g, dtype_name, N)
    dev = "cuda:0"
    a = g.ACAS2DVecEnv(E, N, device=dev, dtype=dtype, seed=77, env_offset=5, config=cfg)
    b = g.ACAS2DVecEnv(E, N, device=dev, dtype=dtype, seed=77, env_offset=5, config=cfg)
    a.reset()
    b.reset()
    gen = torch.Generator(device=dev).manual_seed(11)
    actions = torch.rand(T, E, generator=gen, device=dev, dtype=dtype) * 2 - 1
    out = a.rollout(actions, keep_terminal_obs=True)
    torch.cuda.synchronize()
    dones = 0
    for t in range(T):
        obs, rew, done, infos = b.step(actions[t])
        assert bits_equal(out["obs"][t], obs), t
        assert bits_equal(out["reward"][t], rew) and bits_equal(out["done"][t], done)
        assert bits_equal(out["outcome"][t], infos.outcome)
        d = done
        if bool(d.any()):
            dones += int(d.sum())
            assert bits_equal(out["episode_return"][t][d], infos.episode_return[d])
            assert bits_equal(out["episode_steps"][t][d], infos.episode_steps[d])
            assert bits_equal(out["terminal_observation"][t][d], infos.terminal_observation[d])
    assert dones > 0
    for name in ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi", "trf_v",
                 "steps", "total_reward", "episode"):
        assert bits_equal(getattr(a, name), getattr(b, name)), name
    assert bits_equal(a.outputs["obs"], b.outputs["obs"])          # the latest observation, either way
    # and a second rollout continues from the state the first one left, reusing the buffers
    actions2 = torch.rand(T, E, generator=gen, device=dev, dtype=dtype) * 2 - 1
    out = a.rollout(actions2, out=out)
    for t in range(T):
        obs, rew, done, _ = b.step(actions2[t])
        assert bits_equal(out["obs"][t], obs) and bits_equal(out["done"][t], done)
    with pytest.raises(RuntimeError, match="packed work shape"):
        g.ACAS2DVecEnv(16, 5, device=dev, dtype=dtype).rollout(torch.zeros(2, 16, device=dev, dtype=dtype))


@pytest.mark.parametrize("dtype_name", ("float32", "float64", "float64fast"))
def test_rollout_wraps_and_stores_injected_headings_like_steps(g, dtype_name):
    """aircraft.py:22 wraps a heading (psi % 360) on every step.  Headings injected outside [0, 360)
    are wrapped by the first step and written back; the fused rollout keeps the traffic's sin / cos
    in registers across steps and must still leave the same (wrapped) state behind."""
    E, N, T = 256, 8, 7
    dtype, cfg = _dtype_and_config(g, dtype_name, N)
    own, trf, goal = H.parity_reset_states(cfg, 99, 0, E)
    own[:, 2] += 360.0                       # 357..363 -> 717..723: inside the float32 window (-360, 720)
    own[:, 2] = np.minimum(own[:, 2], 719.0)
    trf[:, :, 2] += 360.0 * (np.arange(N)[None, :] % 2)
    a = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=3, config=cfg)
    b = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=3, config=cfg)
    for v in (a, b):
        v.set_state(own, trf, goal, np.zeros(E, np.int32), observe=False)
    gen = torch.Generator(device="cuda:0").manual_seed(2)
    actions = torch.rand(T, E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
    out = a.rollout(actions)
    for t in range(T):
        obs, rew, done, _ = b.step(actions[t])
        assert bits_equal(out["obs"][t], obs) and bits_equal(out["reward"][t], rew), t
    for name in ("own_psi", "trf_psi", "trf_x", "trf_y", "own_x", "own_y", "steps", "episode"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert float(a.trf_psi.max()) < 360.0 and float(a.own_psi.max()) < 360.0


@pytest.mark.parametrize("dtype_name,N,shapes", (
    ("float32", 8, ("4,2", "8,1", "2,4", "generic,4", "generic,1")),
    ("float32", 64, ("4,16", "8,8", "2,32", "generic,16", "generic,64")),
    ("float64", 8, ("2,4", "4,2", "generic,4")),
    ("float64fast", 8, ("2,4", "4,2", "generic,4"))))
def test_results_do_not_depend_on_the_work_shape(g, dtype_name, N, shapes):
    """How the traffic of an env is spread over lanes (ACAS2D_SHAPE, a tuning knob) must not change a
    single bit: both builds compile with -ffp-contract=off, every variant runs the same IEEE
    operations per aircraft (packed float2 math included), and the reset RNG is keyed per entity."""
    dtype, cfg = _dtype_and_config(g, dtype_name, N)
    E, T = 1536, 120
    gen = torch.Generator(device="cuda:0").manual_seed(4)
    actions = torch.rand(T, E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
    ref = None
    try:
        for sh in shapes:
            os.environ["ACAS2D_SHAPE"] = sh
            v = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=8, config=cfg)
            assert g.native.launch_geometry(E, N, 4 if dtype == torch.float32 else 8)["lanes_per_env"] == int(sh.split(",")[1])
            got = [v.reset().clone()]
            dones = 0
            for t in range(T):
                obs, rew, done, infos = v.step(actions[t])
                got += [obs.clone(), rew.clone(), done.clone(), infos.outcome.clone()]
                dones += int(done.sum())
            got += [v.trf_x.clone(), v.trf_psi.clone(), v.own_psi.clone(), v.total_reward.clone(), v.episode.clone()]
            assert dones > 0
            if ref is None:
                ref = got
            else:
                for k, (a, b) in enumerate(zip(ref, got)):
                    assert bits_equal(a, b), (sh, k)
    finally:
        os.environ.pop("ACAS2D_SHAPE", None)


@pytest.mark.parametrize("dtype_name,N,E,T", (("float32", 8, 4096 + 17, 200), ("float32", 64, 640, 40), ("float32", 3, 2048, 80),
                                               ("float32", 5, 1000, 80), ("float64", 8, 1024, 120), ("float64fast", 8, 1024, 120),
                                               ("float64", 7, 500, 60)))
def test_double_buffered_step_equals_in_place(g, dtype_name, N, E, T):
    """acas2d_step_* with a state_out (read generation g, write generation 1 - g; the VecEnv default) against the
    same steps in place: every observation, reward, mask, side channel and the final state bit for bit -- packed
    and generic work shapes (N = 5, 7), a last wave with padding lanes, resets in both."""
    dtype, cfg = _dtype_and_config(g, dtype_name, N)
    bits = torch.int32 if dtype == torch.float32 else torch.int64

    def same(x, y):                          # bit for bit: a NaN d_cpa (exact parallel flight) equals itself
        return torch.equal(x.view(bits), y.view(bits)) if x.is_floating_point() else torch.equal(x, y)

    gen = torch.Generator(device="cuda:0").manual_seed(11)
    actions = torch.rand(T, E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
    a = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=21, env_offset=3, config=cfg, double_buffer=True)
    b = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=21, env_offset=3, config=cfg, double_buffer=False)
    assert a.double_buffer and not b.double_buffer and a._gen["own_x"].shape[0] == 2 and b._gen["own_x"].shape[0] == 1
    assert same(a.reset(), b.reset())
    dones = 0
    for t in range(T):
        oa, ra, da, ia = a.step(actions[t])
        ob, rb, db, ib = b.step(actions[t])
        assert a.generation == (t + 1) % 2 and b.generation == 0
        assert same(oa, ob) and same(ra, rb) and torch.equal(da, db), t
        for k in ("outcome", "terminal_observation", "episode_return", "episode_steps"):
            assert same(a.outputs[k], b.outputs[k]), (t, k)
        dones += int(da.sum())
        if t in (0, 1, T // 2, T - 1):
            for name in ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi", "trf_v", "steps",
                         "total_reward", "episode"):
                assert same(getattr(a, name), getattr(b, name)), (t, name)
    assert dones > 20
    # a fused rollout and a masked reset act on the LIVE generation, whichever it is
    a.step(actions[0]); b.step(actions[0])
    assert a.generation == (T + 1) % 2
    if N not in (5, 7):                      # (the fused rollout needs a packed work shape)
        ra, rb = a.rollout(actions[:8]), b.rollout(actions[:8])
        assert same(ra["obs"], rb["obs"]) and same(ra["reward"], rb["reward"]) and same(a.trf_x, b.trf_x)
    mask = (torch.arange(E, device="cuda:0") % 3 == 0)
    assert same(a.reset_masked(mask), b.reset_masked(mask)) and same(a.own_psi, b.own_psi)
    oa, _, _, _ = a.step(actions[1]); ob, _, _, _ = b.step(actions[1])
    assert same(oa, ob) and torch.equal(a.steps, b.steps)


@pytest.mark.parametrize("N,E,T,db", ((8, 4096, 150, True), (8, 5120, 150, False), (64, 640, 40, False),
                                      (3, 4096, 80, True), (1, 2048, 560, True)))
def test_consecutive_layout_kernel_equals_the_general_kernel(g, monkeypatch, N, E, T, db):
    """float32 state as ACAS2DVecEnv allocates it (consecutive rows: include/acas2d.h) takes the step kernel whose
    loads all go through preloaded base pointers; ACAS2D_NO_ARENA (read per launch) sends the same state through the
    general kernel.  Every observation, reward, mask, side channel and the final state bit for bit, resets, both
    store policies.  (The kernel assumes whole multiples of eight workgroups; other sizes take the general kernel.)"""
    dtype, cfg = _dtype_and_config(g, "float32", N)
    same = lambda x, y: torch.equal(x.view(torch.int32), y.view(torch.int32)) if x.is_floating_point() else torch.equal(x, y)  # noqa: E731
    gen = torch.Generator(device="cuda:0").manual_seed(12)
    actions = torch.rand(T, E, generator=gen, device="cuda:0") * 2 - 1
    a = g.ACAS2DVecEnv(E, N, device="cuda:0", seed=5, env_offset=9, config=cfg, double_buffer=db)
    b = g.ACAS2DVecEnv(E, N, device="cuda:0", seed=5, env_offset=9, config=cfg, double_buffer=db)
    assert a.consecutive_layout and a.own_y.data_ptr() == a.own_x.data_ptr() + 4 * E
    assert not g.ACAS2DVecEnv(E + 17, N, device="cuda:0").consecutive_layout          # not whole workgroups
    assert not g.ACAS2DVecEnv(64, N, device="cuda:0", dtype=torch.float64).consecutive_layout
    assert not g.ACAS2DVecEnv(64, N, device="cuda:0", auto_reset=False).consecutive_layout
    assert same(a.reset(), b.reset())
    dones = 0
    for t in range(T):
        oa, ra, da, _ = a.step(actions[t])
        monkeypatch.setenv("ACAS2D_NO_ARENA", "1")
        assert not b.consecutive_layout
        ob, rb, db_, _ = b.step(actions[t])
        monkeypatch.delenv("ACAS2D_NO_ARENA")
        assert same(oa, ob) and same(ra, rb) and torch.equal(da, db_), t
        for k in ("outcome", "terminal_observation", "episode_return", "episode_steps"):
            assert same(a.outputs[k], b.outputs[k]), (t, k)
        dones += int(da.sum())
    for name in ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi", "trf_v", "steps",
                 "total_reward", "episode"):
        assert same(getattr(a, name), getattr(b, name)), name
    assert dones > (20 if N > 1 else 0)          # (one traffic aircraft: head-on, the first collisions near step 500)


def test_double_buffered_steps_in_a_replayed_graph(g):
    """A hipGraph holds the state generation it was captured at: an EVEN number of captured steps leaves the live
    generation where the capture found it, and align_generation() puts it back there after an odd number of
    other steps -- the replayed run equals the same steps launched one by one."""
    E, N, CH = 2048, 8, 6
    gen = torch.Generator(device="cuda:0").manual_seed(2)
    actions = torch.rand(CH, E, generator=gen, device="cuda:0") * 2 - 1
    a = g.ACAS2DVecEnv(E, N, device="cuda:0", seed=4, double_buffer=True)
    b = g.ACAS2DVecEnv(E, N, device="cuda:0", seed=4, double_buffer=True)
    a.reset(); b.reset()
    for t in range(3):                      # warm-up before the capture (an odd number: generation 1 is live)
        a.step_from(actions[t]); b.step_from(actions[t])
    torch.cuda.synchronize()
    g0 = a.generation
    assert g0 == 1
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for t in range(CH):
            a.step_from(actions[t])
    assert a.generation == g0               # an even number of steps was captured; nothing ran
    for rep in range(5):
        if rep == 2:                        # an odd number of plain steps in between: the live generation moves on
            a.step_from(actions[0]); b.step_from(actions[0])
            assert a.generation != g0
        a.align_generation(g0)
        assert a.generation == g0
        graph.replay()
        for t in range(CH):
            b.step_from(actions[t])
        torch.cuda.synchronize()
        assert bits_equal(a.outputs["obs"], b.outputs["obs"]) and bits_equal(a.outputs["reward"], b.outputs["reward"])
        for name in ("own_x", "own_psi", "trf_y", "steps", "total_reward", "episode"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (rep, name)
    with pytest.raises(RuntimeError):
        g.ACAS2DVecEnv(64, 1, device="cuda:0", double_buffer=False).align_generation(1)
    # the default is the measured policy: on for launches of one generation of wavefronts (the headline size), off beyond
    assert g.ACAS2DVecEnv(65536, 8, device="cuda:0").double_buffer and g.ACAS2DVecEnv(65536, 8, device="cuda:0", dtype=torch.float64).double_buffer
    assert not g.ACAS2DVecEnv(131072, 8, device="cuda:0").double_buffer and not g.ACAS2DVecEnv(16384, 64, device="cuda:0").double_buffer
    assert not g.ACAS2DVecEnv(64, 1, device="cuda:0", auto_reset=False).double_buffer
    with pytest.raises(ValueError):
        g.ACAS2DVecEnv(64, 1, device="cuda:0", auto_reset=False, double_buffer=True)


def _first_episode(out, E):
    """outcome / game.steps / return of each env's FIRST finished episode in a rollout dict."""
    done = out["done"].cpu().numpy()
    assert done.any(0).all(), "every env must finish at least once"
    t0 = done.argmax(0)
    e = np.arange(E)
    return (out["outcome"].cpu().numpy()[t0, e], out["episode_steps"].cpu().numpy()[t0, e],
            out["episode_return"].cpu().numpy()[t0, e].astype(np.float64))


def test_fused_policy_rollout_reproduces_the_reference_policy_evaluation(g):
    """acas2d_rollout_policy_f64: testing_main.py's whole loop (policy.predict + env.step, 1001 steps,
    100 episodes) in ONE launch with the reference's trained SB3 actor evaluated inside the kernel.
    It must score what the reference recorded for that policy (mean return 1210.069219, mean length
    704.35, 100/100 goals) -- the in-kernel float32 MLP differs from torch's by summation order
    only (~1e-7 per action), so the table is held to 1e-4 relative instead of every digit."""
    pol = g.load_sb3_policy(os.path.join(H.GOLDEN, "ref_policy_best_model.npz"), device="cuda:0")
    own, trf, goal = H.parity_reset_states(g.ACAS2DConfig(), 13, 0, 100)
    v = g.ACAS2DVecEnv(100, 1, device="cuda:0", dtype=torch.float64, auto_reset=True)
    obs0 = v.set_state(own, trf, goal, np.zeros(100, np.int32), observe=True).clone()
    out = v.rollout_policy(pol, 1001)
    a0 = pol.predict(obs0).reshape(-1).to(torch.float64)
    assert float((out["actions"][0] - a0).abs().max()) < 2e-6
    oc, steps, ret = _first_episode(out, 100)
    assert (oc == 1).all()
    H.assert_matches_reference_policy_eval(ret, steps, 2.0 * (steps - 1), tol=1e-4)
    # a loose speed floor: the in-kernel MLP once regressed 7x (its weight loads were hoisted out of the
    # step loop and spilled) without a single wrong bit; 65 536 envs x 100 steps take ~1.5 ms
    big = g.ACAS2DVecEnv(65536, 1, device="cuda:0", dtype=torch.float32, seed=1)
    big.reset()
    o = big.rollout_policy(pol, 100)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    big.rollout_policy(pol, 100, out=o)
    t1.record()
    torch.cuda.synchronize()
    assert t0.elapsed_time(t1) < 6.0, t0.elapsed_time(t1)
    # the convenience wrapper, and the step-by-step evaluation it replaces
    fused = g.evaluate_policy_fused(pol, own, trf, goal)
    assert fused["unfinished"] == 0 and np.array_equal(fused["outcome"], oc) and np.array_equal(fused["steps"], steps)
    ev = g.ACAS2DVecEnv(100, 1, device="cuda:0", dtype=torch.float64, auto_reset=False)
    ev.set_state(own, trf, goal, np.zeros(100, np.int32))
    slow = g.evaluate_policy(ev, pol)
    assert np.array_equal(slow["outcome"], fused["outcome"]) and np.array_equal(slow["steps"], fused["steps"])
    assert np.abs(slow["total_reward"] - fused["total_reward"]).max() < 1e-3
    np.testing.assert_allclose(slow["path_length"], fused["path_length"])


@pytest.mark.parametrize("N,E,T", ((1, 4096, 800), (3, 2048, 120), (8, 2048, 60)))
def test_fused_policy_rollout_equals_policy_then_step(g, N, E, T):
    """float32: the fused launch against torch's policy.predict() + step() per step on a twin env.
    Same env arithmetic (bit-identical given the same actions); the two MLP evaluations differ by
    rounding, so actions are compared to 1e-5 and the trajectories to a tolerance that allows that
    difference to integrate -- and, where an env's actions happened to agree bit for bit all the
    way, exactly."""
    dev = "cuda:0"
    torch.manual_seed(5)
    if N == 1:                                          # the reference's trained policy: reaches the goal
        pol = g.load_sb3_policy(os.path.join(H.GOLDEN, "ref_policy_best_model.npz"), device=dev)
    else:
        pol = g.ActorCritic(5 + 3 * N).to(dev)
        with torch.no_grad():                           # a policy that actually steers (the SB3 init is ~0)
            pol.action_net.weight.mul_(60.0)
    a = g.ACAS2DVecEnv(E, N, device=dev, dtype=torch.float32, seed=21)
    b = g.ACAS2DVecEnv(E, N, device=dev, dtype=torch.float32, seed=21)
    a.reset()
    obs = b.reset().clone()
    out = a.rollout_policy(pol, T)
    same = torch.ones(E, dtype=torch.bool, device=dev)
    worst_a = 0.0
    for t in range(T):
        act = pol.predict(obs).reshape(-1)
        worst_a = max(worst_a, float((out["actions"][t] - act).abs().max()))
        same &= out["actions"][t] == act
        obs, rew, done, infos = b.step(out["actions"][t])          # feed the fused run's own actions
        assert bits_equal(out["obs"][t], obs) and bits_equal(out["reward"][t], rew), t
        assert torch.equal(out["done"][t], done) and torch.equal(out["outcome"][t], infos.outcome), t
        obs = obs.clone()
    assert worst_a < 1e-5, worst_a
    assert float(out["actions"].abs().max()) > 0.2 and int(out["done"].sum()) > 0
    for name in ("own_x", "own_y", "own_psi", "trf_x", "trf_y", "steps", "total_reward", "episode"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert bits_equal(a.outputs["obs"], b.outputs["obs"])
    with pytest.raises(RuntimeError, match="thread-per-env"):
        g.ACAS2DVecEnv(64, 16, device=dev).rollout_policy(g.ActorCritic(53).to(dev), 2)


def test_lazy_infos_and_vecenv_surface(g):
    E, N = 256, 64                          # N = 64: episodes last ~8 steps -> many dones
    env = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=torch.float32, seed=3)
    obs = env.reset()
    assert obs.shape == (E, 5 + 3 * N) and env.observation_space.shape == (5 + 3 * N,)
    assert env.num_envs == E and env.env_is_wrapped(None) == [False] * E
    seen = 0
    for _ in range(10):
        obs, rew, done, infos = env.step(torch.zeros(E, 1, device="cuda:0"))
        assert obs.shape == (E, 5 + 3 * N) and rew.shape == (E,) and done.dtype == torch.bool
        assert len(infos) == E
        dn = done.cpu().numpy()
        for i in np.nonzero(dn)[0][:5]:
            info = infos[int(i)]
            assert set(info) == {"outcome", "episode", "terminal_observation"}
            assert info["episode"]["l"] == info["episode"]["steps"] - 1 >= 1
            assert info["terminal_observation"].shape == (5 + 3 * N,)
            assert info["outcome"] in (1, 2, 3)
            seen += 1
        for i in np.nonzero(~dn)[0][:3]:
            assert infos[int(i)] == {}
    assert seen > 0
    sd = env.state_dict()
    env.step(torch.zeros(E, device="cuda:0"))
    env.load_state_dict(sd)
    assert torch.equal(env.own_x, sd["own_x"])
    with pytest.raises(ValueError):
        env.step(torch.zeros(E + 1, device="cuda:0"))


# ---- BASELINE.json full sizes: oracle spot check + size-independent properties -------------------
@pytest.mark.parametrize("E,N,T", ((65536, 8, 12), (131072, 8, 6), (65536, 64, 6), (4096, 3, 40)))
def test_full_size_f64_vs_oracle(g, O, E, N, T):
    ref = O.OracleEnvs(E, N, seed=13, auto_reset=True)
    env = GpuEngine(g, E, N, auto_reset=True, seed=13)
    ref.reset()
    env.reset()
    rng = np.random.default_rng(2)
    for _ in range(T):
        a = rng.uniform(-1, 1, E)
        o1, r1, d1, oc1, _ = ref.step(a)
        o2, r2, d2, oc2, _ = env.step(a)
        assert np.array_equal(d1, d2) and np.array_equal(oc1, oc2)
        assert np.abs(o2 - o1).max() < 1e-9 and np.abs(r2 - r1).max() < 1e-9
    assert np.array_equal(env.steps, ref.steps) and np.array_equal(env.episode, ref.episode)


@pytest.mark.parametrize("E,N,T", ((4096, 3, 24), (65536, 8, 5), (65536, 64, 3)))
def test_full_size_f32_vs_f64_oracle(g, O, E, N, T):
    """The headline dtype at the three single-GPU BASELINE sizes against the float64 oracle: T steps
    WITH auto-reset, each from the oracle trajectory's state rounded to float32 (so that both sides start every step
    from the identical state and the comparison is the step's, not the accumulated drift's).  done / outcome masks equal
    outside a 1e-3 px band around the thresholds; observations, rewards and positions of the envs that go on within the
    tolerances of test_f32_statistical_single_step_vs_f64_oracle; for the envs that finish: the terminal observation,
    the episode return and length, and the freshly drawn episode (the float32 build draws in float32: positions
    2.5e-4, headings 6e-5 from the float64 draw) with its first observation.  Mismatches are counted and printed."""
    f32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
    ref = O.OracleEnvs(E, N, seed=13, auto_reset=True)
    ref.reset()
    rng = np.random.default_rng(7)
    for _ in range(3 if N == 64 else int(rng.integers(15, 30))):        # mid-episode states (N = 64: episodes last ~8 steps)
        ref.step(rng.uniform(-1, 1, E))
    chk = O.OracleEnvs(E, N, seed=13, auto_reset=True)
    env = GpuEngine(g, E, N, dtype=torch.float32, auto_reset=True, seed=13)
    cfgc = O.default_config()
    col = np.arange(5 + 3 * N)
    cpa, vcl = (col >= 5) & ((col - 5) % 3 == 1), (col >= 5) & ((col - 5) % 3 == 2)
    tot = dict(steps=0, mask_mismatch=0, in_band=0, finished=0, e_obs=0.0, e_cpa=0.0, e_rew=0.0, e_term=0.0, e_fresh_obs=0.0)
    for t in range(T):
        own = f32(np.stack([ref.own_x, ref.own_y, ref.own_psi, ref.own_v], 1))
        trf = f32(np.stack([ref.trf_x, ref.trf_y, ref.trf_psi, ref.trf_v], -1))
        steps, act = ref.steps.copy(), f32(rng.uniform(-1, 1, E))
        chk.set_state(own, trf, None, steps)
        chk.episode[:] = ref.episode
        env.set_state(own, trf, None, steps)
        env.v.episode.copy_(torch.as_tensor(ref.episode.view(np.int32), device="cuda:0"))
        o, r, d, oc, _ = chk.step(act)
        obs, rew, done, outcome, _ = env.step(act)
        d = d.astype(bool)
        # ---- masks: bit-exact outside the band
        ok = ~grazing(np.where(d[:, None], chk.term_obs, o), N, cfgc, 1e-3)
        mism = (done.astype(bool) != d) | (outcome != oc)
        tot["mask_mismatch"] += int((mism & ok).sum()); tot["in_band"] += int((~ok).sum()); tot["steps"] += E
        assert not (mism & ok).any(), (t, int((mism & ok).sum()))
        assert ok.mean() > 0.999
        same = ok & ~mism
        go, fin = same & ~d, same & d
        tot["finished"] += int(fin.sum())
        # ---- envs that go on: the step itself
        v12x = (chk.own_v * np.cos(np.deg2rad(chk.own_psi)))[:, None] - chk.trf_v * np.cos(np.deg2rad(chk.trf_psi))
        v12y = (chk.own_v * np.sin(np.deg2rad(chk.own_psi)))[:, None] - chk.trf_v * np.sin(np.deg2rad(chk.trf_psi))
        well = (np.abs(v12x) > 0.02) & (np.hypot(v12x, v12y) > 2.0)
        near = (o[:, 5::3] * cfgc.d_sep_max) < 16.0
        err = np.abs(obs - o)
        err[:, [1, 4]] = np.minimum(err[:, [1, 4]], 1.0 - err[:, [1, 4]])
        e_plain = err[:, ~(cpa | vcl)][go]
        e_vc, e_cpa = err[:, vcl][go][~near[go]], err[:, cpa][go][well[go]]
        tot["e_obs"] = max(tot["e_obs"], float(e_plain.max()), float(e_vc.max()))
        tot["e_cpa"] = max(tot["e_cpa"], float(e_cpa.max()))
        assert e_plain.max() < 1e-5 and e_vc.max() < 1e-5 and e_cpa.max() < 2e-5, (t, e_plain.max(), e_vc.max(), e_cpa.max())
        nt = go & well[:, 0] & ~near[:, 0]
        e_rew = np.abs(rew[nt] - r[nt])
        tot["e_rew"] = max(tot["e_rew"], float(e_rew.max()))
        # (1e-5 for 99.99 % of the env-steps, the worst below 5e-5: the reward amplifies the d_cpa error up to 39 x --
        #  see test_f32_statistical_single_step_vs_f64_oracle; a percentile needs the samples to carry it)
        assert e_rew.max() < 5e-5 and (e_rew.size < 50000 or np.quantile(e_rew, 0.9999) < 1e-5), (t, e_rew.max())
        assert max(np.abs(env.own_x - chk.own_x)[go].max(), np.abs(env.trf_x - chk.trf_x)[go].max(),
                   np.abs(env.trf_y - chk.trf_y)[go].max()) <= 1.3e-4
        assert np.array_equal(env.steps[go], chk.steps[go])
        # ---- envs that finish: side channels of the finished episode, then the fresh one
        if fin.any():
            te = np.abs(env.term_obs - chk.term_obs)
            te[:, [1, 4]] = np.minimum(te[:, [1, 4]], 1.0 - te[:, [1, 4]])
            tw = te[:, ~(cpa | vcl)][fin]
            tot["e_term"] = max(tot["e_term"], float(tw.max()))
            assert tw.max() < 1e-5
            assert np.array_equal(env.ep_steps[fin], chk.ep_steps[fin]) and np.array_equal(env.episode[fin], chk.episode[fin])
            assert np.abs(env.ep_return - chk.ep_return)[fin].max() <= 1.3e-4 + 1e-5          # one float32 ulp of the +-1000 bonus
            assert max(np.abs(env.trf_x - chk.trf_x)[fin].max(), np.abs(env.trf_y - chk.trf_y)[fin].max()) < 2.5e-4
            for name in ("trf_psi", "own_psi"):
                dpsi = np.abs(getattr(env, name) - getattr(chk, name))[fin]
                assert np.minimum(dpsi, 360 - dpsi).max() < 6e-5, name
            assert np.array_equal(env.steps[fin], chk.steps[fin]) and (env.steps[fin] == 1).all()
            fe = np.abs(obs - o)
            fe[:, [1, 4]] = np.minimum(fe[:, [1, 4]], 1.0 - fe[:, [1, 4]])
            ff = fe[:, ~(cpa | vcl)][fin]                        # (a fresh episode's d_cpa / closing speed: from states 2.5e-4 px apart)
            tot["e_fresh_obs"] = max(tot["e_fresh_obs"], float(ff.max()))
            assert ff.max() < 1e-5
        ref.step(act)
    print("f32 vs f64 oracle at %d x %d over %d steps: %d env-steps, %d finished; mask mismatches outside the 1e-3 band %d "
          "(%d env-steps inside it); max |obs| %.2e, d_cpa %.2e, reward %.2e, terminal obs %.2e, first obs of a fresh "
          "episode %.2e" % (E, N, T, tot["steps"], tot["finished"], tot["mask_mismatch"], tot["in_band"], tot["e_obs"],
                            tot["e_cpa"], tot["e_rew"], tot["e_term"], tot["e_fresh_obs"]))
    assert tot["finished"] > (40 if N == 3 else 300)


@pytest.mark.parametrize("dtype_name", ("float32", "float64"))
def test_full_size_properties(g, dtype_name):
    """65 536 envs x 8 traffic (headline config): invariants every step, bitwise determinism,
    and invariance to sharding (2 shards with env_offset == 1 shard)."""
    dtype = getattr(torch, dtype_name)
    E, N, T = 65536, 8, 40
    dev = "cuda:0"
    full = g.ACAS2DVecEnv(E, N, device=dev, dtype=dtype, seed=13)
    again = g.ACAS2DVecEnv(E, N, device=dev, dtype=dtype, seed=13)
    a_sh = g.ACAS2DVecEnv(40000, N, device=dev, dtype=dtype, seed=13, env_offset=0)
    b_sh = g.ACAS2DVecEnv(E - 40000, N, device=dev, dtype=dtype, seed=13, env_offset=40000)
    for e in (full, again, a_sh, b_sh):
        e.reset()
    gen = torch.Generator(device=dev).manual_seed(0)
    prev_steps = full.steps.clone()
    total_done = 0
    for _ in range(T):
        a = torch.rand(E, generator=gen, device=dev, dtype=dtype) * 2 - 1
        obs, rew, done, infos = full.step(a)
        o2, r2, d2, _ = again.step(a)
        oa, ra, da, _ = a_sh.step(a[:40000].contiguous())
        ob, rb, db, _ = b_sh.step(a[40000:].contiguous())
        assert bits_equal(obs, o2) and bits_equal(rew, r2) and torch.equal(done, d2)
        assert bits_equal(obs, torch.cat([oa, ob])) and bits_equal(rew, torch.cat([ra, rb]))
        assert torch.equal(done, torch.cat([da, db]))
        steps = full.steps
        # steps advance by one, or restart at 1 exactly where done
        assert torch.equal(torch.where(done, torch.ones_like(steps), prev_steps + 1), steps)
        assert torch.equal(done, infos.outcome > 0)
        want = steps.to(dtype) / torch.full_like(obs[:, 0], 1000)                    # true division
        assert torch.equal(obs[:, 0], want) if dtype == torch.float64 else bool(((obs[:, 0] - want).abs() <= 1.2e-7 * want).all())
        psi = full.own_psi
        assert bool(((psi >= 0) & (psi <= 360)).all())
        assert bool(torch.isfinite(obs[:, [0, 1, 2, 3, 4]]).all()) and bool(torch.isfinite(rew).all())
        assert bool((obs[:, 5::3] >= 0).all())                                  # distances
        # a finished env collected the terminal bonus that its outcome implies
        oc = infos.outcome
        assert bool((rew[oc == 2] < -900).all()) and bool((rew[oc == 1] > 900).all())
        assert bool((rew[~done].abs() <= 1.0 + 1e-6).all())
        total_done += int(done.sum())
        prev_steps = steps.clone()
    assert total_done > 0


def test_linearity_of_motion_at_one_million_envs(g):
    """Size-independent physics properties at 1 M envs: with action 0 every aircraft moves
    exactly v*dt = 2 px per step along its heading and headings are unchanged."""
    E, N = 1 << 20, 8
    env = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=torch.float64, seed=21, auto_reset=False)
    env.reset()
    x0, y0, psi0 = env.own_x.clone(), env.own_y.clone(), env.own_psi.clone()
    tx0, ty0, tpsi0 = env.trf_x.clone(), env.trf_y.clone(), env.trf_psi.clone()
    env.step(torch.zeros(E, dtype=torch.float64, device="cuda:0"))
    live = env.status == 0
    d_own = torch.hypot(env.own_x - x0, env.own_y - y0)
    d_trf = torch.hypot(env.trf_x - tx0, env.trf_y - ty0)
    assert float((d_own - 2.0).abs().max()) < 1e-9 and float((d_trf - 2.0).abs().max()) < 1e-9
    assert torch.equal(env.own_psi, psi0) and torch.equal(env.trf_psi, tpsi0)
    rad = torch.deg2rad(psi0)
    assert float((env.own_x - x0 - 2 * torch.cos(rad)).abs().max()) < 1e-9
    assert int(live.sum()) > 0


def test_plain_cpp_host_program_on_the_c_abi_matches_the_python_host(g):
    """examples/c_abi_example.cpp: hipMalloc + acas2d_reset_f32 + acas2d_step_f32 from C++, no Python
    and no torch in the process.  Same seed, same constant action 0, auto-reset: it must count the same
    finished episodes and end on the same observations as ACAS2DVecEnv."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_abi_example")
    subprocess.run(["make", "-C", os.path.join(root, "gym-acas2d_amd", "csrc"), "example"], check=True,
                   capture_output=True)
    E, N, T = 3000, 3, 450
    out = subprocess.run([exe, str(E), str(N), str(T)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(out[0]), int(out[1]), int(out[2])] == [E, N, T]
    v = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=torch.float32, seed=13)
    v.reset()
    zero = torch.zeros(E, device="cuda:0")
    finished, reward_sum = 0, 0.0
    for _ in range(T):
        obs, rew, done, _ = v.step(zero)
        finished += int(done.sum())
        reward_sum += float(rew.double().sum())
    w = (1 + torch.arange(obs.numel(), device="cuda:0") % 7).double()
    checksum = float((obs.reshape(-1).double() * w).sum())
    assert finished > 0 and int(out[3]) == finished
    assert abs(float(out[4]) - reward_sum) <= 1e-9 * abs(reward_sum)
    assert abs(float(out[5]) - checksum) <= 1e-9 * abs(checksum)


def test_c_abi_rejects_bad_arguments_on_gpu_box(g):
    import ctypes as C
    L = g.native.lib()
    assert L.acas2d_step_f32(None, None, None, None, 0, 0, 0, 16, 1, None) == -22
    assert b"NULL" in L.acas2d_last_error()
