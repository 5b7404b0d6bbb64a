"""Shared test machinery: golden-vector loading and engine-agnostic replay loops.

An "engine" here is anything with the OracleEnvs interface (oracle/oracle.py): numpy float64
views  own_x, own_y, own_psi, trf_x, trf_y, steps, total_reward  plus
set_state(own, trf, goal, steps), observe(), step(actions) -> (obs, reward, done, outcome, n).
The CPU oracle implements it natively; tests/test_gpu_parity.py wraps the HIP path in the same
interface so that both are checked by the same code against the same fixtures.
"""
import os
import random

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def parity_reset_states(cfg, seed, skip, count):
    """Episodes number skip .. skip+count-1 of the reference's MT19937 stream after seed."""
    import gym_acas2d_amd as g
    rng = random.Random(seed)
    if skip:
        g.reset_parity.draw_episodes(cfg, skip, rng)
    return g.reset_parity.draw_episodes(cfg, count, rng)


def rollout_as_batch(fx):
    """Reference rollout fixture (one env, reset on done) -> one env per episode:
    acts [T, n_ep], valid [T, n_ep], rows [T] list of fixture row indices ordered by episode."""
    n_ep = len(fx["ep_own"])
    T = int(np.bincount(fx["ep"]).max())
    acts = np.zeros((T, n_ep))
    valid = np.zeros((T, n_ep), bool)
    acts[fx["k"], fx["ep"]] = fx["action"]
    valid[fx["k"], fx["ep"]] = True
    rows = []
    for k in range(T):
        r = np.nonzero(fx["k"] == k)[0]
        rows.append(r[np.argsort(fx["ep"][r])])
    return acts, valid, rows


def replay_rollout(engine, fx):
    """Step `engine` (already holding the fixture's initial states, observe() done) through the
    fixture's action streams.  Returns worst absolute errors and mask mismatch counts."""
    acts, valid, rows = rollout_as_batch(fx)
    res = dict(obs=0.0, reward=0.0, pos=0.0, psi=0.0, total_reward=0.0, done_mismatch=0,
               outcome_mismatch=0, steps_mismatch=0, n=0)
    for k in range(acts.shape[0]):
        obs, reward, done, outcome, _ = engine.step(acts[k])
        sel = np.nonzero(valid[k])[0]
        r = rows[k]
        assert np.array_equal(fx["ep"][r], sel)
        res["obs"] = max(res["obs"], float(np.nanmax(np.abs(obs[sel] - fx["obs"][r]))))
        assert np.array_equal(np.isnan(obs[sel]), np.isnan(fx["obs"][r]))
        res["reward"] = max(res["reward"], float(np.abs(reward[sel] - fx["reward"][r]).max()))
        own = np.stack([engine.own_x, engine.own_y], 1)[sel]
        res["pos"] = max(res["pos"], float(np.abs(own - fx["own"][r][:, :2]).max()),
                         float(np.abs(np.stack([engine.trf_x, engine.trf_y], -1)[sel] - fx["trf_xy"][r]).max()))
        res["psi"] = max(res["psi"], float(np.abs(engine.own_psi[sel] - fx["own"][r][:, 2]).max()))
        res["total_reward"] = max(res["total_reward"],
                                  float(np.abs(engine.total_reward[sel] - fx["total_reward"][r]).max()))
        res["done_mismatch"] += int((done[sel].astype(bool) != fx["done"][r].astype(bool)).sum())
        res["outcome_mismatch"] += int((outcome[sel] != fx["outcome"][r]).sum())
        res["steps_mismatch"] += int((engine.steps[sel] != fx["steps"][r]).sum())
        res["n"] += len(sel)
    return res


def replay_baseline(engine, digest, own, trf):
    """baseline_main.simulate() (baseline_main.py:32-61): constant action 0 until done, for the
    100 episodes of the reference's CSV as one batch of 100 envs.  `engine` must already hold
    the initial states (set_state + observe).  Returns what the CSV digest pins."""
    E = len(own)
    stride = int(digest["stride"])
    outcome = np.zeros(E, np.uint8)
    steps = np.zeros(E, np.int32)
    ret = np.zeros(E)
    last = np.zeros((E, 2))
    sub = np.full_like(digest["own_sub"], np.nan)
    tsub = np.full_like(digest["trf_sub"], np.nan)
    sub[:, 0] = own[:, :2]
    tsub[:, 0] = trf[:, 0, :2]
    first2 = np.zeros((E, 2, 2))
    first2[:, 0] = own[:, :2]
    prev_t = trf[:, 0, :2].copy()
    active = np.ones(E, bool)
    zeros = np.zeros(E)
    for k in range(1, 1001):
        _, _, done, oc, _ = engine.step(zeros)
        pos = np.stack([engine.own_x, engine.own_y], 1)
        if k == 1:
            first2[:, 1] = pos
        fin = active & (done != 0)
        if k % stride == 0:
            # Path[k] = player after k steps; Traffic Paths[k] = traffic BEFORE step k's move
            # (game.py:231-233 logs before :243-245 moves)
            sub[active, k // stride] = pos[active]
            tsub[active, k // stride] = prev_t[active]
        prev_t = np.stack([engine.trf_x[:, 0], engine.trf_y[:, 0]], 1)
        outcome[fin] = oc[fin]
        steps[fin] = engine.steps[fin]
        ret[fin] = engine.total_reward[fin]
        last[fin] = pos[fin]
        active &= ~fin
        if not active.any():
            break
    return dict(outcome=outcome, steps=steps, total_reward=ret, own_last=last, own_sub=sub,
                trf_sub=tsub, own_first2=first2, unfinished=int(active.sum()))


# notebooks/simulation_ACAS2D_PPO_1048576_11_100.ipynb cell 4: `simulation.describe()` of the
# reference's own 100-episode deterministic evaluation of best_model_1048576_11 (testing_main.py)
REF_POLICY_EVAL = {
    "Total Reward": dict(mean=1210.069219, std=69.336537, min=1099.788122, q25=1157.389870,
                         q50=1200.470752, q75=1245.515314, max=1370.264607),
    "Time Steps": dict(mean=704.35, std=73.826082, min=634, q25=656, q50=681, q75=721, max=927),
    "Path Length": dict(mean=1406.70, std=147.652164, min=1266, q25=1310, q50=1360, q75=1440, max=1852),
}


def describe(v):
    v = np.asarray(v, np.float64)
    return dict(mean=v.mean(), std=v.std(ddof=1), min=v.min(), q25=np.percentile(v, 25),
                q50=np.percentile(v, 50), q75=np.percentile(v, 75), max=v.max())


def assert_matches_reference_policy_eval(total_reward, steps, path_length, tol=2e-6):
    """Every printed digit of the reference's table (6 decimals) must be reproduced."""
    got = {"Total Reward": describe(total_reward), "Time Steps": describe(steps),
           "Path Length": describe(path_length)}
    for col, want in REF_POLICY_EVAL.items():
        for k, w in want.items():
            assert abs(got[col][k] - w) <= tol * max(1.0, abs(w)), (col, k, got[col][k], w)
