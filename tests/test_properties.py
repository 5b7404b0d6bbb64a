"""Property tests (hypothesis) of one step on arbitrary states, CPU oracle: the invariants the
reference's state machine guarantees (game.py:194-314), independent of any golden vector."""
import numpy as np
from hypothesis import example, given, settings, strategies as st

coord = st.floats(-300, 1900, allow_nan=False, width=64)
heading = st.floats(0, 360, allow_nan=False, width=64)


# derandomize + database=None: the suite must not depend on a local .hypothesis/ example database;
# the @example is the over-time case round 1's bound missed (steps + 1 = 1002 -> tdf = -0.002).
@settings(max_examples=150, deadline=None, derandomize=True, database=None)
@example(own=(0.0, 176.0, 0.0), trf=[(0.0, 0.0, 0.0)], steps=1001, action=0.0)
@given(own=st.tuples(coord, coord, heading), trf=st.lists(st.tuples(coord, coord, heading), min_size=1, max_size=9),
       steps=st.integers(0, 1002), action=st.floats(-1, 1, allow_nan=False, width=64))
def test_step_invariants(oracle_mod, own, trf, steps, action):
    O = oracle_mod
    N = len(trf)
    env = O.OracleEnvs(1, N)
    env.set_state([[own[0], own[1], own[2], 200.0]], [[[t[0], t[1], t[2], 200.0] for t in trf]], None, [steps])
    x0, y0 = env.own_x[0], env.own_y[0]
    t0 = np.stack([env.trf_x[0], env.trf_y[0]], 1).copy()
    obs, r, done, oc, _ = env.step([action])
    c = env.cfg
    # counters and headings
    assert env.steps[0] == steps + 1 and obs[0, 0] == (steps + 1) / 1000
    assert 0.0 <= env.own_psi[0] <= 360.0 and np.all((env.trf_psi[0] >= 0) & (env.trf_psi[0] <= 360))
    assert abs(((env.own_psi[0] - own[2] + 180) % 360) - 180) <= c.acc_lat_limit / 200.0 + 1e-9     # aircraft.py:20-22
    # every aircraft moved exactly v * dt = 2 px
    assert abs(np.hypot(env.own_x[0] - x0, env.own_y[0] - y0) - 2.0) < 1e-9
    assert np.all(np.abs(np.hypot(env.trf_x[0] - t0[:, 0], env.trf_y[0] - t0[:, 1]) - 2.0) < 1e-9)
    # termination: timeout > collision > goal (game.py:294-314), strict thresholds
    d = np.hypot(env.own_x[0] - env.trf_x[0], env.own_y[0] - env.trf_y[0])
    d_goal = np.hypot(env.own_x[0] - c.goal_x, env.own_y[0] - c.goal_y)
    near = np.abs(d - 96).min() < 1e-6 or abs(d_goal - 144) < 1e-6
    want = 3 if steps + 1 > 1000 else (2 if (d < 96).any() else (1 if d_goal < 144 else 0))
    if not near:
        assert oc[0] == want and bool(done[0]) == (want != 0) and env.status[0] == want
    # observation layout (game.py:199-210): distances non-negative, goal entries consistent
    assert np.all(obs[0, 5::3] >= 0) and abs(obs[0, 3] * c.d_goal_max - d_goal) < 1e-9
    assert abs(obs[0, 1] * 360 - env.own_psi[0]) < 1e-12
    np.testing.assert_allclose(obs[0, 5::3] * c.d_sep_max, d, atol=1e-9)
    # reward: r_step in [0, 1] times tdf = 1 - steps/MAX_STEPS with the incremented counter (game.py:262-263;
    # negative once the counter is past MAX_STEPS: -0.001 on the timeout step itself, lower for a state
    # injected beyond it), plus the terminal bonuses
    if np.isfinite(r[0]) and not near:
        shaped = r[0] - (-1000 if (d < 96).any() else 0) - (1000 if d_goal < 144 else 0)
        tdf = 1.0 - (steps + 1) / c.max_steps
        assert min(0.0, tdf) - 1e-9 <= shaped <= max(0.0, tdf) + 1e-9
