"""Episode records in the reference's CSV layout.

`baseline_main.simulate()` (baseline_main.py:32-74) and `testing_main.simulate()`
(testing_main.py:62-138) roll TEST_EPISODES episodes on one env and dump a pandas DataFrame with
the columns  Episode, Outcome, Total Reward, Time Steps, Path, Traffic Paths  (the test script adds
per-step records that live only in the pygame-side game object).  This module produces the same
table from the single-env adapter, so the reference's notebooks keep working on engine output.
"""
import numpy as np

from .config import OUTCOME_NAMES

BASELINE_COLUMNS = ("Episode", "Outcome", "Total Reward", "Time Steps", "Path", "Traffic Paths")
# testing_main.py:114-138: the baseline columns, Path Length and the thirteen per-step record lists
TESTING_RECORDS = (("psi", "heading_record"), ("d_sep", "d_sep_record"), ("a_lat", "a_lat_record"),
                   ("d_goal", "d_goal_record"), ("delta_heading", "delta_h_goal_record"),
                   ("v_closing", "v_closing_record"), ("d_cpa", "d_cpa_record"), ("d_dev", "d_dev_record"),
                   ("r_d_goal", "step_reward_d_goal_record"), ("r_h_goal", "step_reward_h_goal_record"),
                   ("r_d_cpa", "step_reward_d_cpa_record"), ("r_d_dev", "step_reward_d_dev_record"),
                   ("r_step", "step_reward_record"))
TESTING_COLUMNS = ("Episode", "Outcome", "Total Reward", "Time Steps", "Path Length", "Path", "Traffic Paths") + \
    tuple(c for c, _ in TESTING_RECORDS)


def simulate(env, episodes=100, policy=None, max_steps=None, columns="baseline", initial_states=None):
    """baseline_main.simulate() / testing_main.simulate(): `episodes` x (reset; step until done or
    MAX_STEPS).  `policy(obs)` returns the action array (default: the constant action [0] of
    baseline_main.py:44; testing_main.py:74 passes `model.predict(obs, deterministic=True)[0]`).
    columns="baseline": baseline_main.py:67-74's table; "testing": testing_main.py:114-138's, with Path
    Length and the per-step records.  initial_states: optional list of (own, trf, goal) to replay instead of
    drawing episodes.  Returns a dict of columns (lists), ready for pandas.DataFrame(...)."""
    max_steps = max_steps or env.config.max_steps
    names = TESTING_COLUMNS if columns == "testing" else BASELINE_COLUMNS
    cols = {c: [] for c in names}
    for episode in range(1, episodes + 1):
        obs = env.reset() if initial_states is None else env.reset_to(*initial_states[episode - 1])
        env.game.episode = episode
        for _ in range(max_steps):
            action = np.array([0]) if policy is None else policy(obs)
            obs, _, done, _ = env.step(action)
            if done:
                break
        g = env.game
        cols["Episode"].append(episode)
        cols["Outcome"].append(OUTCOME_NAMES.get(g.outcome))
        cols["Total Reward"].append(g.total_reward)
        cols["Time Steps"].append(g.steps)
        cols["Path"].append(list(g.path))
        cols["Traffic Paths"].append([list(p) for p in g.traffic_paths])
        if columns == "testing":
            cols["Path Length"].append(g.d_path)
            for c, attr in TESTING_RECORDS:
                cols[c].append(list(getattr(g, attr)))
    return cols


def to_csv(cols, path):
    """DataFrame.to_csv(path, index=False) as the reference writes it (baseline_main.py:67-74,
    testing_main.py:114-138)."""
    import pandas as pd
    names = TESTING_COLUMNS if "Path Length" in cols else BASELINE_COLUMNS
    pd.DataFrame({c: cols[c] for c in names}).to_csv(path, index=False)
