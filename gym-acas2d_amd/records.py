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


def simulate(env, episodes=100, policy=None, max_steps=None):
    """baseline_main.simulate(): `episodes` x (reset; step until done or MAX_STEPS).  `policy(obs)`
    returns the action array (default: the constant action [0] of baseline_main.py:44).  Returns a
    dict of columns (lists), ready for pandas.DataFrame(...)."""
    max_steps = max_steps or env.config.max_steps
    cols = {c: [] for c in BASELINE_COLUMNS}
    for episode in range(1, episodes + 1):
        obs = env.reset()
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
    return cols


def to_csv(cols, path):
    """DataFrame.to_csv(path, index=False) as the reference writes it (baseline_main.py:67-74)."""
    import pandas as pd
    pd.DataFrame({c: cols[c] for c in BASELINE_COLUMNS}).to_csv(path, index=False)
