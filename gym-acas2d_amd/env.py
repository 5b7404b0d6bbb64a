"""ACAS2DEnv -- single-env adapter with the reference's exact gym surface
(gym_ACAS2D/envs/environment.py:8-54): old-API ``reset() -> obs`` and
``step(action) -> (obs, reward, done, info)``, numpy float64 in and out, so that the
reference's ``*_main.py`` loops (``baseline_main.py:32-61``, ``testing_main.py:62-105``) run
unchanged.  Arithmetic runs in the same HIP kernel as the batched path (E = 1, float64).

``reset()`` draws the episode from Python's global ``random`` in the reference's draw order
(reset_parity.py), so ``random.seed(13)`` reproduces the reference's episodes.
"""
import random

import numpy as np
import torch

from .config import ACAS2DConfig
from .reset_parity import draw_episode
from .spaces import Box
from .vec_env import ACAS2DVecEnv


class _AircraftView:
    def __init__(self, x, y, psi, v_air):
        self.x, self.y, self.psi, self.v_air = x, y, psi, v_air


class GameView:
    """The attributes of ACAS2DGame that the reference's scripts read after/during an episode
    (testing_main.py:66-105, baseline_main.py:36-58): outcome, steps, total_reward, quit,
    episode, player, traffic, goal_x/goal_y, path, traffic_paths."""

    # attribute name of each record list (game.py:57-75) <- column of ACAS2DVecEnv.trace
    RECORD_LISTS = (("heading_record", 0), ("d_sep_record", 1), ("a_lat_record", 2), ("d_goal_record", 3),
                    ("delta_h_goal_record", 4), ("v_closing_record", 5), ("d_cpa_record", 6), ("d_dev_record", 7),
                    ("step_reward_d_goal_record", 8), ("step_reward_h_goal_record", 9),
                    ("step_reward_d_cpa_record", 10), ("step_reward_d_dev_record", 11), ("step_reward_record", 12))

    def __init__(self, env):
        self._env = env
        self.episode = None
        self.quit = False            # window-close flag of the pygame view (game.py:37,318-321)
        self.manual = False
        self.path = []               # game.py:47,132,231
        self.traffic_paths = []      # game.py:49-50,134-135,232-233 (logged BEFORE traffic moves)
        self.d_path = 0.0            # game.py:45,239: distance covered by the player
        for name, _ in self.RECORD_LISTS:        # game.py:57-75, read by testing_main.py:91-103
            setattr(self, name, [])

    def _append_records(self, row):
        for name, k in self.RECORD_LISTS:
            getattr(self, name).append(float(row[k]))

    def _scalar(self, t):
        return t[0].item()

    @property
    def steps(self):
        return int(self._scalar(self._env._vec.steps))

    @property
    def total_reward(self):
        return float(self._scalar(self._env._vec.total_reward))

    @property
    def outcome(self):
        s = int(self._scalar(self._env._vec.status))
        return None if s == 0 else s

    @property
    def running(self):
        return self.outcome is None

    @property
    def num_traffic(self):
        return self._env.config.n_traffic

    @property
    def goal_x(self):
        return float(self._scalar(self._env._vec.goal_x))

    @property
    def goal_y(self):
        return float(self._scalar(self._env._vec.goal_y))

    @property
    def player(self):
        v = self._env._vec
        return _AircraftView(*(float(self._scalar(t)) for t in (v.own_x, v.own_y, v.own_psi, v.own_v)))

    @property
    def traffic(self):
        v = self._env._vec
        cols = [t[0].cpu().numpy() for t in (v.trf_x, v.trf_y, v.trf_psi, v.trf_v)]
        return [_AircraftView(*(float(c[n]) for c in cols)) for n in range(self.num_traffic)]


class ACAS2DEnv:
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, n_traffic=1, device="cuda", config=None, record_paths=True):
        self._window = None
        self.config = config if config is not None else ACAS2DConfig(n_traffic=n_traffic)
        self._vec = ACAS2DVecEnv(1, device=device, dtype=torch.float64, auto_reset=False,
                                 config=self.config, record_trace=record_paths)
        self.record_paths = record_paths
        lo, hi = self.config.obs_low_high()
        self.observation_space = Box(low=np.array(lo, np.float64), high=np.array(hi, np.float64),
                                     dtype=np.float64)                       # environment.py:18-21
        self.action_space = Box(low=-1, high=1, shape=(1,), dtype=np.float64)   # environment.py:27
        self.game = GameView(self)
        self._new_game()             # the reference constructs a game in __init__ (environment.py:12)

    def reset_to(self, own, trf, goal=None):
        """reset() onto a GIVEN initial state instead of a drawn one (own = (x, y, psi, v), trf [N, 4], goal
        (x, y)): replaying recorded episodes / fixtures.  Returns the first observation."""
        own, trf = np.asarray(own, np.float64), np.asarray(trf, np.float64).reshape(self.config.n_traffic, 4)
        goal = np.asarray(self.config.goal if goal is None else goal, np.float64)
        return self._new_game((own, trf, goal))[0].cpu().numpy().astype(np.float64)

    def _new_game(self, state=None):
        own, trf, goal = draw_episode(self.config, random) if state is None else state
        obs = self._vec.set_state(own[None], trf[None], goal[None], steps=np.zeros(1, np.int32))
        episode = self.game.episode
        self.game = GameView(self)
        self.game.episode = episode
        if self.record_paths:
            # plain Python floats: the reference's CSVs are read back with ast.literal_eval
            self.game.path.append((float(own[0]), float(own[1])))
            self.game.traffic_paths = [[(float(t[0]), float(t[1]))] for t in trf]
            self.game._append_records(self._vec.trace[0].cpu().numpy())          # game.py:132-160
        self._last_trf = trf[:, :2].copy()
        self._last_own = (float(own[0]), float(own[1]))
        return obs

    def reset(self):
        """environment.py:44-48.  NB the reference's observe() in reset() leaves steps == 1."""
        # _new_game() ran observe() through set_state(); undo nothing: steps is now 1 as in the reference
        obs = self._new_game()
        return obs[0].cpu().numpy().astype(np.float64)

    def step(self, action):
        """environment.py:29-42 (without the pygame clock throttle of :31)."""
        a = np.asarray(action, dtype=np.float64).reshape(-1)[:1]
        obs, reward, done, _ = self._vec.step(a)
        v = self._vec
        # ONE device -> host transfer per step: obs, reward, done and (for the records) the positions
        parts = [obs[0], reward[:1], done[:1].to(obs.dtype)]
        if self.record_paths:
            parts += [v.own_x[:1], v.own_y[:1], v.trf_x[0], v.trf_y[0], v.trace[0]]
        host = torch.cat(parts).cpu().numpy()
        D, N = self.config.obs_dim, self.config.n_traffic
        if self.record_paths:
            x, y = float(host[D + 2]), float(host[D + 3])
            self.game.path.append((x, y))
            for n, lst in enumerate(self.game.traffic_paths):       # logged BEFORE the traffic moved
                lst.append((float(self._last_trf[n, 0]), float(self._last_trf[n, 1])))
            self._last_trf = np.stack([host[D + 4:D + 4 + N], host[D + 4 + N:D + 4 + 2 * N]], axis=1)
            self.game._append_records(host[D + 4 + 2 * N:])         # game.py:234-238, :266-276
            # game.py:239 d_path += distance(old, new) = sqrt(dot(d, d)) (kinematics.py:7-13)
            dx, dy = self._last_own[0] - x, self._last_own[1] - y
            self.game.d_path += float(np.sqrt(np.dot([dx, dy], [dx, dy])))
            self._last_own = (x, y)
        return host[:D].astype(np.float64), float(host[D]), bool(host[D + 1] != 0), {}

    def render(self, mode="human"):
        """environment.py:50-51 -> ACAS2DGame.view() (game.py:316-431), on a host copy of the state, off the GPU
        path (render.py).  "human": a pygame window like the reference's (RendererUnavailable if pygame is not
        installed); "rgb_array": the frame as uint8 [1000, 1600, 3] from the NumPy rasteriser.  As in the
        reference, closing the window sets `env.game.quit`."""
        from . import render as R
        scene = R.Scene.from_env(self)
        if mode == "rgb_array":
            return R.rgb_array(scene)
        if mode != "human":
            raise ValueError("render mode %r (metadata['render.modes'] = %s)" % (mode, self.metadata["render.modes"]))
        if self._window is None:
            self._window = R.PygameWindow()
        if not self._window.draw(scene):
            self.game.quit = True                       # game.py:318-321
        return None

    def close(self):
        if self._window is not None:
            self._window.close()
            self._window = None
        self._vec.close()

    def seed(self, seed=None):
        random.seed(seed)
        return [seed]


def register_with_gym():
    """gym.make("ACAS2D-v0") (gym_ACAS2D/__init__.py:3-6) if gym happens to be installed."""
    try:
        from gym.envs.registration import register   # type: ignore
    except Exception:  # noqa: BLE001
        return False
    try:
        register(id="ACAS2D-v0", entry_point="gym_acas2d_amd:ACAS2DEnv")
    except Exception:  # noqa: BLE001  (already registered)
        pass
    return True
