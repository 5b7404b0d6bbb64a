"""ctypes front-end of the CPU oracle (oracle/acas2d_oracle.c).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package never does (and has no CPU fallback).

The constants below restate the reference's settings.py independently of the product's
ACAS2DConfig, so a wrong constant on either side shows up as a parity failure.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACAS2D_ORACLE_LIB: load another build of the oracle (e.g. the -fsanitize=address,undefined one)
_LIB_PATH = os.environ.get("ACAS2D_ORACLE_LIB") or os.path.join(_HERE, "_build", "libacas2d_oracle.so")

G0 = 9.80665  # scipy.constants.g (settings.py:1)


class OracleConfig(C.Structure):
    _fields_ = [("dt", C.c_double), ("acc_lat_limit", C.c_double), ("max_steps", C.c_int32),
                ("_pad", C.c_int32)] + [(n, C.c_double) for n in (
                    "collision_dist", "goal_radius", "safe_distance", "d_goal_max", "d_dev_max",
                    "d_sep_max", "d_cpa_max", "v_closing_max", "rw_d_goal_max", "rw_d_dev_max",
                    "reward_goal", "reward_collision", "own_x0", "own_y0", "own_v",
                    "own_heading_jitter", "goal_x", "goal_y", "t0_x", "t0_y_base", "t0_y_span",
                    "t0_heading_base", "t0_heading_step", "t0_heading_jitter", "tn_x_max",
                    "tn_y_max", "speed_factor_min", "speed_factor_max", "airspeed")]


class OracleState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi",
        "trf_v", "steps", "total_reward", "status", "episode")]


def default_config():
    """settings.py:1-54 + game.py:80-128 + rewards.py:22-23,46-47, evaluated like the reference."""
    MAX_STEPS, WIDTH, HEIGHT, FPS = 1000, 1600, 1000, 100
    AIRCRAFT_SIZE = 24
    COLLISION_RADIUS = 2 * AIRCRAFT_SIZE
    GOAL_RADIUS = 6 * AIRCRAFT_SIZE
    SAFE_DISTANCE = 4 * COLLISION_RADIUS
    AIRSPEED, FMIN, FMAX = 200, 1, 1
    own_x0, own_y0 = COLLISION_RADIUS, HEIGHT / 2
    goal_x, goal_y = WIDTH - GOAL_RADIUS, HEIGHT / 2
    d_goal0 = math.sqrt((own_x0 - goal_x) ** 2 + (own_y0 - goal_y) ** 2)
    rw_d_goal_init = (WIDTH - GOAL_RADIUS) - (2 * AIRCRAFT_SIZE)
    c = OracleConfig()
    c.dt = 1 / FPS
    c.acc_lat_limit = 20 * G0
    c.max_steps = MAX_STEPS
    c.collision_dist = 2 * COLLISION_RADIUS
    c.goal_radius = GOAL_RADIUS
    c.safe_distance = SAFE_DISTANCE
    c.d_goal_max = d_goal0 + (AIRSPEED / FPS) * MAX_STEPS
    c.d_dev_max = (AIRSPEED / FPS) * MAX_STEPS
    c.d_sep_max = float(np.sqrt(WIDTH ** 2 + HEIGHT ** 2) + (2 * (AIRSPEED / FPS) * MAX_STEPS))
    c.d_cpa_max = float(np.sqrt(WIDTH ** 2 + HEIGHT ** 2))
    c.v_closing_max = 2 * (FMAX * AIRSPEED)
    c.rw_d_goal_max = rw_d_goal_init + (AIRSPEED / FPS) * MAX_STEPS
    c.rw_d_dev_max = rw_d_goal_init / 2
    c.reward_goal, c.reward_collision = 1000, -1000
    c.own_x0, c.own_y0, c.own_v, c.own_heading_jitter = own_x0, own_y0, AIRSPEED, 3
    c.goal_x, c.goal_y = goal_x, goal_y
    c.t0_x, c.t0_y_base, c.t0_y_span = WIDTH - COLLISION_RADIUS, COLLISION_RADIUS, HEIGHT - 2 * COLLISION_RADIUS
    c.t0_heading_base, c.t0_heading_step, c.t0_heading_jitter = 145, 70, 15
    c.tn_x_max, c.tn_y_max = WIDTH - AIRCRAFT_SIZE, 3 * HEIGHT / 5
    c.speed_factor_min, c.speed_factor_max, c.airspeed = FMIN, FMAX, AIRSPEED
    return c


def build(force=False):
    if os.environ.get("ACAS2D_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "acas2d_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        d = C.c_double
        for name, n in (("distance", 4), ("relative_angle", 4), ("delta_heading", 2),
                        ("heading_reward", 2), ("closest_approach_reward", 3),
                        ("plan_deviation_reward", 2), ("goal_distance_reward", 2)):
            f = getattr(_lib, "acas2d_oracle_" + name)
            f.restype, f.argtypes = d, [d] * n
        _lib.acas2d_oracle_set_threads.restype = C.c_int
        _lib.acas2d_oracle_set_threads.argtypes = [C.c_int]
        _lib.acas2d_oracle_set_threads(1)                 # the scalar port unless a caller asks otherwise
        _lib.acas2d_oracle_philox4x32.restype = None
        _lib.acas2d_oracle_philox4x32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.acas2d_oracle_reset.restype = None
        _lib.acas2d_oracle_reset.argtypes = [C.POINTER(OracleConfig), C.POINTER(OracleState), C.c_void_p,
                                             C.c_uint64, C.c_int64, C.c_int64, C.c_int32]
        _lib.acas2d_oracle_observe.restype = None
        _lib.acas2d_oracle_observe.argtypes = [C.POINTER(OracleConfig), C.POINTER(OracleState), C.c_void_p,
                                               C.c_int64, C.c_int32]
        _lib.acas2d_oracle_step.restype = C.c_int64
        _lib.acas2d_oracle_step.argtypes = [C.POINTER(OracleConfig), C.POINTER(OracleState)] + \
            [C.c_void_p] * 8 + [C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_int32]
        _lib.acas2d_oracle_single_env_loop.restype = C.c_int64
        _lib.acas2d_oracle_single_env_loop.argtypes = [C.POINTER(OracleConfig), C.POINTER(OracleState), C.c_void_p,
                                                       C.c_int64] + [C.c_void_p] * 7 + [C.c_uint64, C.c_int64, C.c_int32]
    return _lib


def set_threads(n):
    """Host threads for OracleEnvs.step(): 1 = scalar port (default), 0 = all cores.  Returns the count."""
    return int(lib().acas2d_oracle_set_threads(int(n)))


RESET_PHILOX_ROUNDS = 7          # acas2d_oracle.h: ACAS2D_ORACLE_RESET_PHILOX_ROUNDS (the engine's reset RNG)


def philox4x32(ctr, key, rounds=RESET_PHILOX_ROUNDS):
    ctr = np.asarray(ctr, np.uint32)
    key = np.asarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().acas2d_oracle_philox4x32(ctr.ctypes.data, key.ctypes.data, int(rounds), out.ctypes.data)
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data


class OracleEnvs:
    """E independent envs stepped by the C oracle; numpy float64 SoA, traffic block [E][N]."""

    def __init__(self, n_envs, n_traffic, seed=13, env_offset=0, auto_reset=False, config=None):
        self.E, self.N, self.D = int(n_envs), int(n_traffic), 5 + 3 * int(n_traffic)
        self.seed, self.env_offset, self.auto_reset = int(seed), int(env_offset), bool(auto_reset)
        self.cfg = config if config is not None else default_config()
        E, N = self.E, self.N
        f8 = np.float64
        self.own_x, self.own_y, self.own_psi, self.own_v = (np.zeros(E, f8) for _ in range(4))
        self.goal_x, self.goal_y = np.zeros(E, f8), np.zeros(E, f8)
        self.trf_x, self.trf_y, self.trf_psi, self.trf_v = (np.zeros((E, N), f8) for _ in range(4))
        self.steps = np.zeros(E, np.int32)
        self.total_reward = np.zeros(E, f8)
        self.status = np.zeros(E, np.uint8)
        self.episode = np.zeros(E, np.uint32)
        self.obs = np.zeros((E, self.D), f8)
        self.reward = np.zeros(E, f8)
        self.done = np.zeros(E, np.uint8)
        self.outcome = np.zeros(E, np.uint8)
        self.term_obs = np.zeros((E, self.D), f8)
        self.ep_return = np.zeros(E, f8)
        self.ep_steps = np.zeros(E, np.int32)
        self._st = OracleState(*[getattr(self, n).ctypes.data for n, _ in OracleState._fields_])

    def set_state(self, own, trf, goal=None, steps=None):
        """own [E,4] = x,y,psi,v; trf [E,N,4]; goal [E,2] or [2]; steps [E] (value BEFORE observe)."""
        own, trf = np.asarray(own, np.float64), np.asarray(trf, np.float64)
        self.own_x[:], self.own_y[:], self.own_psi[:], self.own_v[:] = own.T
        self.trf_x[:], self.trf_y[:], self.trf_psi[:], self.trf_v[:] = np.moveaxis(trf, -1, 0)
        g = np.array([self.cfg.goal_x, self.cfg.goal_y]) if goal is None else np.asarray(goal, np.float64)
        g = np.broadcast_to(g, (self.E, 2))
        self.goal_x[:], self.goal_y[:] = g[:, 0], g[:, 1]
        self.steps[:] = 0 if steps is None else steps
        self.total_reward[:] = 0
        self.status[:] = 0

    def reset_philox(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().acas2d_oracle_reset(C.byref(self.cfg), C.byref(self._st), _ptr(m), self.seed,
                                  self.env_offset, self.E, self.N)

    def observe(self):
        lib().acas2d_oracle_observe(C.byref(self.cfg), C.byref(self._st), self.obs.ctypes.data,
                                    self.E, self.N)
        return self.obs

    def reset(self):
        self.episode[:] = 0
        self.reset_philox()
        return self.observe()

    def single_env_loop(self, actions):
        """E == 1 only: len(actions) sequential step() calls inside C (auto-reset on done); the state and
        the output buffers end as after the last step.  Returns the number of finished episodes."""
        assert self.E == 1 and self.auto_reset
        a = np.ascontiguousarray(np.asarray(actions, np.float64).reshape(-1))
        return int(lib().acas2d_oracle_single_env_loop(
            C.byref(self.cfg), C.byref(self._st), a.ctypes.data, a.size, self.obs.ctypes.data,
            self.reward.ctypes.data, self.done.ctypes.data, self.outcome.ctypes.data, self.term_obs.ctypes.data,
            self.ep_return.ctypes.data, self.ep_steps.ctypes.data, self.seed, self.env_offset, self.N))

    def step(self, actions):
        a = np.ascontiguousarray(np.asarray(actions, np.float64).reshape(self.E))
        n = lib().acas2d_oracle_step(
            C.byref(self.cfg), C.byref(self._st), a.ctypes.data, self.obs.ctypes.data,
            self.reward.ctypes.data, self.done.ctypes.data, self.outcome.ctypes.data,
            self.term_obs.ctypes.data, self.ep_return.ctypes.data, self.ep_steps.ctypes.data,
            int(self.auto_reset), self.seed, self.env_offset, self.E, self.N)
        return self.obs, self.reward, self.done, self.outcome, n
