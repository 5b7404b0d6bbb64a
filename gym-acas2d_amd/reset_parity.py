"""Host-side "parity reset": the reference's reset distribution drawn from Python's MT19937 in
the reference's exact draw order (ACAS2DGame.__init__, gym_ACAS2D/envs/game.py:41,88-114), so
that ``random.seed(s)`` names the same episodes here as in the reference.

The device reset (csrc/acas2d_kernels.hpp reset_env) draws the same distribution from a
counter-based RNG instead -- a sequential global stream cannot be consumed by 65 536 envs in
parallel.  This module exists for seed-for-seed comparisons against the reference (tests, the
single-env adapter); it is plain host logic and computes no step arithmetic.
"""
import math
import random as _random

import numpy as np


def draw_episode(cfg, rng=_random):
    """One ACAS2DGame() worth of draws.  Returns (own[4], traffic[N,4], goal[2]) float64 with
    rows (x, y, psi, v)."""
    gx, gy = cfg.goal
    sx, sy = cfg.start
    n = rng.randint(cfg.n_traffic, cfg.n_traffic)                        # game.py:41
    rng.uniform(0, 360)                                                  # game.py:88 (discarded)
    h0 = math.degrees(math.atan2(gy - sy, gx - sx) % (2 * math.pi))      # kinematics.py:16-22
    psi = (h0 + rng.uniform(-cfg.player_initial_heading_lim,
                            cfg.player_initial_heading_lim)) % 360       # game.py:91-92
    own = np.array([sx, sy, psi, cfg.airspeed], np.float64)
    trf = np.zeros((n, 4), np.float64)
    for i in range(n):
        if i == 0:
            down = rng.randint(0, 1)                                     # game.py:98
            x = cfg.width - cfg.collision_radius                         # game.py:100
            y = cfg.collision_radius + (down * (cfg.height - (2 * cfg.collision_radius)))
            v = rng.uniform(cfg.airspeed_factor_min, cfg.airspeed_factor_max) * cfg.airspeed
            h = (145 + (down * 70) + rng.uniform(-cfg.traffic_initial_heading_lim,
                                                 cfg.traffic_initial_heading_lim)) % 360
        else:
            x = rng.uniform(0, cfg.width - cfg.aircraft_size)            # game.py:109
            y = rng.uniform(0, 3 * cfg.height / 5)                       # game.py:110
            v = rng.uniform(cfg.airspeed_factor_min, cfg.airspeed_factor_max) * cfg.airspeed
            h = rng.uniform(0, 360)                                      # game.py:114
        trf[i] = (x, y, h, v)
    return own, trf, np.array([gx, gy], np.float64)


def draw_episodes(cfg, count, rng=_random):
    """`count` consecutive games from the stream (env 0 first), stacked."""
    eps = [draw_episode(cfg, rng) for _ in range(count)]
    return (np.stack([e[0] for e in eps]), np.stack([e[1] for e in eps]),
            np.stack([e[2] for e in eps]))
