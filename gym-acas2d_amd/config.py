"""ACAS2DConfig -- every constant the step path reads, as one immutable object.

Restates gym_ACAS2D/settings.py:1-54 (the reference star-imports module constants everywhere)
and the derived normalisers of envs/game.py:120-128 and envs/rewards.py:22-23,46-47.  Defaults
are the reference's values; ``n_traffic`` replaces MIN_TRAFFIC == MAX_TRAFFIC (settings.py:31-32).
"""
import ctypes as C
import dataclasses
import math

G0 = 9.80665  # scipy.constants.g, settings.py:1

OUTCOME_NAMES = {1: "Goal", 2: "Collision", 3: "Timeout"}  # settings.py:6


class CConfig(C.Structure):
    """Mirror of ``struct Acas2dConfig`` in include/acas2d.h (field order is the ABI)."""
    _fields_ = [("dt", C.c_double), ("acc_lat_limit", C.c_double), ("max_steps", C.c_int32),
                ("math", C.c_int32)] + [(n, C.c_double) for n in (
                    "collision_dist", "goal_radius", "safe_distance", "d_goal_max", "d_dev_max",
                    "d_sep_max", "d_cpa_max", "v_closing_max", "rw_d_goal_max", "rw_d_dev_max",
                    "reward_goal", "reward_collision", "own_x0", "own_y0", "own_v", "own_heading0",
                    "own_heading_jitter", "goal_x", "goal_y", "t0_x", "t0_y_base", "t0_y_span",
                    "t0_heading_base", "t0_heading_step", "t0_heading_jitter", "tn_x_max",
                    "tn_y_max", "speed_factor_min", "speed_factor_max", "airspeed")]


@dataclasses.dataclass(frozen=True)
class ACAS2DConfig:
    n_traffic: int = 1                      # settings.py:31-32 (MIN_TRAFFIC == MAX_TRAFFIC)
    max_steps: int = 1000                   # settings.py:9
    width: int = 1600                       # settings.py:15
    height: int = 1000                      # settings.py:16
    fps: int = 100                          # settings.py:17
    aircraft_size: int = 24                 # settings.py:33
    airspeed: float = 200                   # settings.py:39
    airspeed_factor_min: float = 1          # settings.py:40
    airspeed_factor_max: float = 1          # settings.py:41
    acc_lat_limit: float = 20 * G0          # settings.py:42
    player_initial_heading_lim: float = 3   # settings.py:43
    traffic_initial_heading_lim: float = 15  # settings.py:44
    reward_goal: float = 1000               # settings.py:47
    reward_collision: float = -1000         # settings.py:48
    # not a setting of the reference: which formulation the float64 engine runs (include/acas2d.h, ACAS2D_MATH_*).
    # False: the reference's operation order with libm (~1e-13 from the CPU reference);  True: the float32
    # build's algebraic formulation in float64 arithmetic (within 1e-9, more than twice as fast).
    fast_math: bool = False

    def __post_init__(self):
        if self.n_traffic < 1:
            # the reference indexes traffic[0] unconditionally (game.py:254) -> IndexError at N = 0
            raise ValueError("n_traffic must be >= 1 (the reference raises IndexError at 0)")

    # settings.py:34-36
    @property
    def collision_radius(self):
        return 2 * self.aircraft_size

    @property
    def goal_radius(self):
        return 6 * self.aircraft_size

    @property
    def safe_distance(self):
        return 4 * self.collision_radius

    @property
    def obs_dim(self):                      # environment.py:18
        return 5 + 3 * self.n_traffic

    @property
    def dt(self):                           # aircraft.py:18
        return 1 / self.fps

    # game.py:80-87
    @property
    def goal(self):
        return (self.width - self.goal_radius, self.height / 2)

    @property
    def start(self):
        return (self.collision_radius, self.height / 2)

    def to_c(self):
        gx, gy = self.goal
        sx, sy = self.start
        step_len = (self.airspeed / self.fps) * self.max_steps
        diag = math.sqrt(self.width ** 2 + self.height ** 2)
        rw_d_goal_init = (self.width - self.goal_radius) - (2 * self.aircraft_size)   # rewards.py:22,46
        c = CConfig()
        c.dt = self.dt
        c.acc_lat_limit = self.acc_lat_limit
        c.max_steps = self.max_steps
        c.math = 1 if self.fast_math else 0                           # ACAS2D_MATH_FAST / _DEFAULT
        c.collision_dist = 2 * self.collision_radius                  # game.py:187
        c.goal_radius = self.goal_radius
        c.safe_distance = self.safe_distance
        c.d_goal_max = math.sqrt((sx - gx) ** 2 + (sy - gy) ** 2) + step_len   # game.py:120
        c.d_dev_max = step_len                                         # game.py:122
        c.d_sep_max = diag + 2 * step_len                              # game.py:124
        c.d_cpa_max = diag                                             # game.py:126
        c.v_closing_max = 2 * (self.airspeed_factor_max * self.airspeed)   # game.py:128
        c.rw_d_goal_max = rw_d_goal_init + step_len                    # rewards.py:47
        c.rw_d_dev_max = rw_d_goal_init / 2                            # rewards.py:23
        c.reward_goal, c.reward_collision = self.reward_goal, self.reward_collision
        c.own_x0, c.own_y0, c.own_v = sx, sy, self.airspeed
        # kinematics.py:16-22 relative_angle(start -> goal), game.py:91
        c.own_heading0 = math.degrees(math.atan2(gy - sy, gx - sx) % (2 * math.pi))
        c.own_heading_jitter = self.player_initial_heading_lim
        c.goal_x, c.goal_y = gx, gy
        c.t0_x = self.width - self.collision_radius                    # game.py:100
        c.t0_y_base = self.collision_radius                            # game.py:101
        c.t0_y_span = self.height - 2 * self.collision_radius
        c.t0_heading_base, c.t0_heading_step = 145, 70                 # game.py:105
        c.t0_heading_jitter = self.traffic_initial_heading_lim
        c.tn_x_max = self.width - self.aircraft_size                   # game.py:109
        c.tn_y_max = 3 * self.height / 5                               # game.py:110
        c.speed_factor_min, c.speed_factor_max = self.airspeed_factor_min, self.airspeed_factor_max
        c.airspeed = self.airspeed
        return c

    def obs_low_high(self):
        """environment.py:19-20 Box bounds."""
        lo = [0.0, 0.0, -1.0, 0.0, 0.0] + [0.0, -1.0, -1.0] * self.n_traffic
        hi = [1.0] * self.obs_dim
        return lo, hi

    @staticmethod
    def algorithmic_bytes_per_env_step(n_traffic, itemsize):
        """B(N, s) = s (16 + 9 N) + 9 -- SURVEY.md §8d."""
        return itemsize * (16 + 9 * n_traffic) + 9
