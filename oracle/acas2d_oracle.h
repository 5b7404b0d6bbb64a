/*
 * acas2d_oracle.h -- CPU restatement (plain C, float64) of the gym-ACAS2D per-step hot path.
 *
 * TEST INFRASTRUCTURE.  This is the checker, not the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (gym-acas2d_amd/) never calls
 * into this file and has no CPU fallback.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * repository root, i.e. gym_ACAS2D/...).  Parity status: PINNED -- checked against
 *   (1) the reference's own golden file models/logs/baseline_ACAS2D_PPO_11_100.csv (digest in
 *       tests/golden/csv_baseline_digest.npz), and
 *   (2) vectors captured from the unmodified reference by oracle/refharness/capture_golden.py
 *       (tests/golden/ref_*.npz) for N_TRAFFIC in {1, 3, 8, 64}.
 *
 * Data layout (struct of arrays, one entry per env e, traffic block env-major):
 *   own_x/own_y/own_psi/own_v [E], goal_x/goal_y [E], trf_x/trf_y/trf_psi/trf_v [E][N],
 *   steps int32 [E], total_reward [E], status uint8 [E] (0 = running, else outcome 1/2/3),
 *   episode uint32 [E] (reset counter, input to the counter-based reset RNG).
 *   obs [E][5+3N] row-major, reward [E], done uint8 [E], outcome uint8 [E].
 */
#ifndef ACAS2D_ORACLE_H
#define ACAS2D_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* All tunables of settings.py:1-54 plus the per-episode normalisers of game.py:120-128 and
 * rewards.py:22-23,46-47 that are constants under the reference's fixed start/goal. */
typedef struct Acas2dOracleConfig {
    double dt;               /* 1 / FPS                         aircraft.py:18, settings.py:17 */
    double acc_lat_limit;    /* ACC_LAT_LIMIT = 20 g            settings.py:42 */
    int32_t max_steps;       /* MAX_STEPS                       settings.py:9  */
    int32_t _pad;
    double collision_dist;   /* 2 * COLLISION_RADIUS            game.py:187    */
    double goal_radius;      /* GOAL_RADIUS                     game.py:192    */
    double safe_distance;    /* SAFE_DISTANCE                   rewards.py:16  */
    double d_goal_max;       /* obs normaliser                  game.py:120    */
    double d_dev_max;        /* obs normaliser                  game.py:122    */
    double d_sep_max;        /* obs normaliser                  game.py:124    */
    double d_cpa_max;        /* obs normaliser                  game.py:126    */
    double v_closing_max;    /* obs normaliser                  game.py:128    */
    double rw_d_goal_max;    /* reward-side d_goal_max          rewards.py:46-47 */
    double rw_d_dev_max;     /* reward-side d_dev_max           rewards.py:22-23 */
    double reward_goal;      /* REWARD_GOAL                     settings.py:47 */
    double reward_collision; /* REWARD_COLLISION                settings.py:48 */
    /* reset distribution, game.py:80-116 */
    double own_x0, own_y0, own_v;        /* game.py:85-87   */
    double own_heading_jitter;           /* settings.py:43  */
    double goal_x, goal_y;               /* game.py:80-81   */
    double t0_x, t0_y_base, t0_y_span;   /* game.py:100-101 */
    double t0_heading_base, t0_heading_step, t0_heading_jitter; /* game.py:105-106 */
    double tn_x_max, tn_y_max;           /* game.py:109-110 */
    double speed_factor_min, speed_factor_max, airspeed;        /* game.py:103,112 */
} Acas2dOracleConfig;

typedef struct Acas2dOracleState {
    double *own_x, *own_y, *own_psi, *own_v;
    double *goal_x, *goal_y;
    double *trf_x, *trf_y, *trf_psi, *trf_v;
    int32_t *steps;
    double *total_reward;
    uint8_t *status;
    uint32_t *episode;
} Acas2dOracleState;

/* Philox4x32-R (Salmon et al., SC'11; Random123) -- the counter-based RNG of the build-defined device
 * reset: the engine and this oracle draw with R = 7 rounds, the fewest Random123 publishes known-answer
 * vectors for (tests/test_oracle_golden.py checks R = 7 and R = 10 against them). */
#define ACAS2D_ORACLE_RESET_PHILOX_ROUNDS 7
void acas2d_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);

/* Reset envs [0, n_envs) with the distribution of game.py:80-116 using Philox keyed on
 * (seed, env_offset + e, episode[e]); sets steps = 0, total_reward = 0, status = 0.  If `mask`
 * is non-NULL only envs with mask[e] != 0 are reset.  Does NOT compute the first observation. */
void acas2d_oracle_reset(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                         const uint8_t *mask, uint64_t seed, int64_t env_offset,
                         int64_t n_envs, int32_t n_traffic);

/* game.py:194-220 observe(): steps += 1, then the 5+3N vector (a_lat = 0, as at reset). */
void acas2d_oracle_observe(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                           double *obs, int64_t n_envs, int32_t n_traffic);

/* environment.py:29-42 step(): action -> observe -> evaluate -> is_done for every env.
 * auto_reset != 0 adds SB3-VecEnv semantics: on done, ep_return/ep_steps/term_obs (each may be
 * NULL) receive the finished episode's total reward / game.steps / last observation, episode[e]
 * is incremented, the env is reset (as acas2d_oracle_reset) and obs[e] is the new episode's
 * first observation.  auto_reset == 0: status[e] latches the outcome and later steps freeze the
 * traffic (game.py:243-245).  Returns the number of envs that finished. */
int64_t acas2d_oracle_step(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                           const double *actions, double *obs, double *reward, uint8_t *done,
                           uint8_t *outcome, double *term_obs, double *ep_return,
                           int32_t *ep_steps, int32_t auto_reset, uint64_t seed,
                           int64_t env_offset, int64_t n_envs, int32_t n_traffic);

/* One env, n_steps sequential acas2d_oracle_step(n_envs = 1, auto_reset = 1) calls with actions[t]:
 * the reference's single-env `env.step(a)` loop (baseline_main.py:39-61) without the interpreter. */
int64_t acas2d_oracle_single_env_loop(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                                      const double *actions, int64_t n_steps, double *obs, double *reward,
                                      uint8_t *done, uint8_t *outcome, double *term_obs,
                                      double *ep_return, int32_t *ep_steps, uint64_t seed,
                                      int64_t env_offset, int32_t n_traffic);

/* L1 functions exported one by one for known-answer tests. */
double acas2d_oracle_distance(double x1, double y1, double x2, double y2);        /* kinematics.py:7-13  */
double acas2d_oracle_relative_angle(double x1, double y1, double x2, double y2);  /* kinematics.py:16-22 */
double acas2d_oracle_delta_heading(double psi, double phi);                       /* kinematics.py:82-83 */
double acas2d_oracle_heading_reward(double psi, double phi);                      /* rewards.py:5-9   */
double acas2d_oracle_closest_approach_reward(double v_closing, double d_cpa, double safe_distance); /* rewards.py:12-16 */
double acas2d_oracle_plan_deviation_reward(double d_dev, double d_dev_max);       /* rewards.py:19-27 */
double acas2d_oracle_goal_distance_reward(double d_goal, double d_goal_max);      /* rewards.py:44-50 */

#ifdef __cplusplus
}
#endif
/* host threads used by acas2d_oracle_step(): 1 = scalar port, 0 = all cores; returns the number in effect */
int acas2d_oracle_set_threads(int n);

#endif
