/*
 * acas2d.h -- C ABI of the MI355X-native batched ACAS2D step engine (libacas2d_hip.so).
 *
 * The reference (Christos-14/gym-ACAS2D) is pure Python and has no FFI; the boundary it exposes
 * for this path is the gym.Env surface  ACAS2DEnv.reset() / ACAS2DEnv.step(action)
 * (gym_ACAS2D/envs/environment.py:29-48).  Each entry point below names the reference interface
 * it replaces.  The Python host (gym-acas2d_amd/) binds these with ctypes; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary;
 *   - every pointer in Acas2dState / Acas2dStepIO is a DEVICE pointer owned by the caller; the
 *     library allocates nothing, keeps no global state besides a thread-local error string;
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) and are safe to capture into a hipGraph;
 *   - return value: 0 on success, negative ACAS2D_E* on failure, text via acas2d_last_error();
 *     no exceptions cross the ABI.  Where the reference raises ValueError for NaN headings
 *     (rewards.py:6-9) the engine propagates NaN instead;
 *   - `_f32` / `_f64` give the element type of every floating-point buffer (`void*` fields).
 *
 * Data layout in HBM (struct of arrays; E = n_envs, N = n_traffic, D = 5 + 3N):
 *   own_x, own_y, own_psi, own_v           T[E]      player aircraft      (aircraft.py:8-14)
 *   goal_x, goal_y                         T[E]      goal position        (game.py:80-81)
 *   trf_x, trf_y, trf_psi, trf_v           T[E][N]   traffic block, env-major so that one env's
 *                                                    block is contiguous  (game.py:96-116)
 *   steps                                  i32[E]    game.steps           (game.py:30,197)
 *   total_reward                           T[E]      game.total_reward    (game.py:32,287)
 *   status                                 u8[E]     0 = running, else latched outcome
 *                                                    (game.running/outcome, game.py:36,39)
 *   episode                                u32[E]    reset counter (input of the reset RNG)
 *   actions T[E]; obs T[E][D] row-major; reward T[E]; done u8[E]; outcome u8[E]
 */
#ifndef ACAS2D_H
#define ACAS2D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACAS2D_ABI_VERSION 7

/* error codes */
#define ACAS2D_OK 0
#define ACAS2D_EINVAL (-22)  /* bad argument (NULL pointer, n_traffic < 1, ...) */
#define ACAS2D_EHIP (-5)     /* HIP runtime error at launch */

/* step flags */
#define ACAS2D_AUTO_RESET 1u /* SB3 VecEnv semantics: reset finished envs inside the step */

/* Acas2dConfig.math.  The float32 entry points always run the FAST formulation (algebraically identical to
 * the reference, fewer roundings, hardware transcendentals; 1e-5-grade observations).  The float64 entry points
 * run  DEFAULT: the reference's operation order literally, with libm (agrees with the CPU reference to ~1e-13:
 *               the parity mode);
 *      FAST:    the FAST formulation in float64 arithmetic -- d_cpa and d_dev in their algebraic forms (no
 *               atan2 / atan / sin per aircraft), reciprocal multiplications, a range-limited sincos -- within
 *               1e-9 of the reference on every fixture (tests/test_gpu_parity.py), at more than twice the rate.
 *               One documented difference: the SIGN of a d_cpa observation entry is unspecified where the
 *               reference's relative velocity component v12x is an exact 0 (|v12x| < 1e-9: equal airspeeds with
 *               parallel or mirror-image headings) -- kinematics.py:47 takes arctan(v12y / v12x), whose sign there
 *               is a coin toss of libm's cos; the magnitude still agrees to 1e-9 and DEFAULT reproduces the sign. */
#define ACAS2D_MATH_DEFAULT 0
#define ACAS2D_MATH_FAST 1

/* outcome codes = settings.py:6 OUTCOME_NAMES */
#define ACAS2D_OUTCOME_NONE 0
#define ACAS2D_OUTCOME_GOAL 1
#define ACAS2D_OUTCOME_COLLISION 2
#define ACAS2D_OUTCOME_TIMEOUT 3

/* Every tunable of gym_ACAS2D/settings.py:1-54 that the step path reads, plus the normalisers
 * of game.py:120-128 and rewards.py:22-23,46-47 (constants under the reference's fixed start and
 * goal).  Always float64 here; the f32 entry points round each field once on the host. */
typedef struct Acas2dConfig {
    double dt;               /* 1 / FPS                          aircraft.py:18        */
    double acc_lat_limit;    /* ACC_LAT_LIMIT                    settings.py:42        */
    int32_t max_steps;       /* MAX_STEPS                        settings.py:9         */
    int32_t math;            /* ACAS2D_MATH_*: which formulation the float64 entry points run */
    double collision_dist;   /* 2 * COLLISION_RADIUS             game.py:187           */
    double goal_radius;      /* GOAL_RADIUS                      game.py:192           */
    double safe_distance;    /* SAFE_DISTANCE                    rewards.py:16         */
    double d_goal_max;       /* obs normaliser                   game.py:120           */
    double d_dev_max;        /* obs normaliser                   game.py:122           */
    double d_sep_max;        /* obs normaliser                   game.py:124           */
    double d_cpa_max;        /* obs normaliser                   game.py:126           */
    double v_closing_max;    /* obs normaliser                   game.py:128           */
    double rw_d_goal_max;    /* reward-side d_goal_max           rewards.py:46-47      */
    double rw_d_dev_max;     /* reward-side d_dev_max            rewards.py:22-23      */
    double reward_goal;      /* REWARD_GOAL                      settings.py:47        */
    double reward_collision; /* REWARD_COLLISION                 settings.py:48        */
    /* reset distribution, game.py:80-116 */
    double own_x0, own_y0, own_v;       /* game.py:85-87                                */
    double own_heading0;                /* relative_angle(start -> goal), game.py:91    */
    double own_heading_jitter;          /* PLAYER_INITIAL_HEADING_LIM, settings.py:43   */
    double goal_x, goal_y;              /* game.py:80-81                                */
    double t0_x, t0_y_base, t0_y_span;  /* game.py:100-101                              */
    double t0_heading_base, t0_heading_step, t0_heading_jitter; /* game.py:105-106      */
    double tn_x_max, tn_y_max;          /* game.py:109-110                              */
    double speed_factor_min, speed_factor_max, airspeed;        /* game.py:103,112      */
} Acas2dConfig;

/* Per-env state, device pointers (element type T = float for _f32, double for _f64). */
typedef struct Acas2dState {
    void *own_x, *own_y, *own_psi, *own_v;
    void *goal_x, *goal_y;
    void *trf_x, *trf_y, *trf_psi, *trf_v;
    int32_t *steps;
    void *total_reward;
    uint8_t *status;
    uint32_t *episode;
    /* Optional (NULL = off): T[E][16], the per-step record row behind testing_main.py:114-138's CSV columns
     * (the lists ACAS2DGame appends to, game.py:132-160, :231-241, :266-276):
     *   [0] psi  [1] d_sep (minimum separation AFTER the player moved and BEFORE the traffic did, :236-237)
     *   [2] a_lat  [3] d_goal  [4] delta_heading  [5] v_closing  [6] d_cpa  [7] d_dev
     *   [8] r_d_goal  [9] r_h_goal  [10] r_d_cpa  [11] r_d_dev  [12] r_step (step reward before the terminal
     *   bonuses; the undiscounted step_reward_5 in the row acas2d_reset_* writes, :160)  [13..15] zero.
     * Written by acas2d_step_* WITHOUT ACAS2D_AUTO_RESET (the single-env semantics those scripts run) and by
     * acas2d_reset_* when it computes an observation. */
    void *trace;
} Acas2dState;

/* Inputs / outputs of one step, device pointers.  term_obs, ep_return, ep_steps may be NULL. */
typedef struct Acas2dStepIO {
    const void *actions; /* T[E]      action[0] in [-1, 1]              game.py:225           */
    void *obs;           /* T[E][D]   observe()                         game.py:194-220       */
    void *reward;        /* T[E]      evaluate()                        game.py:249-292       */
    uint8_t *done;       /* u8[E]     is_done()                         game.py:294-314       */
    uint8_t *outcome;    /* u8[E]     game.outcome of THIS step (0 while running)             */
    void *term_obs;      /* T[E][D]   AUTO_RESET: last obs of a finished episode (rows of
                                      envs that did not finish are left untouched)            */
    void *ep_return;     /* T[E]      AUTO_RESET: game.total_reward at done                   */
    int32_t *ep_steps;   /* i32[E]    AUTO_RESET: game.steps at done (= step() calls + 1)     */
} Acas2dStepIO;

int acas2d_abi_version(void);
size_t acas2d_config_size(void);      /* sizeof(Acas2dConfig): layout check for bindings */
size_t acas2d_state_size(void);       /* sizeof(Acas2dState) */

const char *acas2d_last_error(void);  /* thread-local; valid until the next failing call  */

/*
 * acas2d_step_*: replaces ACAS2DEnv.step(action) (environment.py:29-42) for n_envs independent
 * envs: game.action -> game.observe -> game.evaluate -> game.is_done, one kernel launch.
 *   flags & ACAS2D_AUTO_RESET: finished envs store term_obs/ep_return/ep_steps, bump episode[e],
 *     are re-initialised with the distribution of game.py:80-116 from the counter-based RNG
 *     Philox4x32-7(key = seed, counter = (env_offset + e, episode[e], entity)) and return the
 *     new episode's first observation in obs (SB3 DummyVecEnv.step_wait semantics).
 *   otherwise: status[e] latches the outcome; stepping a finished env keeps moving the player
 *     but freezes its traffic (game.py:243-245).
 * env_offset = global index of env 0 of this shard (results are invariant to the sharding).
 *
 * state_out (ABI 6): NULL or == state: the step updates `state` in place.  Otherwise DOUBLE-BUFFERED state: the
 *   step reads `state` and writes the arrays it rewrites for every env -- own_x, own_y, own_psi, steps,
 *   total_reward, trf_x, trf_y -- into state_out's buffers instead (the caller then passes the two structs the
 *   other way round at the next step; results are bit-identical to stepping in place).  Why: a store that hits
 *   a cache line its launch loaded stays dirty in the XCD's L2 until the write-back at the END of the kernel,
 *   whereas stores to the other generation stream out during it (65 536 x 8 float32: 4.23 vs 4.65 us per launch).
 *   Layout contract: state_out's own_x, own_y, own_psi, steps and total_reward lie at ONE element offset from
 *   state's, its trf_x and trf_y at one (e.g. every such array allocated as [2][E] / [2][E][N], the two structs
 *   pointing at the two halves), without overlap and less than 2^31 elements away; every other field (own_v,
 *   goal_*, trf_psi, trf_v, status, episode, trace: changed at a reset only, in place) is the SAME buffer in
 *   both structs.  Needs ACAS2D_AUTO_RESET.  A hipGraph that captures an odd number of steps must not be
 *   replayed twice in a row (each replay would read the generation the previous one also read).
 *
 * Consecutive layout (ABI 7; optional, detected per launch, results identical either way): when the float32 arrays a
 *   step READS are consecutive rows of four blocks --
 *       own_x, own_y, own_psi, total_reward, steps     T[5][E]        own_v, goal_x, goal_y, episode    T[4][E]
 *       trf_x, trf_y                                   T[2][E][N]     trf_psi, trf_v                    T[2][E][N]
 *   (i.e. own_y == own_x + E, ..., (void*)steps == own_x + 4 E, trf_y == trf_x + E N, ...; a second generation is then
 *   a whole second T[5][E] / T[2][E][N] block, which satisfies state_out's contract above) -- five base pointers and the
 *   env count name every input of the step.  gfx950 hands the first 14 dwords of a kernel's arguments to each wavefront
 *   in registers, so the auto-reset step then issues ALL its loads with its first instructions instead of behind a
 *   scalar-load round trip to the argument segment (65 536 x 8: 5.19 -> 4.7 us per launch, 4.36 -> 3.8 where no env
 *   finishes).  Needs n_traffic with a packed work shape, E N < 2^29, and n_envs a whole multiple of eight workgroups'
 *   envs (1 024 at n_traffic = 8: the kernel then has no bounds checks).  acas2d_state_is_consecutive() tells whether
 *   a state qualifies; any other layout or size runs the general kernel.
 *
 * n_envs < 2^31 per call (shard beyond that: env_offset).
 */
int acas2d_step_f32(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dState *state_out,
                    const Acas2dStepIO *io, uint32_t flags, uint64_t seed, int64_t env_offset,
                    int64_t n_envs, int32_t n_traffic, void *stream);
int acas2d_step_f64(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dState *state_out,
                    const Acas2dStepIO *io, uint32_t flags, uint64_t seed, int64_t env_offset,
                    int64_t n_envs, int32_t n_traffic, void *stream);

/*
 * acas2d_rollout_*: n_steps consecutive ACAS2DEnv.step() calls fused into ONE launch -- the inner
 * loop of a rollout collector (baseline_main.py:39-61 / testing_main.py:69-105 with the actions
 * known up front; SB3's collect_rollouts once the policy runs on the device).  The state stays in
 * registers, step t reads actions[t][E] and writes obs[t][E][D], reward[t][E], done[t][E],
 * outcome[t][E]; term_obs [t][E][D], ep_return [t][E], ep_steps [t][E] are optional (NULL) and
 * written only where done[t][e].  ACAS2D_AUTO_RESET semantics always.  Bit-identical to n_steps
 * acas2d_step_* calls.  Needs a packed work shape: n_traffic in {1, 2, 3} or a multiple of
 * 16 / sizeof(T) that tiles a wave (4, 8, 16, 32, 64 for f32), else ACAS2D_EINVAL.
 */
int acas2d_rollout_f32(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dStepIO *io,
                       int32_t n_steps, uint64_t seed, int64_t env_offset, int64_t n_envs,
                       int32_t n_traffic, void *stream);
int acas2d_rollout_f64(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dStepIO *io,
                       int32_t n_steps, uint64_t seed, int64_t env_offset, int64_t n_envs,
                       int32_t n_traffic, void *stream);

/*
 * acas2d_rollout_policy_*: the rollout above with the policy evaluated INSIDE the kernel -- the
 * whole loop of testing_main.py:69-105 (`action, _ = model.predict(obs, deterministic=True);
 * obs, reward, done, info = env.step(action)`) in one launch.  The policy is Stable-Baselines3
 * 1.1.0's MlpPolicy actor as stored in the reference's model zips (policy.pth:
 * mlp_extractor.policy_net.{0,2}, action_net): obs -> Linear(D,64) tanh -> Linear(64,64) tanh ->
 * Linear(64,1); the deterministic action is the mean clipped to [-1, 1].  float32 weights and
 * float32 arithmetic in both element types (policy.predict() casts the observation to float32).
 *   obs_in          T[E][D]  the observation the first action is taken on (reset()'s / the last step's)
 *   io->actions     T[n_steps][E]  OUTPUT here: the action each step took
 *   everything else as acas2d_rollout_*.
 * Needs a thread-per-env work shape: n_traffic in {1, 2, 3, 4, 8} (f32) / {1, 2, 3} (f64).
 */
typedef struct Acas2dPolicy {
    const void *w1t, *b1;    /* float[D][64]  = policy_net.0.weight TRANSPOSED, float[64] */
    const void *w2t, *b2;    /* float[64][64] = policy_net.2.weight TRANSPOSED, float[64] */
    const void *w3, *b3;     /* float[64]     = action_net.weight,              float[1]  */
    int32_t hidden;          /* 64 */
    int32_t _pad;
} Acas2dPolicy;

int acas2d_rollout_policy_f32(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dStepIO *io,
                              const Acas2dPolicy *policy, const void *obs_in, int32_t n_steps,
                              uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                              void *stream);
int acas2d_rollout_policy_f64(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dStepIO *io,
                              const Acas2dPolicy *policy, const void *obs_in, int32_t n_steps,
                              uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                              void *stream);

/*
 * acas2d_collect_*: the collector of one PPO iteration in ONE launch -- SB3 1.1.0's `collect_rollouts` as
 * training_main.py:44-52 runs it through `PPO.learn()`: for n_steps steps, `actions, values, log_probs =
 * policy(obs)` (the action DRAWN from N(mean, exp(log_std))), `env.step(clip(actions, -1, 1))`.  acas2d_rollout_policy_*
 * with the value net and the Gaussian sampling inside the kernel:
 *   actor, obs_in, io                 as acas2d_rollout_policy_*; io->actions[t][e] receives the RAW (unclipped) action
 *   v1t .. vb3                        the value net, same layout as the actor (mlp_extractor.value_net.{0,2}, value_net)
 *   log_std                           float[1]
 *   values, logp                      T[n_steps][E] outputs: V(obs_t), log N(action_t; mean_t, exp(log_std))
 *   noise_seed, noise_step            eps ~ N(0, 1) comes from Philox4x32-7(key = noise_seed, counter = (env_offset + e,
 *                                     noise_step + t, tag)) by Box-Muller: the stream depends on the global env index and
 *                                     the step number only (pass the number of steps collected so far)
 * A non-finite observation entry (the reference's NaN d_cpa in exact parallel flight, kinematics.py:48) reaches the
 * two networks as 0; the observation itself is stored as it is.  Same work shapes as acas2d_rollout_policy_*.
 */
typedef struct Acas2dActorCritic {
    Acas2dPolicy actor;
    const void *v1t, *vb1;   /* float[D][64]  = value_net.0.weight TRANSPOSED, float[64] */
    const void *v2t, *vb2;   /* float[64][64] = value_net.2.weight TRANSPOSED, float[64] */
    const void *v3, *vb3;    /* float[64]     = value_net.weight,              float[1]  */
    const void *log_std;     /* float[1] */
    void *values, *logp;     /* T[n_steps][E] */
    uint64_t noise_seed;
    uint32_t noise_step;
    uint32_t _pad;
} Acas2dActorCritic;

int acas2d_collect_f32(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dStepIO *io,
                       const Acas2dActorCritic *ac, const void *obs_in, int32_t n_steps, uint64_t seed,
                       int64_t env_offset, int64_t n_envs, int32_t n_traffic, void *stream);
int acas2d_collect_f64(const Acas2dConfig *cfg, const Acas2dState *state, const Acas2dStepIO *io,
                       const Acas2dActorCritic *ac, const void *obs_in, int32_t n_steps, uint64_t seed,
                       int64_t env_offset, int64_t n_envs, int32_t n_traffic, void *stream);

/*
 * acas2d_ppo_update_f32: ONE minibatch update of SB3 1.1.0's PPO.train() for the MlpPolicy actor-critic (the update
 * half of `PPO('MlpPolicy', env).learn()`, training_main.py:44-52) as two launches: forward + PPO loss + backward of
 * both 2 x 64 tanh networks on the rows idx[0 .. n_rows) of the rollout buffer (advantages normalised over the
 * minibatch, clipped surrogate, MSE value loss without clipping, entropy of the state-independent Gaussian), then
 * clip_grad_norm_ + Adam on the 13 parameter tensors IN PLACE.  Parameters in torch's own layouts ([out][in]), all
 * float32.  grad / adam_m / adam_v: acas2d_ppo_workspace_floats(obs_dim) floats each, zero before the first call
 * (grad is left zero by every call); adam_step: int32[1], 0 before the first call; stats: float[8], zero before
 * the first call -- [2] gradient norm, [4] policy loss, [5] value loss of the last minibatch.
 * obs_dim in {8, 11, 14, 17, 29} (n_traffic 1, 2, 3, 4, 8).  max_grad_norm < 0 (tests): only the gradient is
 * computed and left in `grad` (actor w1 b1 w2 b2 w3 b3, critic likewise, log_std), nothing is applied.
 * Run-to-run: the per-wave partial gradients are added to `grad` with float atomics, whose order is not fixed, so two
 * runs of the same update agree to float32 rounding of the sums (~1e-7 relative), not bit for bit -- unlike the env
 * kernels, which are bitwise deterministic.  The gradient kernel uses 70 - 75 KB of LDS per workgroup (gfx950 has 160 KB;
 * checked against the device at the first call, ACAS2D_EINVAL where it does not fit).
 */
typedef struct Acas2dPpoUpdate {
    void *actor_w1, *actor_b1, *actor_w2, *actor_b2, *actor_w3, *actor_b3;       /* mlp_extractor.policy_net.{0,2}, action_net */
    void *critic_w1, *critic_b1, *critic_w2, *critic_b2, *critic_w3, *critic_b3; /* mlp_extractor.value_net.{0,2}, value_net  */
    void *log_std;
    const void *obs;                 /* float[n][obs_dim]: the rollout buffer, flat */
    const void *act, *old_logp, *adv, *ret;   /* float[n] */
    const int64_t *idx;              /* int64[n_rows]: the minibatch */
    int32_t n_rows, obs_dim;
    float clip_range, vf_coef, ent_coef, max_grad_norm;
    float learning_rate, beta1, beta2, adam_eps;
    void *grad, *adam_m, *adam_v;
    int32_t *adam_step;
    void *stats;
} Acas2dPpoUpdate;

int acas2d_ppo_workspace_floats(int32_t obs_dim);
int acas2d_ppo_update_f32(const Acas2dPpoUpdate *u, void *stream);

/*
 * acas2d_reset_*: replaces ACAS2DEnv.reset() (environment.py:44-48 -> ACAS2DGame.__init__,
 * game.py:28-41,80-116, then observe()).  For every env with mask[e] != 0 (mask == NULL: all):
 *   do_init != 0: draw a fresh episode from the Philox stream described above
 *                 (episode[e] is read, not modified), steps = 0, total_reward = 0, status = 0;
 *   do_init == 0: keep the state the caller wrote into the buffers (oracle-state injection /
 *                 host-side MT19937 "parity reset"), only zero total_reward and status;
 * then, if obs != NULL, run observe(): steps += 1 and the first observation into obs[e].
 */
int acas2d_reset_f32(const Acas2dConfig *cfg, const Acas2dState *state, const uint8_t *mask,
                     void *obs, int32_t do_init, uint64_t seed, int64_t env_offset,
                     int64_t n_envs, int32_t n_traffic, void *stream);
int acas2d_reset_f64(const Acas2dConfig *cfg, const Acas2dState *state, const uint8_t *mask,
                     void *obs, int32_t do_init, uint64_t seed, int64_t env_offset,
                     int64_t n_envs, int32_t n_traffic, void *stream);

/* 1 if acas2d_step_* with ACAS2D_AUTO_RESET takes the consecutive-layout kernel for this state (see acas2d_step_*),
 * else 0.  Informational (bench / tests); never an error. */
int acas2d_state_is_consecutive(const Acas2dState *state, int64_t n_envs, int32_t n_traffic, int32_t elem_size);

/*
 * Launch geometry chosen for (n_envs, n_traffic, elem_size = 4 | 8): lanes per env (power of two
 * <= 64), traffic aircraft per lane (-1: generic strided walk), threads per workgroup and number
 * of workgroups.  Informational (bench / DESIGN.md); output pointers may be NULL.
 */
int acas2d_launch_geometry(int64_t n_envs, int32_t n_traffic, int32_t elem_size,
                           int32_t *lanes_per_env, int32_t *traffic_per_lane,
                           int32_t *block_threads, int64_t *grid_blocks);

#ifdef __cplusplus
}
#endif
#endif /* ACAS2D_H */
