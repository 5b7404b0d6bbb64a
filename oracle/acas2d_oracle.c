/*
 * acas2d_oracle.c -- CPU restatement (plain C, float64) of the gym-ACAS2D per-step hot path.
 *
 * TEST INFRASTRUCTURE -- see acas2d_oracle.h for who may use this and how it is pinned.
 * Written from the reference's behaviour, one scalar env at a time, in the reference's own
 * operation order (so that float64 results agree to the last bit wherever the reference uses
 * libm, and to <= 1 ulp where it uses NumPy's own arctan):
 *
 *   - math.cos / math.sin / math.atan2 / np.sin / np.cos == glibc here (probed 50 000/50 000);
 *   - np.linalg.norm(p1 - p2, 2) == sqrt(fma(dy, dy, dx * dx)) and
 *     np.dot(a, b) == fma(a1, b1, a0 * b0) on this host (OpenBLAS ddot, probed 50 000/50 000);
 *   - np.arctan differs from glibc atan in ~0.1 % of arguments by 1 ulp (only feeds d_cpa).
 *
 * Compile with -ffp-contract=off so that the ONLY fused multiply-adds are the explicit fma()s.
 */
#include "acas2d_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- Python float semantics ------------------------------------------------------------- */

/* CPython float_rem / NumPy npy_remainder: result takes the sign of the divisor.
 * Used by aircraft.py:22, kinematics.py:20,58,69, game.py:91,106. */
static double py_mod(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* Python builtin min(1, v): returns v only if v < 1 (so min(1, nan) == 1). rewards.py:16,48 */
static double py_min1(double v) { return (v < 1.0) ? v : 1.0; }

/* (deg / 360.0) * 2 * math.pi, evaluated left to right. aircraft.py:23, kinematics.py:29,33,46 */
static double deg2rad_ref(double deg) { return ((deg / 360.0) * 2.0) * M_PI; }

/* ---- L1: kinematics.py ------------------------------------------------------------------- */

/* kinematics.py:7-13  np.linalg.norm(p1 - p2, 2) = sqrt(dot(d, d)), dot with one fma. */
double acas2d_oracle_distance(double x1, double y1, double x2, double y2)
{
    double dx = x1 - x2, dy = y1 - y2;
    return sqrt(fma(dy, dy, dx * dx));
}

/* kinematics.py:16-22  math.degrees(math.atan2(dy, dx) % (2 pi)); degrees(x) = x * (180/pi). */
double acas2d_oracle_relative_angle(double x1, double y1, double x2, double y2)
{
    double dx = x2 - x1, dy = y2 - y1;
    double rads = py_mod(atan2(dy, dx), 2.0 * M_PI);
    return rads * (180.0 / M_PI);
}

/* kinematics.py:82-83  builtin min(a, b): b only if b < a. */
double acas2d_oracle_delta_heading(double psi, double phi)
{
    double a = fabs(psi - phi), b = 360.0 - fabs(psi - phi);
    return (b < a) ? b : a;
}

typedef struct { double x, y, v, psi, a_lat; } aircraft_t;

/* aircraft.py:16-26  Aircraft.update_state(). */
static void update_state(aircraft_t *ac, double dt)
{
    double psi_dot = ac->a_lat / (ac->v * dt);
    ac->psi = py_mod(ac->psi + (psi_dot * dt), 360.0);
    double psi_rad = deg2rad_ref(ac->psi);
    ac->x = ac->x + ((ac->v * cos(psi_rad)) * dt);
    ac->y = ac->y + ((ac->v * sin(psi_rad)) * dt);
}

/* kinematics.py:40-49  distance_closest_approach() (with relative_speed, :25-37, inlined).
 * h_rel is a plain arctan of a quotient: v12x == 0 gives +-pi/2, 0/0 gives NaN. Signed. */
static double dist_closest_approach(const aircraft_t *a1, const aircraft_t *a2)
{
    double d = acas2d_oracle_distance(a1->x, a1->y, a2->x, a2->y);
    double a_rel = acas2d_oracle_relative_angle(a1->x, a1->y, a2->x, a2->y);
    double a_rel_rad = deg2rad_ref(a_rel);
    double psi1_rad = deg2rad_ref(a1->psi), psi2_rad = deg2rad_ref(a2->psi);
    double v12x = a1->v * cos(psi1_rad) - a2->v * cos(psi2_rad);
    double v12y = a1->v * sin(psi1_rad) - a2->v * sin(psi2_rad);
    double h_rel_rad = atan(v12y / v12x);
    return d * sin(a_rel_rad - h_rel_rad);
}

/* kinematics.py:52-79  closing_speed(): both aircraft projected one step ahead with
 * psi_dot = a_lat / v (no /dt, unlike aircraft.py:20); v2's y component uses aircraft ONE's
 * airspeed (:74) -- kept.  c > 0 <=> separating. */
static double closing_speed(const aircraft_t *a1, const aircraft_t *a2, double dt)
{
    double psi_1 = py_mod(a1->psi + ((a1->a_lat / a1->v) * dt), 360.0);
    double r1 = deg2rad_ref(psi_1);
    double v1x = (a1->v * cos(r1)) * dt, v1y = (a1->v * sin(r1)) * dt;
    double x1 = a1->x + v1x, y1 = a1->y + v1y;

    double psi_2 = py_mod(a2->psi + ((a2->a_lat / a2->v) * dt), 360.0);
    double r2 = deg2rad_ref(psi_2);
    double x2 = a2->x + ((a2->v * cos(r2)) * dt), y2 = a2->y + ((a2->v * sin(r2)) * dt);
    double v2x = (a2->v * cos(r2)) * dt, v2y = (a1->v * sin(r2)) * dt;

    double ax = v1x - v2x, ay = v1y - v2y;   /* v1 - v2 */
    double bx = x1 - x2, by = y1 - y2;       /* p1 - p2 */
    double dot = fma(ay, by, ax * bx);
    return (dot / acas2d_oracle_distance(x1, y1, x2, y2)) / dt;
}

/* ---- L1: rewards.py ---------------------------------------------------------------------- */

/* rewards.py:5-9.  The reference raises ValueError outside [0, 360] (NaN only); here NaN
 * propagates instead. */
double acas2d_oracle_heading_reward(double psi, double phi)
{
    return pow(1.0 - acas2d_oracle_delta_heading(psi, phi) / 180.0, 4.0);
}

/* rewards.py:12-16 */
double acas2d_oracle_closest_approach_reward(double v_closing, double d_cpa, double safe_distance)
{
    if (v_closing > 0.0) return 1.0;
    return py_min1(pow(d_cpa / safe_distance, 4.0));
}

/* rewards.py:19-27 */
double acas2d_oracle_plan_deviation_reward(double d_dev, double d_dev_max)
{
    d_dev = fabs(d_dev);
    if (d_dev > d_dev_max) return 0.0;
    return pow(1.0 - d_dev / d_dev_max, 0.5);
}

/* rewards.py:44-50 (ValueError for d_goal < 0 is unreachable: a norm is never negative). */
double acas2d_oracle_goal_distance_reward(double d_goal, double d_goal_max)
{
    return py_min1(pow(1.0 - d_goal / d_goal_max, 4.0));
}

/* rewards.py:53-60  step_reward_5 */
static double step_reward_5(const Acas2dOracleConfig *c, double v_closing, double psi, double phi,
                            double d_cpa, double d_goal, double d_dev)
{
    if (v_closing <= 0.0)
        return acas2d_oracle_heading_reward(psi, phi) *
               acas2d_oracle_closest_approach_reward(v_closing, d_cpa, c->safe_distance) *
               acas2d_oracle_plan_deviation_reward(d_dev, c->rw_d_dev_max);
    return acas2d_oracle_heading_reward(psi, phi) *
           acas2d_oracle_goal_distance_reward(d_goal, c->rw_d_goal_max);
}

/* ---- counter-based reset RNG (build-defined; the reference uses global MT19937) ---------- */

void acas2d_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 32-bit word -> uniform in (0, 1): (w + 0.5) * 2^-32, exact in float64. */
static double u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
/* random.uniform(a, b) = a + (b - a) * random() */
static double uniform(double a, double b, double u) { return a + (b - a) * u; }

/* One Philox block per entity: counter = (env_lo, env_hi, episode, entity), entity 0 = player,
 * 1 + n = traffic n; key = seed.  Words: w0 -> x (bit 31: starts_down for traffic 0), w1 -> y,
 * w2 -> heading, w3 -> airspeed factor.  Distribution: game.py:80-116. */
static void reset_env(const Acas2dOracleConfig *c, const Acas2dOracleState *st, uint64_t seed,
                      uint64_t gid, int64_t e, int32_t N)
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), st->episode[e], 0u}, w[4];

    acas2d_oracle_philox4x32(ctr, key, ACAS2D_ORACLE_RESET_PHILOX_ROUNDS, w);
    st->goal_x[e] = c->goal_x;                                            /* game.py:80-81 */
    st->goal_y[e] = c->goal_y;
    st->own_x[e] = c->own_x0;                                             /* game.py:85-87 */
    st->own_y[e] = c->own_y0;
    st->own_v[e] = c->own_v;
    st->own_psi[e] = py_mod(acas2d_oracle_relative_angle(c->own_x0, c->own_y0, c->goal_x, c->goal_y) +
                            uniform(-c->own_heading_jitter, c->own_heading_jitter, u01(w[2])),
                            360.0);                                       /* game.py:91-92 */
    for (int32_t n = 0; n < N; ++n) {
        ctr[3] = 1u + (uint32_t)n;
        acas2d_oracle_philox4x32(ctr, key, ACAS2D_ORACLE_RESET_PHILOX_ROUNDS, w);
        double x, y, psi;
        double v = uniform(c->speed_factor_min, c->speed_factor_max, u01(w[3])) * c->airspeed;
        if (n == 0) {                                                     /* game.py:97-106 */
            double down = (double)(w[0] >> 31);
            x = c->t0_x;
            y = c->t0_y_base + (down * c->t0_y_span);
            psi = py_mod(c->t0_heading_base + (down * c->t0_heading_step) +
                         uniform(-c->t0_heading_jitter, c->t0_heading_jitter, u01(w[2])), 360.0);
        } else {                                                          /* game.py:107-114 */
            x = uniform(0.0, c->tn_x_max, u01(w[0]));
            y = uniform(0.0, c->tn_y_max, u01(w[1]));
            psi = uniform(0.0, 360.0, u01(w[2]));
        }
        st->trf_x[e * N + n] = x;
        st->trf_y[e * N + n] = y;
        st->trf_psi[e * N + n] = psi;
        st->trf_v[e * N + n] = v;
    }
    st->steps[e] = 0;                                                     /* game.py:28-41 */
    st->total_reward[e] = 0.0;
    st->status[e] = 0;
}

void acas2d_oracle_reset(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                         const uint8_t *mask, uint64_t seed, int64_t env_offset,
                         int64_t n_envs, int32_t n_traffic)
{
    for (int64_t e = 0; e < n_envs; ++e)
        if (!mask || mask[e]) reset_env(cfg, st, seed, (uint64_t)(env_offset + e), e, n_traffic);
}

/* ---- L2: game.py ------------------------------------------------------------------------- */

static aircraft_t load_own(const Acas2dOracleState *st, int64_t e, double a_lat)
{
    aircraft_t a = {st->own_x[e], st->own_y[e], st->own_v[e], st->own_psi[e], a_lat};
    return a;
}

static aircraft_t load_trf(const Acas2dOracleState *st, int64_t e, int32_t N, int32_t n)
{
    aircraft_t a = {st->trf_x[e * N + n], st->trf_y[e * N + n], st->trf_v[e * N + n],
                    st->trf_psi[e * N + n], 0.0};
    return a;
}

/* game.py:194-220 observe() for one env; `own` carries the a_lat that closing_speed reads. */
static void observe_env(const Acas2dOracleConfig *c, const Acas2dOracleState *st, int64_t e,
                        int32_t N, const aircraft_t *own, double *obs)
{
    st->steps[e] += 1;                                                    /* game.py:197 */
    double d_goal = acas2d_oracle_distance(own->x, own->y, st->goal_x[e], st->goal_y[e]);
    double h_goal = acas2d_oracle_relative_angle(own->x, own->y, st->goal_x[e], st->goal_y[e]);
    double d_dev = d_goal * sin(deg2rad_ref(h_goal));                     /* game.py:175-180 */
    obs[0] = (double)st->steps[e] / (double)c->max_steps;                 /* game.py:199 */
    obs[1] = own->psi / 360.0;
    obs[2] = d_dev / c->d_dev_max;
    obs[3] = d_goal / c->d_goal_max;
    obs[4] = h_goal / 360.0;
    for (int32_t n = 0; n < N; ++n) {                                     /* game.py:205-210 */
        aircraft_t t = load_trf(st, e, N, n);
        obs[5 + 3 * n + 0] = acas2d_oracle_distance(own->x, own->y, t.x, t.y) / c->d_sep_max;
        obs[5 + 3 * n + 1] = dist_closest_approach(own, &t) / c->d_cpa_max;
        obs[5 + 3 * n + 2] = closing_speed(own, &t, c->dt) / c->v_closing_max;
    }
}

void acas2d_oracle_observe(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                           double *obs, int64_t n_envs, int32_t n_traffic)
{
    const int32_t D = 5 + 3 * n_traffic;
    for (int64_t e = 0; e < n_envs; ++e) {
        aircraft_t own = load_own(st, e, 0.0);
        observe_env(cfg, st, e, n_traffic, &own, obs + e * D);
    }
}

/* game.py:185-189 */
static int detect_collisions(const Acas2dOracleConfig *c, const Acas2dOracleState *st, int64_t e,
                             int32_t N, const aircraft_t *own)
{
    for (int32_t n = 0; n < N; ++n)
        if (acas2d_oracle_distance(own->x, own->y, st->trf_x[e * N + n], st->trf_y[e * N + n]) <
            c->collision_dist)
            return 1;
    return 0;
}

int64_t acas2d_oracle_step(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                           const double *actions, double *obs, double *reward, uint8_t *done,
                           uint8_t *outcome, double *term_obs, double *ep_return,
                           int32_t *ep_steps, int32_t auto_reset, uint64_t seed,
                           int64_t env_offset, int64_t n_envs, int32_t n_traffic)
{
    const int32_t N = n_traffic, D = 5 + 3 * n_traffic;
    int64_t n_done = 0;
    /* envs are independent (no cross-env reads anywhere in the reference): with -fopenmp the batch is
     * spread over the host cores (acas2d_oracle_set_threads), the result is the same bit for bit */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : n_done)
#endif
    for (int64_t e = 0; e < n_envs; ++e) {
        double *o = obs + e * D;

        /* ---- game.py:222-247 action() ---- */
        aircraft_t own = load_own(st, e, actions[e] * cfg->acc_lat_limit);    /* :225 */
        update_state(&own, cfg->dt);                                          /* :229 */
        st->own_x[e] = own.x; st->own_y[e] = own.y; st->own_psi[e] = own.psi;
        if (st->status[e] == 0) {                                             /* :243-245 */
            for (int32_t n = 0; n < N; ++n) {
                aircraft_t t = load_trf(st, e, N, n);
                update_state(&t, cfg->dt);
                st->trf_x[e * N + n] = t.x; st->trf_y[e * N + n] = t.y;
                st->trf_psi[e * N + n] = t.psi;
            }
        }

        /* ---- game.py:194-220 observe() ---- */
        observe_env(cfg, st, e, N, &own, o);

        /* ---- game.py:249-292 evaluate(): traffic[0] only (:254-255) ---- */
        aircraft_t t0 = load_trf(st, e, N, 0);
        double phi = acas2d_oracle_relative_angle(own.x, own.y, st->goal_x[e], st->goal_y[e]);
        double v_closing = closing_speed(&own, &t0, cfg->dt);
        double d_cpa = dist_closest_approach(&own, &t0);
        double d_goal = acas2d_oracle_distance(own.x, own.y, st->goal_x[e], st->goal_y[e]);
        double d_dev = d_goal * sin(deg2rad_ref(phi));
        double r_step = step_reward_5(cfg, v_closing, own.psi, phi, d_cpa, d_goal, d_dev);
        double tdf = 1.0 - ((double)st->steps[e] / (double)cfg->max_steps);   /* :262 */
        double r = r_step * tdf;
        int collided = detect_collisions(cfg, st, e, N, &own);
        int at_goal = d_goal < cfg->goal_radius;                              /* :191-192 */
        if (collided) r += cfg->reward_collision;                             /* :279-280 */
        if (at_goal) r += cfg->reward_goal;                                   /* :283-284 */
        st->total_reward[e] += r;                                             /* :287 */
        reward[e] = r;

        /* ---- game.py:294-314 is_done(): timeout > collision > goal ---- */
        uint8_t oc = 0;
        if (st->steps[e] > cfg->max_steps) oc = 3;
        else if (collided) oc = 2;
        else if (at_goal) oc = 1;
        done[e] = oc != 0;
        outcome[e] = oc;
        if (!oc) continue;
        ++n_done;
        if (!auto_reset) { st->status[e] = oc; continue; }

        /* ---- SB3 DummyVecEnv.step_wait semantics (build-defined) ---- */
        if (term_obs) memcpy(term_obs + e * D, o, sizeof(double) * (size_t)D);
        if (ep_return) ep_return[e] = st->total_reward[e];
        if (ep_steps) ep_steps[e] = st->steps[e];
        st->episode[e] += 1u;
        reset_env(cfg, st, seed, (uint64_t)(env_offset + e), e, N);
        aircraft_t fresh = load_own(st, e, 0.0);
        observe_env(cfg, st, e, N, &fresh, o);                                /* environment.py:44-48 */
    }
    return n_done;
}

/* The like-for-like CPU line of BASELINE.md section 3: ONE env stepped n_steps times in sequence,
 * one acas2d_oracle_step(n_envs = 1) per iteration with action actions[t] and auto-reset on done --
 * the loop of baseline_main.py:39-61 / the reference's `for t in range(MAX_STEPS): env.step(a)`
 * without the interpreter around it.  `st`, obs .. ep_steps are the one-env buffers of
 * acas2d_oracle_step().  Returns the number of episodes that finished. */
int64_t acas2d_oracle_single_env_loop(const Acas2dOracleConfig *cfg, const Acas2dOracleState *st,
                                      const double *actions, int64_t n_steps, double *obs, double *reward,
                                      uint8_t *done, uint8_t *outcome, double *term_obs,
                                      double *ep_return, int32_t *ep_steps, uint64_t seed,
                                      int64_t env_offset, int32_t n_traffic)
{
    int64_t finished = 0;
    for (int64_t t = 0; t < n_steps; ++t)
        finished += acas2d_oracle_step(cfg, st, actions + t, obs, reward, done, outcome, term_obs, ep_return,
                                       ep_steps, 1, seed, env_offset, 1, n_traffic);
    return finished;
}

/* number of host threads acas2d_oracle_step() uses (1 = the scalar port the bench reports as
 * cpu_baseline; 0 = all cores).  Returns the number in effect; always 1 without OpenMP. */
#ifdef _OPENMP
#include <omp.h>
#endif
int acas2d_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n <= 0) n = omp_get_num_procs();
    omp_set_num_threads(n);
    return n;
#else
    (void)n;
    return 1;
#endif
}
