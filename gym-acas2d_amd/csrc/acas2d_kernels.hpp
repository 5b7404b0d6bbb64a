// acas2d_kernels.hpp -- CDNA4 (gfx950) device code of the batched ACAS2D step engine.
//
// One launch advances every env by one ACAS2DEnv.step() (reference: gym_ACAS2D/envs/
// environment.py:29-42).  Work decomposition: an env is owned by a GROUP of G consecutive lanes
// (G = power of two, 1..64, chosen from n_traffic on the host); lane j of the group owns traffic
// aircraft j, j+G, ...  With the env-major traffic block  trf_*[E][N]  a wave's loads and stores
// of the traffic block are fully coalesced (lane-linear addresses), the per-env scalars are
// same-address broadcasts, and the only cross-lane traffic is a log2(G)-step reduce of the
// collision predicate plus one broadcast of traffic[0]'s closing speed / d_cpa for the reward.
//
// The path is HBM-bound by design (no dense contraction -> no MFMA): B(N, s) = s(16 + 9N) + 9
// algorithmic bytes per env-step.
//
// Arithmetic follows the reference's operation order (cited per function) so that the float64
// instantiation agrees with the CPU reference to rounding of the device libm; the float32
// instantiation is the throughput mode.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acas2d.h"

namespace acas2d {

constexpr int kBlock = 256;

// ---- launch-constant parameters, already rounded to T on the host -----------------------------
template <typename T>
struct Params {
    T dt, acc_lat_limit, collision_dist, goal_radius, safe_distance;
    T d_goal_max, d_dev_max, d_sep_max, d_cpa_max, v_closing_max;
    T rw_d_goal_max, rw_d_dev_max, reward_goal, reward_collision;
    int32_t max_steps;
};

// reset distribution (game.py:80-116); evaluated in float64 for both instantiations so that a
// seed names the same episode in f32 and f64, then rounded to T once.
struct ResetParams {
    double own_x0, own_y0, own_v, own_heading0, own_heading_jitter, goal_x, goal_y;
    double t0_x, t0_y_base, t0_y_span, t0_heading_base, t0_heading_step, t0_heading_jitter;
    double tn_x_max, tn_y_max, speed_factor_min, speed_factor_max, airspeed;
};

template <typename T>
struct State {
    T *own_x, *own_y, *own_psi, *own_v, *goal_x, *goal_y;
    T *trf_x, *trf_y, *trf_psi, *trf_v;
    int32_t* steps;
    T* total_reward;
    uint8_t* status;
    uint32_t* episode;
};

template <typename T>
struct StepIO {
    const T* actions;
    T *obs, *reward;
    uint8_t *done, *outcome;
    T *term_obs, *ep_return;
    int32_t* ep_steps;
};

// ---- scalar math, one overload set per element type ----------------------------------------------
__device__ __forceinline__ void m_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void m_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ float m_sin(float x) { return sinf(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double m_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float m_atan(float x) { return atanf(x); }
__device__ __forceinline__ double m_atan(double x) { return atan(x); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ float m_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double m_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float m_fmod(float a, float b) { return fmodf(a, b); }
__device__ __forceinline__ double m_fmod(double a, double b) { return fmod(a, b); }

template <typename T> struct Const;
template <> struct Const<float> {
    static constexpr float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f,
                           rad2deg = 57.29577951308232087680f;
};
template <> struct Const<double> {
    static constexpr double pi = 3.14159265358979323846, two_pi = 6.28318530717958647692,
                            rad2deg = 57.29577951308232087680;
};

// Python float `a % 360` (CPython float_rem == NumPy remainder): sign of the divisor.
// aircraft.py:22, kinematics.py:58,69.  Headings move by < 1 degree per step, so the window
// (-360, 720) is the common case and is exact (Sterbenz); anything else takes fmod.
template <typename T>
__device__ __forceinline__ T py_mod360(T a) {
    const T m = T(360);
    if (a >= T(0) && a < m) return a;
    if (a >= m && a < T(720)) return a - m;
    if (a < T(0) && a > -m) return a + m;          // fmod(a, 360) == a, then += 360
    T r = m_fmod(a, m);                             // general case (and NaN)
    if (r != T(0)) { if (r < T(0)) r += m; } else { r = T(0); }
    return r;
}

// (deg / 360.0) * 2 * math.pi, left to right.  aircraft.py:23, kinematics.py:29,33,46,59,70
template <typename T>
__device__ __forceinline__ T deg2rad_ref(T deg) { return ((deg / T(360)) * T(2)) * Const<T>::pi; }

// kinematics.py:7-13  np.linalg.norm(p1 - p2, 2) == sqrt(fma(dy, dy, dx * dx)) (OpenBLAS ddot).
template <typename T>
__device__ __forceinline__ T distance(T x1, T y1, T x2, T y2) {
    T dx = x1 - x2, dy = y1 - y2;
    return m_sqrt(m_fma(dy, dy, dx * dx));
}

// kinematics.py:16-22  degrees(atan2(dy, dx) % (2 pi)); atan2 is in [-pi, pi] so the Python
// modulo is `r < 0 ? r + 2 pi : r` (and -0.0 -> +0.0).
template <typename T>
__device__ __forceinline__ T relative_angle(T x1, T y1, T x2, T y2) {
    T r = m_atan2(y2 - y1, x2 - x1);
    r = (r < T(0)) ? r + Const<T>::two_pi : (r == T(0) ? T(0) : r);
    return r * Const<T>::rad2deg;
}

// kinematics.py:82-83  builtin min(a, b) -> b only if b < a
template <typename T>
__device__ __forceinline__ T delta_heading(T psi, T phi) {
    T a = m_abs(psi - phi), b = T(360) - m_abs(psi - phi);
    return (b < a) ? b : a;
}

template <typename T> __device__ __forceinline__ T pow4(T x) { T x2 = x * x; return x2 * x2; }
template <typename T> __device__ __forceinline__ T py_min1(T v) { return (v < T(1)) ? v : T(1); }

// rewards.py:53-60 step_reward_5 with :5-9, :12-16, :19-27, :44-50 inlined.
template <typename T>
__device__ __forceinline__ T step_reward_5(const Params<T>& p, T v_closing, T psi, T phi, T d_cpa,
                                            T d_goal, T d_dev) {
    T hr = pow4(T(1) - delta_heading(psi, phi) / T(180));
    if (v_closing <= T(0)) {
        T car = py_min1(pow4(d_cpa / p.safe_distance));
        T ad = m_abs(d_dev);
        T pdr = (ad > p.rw_d_dev_max) ? T(0) : m_sqrt(T(1) - ad / p.rw_d_dev_max);
        return hr * car * pdr;
    }
    return hr * py_min1(pow4(T(1) - d_goal / p.rw_d_goal_max));
}

// ---- cross-lane helpers within a group of G lanes ------------------------------------------------
template <int G>
__device__ __forceinline__ int group_or(int v) {
#pragma unroll
    for (int m = 1; m < G; m <<= 1) v |= __shfl_xor(v, m, 64);
    return v;
}
template <int G, typename T>
__device__ __forceinline__ T group_bcast0(T v) {
    if constexpr (G == 1) return v;
    return __shfl(v, (int)(threadIdx.x & 63u) & ~(G - 1), 64);
}

// XCD-aware block remap: blocks are dealt round-robin over the 8 XCDs, so give each XCD one
// contiguous eighth of the env range (its L2 then sees whole cache lines and the same envs on
// every step).  Speed only -- any placement is correct.
__device__ __forceinline__ int64_t remap_block() {
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    if ((nb & 7u) == 0u) return (int64_t)(b & 7u) * (nb >> 3) + (b >> 3);
    return b;
}

// ---- Philox4x32-10 counter-based reset RNG ---------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c.x), l0 = 0xD2511F53u * c.x;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c.z), l1 = 0xCD9E8D57u * c.z;
        c = U4{h1 ^ c.y ^ k0, l1, h0 ^ c.w ^ k1, l0};
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ double u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
__device__ __forceinline__ double uniform(double a, double b, double u) {
#pragma clang fp contract(off)   // same rounding in the f32 (contract=fast) and f64 builds
    return a + (b - a) * u;
}

// ---- one env's player, as every lane of its group sees it ------------------------------------------
template <typename T>
struct Own {
    T x, y, psi, v, a_lat, gx, gy;
};

// What the reward / termination need back from observe().
template <typename T>
struct Seen {
    T d_goal, h_goal, d_dev, v_closing0, d_cpa0;
    int collided;
};

// game.py:194-220 observe() (+ the traffic half of game.py:222-247 action() when MOVE is set):
// every lane computes the player-side terms (same-address inputs, identical results), lane j
// walks traffic j, j+G, ... and writes its three observation entries.  `steps` is the already
// incremented counter.
template <typename T, int G>
__device__ __forceinline__ Seen<T> observe(const Params<T>& p, const State<T>& s, const Own<T>& o,
                                           int64_t e, int j, int N, int32_t steps, bool move,
                                           T* __restrict__ obs_row) {
    Seen<T> r;
    // relative_speed() player terms, kinematics.py:27-29,35-36
    T so, co;
    m_sincos(deg2rad_ref(o.psi), &so, &co);
    // closing_speed() player projection, kinematics.py:56-65 (psi_dot = a_lat / v: no /dt here)
    T psi1 = py_mod360(o.psi + ((o.a_lat / o.v) * p.dt));
    T s1, c1;
    m_sincos(deg2rad_ref(psi1), &s1, &c1);
    T v1x = (o.v * c1) * p.dt, v1y = (o.v * s1) * p.dt;
    T x1 = o.x + v1x, y1 = o.y + v1y;
    // game.py:168-180
    r.d_goal = distance(o.x, o.y, o.gx, o.gy);
    r.h_goal = relative_angle(o.x, o.y, o.gx, o.gy);
    r.d_dev = r.d_goal * m_sin(deg2rad_ref(r.h_goal));

    int coll = 0;
    T vc0 = T(0), dc0 = T(0);
    for (int n = j; n < N; n += G) {
        const int64_t i = e * N + n;
        T tx = s.trf_x[i], ty = s.trf_y[i], tpsi = s.trf_psi[i], tv = s.trf_v[i];
        // aircraft.py:16-26 with a_lat = 0: psi = psi % 360, then the Euler step
        T tpsi_w = py_mod360(tpsi);
        T st, ct;
        m_sincos(deg2rad_ref(tpsi_w), &st, &ct);
        if (move) {
            tx = tx + ((tv * ct) * p.dt);
            ty = ty + ((tv * st) * p.dt);
            s.trf_x[i] = tx;
            s.trf_y[i] = ty;
            if (tpsi_w != tpsi) s.trf_psi[i] = tpsi_w;   // only ever for injected headings >= 360
        }
        // game.py:205-210
        T d = distance(o.x, o.y, tx, ty);
        coll |= (d < p.collision_dist) ? 1 : 0;            // game.py:185-189
        // kinematics.py:40-49 distance_closest_approach (arctan of a quotient, signed result)
        T a_rel_rad = deg2rad_ref(relative_angle(o.x, o.y, tx, ty));
        T v12x = o.v * co - tv * ct, v12y = o.v * so - tv * st;
        T dca = d * m_sin(a_rel_rad - m_atan(v12y / v12x));
        // kinematics.py:52-79 closing_speed; v2.y uses the PLAYER's airspeed (:74), kept
        T v2x = (tv * ct) * p.dt, v2y = (o.v * st) * p.dt;
        T x2 = tx + v2x, y2 = ty + ((tv * st) * p.dt);
        T ax = v1x - v2x, ay = v1y - v2y, bx = x1 - x2, by = y1 - y2;
        T c = (m_fma(ay, by, ax * bx) / distance(x1, y1, x2, y2)) / p.dt;
        T* q = obs_row + 5 + 3 * n;
        q[0] = d / p.d_sep_max;
        q[1] = dca / p.d_cpa_max;
        q[2] = c / p.v_closing_max;
        if (n == 0) { vc0 = c; dc0 = dca; }
    }
    r.collided = group_or<G>(coll);
    r.v_closing0 = group_bcast0<G>(vc0);                  // evaluate() reads traffic[0] only,
    r.d_cpa0 = group_bcast0<G>(dc0);                      // game.py:254-255
    if (j == 0) {                                          // game.py:199-203
        obs_row[0] = (T)steps / (T)p.max_steps;
        obs_row[1] = o.psi / T(360);
        obs_row[2] = r.d_dev / p.d_dev_max;
        obs_row[3] = r.d_goal / p.d_goal_max;
        obs_row[4] = r.h_goal / T(360);
    }
    return r;
}

// ACAS2DGame.__init__ reset distribution, game.py:80-116, from one Philox block per entity:
// counter = (env_lo, env_hi, episode, entity) with entity 0 = player, 1 + n = traffic n;
// words: x (bit 31 of it = starts_down for traffic 0), y, heading, airspeed factor.
template <typename T, int G>
__device__ __forceinline__ Own<T> reset_env(const ResetParams& rp, const State<T>& s, uint32_t k0,
                                            uint32_t k1, uint64_t gid, uint32_t episode, int64_t e,
                                            int j, int N) {
#pragma clang fp contract(off)   // float64 here in both builds: a seed names the same episode
    const uint32_t g_lo = (uint32_t)gid, g_hi = (uint32_t)(gid >> 32);
    U4 w = philox4x32_10(U4{g_lo, g_hi, episode, 0u}, k0, k1);
    Own<T> o;
    o.x = (T)rp.own_x0;
    o.y = (T)rp.own_y0;
    o.v = (T)rp.own_v;
    o.psi = (T)py_mod360(rp.own_heading0 + uniform(-rp.own_heading_jitter, rp.own_heading_jitter, u01(w.z)));
    o.gx = (T)rp.goal_x;
    o.gy = (T)rp.goal_y;
    o.a_lat = T(0);
    for (int n = j; n < N; n += G) {
        w = philox4x32_10(U4{g_lo, g_hi, episode, 1u + (uint32_t)n}, k0, k1);
        double x, y, psi;
        double v = uniform(rp.speed_factor_min, rp.speed_factor_max, u01(w.w)) * rp.airspeed;
        if (n == 0) {
            double down = (double)(w.x >> 31);
            x = rp.t0_x;
            y = rp.t0_y_base + (down * rp.t0_y_span);
            psi = py_mod360(rp.t0_heading_base + (down * rp.t0_heading_step) +
                            uniform(-rp.t0_heading_jitter, rp.t0_heading_jitter, u01(w.z)));
        } else {
            x = uniform(0.0, rp.tn_x_max, u01(w.x));
            y = uniform(0.0, rp.tn_y_max, u01(w.y));
            psi = uniform(0.0, 360.0, u01(w.z));
        }
        const int64_t i = e * N + n;
        s.trf_x[i] = (T)x;
        s.trf_y[i] = (T)y;
        s.trf_psi[i] = (T)psi;
        s.trf_v[i] = (T)v;
    }
    if (j == 0) {
        s.own_x[e] = o.x; s.own_y[e] = o.y; s.own_psi[e] = o.psi; s.own_v[e] = o.v;
        s.goal_x[e] = o.gx; s.goal_y[e] = o.gy;
    }
    return o;
}

// ---- kernels ------------------------------------------------------------------------------------------

// ACAS2DEnv.step(), environment.py:29-42.
template <typename T, int G, bool AUTO_RESET>
__global__ __launch_bounds__(kBlock) void step_kernel(Params<T> p, ResetParams rp, State<T> s,
                                                      StepIO<T> io, uint32_t k0, uint32_t k1,
                                                      int64_t env_offset, int64_t n_envs, int N) {
    const int64_t tid = remap_block() * kBlock + threadIdx.x;
    const int64_t e = tid / G;
    const int j = (int)(tid % G);
    if (e >= n_envs) return;                      // whole groups leave together (G divides 256)
    const int D = 5 + 3 * N;
    T* obs_row = io.obs + e * D;

    Own<T> o{s.own_x[e], s.own_y[e], s.own_psi[e], s.own_v[e], T(0), s.goal_x[e], s.goal_y[e]};
    int32_t steps = s.steps[e];
    bool frozen = false;
    if constexpr (!AUTO_RESET) frozen = s.status[e] != 0;   // game.py:243-245

    // game.py:225 + aircraft.py:16-26 for the player
    o.a_lat = io.actions[e] * p.acc_lat_limit;
    {
        T psi_dot = o.a_lat / (o.v * p.dt);
        o.psi = py_mod360(o.psi + (psi_dot * p.dt));
        T sn, cs;
        m_sincos(deg2rad_ref(o.psi), &sn, &cs);
        o.x = o.x + ((o.v * cs) * p.dt);
        o.y = o.y + ((o.v * sn) * p.dt);
    }
    steps += 1;                                                       // game.py:197
    Seen<T> r = observe<T, G>(p, s, o, e, j, N, steps, !frozen, obs_row);

    // game.py:249-292 evaluate()
    T rw = step_reward_5(p, r.v_closing0, o.psi, r.h_goal, r.d_cpa0, r.d_goal, r.d_dev);
    rw = rw * (T(1) - ((T)steps / (T)p.max_steps));                  // :262-263
    const bool at_goal = r.d_goal < p.goal_radius;                    // :191-192
    if (r.collided) rw += p.reward_collision;                         // :279-280
    if (at_goal) rw += p.reward_goal;                                 // :283-284
    // game.py:294-314 is_done(): timeout > collision > goal
    const uint8_t oc = (steps > p.max_steps) ? 3 : (r.collided ? 2 : (at_goal ? 1 : 0));
    T total = T(0);
    if (j == 0) {
        total = s.total_reward[e] + rw;                               // :287
        io.reward[e] = rw;
        io.done[e] = oc != 0;
        io.outcome[e] = oc;
    }
    if (oc == 0 || !AUTO_RESET) {
        if (j == 0) {
            s.own_x[e] = o.x; s.own_y[e] = o.y; s.own_psi[e] = o.psi;
            s.steps[e] = steps;
            s.total_reward[e] = total;
            if constexpr (!AUTO_RESET) { if (oc) s.status[e] = oc; }
        }
        return;
    }

    // ---- SB3 DummyVecEnv.step_wait semantics for a finished env (group-uniform branch) ----
    if (io.term_obs) {                        // every lane copies exactly the entries it wrote
        T* t_row = io.term_obs + e * D;
        for (int n = j; n < N; n += G)
            for (int k = 0; k < 3; ++k) t_row[5 + 3 * n + k] = obs_row[5 + 3 * n + k];
        if (j == 0)
            for (int k = 0; k < 5; ++k) t_row[k] = obs_row[k];
    }
    const uint32_t episode = s.episode[e] + 1u;
    if (j == 0) {
        if (io.ep_return) io.ep_return[e] = total;
        if (io.ep_steps) io.ep_steps[e] = steps;
        s.episode[e] = episode;
        s.steps[e] = 1;                                               // environment.py:47
        s.total_reward[e] = T(0);
    }
    Own<T> fresh = reset_env<T, G>(rp, s, k0, k1, (uint64_t)(env_offset + e), episode, e, j, N);
    observe<T, G>(p, s, fresh, e, j, N, 1, false, obs_row);
}

// ACAS2DEnv.reset(), environment.py:44-48.
template <typename T, int G>
__global__ __launch_bounds__(kBlock) void reset_kernel(Params<T> p, ResetParams rp, State<T> s,
                                                       const uint8_t* __restrict__ mask, T* obs,
                                                       int do_init, uint32_t k0, uint32_t k1,
                                                       int64_t env_offset, int64_t n_envs, int N) {
    const int64_t tid = remap_block() * kBlock + threadIdx.x;
    const int64_t e = tid / G;
    const int j = (int)(tid % G);
    if (e >= n_envs) return;
    if (mask && !mask[e]) return;
    Own<T> o;
    int32_t steps;
    if (do_init) {
        o = reset_env<T, G>(rp, s, k0, k1, (uint64_t)(env_offset + e), s.episode[e], e, j, N);
        steps = 0;
    } else {
        o = Own<T>{s.own_x[e], s.own_y[e], s.own_psi[e], s.own_v[e], T(0), s.goal_x[e], s.goal_y[e]};
        steps = s.steps[e];
    }
    if (obs) {
        steps += 1;
        observe<T, G>(p, s, o, e, j, N, steps, false, obs + e * (5 + 3 * N));
    }
    if (j == 0) {
        s.steps[e] = steps;
        s.total_reward[e] = T(0);
        s.status[e] = 0;
    }
}

// ---- host-side launchers (instantiated per element type in acas2d_f32.hip / acas2d_f64.hip) ----
int lanes_per_env(int n_traffic);

template <typename T>
int launch_step(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io, uint32_t flags,
                uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream);
template <typename T>
int launch_reset(const Acas2dConfig* cfg, const Acas2dState* st, const uint8_t* mask, void* obs,
                 int32_t do_init, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                 hipStream_t stream);

void set_error(const char* fmt, ...);

}  // namespace acas2d
