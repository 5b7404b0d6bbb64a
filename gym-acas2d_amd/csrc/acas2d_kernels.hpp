// acas2d_kernels.hpp -- CDNA4 (gfx950) device code of the batched ACAS2D step engine.
//
// One launch advances every env by one ACAS2DEnv.step() (reference: gym_ACAS2D/envs/
// environment.py:29-42).
//
// Work decomposition.  An env is owned by a GROUP of G consecutive lanes (G = power of two,
// 1..64); lane j of the group owns C traffic aircraft:
//   packed  (N == C*G): the contiguous run [jC, jC+C) of the env's traffic block, moved with one
//           16-byte (C*sizeof(T)) vector load / store per state field -- lane-linear addresses,
//           1 KiB per wave-instruction, the coalescing sweet spot of the HBM path;
//   generic (any N, C = 1): traffic j, j+G, j+2G, ... with dword accesses.
// The traffic block is env-major, trf_*[E][N], so one env's block is contiguous.  Per-env
// scalars are same-address broadcast loads.  Observations are staged in a per-wavefront LDS tile
// (rows of D = 5+3N values, exactly the wave's contiguous slice of obs[E][D]) and flushed with
// lane-linear 16-byte stores; the tile also feeds terminal_observation on auto-reset.  The only
// cross-lane traffic is a log2(G)-step OR-reduce of the collision predicate (min-distance < 96)
// and one broadcast of traffic[0]'s closing speed / d_cpa for the reward.
//
// The path is HBM-bound by design (no dense contraction -> no MFMA): B(N, s) = s(16 + 9N) + 9
// algorithmic bytes per env-step.
//
// Two formulations of the same arithmetic, selected at compile time:
//   EXACT (float64 build)  the reference's operation order, literally, with device libm --
//          agrees with the CPU reference to ~1e-13;
//   FAST  (float32 build)  algebraically identical, fewer roundings and no libm calls:
//          d_dev = goal_y - y (= d_goal sin(atan2(dy, dx))), d_cpa = sign(v12x) (dy v12x - dx v12y)
//          / |v12| (= d sin(a_rel - arctan(v12y / v12x)), incl. the sign quirk of the plain
//          arctan), headings handled in revolutions so that v_sin_f32 / v_cos_f32 (1.3e-7 abs,
//          measured) need no range reduction, atan2 as a degree-15 odd minimax polynomial
//          (2.6e-8 rev), divisions by constants as multiplications, v_rcp / v_rsq / v_sqrt.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "acas2d.h"

namespace acas2d {

#ifndef ACAS2D_BLOCK
#define ACAS2D_BLOCK 256
#endif
constexpr int kBlock = ACAS2D_BLOCK;      // threads per workgroup (tuning knob; 256 measured best)
constexpr int kWavesPerBlock = kBlock / 64;

// constant address space: wave-uniform loads through it are scalar loads (s_load_*)
#define ACAS2D_AS4 __attribute__((address_space(4)))

#include "acas2d_diag.hpp"   // ACAS2D_STAMP(): empty in the product build

// ---- launch-constant parameters, already rounded to T on the host -----------------------------
template <typename T>
struct Params {
    T dt, acc_lat_limit, collision_dist, goal_radius, safe_distance;
    T d_goal_max, d_dev_max, d_sep_max, d_cpa_max, v_closing_max;
    T rw_d_goal_max, rw_d_dev_max, reward_goal, reward_collision;
    // reciprocals for the FAST formulation
    T inv_dt, inv_d_goal_max, inv_d_dev_max, inv_d_sep_max, inv_d_cpa_max, inv_v_closing_max;
    T inv_rw_d_goal_max, inv_rw_d_dev_max, inv_safe_distance, inv_max_steps;
    int32_t max_steps;
};

// Kernel arguments live in the kernarg segment; under SGPR pressure hipcc re-fetches them with
// s_load + s_waitcnt lgkmcnt(0) at every use (8+ full scalar-cache round trips in the step
// kernel's main path, each also draining the LDS queue).  Pinning the launch constants into
// VGPRs once at kernel entry removes every one of those stalls.
template <typename T>
__device__ __forceinline__ void pin_vgpr(T& x) { asm volatile("" : "+v"(x)); }
template <typename T>
__device__ __forceinline__ Params<T> pinned(Params<T> p) {
    pin_vgpr(p.dt); pin_vgpr(p.acc_lat_limit); pin_vgpr(p.collision_dist); pin_vgpr(p.goal_radius);
    pin_vgpr(p.safe_distance); pin_vgpr(p.d_goal_max); pin_vgpr(p.d_dev_max); pin_vgpr(p.d_sep_max);
    pin_vgpr(p.d_cpa_max); pin_vgpr(p.v_closing_max); pin_vgpr(p.rw_d_goal_max); pin_vgpr(p.rw_d_dev_max);
    pin_vgpr(p.reward_goal); pin_vgpr(p.reward_collision); pin_vgpr(p.inv_dt); pin_vgpr(p.inv_d_goal_max);
    pin_vgpr(p.inv_d_dev_max); pin_vgpr(p.inv_d_sep_max); pin_vgpr(p.inv_d_cpa_max);
    pin_vgpr(p.inv_v_closing_max); pin_vgpr(p.inv_rw_d_goal_max); pin_vgpr(p.inv_rw_d_dev_max);
    pin_vgpr(p.inv_safe_distance); pin_vgpr(p.inv_max_steps);
    return p;
}
// The step kernel's set: only what its formulation reads -- FAST multiplies by the reciprocals the host prepared, the
// reference operation order divides by the constants themselves (a pinned constant nobody reads still costs its scalar
// load, its registers and a v_mov in every wave; an unpinned one that IS read would merely be fetched at its use).
template <typename T, bool FAST>
__device__ __forceinline__ Params<T> pinned_for(Params<T> p) {
    pin_vgpr(p.dt); pin_vgpr(p.acc_lat_limit); pin_vgpr(p.collision_dist); pin_vgpr(p.goal_radius);
    pin_vgpr(p.rw_d_dev_max); pin_vgpr(p.reward_goal); pin_vgpr(p.reward_collision);
    if constexpr (FAST) {
        pin_vgpr(p.inv_dt); pin_vgpr(p.inv_d_goal_max); pin_vgpr(p.inv_d_dev_max); pin_vgpr(p.inv_d_sep_max);
        pin_vgpr(p.inv_d_cpa_max); pin_vgpr(p.inv_v_closing_max); pin_vgpr(p.inv_rw_d_goal_max);
        pin_vgpr(p.inv_rw_d_dev_max); pin_vgpr(p.inv_safe_distance); pin_vgpr(p.inv_max_steps);
    } else {
        pin_vgpr(p.safe_distance); pin_vgpr(p.d_goal_max); pin_vgpr(p.d_dev_max); pin_vgpr(p.d_sep_max);
        pin_vgpr(p.d_cpa_max); pin_vgpr(p.v_closing_max); pin_vgpr(p.rw_d_goal_max);
    }
    return p;
}

// reset distribution (game.py:80-116).  R = the type the constants arrive in; reset_entity() evaluates the draws
// in the ELEMENT type: (seed, global env index, episode counter) names one episode per element type, and the
// float32 episode equals the float64 one up to float32 rounding (24 random bits per uniform instead of 32).
template <typename R, typename GT = R>
struct ResetParamsT {
    R own_x0, own_y0, own_v, own_heading0, own_heading_jitter, goal_x, goal_y;
    R t0_x, t0_y_base, t0_y_span, t0_heading_base, t0_heading_step, t0_heading_jitter;
    R tn_x_max, tn_y_max, speed_factor_min, speed_factor_max, airspeed;
    // A fresh episode's player stands at (own_x0, own_y0) with the goal at (goal_x, goal_y): its goal distance,
    // goal bearing (degrees) and deviation are constants of the config, evaluated once on the host
    // (make_reset_params) for the FAST formulation's first observation -- see own_context_fresh().
    GT d_goal0, h_goal0, d_dev0;                          // in the element type also where the draws are float64
};
// reset_kernel, the in-step reset and the fused rollout all draw through reset_entity<T>() (bit-identical per
// element type: test_f32_reset_names_the_same_episodes).  The per-step float32 kernel gets the constants rounded
// on the host (18 SGPRs instead of 36 and no v_cvt_f32_f64 on the reset path); reset_kernel and the fused
// rollout take them as float64 and convert at the use (with the float set the compiler settles on a register
// allocation for the rollout kernel that runs 7 % slower -- measured, 3.07e10 vs 3.3e10 env-steps/s).
template <typename T, bool ROLLOUT>
using StepResetParams = ResetParamsT<typename std::conditional<ROLLOUT, double, T>::type, T>;

// Per-step record row behind testing_main.py:114-138's CSV columns (the lists ACAS2DGame appends to at
// game.py:132-160, :231-241, :266-276): psi, d_sep, a_lat, d_goal, delta_heading, v_closing, d_cpa, d_dev,
// r_d_goal, r_h_goal, r_d_cpa, r_d_dev, r_step; three spare values.
constexpr int kTraceWidth = 16;

template <typename T>
struct State {
    T *own_x, *own_y, *own_psi, *own_v, *goal_x, *goal_y;
    T *trf_x, *trf_y, *trf_psi, *trf_v;
    int32_t* steps;
    T* total_reward;
    uint8_t* status;
    uint32_t* episode;
    T* trace;                 // optional [E][kTraceWidth] (latching kernels and reset only), see write_trace()
    // Double-buffered state (acas2d_step_* with a `state_out`): the arrays a step REWRITES for every env -- own_x,
    // own_y, own_psi, steps, total_reward, trf_x, trf_y -- are read here and written `w_env` (per-env arrays) /
    // `w_trf` (traffic arrays) ELEMENTS further on, i.e. into the other generation's buffers (0 = in place).  See
    // put_env() / put_trf() and "store policies" below for why.
    int32_t w_env, w_trf;
};

template <typename T>
struct StepIO {
    const T* actions;
    T *obs, *reward;
    uint8_t *done, *outcome;
    T *term_obs, *ep_return;
    int32_t* ep_steps;
};

// Rebase every array on a wave's first env (`e0` wave-uniform, held in SGPRs): inside the kernels
// all indexing is then  scalar 64-bit base + 32-bit per-lane offset  (global_load ... v_off, s[base]),
// with no 64-bit VALU address arithmetic.
template <typename T>
__device__ __forceinline__ State<T> rebase(const State<T>& s, int64_t e0, int N) {
    return State<T>{s.own_x + e0, s.own_y + e0, s.own_psi + e0, s.own_v + e0, s.goal_x + e0, s.goal_y + e0,
                    s.trf_x + e0 * N, s.trf_y + e0 * N, s.trf_psi + e0 * N, s.trf_v + e0 * N,
                    s.steps + e0, s.total_reward + e0, s.status + e0, s.episode + e0,
                    s.trace ? s.trace + e0 * kTraceWidth : nullptr, s.w_env, s.w_trf};
}
template <typename T>
__device__ __forceinline__ StepIO<T> rebase(const StepIO<T>& io, int64_t e0, int D) {
    return StepIO<T>{io.actions + e0, io.obs + e0 * D, io.reward + e0, io.done + e0, io.outcome + e0,
                     io.term_obs ? io.term_obs + e0 * D : nullptr, io.ep_return ? io.ep_return + e0 : nullptr,
                     io.ep_steps ? io.ep_steps + e0 : nullptr};
}
// Wave index within the workgroup as a scalar (threadIdx.x >> 6 is wave-uniform, but only
// readfirstlane tells the compiler so).
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

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
__device__ __forceinline__ float m_copysign(float a, float b) { return copysignf(a, b); }
__device__ __forceinline__ double m_copysign(double a, double b) { return copysign(a, b); }

// hardware fast paths (FAST formulation; the double overloads keep FAST usable for T = double)
__device__ __forceinline__ float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// float64 FAST: v_rcp_f64 seed + two Newton steps (1e-16 relative) instead of the full IEEE division sequence
__device__ __forceinline__ double f_rcp(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double y1 = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    const double y2 = __builtin_fma(__builtin_fma(-x, y1, 1.0), y1, y1);
    return (x != 0.0 && x - x == 0.0) ? y2 : y;          // 0, inf, NaN: the seed already is 1 / x
}
__device__ __forceinline__ float f_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// float64 FAST: v_rsq_f64 (a ~2^-26 seed) and two Newton steps y <- y (1.5 - 0.5 x y^2): ~12 instructions and
// 1e-16 relative instead of a correctly rounded sqrt followed by a division (~45).  x = 0 -> inf, x = NaN -> NaN
// as 1 / sqrt(x) gives them (the Newton steps are skipped where they would produce inf * 0).
__device__ __forceinline__ double f_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    const double y1 = y * __builtin_fma(-hx * y, y, 1.5);
    const double y2 = y1 * __builtin_fma(-hx * y1, y1, 1.5);
    return (x > 0.0 && x < __builtin_inf()) ? y2 : y;
}
__device__ __forceinline__ float f_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// float64 FAST: sqrt(x) = x rsq(x) with one correction step (error below 1 ulp; libm's is correctly rounded at
// three times the instructions).  0 -> 0, inf -> inf, negative / NaN -> NaN.
__device__ __forceinline__ double f_rsq(double x);
__device__ __forceinline__ double f_sqrt(double x) {
    const double y = f_rsq(x), s = x * y;
    const double r = __builtin_fma(__builtin_fma(-s, s, x), 0.5 * y, s);
    return (x > 0.0 && x < __builtin_inf()) ? r : (x == 0.0 ? x : (x < 0.0 ? __builtin_nan("") : x));
}
// sin / cos of an angle given in REVOLUTIONS (v_sin_f32 / v_cos_f32 compute sin(2 pi x))
__device__ __forceinline__ void f_sincos_rev(float r, float* s, float* c) {
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}
// float64 FAST: headings are a fraction of a revolution away from [0, 1], so the quadrant comes off exactly
// (4 r and the subtraction are exact) and what is left is |theta| <= pi/4, where the Taylor series to
// theta^15 / theta^16 are below 5e-17 -- no Payne-Hanek path, no table, ~40 instructions instead of ocml's
// full-range sincos.  Absolute error ~1e-16 (the reference's own argument (psi / 360) 2 pi carries 1e-15).
__device__ __forceinline__ void f_sincos_rev(double r, double* s, double* c) {
    const double q = __builtin_rint(r * 4.0);
    const double t = __builtin_fma(r, 4.0, -q) * 1.57079632679489661923;
    const double u = t * t;
    double ps = -1.0 / 1307674368000.0;
    ps = __builtin_fma(ps, u, 1.0 / 6227020800.0);
    ps = __builtin_fma(ps, u, -1.0 / 39916800.0);
    ps = __builtin_fma(ps, u, 1.0 / 362880.0);
    ps = __builtin_fma(ps, u, -1.0 / 5040.0);
    ps = __builtin_fma(ps, u, 1.0 / 120.0);
    ps = __builtin_fma(ps, u, -1.0 / 6.0);
    const double sn = __builtin_fma(t * u, ps, t);
    double pc = 1.0 / 20922789888000.0;
    pc = __builtin_fma(pc, u, -1.0 / 87178291200.0);
    pc = __builtin_fma(pc, u, 1.0 / 479001600.0);
    pc = __builtin_fma(pc, u, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, u, 1.0 / 40320.0);
    pc = __builtin_fma(pc, u, -1.0 / 720.0);
    pc = __builtin_fma(pc, u, 1.0 / 24.0);
    pc = __builtin_fma(pc, u, -0.5);
    const double cs = __builtin_fma(u, pc, 1.0);
    const int qi = (int)q;
    const double ss = (qi & 1) ? cs : sn, cc = (qi & 1) ? sn : cs;    // sin / cos of theta + qi pi / 2
    *s = (qi & 2) ? -ss : ss;
    *c = ((qi + 1) & 2) ? -cc : cc;
}

template <typename T> struct Const;
template <> struct Const<float> {
    static constexpr float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f,
                           rad2deg = 57.29577951308232087680f, inv360 = 1.0f / 360.0f;
};
template <> struct Const<double> {
    static constexpr double pi = 3.14159265358979323846, two_pi = 6.28318530717958647692,
                            rad2deg = 57.29577951308232087680, inv360 = 1.0 / 360.0;
};

// Python float `a % 360` (CPython float_rem == NumPy remainder): sign of the divisor.
// aircraft.py:22, kinematics.py:58,69.  Headings move by < 1 degree per step, so the window
// (-360, 720) is the common case and is exact (Sterbenz); anything else takes fmod.
template <typename T>
__device__ __forceinline__ T py_mod360(T a) {
    const T m = T(360);
    // common window, branch-free and exact: [0,360) -> a; [360,720) -> a - 360 (Sterbenz);
    // (-360,0) -> fmod(a,360) == a, then += 360 exactly as CPython does
    T r = a;
    r = (a >= m) ? a - m : r;
    r = (a < T(0)) ? a + m : r;
    if (__builtin_expect(!(a > -m && a < T(720)), 0)) {   // injected headings / NaN only
        r = m_fmod(a, m);
        if (r != T(0)) { if (r < T(0)) r += m; } else { r = T(0); }
    }
    return r;
}

// FAST formulation: the branch-free window alone.  Exact for a in (-360, 720), i.e. for every
// heading the engine itself produces (|d psi| < 1 degree per step from [0, 360]); headings injected
// outside that window are wrapped by the EXACT (float64) build only.
template <typename T>
__device__ __forceinline__ T wrap360_window(T a) {
    const T m = T(360);
    T r = a;
    r = (a >= m) ? a - m : r;
    r = (a < T(0)) ? a + m : r;
    return r;
}
// The FAST formulation's wrap: the window alone in float32; the float64 build keeps the general form (its
// rarely taken fmod branch costs nothing measurable there, and injected headings stay covered).
template <typename T>
__device__ __forceinline__ T wrap_fast(T a) {
    if constexpr (sizeof(T) == 4) return wrap360_window(a);
    else return py_mod360(a);
}
template <typename T, bool FAST>
__device__ __forceinline__ T wrap360(T a) {
    if constexpr (FAST) return wrap_fast(a);
    else return py_mod360(a);
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

// FAST: atan2(y, x) mod 2 pi, in REVOLUTIONS [0, 1].  Octant reduction + odd minimax polynomial
// of degree 15 for atan(t) / (2 pi), t in [0, 1] (max error 2.6e-8 rev = 9.4e-6 degrees in
// float32, fitted offline).  atan2(+-0, +-0) = 0 like C; y = -0 never counts as negative, which
// reproduces `atan2 % (2 pi)` mapping -0.0 to +0.0.
template <typename T>
__device__ __forceinline__ T atan2_rev(T y, T x) {
    const T ax = m_abs(x), ay = m_abs(y);
    const T mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const T t = (mx == T(0)) ? T(0) : mn * f_rcp(mx);
    const T u = t * t;
    T p = T(-6.4530050760e-04);
    p = m_fma(p, u, T(3.4795833267e-03));
    p = m_fma(p, u, T(-8.8987017304e-03));
    p = m_fma(p, u, T(1.5346017534e-02));
    p = m_fma(p, u, T(-2.2136264969e-02));
    p = m_fma(p, u, T(3.1745943883e-02));
    p = m_fma(p, u, T(-5.3046120845e-02));
    p = m_fma(p, u, T(1.5915483734e-01));
    p = p * t;
    if (ay > ax) p = T(0.25) - p;
    if (x < T(0)) p = T(0.5) - p;
    if (y < T(0)) p = T(1) - p;
    return p;
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
template <typename T, bool FAST>
__device__ __forceinline__ T step_reward_5(const Params<T>& p, T v_closing, T psi, T phi, T d_cpa,
                                            T d_goal, T d_dev) {
    if constexpr (FAST) {
        // every 1 - a*b below is an EXPLICIT fma.  (Under -ffp-contract=fast, which this build used to
        // have, different instantiations of the kernel -- work shapes, rollout / policy variants --
        // fused some of them and not others, and the same state gave rewards one ulp apart depending
        // on the shape that stepped it; the build is -ffp-contract=off now, belt and braces.)
        T hr = pow4(m_fma(-delta_heading(psi, phi), T(1.0 / 180.0), T(1)));
        if (v_closing <= T(0)) {
            T car = py_min1(pow4(d_cpa * p.inv_safe_distance));
            T ad = m_abs(d_dev);
            T pdr = (ad > p.rw_d_dev_max) ? T(0) : f_sqrt(m_fma(-ad, p.inv_rw_d_dev_max, T(1)));
            return (hr * car) * pdr;
        }
        return hr * py_min1(pow4(m_fma(-d_goal, p.inv_rw_d_goal_max, T(1))));
    } else {
        T hr = pow4(T(1) - delta_heading(psi, phi) / T(180));
        if (v_closing <= T(0)) {
            T car = py_min1(pow4(d_cpa / p.safe_distance));
            T ad = m_abs(d_dev);
            T pdr = (ad > p.rw_d_dev_max) ? T(0) : m_sqrt(T(1) - ad / p.rw_d_dev_max);
            return hr * car * pdr;
        }
        return hr * py_min1(pow4(T(1) - d_goal / p.rw_d_goal_max));
    }
}

// The four sub-rewards the reference logs next to step_reward_5 (game.py:156-159, :272-275):
// goal_distance_reward rewards.py:44-50, heading_reward :5-9, closest_approach_reward :12-16,
// plan_deviation_reward :19-27 -- each in the formulation of the build (see step_reward_5 above).
template <typename T, bool FAST>
__device__ __forceinline__ void reward_parts(const Params<T>& p, T v_closing, T psi, T phi, T d_cpa, T d_goal,
                                             T d_dev, T& r_d_goal, T& r_h_goal, T& r_d_cpa, T& r_d_dev) {
    const T ad = m_abs(d_dev);
    if constexpr (FAST) {
        r_d_goal = py_min1(pow4(m_fma(-d_goal, p.inv_rw_d_goal_max, T(1))));
        r_h_goal = pow4(m_fma(-delta_heading(psi, phi), T(1.0 / 180.0), T(1)));
        r_d_cpa = (v_closing > T(0)) ? T(1) : py_min1(pow4(d_cpa * p.inv_safe_distance));
        r_d_dev = (ad > p.rw_d_dev_max) ? T(0) : f_sqrt(m_fma(-ad, p.inv_rw_d_dev_max, T(1)));
    } else {
        r_d_goal = py_min1(pow4(T(1) - d_goal / p.rw_d_goal_max));
        r_h_goal = pow4(T(1) - delta_heading(psi, phi) / T(180));
        r_d_cpa = (v_closing > T(0)) ? T(1) : py_min1(pow4(d_cpa / p.safe_distance));
        r_d_dev = (ad > p.rw_d_dev_max) ? T(0) : m_sqrt(T(1) - ad / p.rw_d_dev_max);
    }
}

// One row of the record table: `r_step` is what the reference appends to step_reward_record -- the
// undiscounted step_reward_5 at construction (game.py:160), r_step * tdf in evaluate() (:276).
template <typename T, bool FAST>
__device__ __forceinline__ void write_trace(const Params<T>& p, T* row, T psi, T d_sep, T a_lat, T h_goal, T d_goal,
                                            T d_dev, T v_closing, T d_cpa, T r_step) {
    T a, b, c, d;
    reward_parts<T, FAST>(p, v_closing, psi, h_goal, d_cpa, d_goal, d_dev, a, b, c, d);
    row[0] = psi; row[1] = d_sep; row[2] = a_lat; row[3] = d_goal; row[4] = delta_heading(psi, h_goal);
    row[5] = v_closing; row[6] = d_cpa; row[7] = d_dev; row[8] = a; row[9] = b; row[10] = c; row[11] = d;
    row[12] = r_step; row[13] = T(0); row[14] = T(0); row[15] = T(0);
}

// ---- cross-lane helpers within a group of G lanes ------------------------------------------------
// Groups of 2 or 4 lanes sit inside a DPP quad: the exchange is one v_mov_b32 with a quad_perm
// modifier (no LDS crossbar round trip as with ds_bpermute, which cost the headline (4,2) shape
// three lgkmcnt waits in the middle of its arithmetic).  quad_perm [a,b,c,d] = a | b<<2 | c<<4 | d<<6.
template <int CTRL>
__device__ __forceinline__ int quad_perm(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int G>
__device__ __forceinline__ int group_or(int v) {
    if constexpr (G == 2) return v | quad_perm<0xB1>(v);                      // [1,0,3,2]
    else if constexpr (G == 4) { v |= quad_perm<0xB1>(v); return v | quad_perm<0x4E>(v); }   // then [2,3,0,1]
    else {
#pragma unroll
        for (int m = 1; m < G; m <<= 1) v |= __shfl_xor(v, m, 64);
        return v;
    }
}
template <int G>
__device__ __forceinline__ float group_bcast0(float v) {
    if constexpr (G == 1) return v;
    else if constexpr (G == 2) return __int_as_float(quad_perm<0xA0>(__float_as_int(v)));     // [0,0,2,2]
    else if constexpr (G == 4) return __int_as_float(quad_perm<0x00>(__float_as_int(v)));     // [0,0,0,0]
    else return __shfl(v, (int)(threadIdx.x & 63u) & ~(G - 1), 64);
}
template <int G>
__device__ __forceinline__ double group_bcast0(double v) {
    if constexpr (G == 1) return v;
    else if constexpr (G == 2 || G == 4) {
        constexpr int CTRL = G == 2 ? 0xA0 : 0x00;
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)quad_perm<CTRL>((int)(unsigned)b), hi = (unsigned)quad_perm<CTRL>((int)(unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    } else return __shfl(v, (int)(threadIdx.x & 63u) & ~(G - 1), 64);
}

// Value of `v` in lane `src` (wave-uniform, so a v_readlane instead of a ds_bpermute round trip).
__device__ __forceinline__ int lane_value(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_value(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double lane_value(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Orders this wave's LDS writes before its later LDS reads.  The tile is private to the wave and
// a wave's DS instructions execute in issue order, so only the COMPILER must be held back: a
// memory clobber + wave_barrier.  (NOT __builtin_amdgcn_fence(..., "wavefront"): hipcc lowers that
// with s_waitcnt vmcnt(0), which stalled a resetting wave on the acknowledgement of every store
// it had in flight -- in the middle of the chip-wide write burst.)
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// XCD-aware block remap: blocks are dealt round-robin over the 8 XCDs, so give each XCD one
// contiguous eighth of the env range (its L2 then sees whole cache lines and the same envs on
// every step).  Speed only -- any placement is correct.
__device__ __forceinline__ uint32_t remap_block_of(uint32_t nb) {
    const uint32_t b = blockIdx.x;
    if ((nb & 7u) == 0u) return (b & 7u) * (nb >> 3) + (b >> 3);
    return b;
}
__device__ __forceinline__ int64_t remap_block() { return remap_block_of(gridDim.x); }

// ---- Philox4x32-7 counter-based reset RNG -----------------------------------------------------------
// Seven rounds: the fewest for which Random123 (Salmon et al., SC'11) publishes known-answer vectors and reports the
// generator Crush-resistant (ten is its default, with a safety margin this use does not need: the words only place
// aircraft).  A round is two dependent v_mad_u64_u32 on the wave that resets an env, i.e. on the launch's critical
// path: 10 -> 7 rounds measured 5.43 -> 5.22 us per launch at 65 536 x 8 (tools/ab.sh, one box).  The CPU checker of the
// test suite draws with the same seven rounds; both are pinned by the published vectors (the known-answer test under tests/).
struct U4 { uint32_t x, y, z, w; };
#ifndef ACAS2D_PHILOX_ROUNDS
#define ACAS2D_PHILOX_ROUNDS 7           // anything else: diagnostic builds only (the reset chain's latency knob)
#endif
__device__ __forceinline__ U4 philox4x32(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ACAS2D_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;   // v_mad_u64_u32
        c = U4{(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ double u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
__device__ __forceinline__ double uniform(double a, double b, double u) {
#pragma clang fp contract(off)   // same rounding in both builds whatever their -ffp-contract
    return a + (b - a) * u;
}
// float32 build, in-step reset only: the same draws evaluated in float32 (24 random bits per
// uniform) -- equal to the float64 evaluation rounded to float32 up to 1-2 ulp, at a fraction of
// the latency of the float64 path on the kernel's critical tail.
__device__ __forceinline__ float u01f(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float uniformf(float a, float b, float u) { return fmaf(b - a, u, a); }

// ---- store policies -----------------------------------------------------------------------------------------
// A plain store leaves its line dirty in the XCD's L2, and every dirty line is written back when the kernel ends
// (L2s are not coherent across XCDs, so a kernel boundary flushes them) -- AFTER the last wave has finished, with
// nothing left to overlap it.  Measured at 65 536 x 8, reset-free (tools/run_variants.sh, one box):
//   observations (7.6 MB, never re-read on the device)   plain 7.1 us per launch   non-temporal 4.8
//   + state (traffic x / y, per-env scalars: 5.9 MB), IN PLACE   non-temporal 4.7 - 4.9 (no gain: a store that hits a
//       line the launch LOADED earlier leaves it dirty whatever its hint)
//   + state into ANOTHER buffer (double-buffered state: read generation g, write generation 1 - g; State::w_env /
//       w_trf, acas2d_step_*'s state_out)   non-temporal 4.23 against 4.65 in place, same box -- the 5.9 MB leave
//       the L2s during the launch like the observations do, instead of in the write-back after its last wave
//   everything write-through (sc1)                        4.3 -- but UNSAFE and therefore not used: the kernel's
//       completion does not wait for write-through stores still in flight (with or without s_waitcnt vmcnt(0)
//       before s_endpgm, agent or system scope), and a device-to-host copy right behind the launch read stale
//       observation rows in one run out of four (the full-size float64 parity test of tests/test_gpu_parity.py).
// So: observations non-temporal; state non-temporal and, where the caller passes a second generation, out of place.

// ---- per-lane vector of C traffic values ---------------------------------------------------------
template <typename T, int C>
struct alignas((C * sizeof(T)) % 16 == 0 ? 16 : ((C * sizeof(T)) % 8 == 0 ? 8 : sizeof(T))) Vec {
    T v[C];
};

template <typename T, int W>
__device__ __forceinline__ void store_chunk(Vec<T, W>* dst, const Vec<T, W>& v);

// Stores into the arrays a step rewrites for every env, to the WRITE generation (State::w_env / w_trf), non-temporal.
// `e` / `i0`: the element index the same value was read at.
template <typename T, typename U>
__device__ __forceinline__ void put_env(const State<T>& s, U* base, int e, U v) {
    __builtin_nontemporal_store(v, base + (e + s.w_env));
}
template <typename T>
__device__ __forceinline__ void put_trf1(const State<T>& s, T* base, int i, T v) {
    __builtin_nontemporal_store(v, base + (i + s.w_trf));
}
template <typename T, int C>
__device__ __forceinline__ void put_trf(const State<T>& s, T* base, int i0, const Vec<T, C>& v) {
    Vec<T, C>* dst = reinterpret_cast<Vec<T, C>*>(base + (i0 + s.w_trf));
    if constexpr ((C & (C - 1)) == 0) store_chunk<T, C>(dst, v);
    else *dst = v;                                          // 3 aircraft per lane: one 12- / 24-byte store
}

// ---- one env's player, as every lane of its group sees it ------------------------------------------
// A product rounded to T that the compiler can never contract into an fma, whatever the build's
// -ffp-contract setting (fast lets the backend fuse across statements whatever a pragma says; an
// empty asm is the one thing it cannot see through).  The reference's differences of such products --
// relative velocity v1 cos(psi1) - v2 cos(psi2) (kinematics.py:35-36), the velocity difference of
// the closing speed (:70-76) -- cancel to EXACTLY 0 in parallel flight: 0/0 -> NaN d_cpa, closing
// speed 0 -> the `v_closing <= 0` reward branch.  An fma keeps one product unrounded and leaves a
// ~1e-6 residual that flips both (found by test_f32_reproduces_the_reference_nan_pattern).
template <typename T>
__device__ __forceinline__ T rounded(T x) {
    asm("" : "+v"(x));
    return x;
}

template <typename T>
struct Own {
    T x, y, psi, v, a_lat, gx, gy;
};

// Player-side terms shared by every traffic aircraft of the env.
template <typename T>
struct OwnCtx {
    T x, y;
    T co, so;             // cos / sin of the player's heading                (kinematics.py:35-36)
    T v1x, v1y, x1, y1;   // closing_speed(): one-step-ahead projection        (kinematics.py:56-65)
    T d_goal, h_goal, d_dev;
    T v;                  // (last: next to x and y, hipcc slices the three into one vector and extracts (y, v) through scratch)
};

// The player-side terms are broadcast operands of the packed float2 arithmetic.  Left to itself hipcc gathers
// neighbouring fields of the struct into a vector and extracts the pairs it wants from it THROUGH SCRATCH MEMORY
// (scratch_store_dwordx4 + three overlapping scratch_load_dwordx2 in the middle of the step); an empty asm per
// field keeps every one of them an independent scalar.
template <typename T>
__device__ __forceinline__ void scalars(OwnCtx<T>& c) {
    asm("" : "+v"(c.x)); asm("" : "+v"(c.y)); asm("" : "+v"(c.v)); asm("" : "+v"(c.co)); asm("" : "+v"(c.so));
    asm("" : "+v"(c.v1x)); asm("" : "+v"(c.v1y)); asm("" : "+v"(c.x1)); asm("" : "+v"(c.y1));
}

// What the reward / termination need back from observe().
template <typename T>
struct Seen {
    T d_goal, h_goal, d_dev, v_closing0, d_cpa0;
    int collided;
};

// ZERO_ACTION (a freshly reset episode: a_lat = 0, heading in [0, 360)): the one-step-ahead heading
// of closing_speed() is the heading itself, bit for bit, so its sin / cos are not computed twice.
template <typename T, bool FAST, bool ZERO_ACTION = false>
__device__ __forceinline__ OwnCtx<T> own_context(const Params<T>& p, const Own<T>& o) {
    OwnCtx<T> c;
    c.x = o.x; c.y = o.y; c.v = o.v;
    if constexpr (FAST) {
        f_sincos_rev(o.psi * Const<T>::inv360, &c.so, &c.co);
        T s1 = c.so, c1 = c.co;
        if constexpr (!ZERO_ACTION) {
            T psi1 = wrap_fast(o.psi + (o.a_lat * f_rcp(o.v)) * p.dt);
            f_sincos_rev(psi1 * Const<T>::inv360, &s1, &c1);
        }
        const T vdt = o.v * p.dt;
        c.v1x = rounded(vdt * c1); c.v1y = rounded(vdt * s1);
        c.x1 = o.x + c.v1x; c.y1 = o.y + c.v1y;
        const T gdx = o.gx - o.x, gdy = o.gy - o.y;
        c.d_goal = f_sqrt(m_fma(gdy, gdy, gdx * gdx));
        // materialised: evaluate() subtracts it from psi.  (float64: the degree-15 polynomial is a float32-grade
        // 2.6e-8 rev; one libm atan2 per lane and step keeps the heading at 1e-13 degrees.  A float64 polynomial --
        // folded at pi/8, degree 21, 8e-17 rad, 5.7e-14 degrees -- was measured on one box: 9.04 against 9.21 us per launch
        // where nothing finishes, but 11.6 - 11.7 against 11.1 with resets: not kept)
        if constexpr (sizeof(T) == 4) c.h_goal = rounded(atan2_rev(gdy, gdx) * T(360));
        else c.h_goal = relative_angle(o.x, o.y, o.gx, o.gy);
        c.d_dev = gdy;            // d_goal * sin(atan2(gdy, gdx)) == gdy          (game.py:175-180)
        scalars(c);
    } else {
        m_sincos(deg2rad_ref(o.psi), &c.so, &c.co);
        T s1 = c.so, c1 = c.co;
        if constexpr (!ZERO_ACTION) {
            // psi_dot = a_lat / v: no /dt here, unlike aircraft.py:20
            T psi1 = py_mod360(o.psi + ((o.a_lat / o.v) * p.dt));
            m_sincos(deg2rad_ref(psi1), &s1, &c1);
        }
        c.v1x = (o.v * c1) * p.dt; c.v1y = (o.v * s1) * p.dt;
        c.x1 = o.x + c.v1x; c.y1 = o.y + c.v1y;
        c.d_goal = distance(o.x, o.y, o.gx, o.gy);                                // game.py:168-169
        c.h_goal = relative_angle(o.x, o.y, o.gx, o.gy);                          // game.py:171-173
        c.d_dev = c.d_goal * m_sin(deg2rad_ref(c.h_goal));                        // game.py:175-180
    }
    return c;
}

// The player side of a freshly drawn episode's first observation (a_lat = 0, heading in [0, 360): the
// one-step-ahead heading of closing_speed() is the heading itself, bit for bit, so its sin / cos are not
// computed twice).  FAST: the three goal terms do not depend on the drawn heading -- the host evaluated them
// (ResetParamsT::d_goal0 / h_goal0 / d_dev0), which takes a square root and an arctangent off the tail of
// every wave that resets an env (65 536 x 8: float32 5.88 -> 5.74 us per launch, float64 12.1 -> 11.5).
// EXACT keeps the reference's operations on the device.
template <typename T, bool FAST, typename R>
__device__ __forceinline__ OwnCtx<T> own_context_fresh(const Params<T>& p, const R& rp, const Own<T>& o) {
    if constexpr (FAST) {
        OwnCtx<T> c;
        c.x = o.x; c.y = o.y; c.v = o.v;
        f_sincos_rev(o.psi * Const<T>::inv360, &c.so, &c.co);
        const T vdt = o.v * p.dt;
        c.v1x = rounded(vdt * c.co); c.v1y = rounded(vdt * c.so);
        c.x1 = o.x + c.v1x; c.y1 = o.y + c.v1y;
        c.d_goal = (T)rp.d_goal0; c.h_goal = (T)rp.h_goal0; c.d_dev = (T)rp.d_dev0;
        scalars(c);
        return c;
    } else {
        return own_context<T, FAST, true>(p, o);
    }
}

// One traffic aircraft, part 1 -- the traffic half of game.py:222-247 action(): wrap the heading,
// sin / cos of it, Euler step when `move`.  In/out: tx, ty (moved), tpsi (wrapped); out: st, ct.
template <typename T, bool FAST>
__device__ __forceinline__ void traffic_move(const Params<T>& p, bool move, T& tx, T& ty, T& tpsi, T tv,
                                             T& st, T& ct) {
    // aircraft.py:16-26 with a_lat = 0: psi = psi % 360, then the Euler step
    tpsi = wrap360<T, FAST>(tpsi);
    if constexpr (FAST) {
        f_sincos_rev(tpsi * Const<T>::inv360, &st, &ct);
        const T tvdt = tv * p.dt;
        if (move) { tx = m_fma(tvdt, ct, tx); ty = m_fma(tvdt, st, ty); }
    } else {
        m_sincos(deg2rad_ref(tpsi), &st, &ct);
        if (move) {
            tx = tx + ((tv * ct) * p.dt);
            ty = ty + ((tv * st) * p.dt);
        }
    }
}

// Part 2a -- the distance to the player (game.py:205 / :185-189): the collision test needs nothing else, so it is
// computed for all of a lane's aircraft BEFORE the rest of their observation entries (see observe(): the outcome of
// the step is known at that point).
template <typename T, bool FAST>
__device__ __forceinline__ T traffic_dist(const OwnCtx<T>& c, T tx, T ty) {
    if constexpr (FAST) {
        const T dx = tx - c.x, dy = ty - c.y;
        return f_sqrt(m_fma(dy, dy, dx * dx));
    } else {
        return distance(c.x, c.y, tx, ty);
    }
}

// Part 2b -- the other two raw (un-normalised) values behind the observation entries of game.py:205-210 for an
// aircraft at (tx, ty), `d` = traffic_dist() away, with heading sin / cos (st, ct).
template <typename T, bool FAST>
__device__ __forceinline__ void traffic_observe(const Params<T>& p, const OwnCtx<T>& c, T tx, T ty, T tv,
                                                T st, T ct, T d, T& dca, T& vc) {
    if constexpr (FAST) {
        const T tvdt = tv * p.dt;
        const T v2x = rounded(tvdt * ct), v2yt = tvdt * st;
        const T dx = tx - c.x, dy = ty - c.y;
        // kinematics.py:40-49: d sin(a_rel - arctan(v12y / v12x))
        //   == sign(v12x) (dy v12x - dx v12y) / |v12|   (sign bit of v12x, so -0.0 counts as
        //   negative like the quotient's; v12 == 0 gives 0 * inf = NaN like the reference's 0/0)
        const T v12x = rounded(c.v * c.co) - rounded(tv * ct), v12y = rounded(c.v * c.so) - rounded(tv * st);
        const T cross = m_fma(dy, v12x, -(dx * v12y));
        dca = (m_copysign(T(1), v12x) * cross) * f_rsq(m_fma(v12y, v12y, v12x * v12x));
        // kinematics.py:52-79; v2.y uses the PLAYER's airspeed (:74), kept
        const T v2y = rounded((c.v * p.dt) * st);
        const T x2 = tx + v2x, y2 = ty + v2yt;
        const T ax = c.v1x - v2x, ay = c.v1y - v2y, bx = c.x1 - x2, by = c.y1 - y2;
        vc = (m_fma(ay, by, ax * bx) * f_rsq(m_fma(by, by, bx * bx))) * p.inv_dt;
    } else {
        // kinematics.py:40-49 distance_closest_approach (arctan of a quotient, signed result)
        const T a_rel_rad = deg2rad_ref(relative_angle(c.x, c.y, tx, ty));
        const T v12x = c.v * c.co - tv * ct, v12y = c.v * c.so - tv * st;
        dca = d * m_sin(a_rel_rad - m_atan(v12y / v12x));
        // kinematics.py:52-79 closing_speed; v2.y uses the PLAYER's airspeed (:74), kept
        const T v2x = (tv * ct) * p.dt, v2y = (c.v * st) * p.dt;
        const T x2 = tx + v2x, y2 = ty + ((tv * st) * p.dt);
        const T ax = c.v1x - v2x, ay = c.v1y - v2y, bx = c.x1 - x2, by = c.y1 - y2;
        vc = (m_fma(ay, by, ax * bx) / distance(c.x1, c.y1, x2, y2)) / p.dt;
    }
}

// Both parts for one aircraft (generic walk, reset path).
template <typename T, bool FAST>
__device__ __forceinline__ void traffic_step(const Params<T>& p, const OwnCtx<T>& c, bool move, T& tx,
                                             T& ty, T& tpsi, T tv, T& d, T& dca, T& vc) {
    T st, ct;
    traffic_move<T, FAST>(p, move, tx, ty, tpsi, tv, st, ct);
    d = traffic_dist<T, FAST>(c, tx, ty);
    traffic_observe<T, FAST>(p, c, tx, ty, tv, st, ct, d, dca, vc);
}

// ---- two traffic aircraft per call (float32 throughput build) ---------------------------------------
// With 2 wavefronts per SIMD at the headline size the step kernel's arithmetic phase is bound by
// VALU ISSUE (409 VALU instructions per wave x 4 cycles x 2 waves), not by latency.  gfx950 has
// packed float32 math -- v_pk_mul / add / fma_f32 do two floats per lane in one issue slot -- and a
// lane's C traffic aircraft run the same arithmetic, so the packed shapes with an even C walk their
// traffic in PAIRS on float2 operands: the same IEEE operations in the same order as the scalar
// functions above (bit-identical per aircraft), half the issue slots for everything that is not a
// transcendental, a compare or a select.
typedef float F2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ F2 m_fma(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ F2 f_sqrt(F2 x) { return F2{f_sqrt(x.x), f_sqrt(x.y)}; }
__device__ __forceinline__ F2 f_rsq(F2 x) { return F2{f_rsq(x.x), f_rsq(x.y)}; }

// traffic_move() in two halves: the heading part (wrap, sin, cos) and the Euler step.  Traffic flies
// straight (a_lat = 0, game.py:232-241), so the heading part gives the same result every step of an
// episode; the fused rollout keeps it in registers (TrigCache) and redoes it only after a reset.
__device__ __forceinline__ void traffic_trig2(F2& tpsi, F2& st, F2& ct) {
    tpsi = F2{wrap360_window(tpsi.x), wrap360_window(tpsi.y)};
    const F2 r = tpsi * Const<float>::inv360;
    st = F2{__builtin_amdgcn_sinf(r.x), __builtin_amdgcn_sinf(r.y)};
    ct = F2{__builtin_amdgcn_cosf(r.x), __builtin_amdgcn_cosf(r.y)};
}
__device__ __forceinline__ void traffic_advance2(const Params<float>& p, bool move, F2& tx, F2& ty, F2 tv, F2 st, F2 ct) {
    const F2 tvdt = tv * p.dt;
    if (move) { tx = m_fma(tvdt, ct, tx); ty = m_fma(tvdt, st, ty); }
}
template <typename T, int C>
struct TrigCache {
    T st[C], ct[C];
    bool valid = false;      // st / ct belong to the headings in the lane's registers
    bool dirty = false;      // a wrap changed a heading that has not been stored yet
};

__device__ __forceinline__ F2 traffic_dist2(const OwnCtx<float>& c, F2 tx, F2 ty) {
    const F2 dx = tx - c.x, dy = ty - c.y;
    return f_sqrt(m_fma(dy, dy, dx * dx));
}
__device__ __forceinline__ void traffic_observe2(const Params<float>& p, const OwnCtx<float>& c, F2 tx, F2 ty, F2 tv,
                                                 F2 st, F2 ct, F2& dca, F2& vc) {
    const F2 tvdt = tv * p.dt;
    const F2 v2x = rounded(tvdt * ct), v2yt = tvdt * st;
    const F2 dx = tx - c.x, dy = ty - c.y;
    const F2 v12x = rounded(c.v * c.co) - rounded(tv * ct), v12y = rounded(c.v * c.so) - rounded(tv * st);
    const F2 cross = m_fma(dy, v12x, -(dx * v12y));
    const F2 sgn = F2{m_copysign(1.0f, v12x.x), m_copysign(1.0f, v12x.y)};
    dca = (sgn * cross) * f_rsq(m_fma(v12y, v12y, v12x * v12x));
    const F2 v2y = rounded((c.v * p.dt) * st);
    const F2 x2 = tx + v2x, y2 = ty + v2yt;
    const F2 ax = c.v1x - v2x, ay = c.v1y - v2y, bx = c.x1 - x2, by = c.y1 - y2;
    vc = (m_fma(ay, by, ax * bx) * f_rsq(m_fma(by, by, bx * bx))) * p.inv_dt;
}

// game.py:199-203: the five player entries of the observation into the LDS row.
template <typename T, bool FAST>
__device__ __forceinline__ void put_own_obs(const Params<T>& p, T* row, int32_t steps, T psi, const OwnCtx<T>& c) {
    if constexpr (FAST) {
        row[0] = (T)steps * p.inv_max_steps;
        row[1] = psi * Const<T>::inv360;
        row[2] = c.d_dev * p.inv_d_dev_max;
        row[3] = c.d_goal * p.inv_d_goal_max;
        row[4] = c.h_goal * Const<T>::inv360;
    } else {
        row[0] = (T)steps / (T)p.max_steps;
        row[1] = psi / T(360);
        row[2] = c.d_dev / p.d_dev_max;
        row[3] = c.d_goal / p.d_goal_max;
        row[4] = c.h_goal / T(360);
    }
}

// game.py:205-210: the three normalised entries of one traffic aircraft into the LDS row.
template <typename T, bool FAST>
__device__ __forceinline__ void put_traffic_obs(const Params<T>& p, T* q, T d, T dca, T vc) {
    if constexpr (FAST) {
        q[0] = d * p.inv_d_sep_max; q[1] = dca * p.inv_d_cpa_max; q[2] = vc * p.inv_v_closing_max;
    } else {
        q[0] = d / p.d_sep_max; q[1] = dca / p.d_cpa_max; q[2] = vc / p.v_closing_max;
    }
}

// The C traffic aircraft of one lane (packed shapes), loaded / stored as 16-byte vectors.
template <typename T, int C>
struct Traffic {
    Vec<T, C> x, y, psi, v;
};
template <typename T, int C>
__device__ __forceinline__ Traffic<T, C> load_traffic(const T* trf_psi, const T* trf_v, const T* trf_x, const T* trf_y, int i0) {
    using V = Vec<T, C>;
    Traffic<T, C> t;
    // headings and speeds first: sin / cos and v dt start while the positions are still landing
    t.psi = *reinterpret_cast<const V*>(trf_psi + i0);
    t.v = *reinterpret_cast<const V*>(trf_v + i0);
    // (non-temporal LOADS of the positions measured slower in round 3: 5.55 against 5.21 us per launch, 9.1 against 7.0 at
    //  131 072 envs -- profiles/r03_ab_partial_double_buffer_nt_loads_shapes.txt)
    t.x = *reinterpret_cast<const V*>(trf_x + i0);
    t.y = *reinterpret_cast<const V*>(trf_y + i0);
    return t;
}
template <typename T, int C>
__device__ __forceinline__ Traffic<T, C> load_traffic(const State<T>& s, int i0) {
    return load_traffic<T, C>(s.trf_psi, s.trf_v, s.trf_x, s.trf_y, i0);
}

// game.py:162-166 minimum_separation() as action() logs it (game.py:236-237): the player has moved, the
// traffic has not yet.  Record rows only (the latching kernels / reset when a trace buffer is given).
template <typename T, int C, int G, bool PACKED>
__device__ __forceinline__ T minimum_separation(const State<T>& s, const Own<T>& o, const Traffic<T, C>& tr, int e,
                                                int j, int N) {
    T m = T(1) / T(0);                                     // float("inf") for an env without traffic
    if constexpr (PACKED) {
#pragma unroll
        for (int k = 0; k < C; ++k) { const T d = distance(o.x, o.y, tr.x.v[k], tr.y.v[k]); m = d < m ? d : m; }
    } else {
        for (int n = j; n < N; n += G) { const T d = distance(o.x, o.y, s.trf_x[e * N + n], s.trf_y[e * N + n]); m = d < m ? d : m; }
    }
#pragma unroll
    for (int w = 1; w < G; w <<= 1) { const T q = __shfl_xor(m, w, 64); m = q < m ? q : m; }
    return m;
}

// game.py:194-220 observe() (+ the traffic half of action() when `move`): every lane computes the
// player-side terms (same-address inputs, identical results); lane j walks its traffic and writes
// the observation entries into the wave's LDS tile row.  `steps` is the incremented counter.
// Packed shapes get their traffic in registers (`tr`, loaded by the caller up front so that all
// of a wave's loads are in flight together) and write the moved block back; the generic walk
// loads / stores aircraft by aircraft.
// `after_own(c)` runs between the player side and the traffic side (the record rows' minimum separation).
struct NoHook { template <typename X> __device__ __forceinline__ void operator()(const X&) const {} };
template <typename T, int C, int G, bool PACKED, bool FAST, typename Hook = NoHook>
__device__ __forceinline__ Seen<T> observe(const Params<T>& p, const State<T>& s, const Own<T>& o,
                                           int e, int j, int N, int32_t steps, bool move,
                                           Traffic<T, C>& tr, T* __restrict__ row, bool store_traffic = true,
                                           TrigCache<T, C>* tc = nullptr, Hook after_own = Hook()) {
    const OwnCtx<T> c = own_context<T, FAST>(p, o);
    // Keep the traffic arithmetic below this line: the player side above needs only the scalars, which
    // were requested first, so it runs under s_waitcnt vmcnt(11..5) while the traffic vectors land.
    if constexpr (PACKED) __builtin_amdgcn_sched_barrier(0);
    after_own(c);
    Seen<T> r;
    r.d_goal = c.d_goal; r.h_goal = c.h_goal; r.d_dev = c.d_dev;
    int coll = 0;
    T vc0 = T(0), dc0 = T(0);
    if constexpr (PACKED) {
        using V = Vec<T, C>;
        const int i0 = e * N + j * C;
        // Move the whole block first and store it at once: the state write-back (half of the
        // kernel's store bytes besides obs) then drains underneath the observation arithmetic
        // instead of joining the write burst at the end of the wave.
        constexpr bool PAIRS = FAST && sizeof(T) == 4 && C % 2 == 0;    // see traffic_trig2() / traffic_observe2()
        bool psi_changed = false;
        T st[C], ct[C];
        if constexpr (PAIRS) {
            if (tc != nullptr && tc->valid) {
#pragma unroll
                for (int k = 0; k < C; ++k) { st[k] = tc->st[k]; ct[k] = tc->ct[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < C; k += 2) {
                    F2 ps{tr.psi.v[k], tr.psi.v[k + 1]}, s2, c2;
                    const F2 ps_in = ps;
                    traffic_trig2(ps, s2, c2);
                    tr.psi.v[k] = ps.x; tr.psi.v[k + 1] = ps.y;
                    st[k] = s2.x; st[k + 1] = s2.y; ct[k] = c2.x; ct[k + 1] = c2.y;
                    psi_changed |= (ps.x != ps_in.x) | (ps.y != ps_in.y);
                }
                if (tc != nullptr) {
#pragma unroll
                    for (int k = 0; k < C; ++k) { tc->st[k] = st[k]; tc->ct[k] = ct[k]; }
                    tc->valid = true;
                }
            }
#pragma unroll
            for (int k = 0; k < C; k += 2) {
                F2 x{tr.x.v[k], tr.x.v[k + 1]}, y{tr.y.v[k], tr.y.v[k + 1]};
                traffic_advance2(p, move, x, y, F2{tr.v.v[k], tr.v.v[k + 1]}, F2{st[k], st[k + 1]}, F2{ct[k], ct[k + 1]});
                tr.x.v[k] = x.x; tr.x.v[k + 1] = x.y; tr.y.v[k] = y.x; tr.y.v[k + 1] = y.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const T psi_in = tr.psi.v[k];
                traffic_move<T, FAST>(p, move, tr.x.v[k], tr.y.v[k], tr.psi.v[k], tr.v.v[k], st[k], ct[k]);
                psi_changed |= (tr.psi.v[k] != psi_in);
            }
        }
        // rollout: a wrapped heading waits in registers for the last step's store
        if (tc != nullptr) { tc->dirty |= psi_changed; psi_changed = tc->dirty; }
        if (move && store_traffic) {
            put_trf<T, C>(s, s.trf_x, i0, tr.x);
            put_trf<T, C>(s, s.trf_y, i0, tr.y);
            if (psi_changed) {                                                  // injected headings >= 360 only
                *reinterpret_cast<V*>(s.trf_psi + i0) = tr.psi;
                if (tc != nullptr) tc->dirty = false;
            }
        }
        T dist[C];
        if constexpr (PAIRS) {
#pragma unroll
            for (int k = 0; k < C; k += 2) {
                const F2 d = traffic_dist2(c, F2{tr.x.v[k], tr.x.v[k + 1]}, F2{tr.y.v[k], tr.y.v[k + 1]});
                coll |= ((d.x < p.collision_dist) | (d.y < p.collision_dist)) ? 1 : 0;   // game.py:185-189
                dist[k] = d.x; dist[k + 1] = d.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                dist[k] = traffic_dist<T, FAST>(c, tr.x.v[k], tr.y.v[k]);
                coll |= (dist[k] < p.collision_dist) ? 1 : 0;                    // game.py:185-189
            }
        }
        // (the step's outcome is decided here: d_goal, the step counter and this OR -- before the closest-approach /
        //  closing-speed arithmetic of the observation)
        r.collided = group_or<G>(coll);
        if constexpr (PAIRS) {
#pragma unroll
            for (int k = 0; k < C; k += 2) {
                F2 dca, vc;
                const F2 d{dist[k], dist[k + 1]};
                traffic_observe2(p, c, F2{tr.x.v[k], tr.x.v[k + 1]}, F2{tr.y.v[k], tr.y.v[k + 1]},
                                 F2{tr.v.v[k], tr.v.v[k + 1]}, F2{st[k], st[k + 1]}, F2{ct[k], ct[k + 1]}, dca, vc);
                const F2 dn = d * p.inv_d_sep_max, cn = dca * p.inv_d_cpa_max, vn = vc * p.inv_v_closing_max;
                T* q = row + 5 + 3 * (j * C + k);                                 // game.py:205-210
                q[0] = dn.x; q[1] = cn.x; q[2] = vn.x; q[3] = dn.y; q[4] = cn.y; q[5] = vn.y;
                if (k == 0) { vc0 = vc.x; dc0 = dca.x; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                T dca, vc;
                traffic_observe<T, FAST>(p, c, tr.x.v[k], tr.y.v[k], tr.v.v[k], st[k], ct[k], dist[k], dca, vc);
                put_traffic_obs<T, FAST>(p, row + 5 + 3 * (j * C + k), dist[k], dca, vc);
                if (k == 0) { vc0 = vc; dc0 = dca; }
            }
        }
    } else {
        for (int n = j; n < N; n += G) {
            const int i = e * N + n;
            T tx = s.trf_x[i], ty = s.trf_y[i], tpsi = s.trf_psi[i];
            const T tv = s.trf_v[i], psi_in = tpsi;
            T d, dca, vc;
            traffic_step<T, FAST>(p, c, move, tx, ty, tpsi, tv, d, dca, vc);
            if (move) {
                put_trf1(s, s.trf_x, i, tx);
                put_trf1(s, s.trf_y, i, ty);
                if (tpsi != psi_in) s.trf_psi[i] = tpsi;
            }
            coll |= (d < p.collision_dist) ? 1 : 0;                          // game.py:185-189
            put_traffic_obs<T, FAST>(p, row + 5 + 3 * n, d, dca, vc);
            if (n == 0) { vc0 = vc; dc0 = dca; }
        }
        r.collided = group_or<G>(coll);
    }
    r.v_closing0 = group_bcast0<G>(vc0);                  // evaluate() reads traffic[0] only,
    r.d_cpa0 = group_bcast0<G>(dc0);                      // game.py:254-255
    if (j == 0) put_own_obs<T, FAST>(p, row, steps, o.psi, c);
    return r;
}

// ACAS2DGame.__init__ reset distribution, game.py:80-116, from one Philox block per ENTITY:
// counter = (env_lo, env_hi, episode, entity) with entity 0 = the player (only its heading is random,
// returned in opsi; game.py:85-92), entity n + 1 = traffic n (game.py:96-116); words: x (bit 31 of it =
// starts_down for traffic 0), y, heading, airspeed factor.  The ONE formulation of the distribution per
// element type: reset(), the in-step auto-reset and the speculative generation all call it, so (seed, global env
// index, episode counter) names the same episode bit for bit whichever path draws it.  float64 draws use
// 32 random bits per uniform; the float32 build draws in float32 from 24 bits (equal to the float64
// evaluation rounded to float32 up to 1-2 ulp, at a fraction of its latency).
template <typename T, typename R>
__device__ __forceinline__ void reset_entity(const R& rp, uint32_t k0, uint32_t k1, uint32_t g_lo,
                                             uint32_t g_hi, uint32_t episode, int ent, T& ox, T& oy, T& opsi,
                                             T& ov) {
#pragma clang fp contract(off)
    const U4 w = philox4x32(U4{g_lo, g_hi, episode, (uint32_t)ent}, k0, k1);
    if constexpr (sizeof(T) == 4) {
        const bool own = ent == 0, first = ent == 1;
        const float down = (float)(w.x >> 31), u_psi = u01f(w.z);
        const float jitter = own ? (float)rp.own_heading_jitter : (float)rp.t0_heading_jitter;
        const float base = own ? (float)rp.own_heading0 : fmaf(down, (float)rp.t0_heading_step, (float)rp.t0_heading_base);
        const float psi_special = wrap360_window(base + uniformf(-jitter, jitter, u_psi));
        opsi = (own || first) ? psi_special : 360.0f * u_psi;
        ox = first ? (float)rp.t0_x : (float)rp.tn_x_max * u01f(w.x);
        oy = first ? fmaf(down, (float)rp.t0_y_span, (float)rp.t0_y_base) : (float)rp.tn_y_max * u01f(w.y);
        ov = uniformf((float)rp.speed_factor_min, (float)rp.speed_factor_max, u01f(w.w)) * (float)rp.airspeed;
        return;
    }
    const double u_psi = u01(w.z);
    const bool own = ent == 0, first = ent == 1;
    const double down = (double)(w.x >> 31);
    // heading: base + jitter * (2u - 1) for the player / traffic 0, 360 u for the rest
    const double jitter = own ? rp.own_heading_jitter : rp.t0_heading_jitter;
    const double base = own ? rp.own_heading0 : (rp.t0_heading_base + (down * rp.t0_heading_step));
    const double psi_special = py_mod360(base + uniform(-jitter, jitter, u_psi));
    const double psi = (own || first) ? psi_special : uniform(0.0, 360.0, u_psi);
    const double x = first ? rp.t0_x : uniform(0.0, rp.tn_x_max, u01(w.x));
    const double y = first ? (rp.t0_y_base + (down * rp.t0_y_span)) : uniform(0.0, rp.tn_y_max, u01(w.y));
    const double v = uniform(rp.speed_factor_min, rp.speed_factor_max, u01(w.w)) * rp.airspeed;
    ox = (T)x; oy = (T)y; opsi = (T)psi; ov = (T)v;
}

// reset() for one env by its owner group: lane j draws its own traffic aircraft (and, redundantly, the
// player's heading), stores the new state and returns the player.
template <typename T, int C, int G, bool PACKED, typename R>
__device__ __forceinline__ Own<T> reset_env(const R& rp, const State<T>& s, uint32_t k0,
                                            uint32_t k1, uint64_t gid, uint32_t episode, int e,
                                            int j, int N, Traffic<T, C>& tr) {
    const uint32_t g_lo = (uint32_t)gid, g_hi = (uint32_t)(gid >> 32);
    T ux, uy, psi_own, uv;
    reset_entity<T, R>(rp, k0, k1, g_lo, g_hi, episode, 0, ux, uy, psi_own, uv);
    const Own<T> o{(T)rp.own_x0, (T)rp.own_y0, psi_own, (T)rp.own_v, T(0), (T)rp.goal_x, (T)rp.goal_y};
    if constexpr (PACKED) {
        using V = Vec<T, C>;
#pragma unroll
        for (int k = 0; k < C; ++k)
            reset_entity<T, R>(rp, k0, k1, g_lo, g_hi, episode, 1 + j * C + k, tr.x.v[k], tr.y.v[k], tr.psi.v[k], tr.v.v[k]);
        const int i0 = e * N + j * C;
        *reinterpret_cast<V*>(s.trf_x + i0) = tr.x;
        *reinterpret_cast<V*>(s.trf_y + i0) = tr.y;
        *reinterpret_cast<V*>(s.trf_psi + i0) = tr.psi;
        *reinterpret_cast<V*>(s.trf_v + i0) = tr.v;
    } else {
        for (int n = j; n < N; n += G) {
            const int i = e * N + n;
            T x, y, psi, v;
            reset_entity<T, R>(rp, k0, k1, g_lo, g_hi, episode, n + 1, x, y, psi, v);
            s.trf_x[i] = x; s.trf_y[i] = y; s.trf_psi[i] = psi; s.trf_v[i] = v;
        }
    }
    if (j == 0) {
        s.own_x[e] = o.x; s.own_y[e] = o.y; s.own_psi[e] = o.psi; s.own_v[e] = o.v;
        s.goal_x[e] = o.gx; s.goal_y[e] = o.gy;
    }
    return o;
}

// Wave-cooperative reset of ONE finished env inside the step kernel (SB3 DummyVecEnv.step_wait
// semantics).  The owner group alone would run 1 + C Philox blocks and a whole observe()
// back to back while the rest of the chip waits for this wave (the kernel ends with its slowest
// wave); instead all 64 lanes of the wave take one ENTITY each -- lane 0 the player, lane n the
// traffic aircraft n-1 (strided by 64 beyond that) -- so the new episode costs one Philox block
// and one traffic_step() of latency.  Must be called by the whole wave (wave-uniform arguments).
//
// HANDOFF (packed shapes): the new state goes to the wave's LDS `scratch` (x[N] y[N] psi[N] v[N]
// own_psi) and nothing but term_obs is stored to memory here -- the owner lanes pick the state up
// and it leaves with the wave's ordinary, coalesced state stores and its single tile flush.  Without
// it (generic walk) the entity lanes store the new state themselves and the caller re-flushes the row.
template <typename T, bool FAST, int NS, bool HANDOFF, typename R>
__device__ __forceinline__ void wave_reset_env(const Params<T>& p, const R& rp, const State<T>& s,
                                               const StepIO<T>& io, uint32_t k0, uint32_t k1, uint64_t gid,
                                               int e, int N_dyn, int lane, T total, int32_t steps,
                                               uint32_t episode_prev, T* __restrict__ row,
                                               T* __restrict__ scratch) {
    const int N = NS > 0 ? NS : N_dyn;                   // compile-time for packed shapes
    const int D = 5 + 3 * N;
#ifdef ACAS2D_STAMPS
    const int64_t wave_dbg = remap_block() * kWavesPerBlock + (threadIdx.x >> 6);
#endif
    ACAS2D_STAMP(8, wave_dbg, lane, false);
    // The finished episode's last observation: read from the row now, stored after the Philox block
    // (the LDS round trip of the read then hides under it instead of standing in front of it).
    constexpr bool kOneTermPass = NS > 0 && 5 + 3 * NS <= 64;
    T term_v = T(0);
    if (io.term_obs) {
        if constexpr (kOneTermPass) { if (lane < D) term_v = row[lane]; }
        else { T* t_row = io.term_obs + e * D; for (int i = lane; i < D; i += 64) t_row[i] = row[i]; }
    }
    const uint32_t episode = episode_prev + 1u;
    wave_lds_fence();                                    // row reads precede its rewrite below
    const uint32_t g_lo = (uint32_t)gid, g_hi = (uint32_t)(gid >> 32);

    // entity `lane`: the player (lane 0) or traffic lane-1; further traffic in strides of 64.
    // ONE Philox block per lane (a divergent player / traffic split would run two back to back).
    T tx = T(0), ty = T(0), tpsi = T(0), tv = T(0), psi_own = T(0);
    if (lane <= N) {
        reset_entity<T, R>(rp, k0, k1, g_lo, g_hi, episode, lane, tx, ty, tpsi, tv);
        if (lane == 0) {
            psi_own = tpsi;
            if constexpr (HANDOFF) scratch[4 * N] = tpsi;
        } else if constexpr (HANDOFF) {                  // owner lanes pick the state up from LDS
            const int n = lane - 1;
            scratch[n] = tx; scratch[N + n] = ty; scratch[2 * N + n] = tpsi; scratch[3 * N + n] = tv;
        } else {
            const int i = e * N + (lane - 1);
            s.trf_x[i + s.w_trf] = tx; s.trf_y[i + s.w_trf] = ty; s.trf_psi[i] = tpsi; s.trf_v[i] = tv;
        }
    }
    if constexpr (NS == 0 || NS > 63) {
        for (int n = lane + 63; n < N; n += 64) {        // N > 63 only
            T x, y, ps, v;
            reset_entity<T, R>(rp, k0, k1, g_lo, g_hi, episode, n + 1, x, y, ps, v);
            if constexpr (HANDOFF) {
                scratch[n] = x; scratch[N + n] = y; scratch[2 * N + n] = ps; scratch[3 * N + n] = v;
            } else {
                const int i = e * N + n;
                s.trf_x[i + s.w_trf] = x; s.trf_y[i + s.w_trf] = y; s.trf_psi[i] = ps; s.trf_v[i] = v;
            }
        }
    }
    ACAS2D_STAMP(9, wave_dbg, lane, false);
    if constexpr (kOneTermPass) { if (io.term_obs && lane < D) (io.term_obs + e * D)[lane] = term_v; }
    psi_own = lane_value(psi_own, 0);
    const Own<T> o{(T)rp.own_x0, (T)rp.own_y0, psi_own, (T)rp.own_v, T(0), (T)rp.goal_x, (T)rp.goal_y};
    const OwnCtx<T> c = own_context_fresh<T, FAST>(p, rp, o);

    ACAS2D_STAMP(10, wave_dbg, lane, false);
    // environment.py:44-48: the new episode's first observation (steps becomes 1)
    if (lane >= 1 && lane <= N) {
        T d, dca, vc;
        traffic_step<T, FAST>(p, c, false, tx, ty, tpsi, tv, d, dca, vc);
        put_traffic_obs<T, FAST>(p, row + 5 + 3 * (lane - 1), d, dca, vc);
    }
    if constexpr (NS == 0 || NS > 63) {
        for (int n = lane + 63; n < N; n += 64) {
            T x, y, ps, v;                                         // written by this lane above
            if constexpr (HANDOFF) { x = scratch[n]; y = scratch[N + n]; ps = scratch[2 * N + n]; v = scratch[3 * N + n]; }
            else { const int i = e * N + n; x = s.trf_x[i + s.w_trf]; y = s.trf_y[i + s.w_trf]; ps = s.trf_psi[i]; v = s.trf_v[i]; }
            T d, dca, vc;
            traffic_step<T, FAST>(p, c, false, x, y, ps, v, d, dca, vc);
            put_traffic_obs<T, FAST>(p, row + 5 + 3 * n, d, dca, vc);
        }
    }
    if (lane == 0) {
        if constexpr (!HANDOFF) {
            if (io.ep_return) io.ep_return[e] = total;
            if (io.ep_steps) io.ep_steps[e] = steps;
            s.episode[e] = episode;
            put_env(s, s.own_x, e, o.x); put_env(s, s.own_y, e, o.y); put_env(s, s.own_psi, e, o.psi); s.own_v[e] = o.v;
            s.goal_x[e] = o.gx; s.goal_y[e] = o.gy;
            put_env(s, s.steps, e, (int32_t)1);                           // environment.py:47
            put_env(s, s.total_reward, e, T(0));
        }
        put_own_obs<T, FAST>(p, row, 1, o.psi, c);
    }
    ACAS2D_STAMP(11, wave_dbg, lane, false);
}

// Several finished envs of one wave reset SIDE BY SIDE (packed shapes with N + 1 <= 32).  A wave that
// holds two finished envs used to reset them one after the other -- and with ~190 resets per step over
// 2 048 waves there is such a wave in almost every launch, so the kernel ended two reset chains after
// everybody else (limiting the resets to one per wave and step, as an experiment: 6.02 -> 5.48 us).
// The 64 lanes are cut into SLOTS = 64 / STRIDE slots of STRIDE = pow2 >= N + 1 lanes; slot k takes the
// k-th finished env, lane `ent` of a slot the entity `ent` (0 the player, n + 1 traffic n): the same
// per-lane functions as wave_reset_env(), hence the same bits, for up to SLOTS envs in the time of one.
// Each slot leaves its state in its own 4N+1-value scratch and the first observation in its env's row.
template <int NS> struct ResetSlots {
    static constexpr int ENT = NS + 1;
    static constexpr int STRIDE = ENT <= 2 ? 2 : ENT <= 4 ? 4 : ENT <= 8 ? 8 : ENT <= 16 ? 16 : ENT <= 32 ? 32 : 64;
    static constexpr int SLOTS = 64 / STRIDE;
};
// One reset slot in the wave's LDS, in values of T: the new traffic block x[N] y[N] psi[N] v[N] (each 16-byte
// aligned for the packed shapes' vector reads) and the player's heading.  Slots start 16-byte aligned.
template <typename T, int NS> struct SlotLayout {
    static constexpr int W = 16 / (int)sizeof(T);
    static constexpr int OWN_PSI = 4 * NS;
    static constexpr int STRIDE = (4 * NS + 1 + W - 1) / W * W;
};

// Resets the envs named by the lowest min(SLOTS, popcount(dm)) bits of `dm` (bit = lane of the env's
// group leader, i.e. el * G); returns the mask of those bits.  Owner lanes find their slot as the rank of
// their bit in the returned mask.  Whole wave, wave-uniform arguments.
template <typename T, bool FAST, int NS, int G, typename R>
__device__ __forceinline__ unsigned long long wave_reset_slots(const Params<T>& p, const R& rp,
                                                               const StepIO<T>& io, uint32_t k0, uint32_t k1,
                                                               uint64_t gid_wave, unsigned long long dm, int lane,
                                                               uint32_t episode_lane, T* __restrict__ tile,
                                                               T* __restrict__ scratch) {
    using RS = ResetSlots<NS>;
    constexpr int N = NS, D = 5 + 3 * NS, SCR = SlotLayout<T, NS>::STRIDE;
    const int slot = lane / RS::STRIDE, ent = lane % RS::STRIDE;
    // slot k <- the k-th set bit of dm
    unsigned long long taken = 0;
    int src = 0;                                          // the lane of my slot's env's group leader
    uint32_t episode_prev = 0;
    bool have = false;
    for (int k = 0; k < RS::SLOTS && dm != 0; ++k) {       // wave-uniform, usually one or two trips
        const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)dm) - 1);
        dm &= dm - 1;
        taken |= 1ull << b;
        const uint32_t ep_b = (uint32_t)lane_value((int)episode_lane, b);    // v_readlane: b is wave-uniform
        if (slot == k) { src = b; have = true; episode_prev = ep_b; }
    }
    const int e = src / G;                                // my slot's env (in the wave)
    T* row = tile + e * D;
    T* scr = scratch + slot * SCR;
    // the finished episodes' last observations: LDS reads now, stores after the Philox block
    constexpr int TERM_PASSES = (D + RS::STRIDE - 1) / RS::STRIDE;
    T term_v[TERM_PASSES];
    if (io.term_obs) {
#pragma unroll
        for (int q = 0; q < TERM_PASSES; ++q) {
            const int i = ent + q * RS::STRIDE;
            term_v[q] = (have && i < D) ? row[i] : T(0);
        }
    }
    const uint32_t episode = episode_prev + 1u;
    wave_lds_fence();                                     // row reads precede their rewrite below
    const uint64_t gid = gid_wave + (uint64_t)e;
    T tx = T(0), ty = T(0), tpsi = T(0), tv = T(0);
    const bool mine = have && ent <= N;
    if (mine) {
        reset_entity<T, R>(rp, k0, k1, (uint32_t)gid, (uint32_t)(gid >> 32), episode, ent, tx, ty, tpsi, tv);
        if (ent == 0) scr[4 * N] = tpsi;
        else { const int n = ent - 1; scr[n] = tx; scr[N + n] = ty; scr[2 * N + n] = tpsi; scr[3 * N + n] = tv; }
    }
    if (io.term_obs) {
#pragma unroll
        for (int q = 0; q < TERM_PASSES; ++q) {
            const int i = ent + q * RS::STRIDE;
            if (have && i < D) (io.term_obs + e * D)[i] = term_v[q];
        }
    }
    // every lane of a slot needs its player's heading (lane 0 of the slot): a 16-lane slot is one DPP
    // row (row_newbcast:0), smaller slots sit inside a quad; otherwise through the slot's scratch
    T psi_own;
    if constexpr (sizeof(T) == 4 && RS::STRIDE == 16)
        psi_own = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int((float)tpsi), 0x150, 0xf, 0xf, false));
    else if constexpr (sizeof(T) == 4 && RS::STRIDE == 4)
        psi_own = __int_as_float(quad_perm<0x00>(__float_as_int((float)tpsi)));
    else if constexpr (sizeof(T) == 4 && RS::STRIDE == 2)
        psi_own = __int_as_float(quad_perm<0xA0>(__float_as_int((float)tpsi)));
    else {
        wave_lds_fence();                                 // the player's heading of every slot is in its scratch
        psi_own = have ? scr[4 * N] : T(0);
    }
    const Own<T> o{(T)rp.own_x0, (T)rp.own_y0, psi_own, (T)rp.own_v, T(0), (T)rp.goal_x, (T)rp.goal_y};
    const OwnCtx<T> c = own_context_fresh<T, FAST>(p, rp, o);
    // environment.py:44-48: the new episodes' first observations (steps becomes 1)
    if (mine) {
        if (ent >= 1) {
            T d, dca, vc;
            traffic_step<T, FAST>(p, c, false, tx, ty, tpsi, tv, d, dca, vc);
            put_traffic_obs<T, FAST>(p, row + 5 + 3 * (ent - 1), d, dca, vc);
        } else {
            put_own_obs<T, FAST>(p, row, 1, o.psi, c);
        }
    }
    return taken;
}

// Flush the wave's LDS tile (`count` values, the contiguous slice dst[0 .. count) of obs[E][D])
// with lane-linear stores: 16 bytes per lane where the slice is 16-byte aligned, else one value.
// Chunk c (16 bytes, or one value on the unaligned path) is always written by lane c % 64, so a
// later flush_rows() of the same tile rewrites every address from the SAME work-item (program
// order, no cross-lane store ordering assumed).
// One 16-byte observation chunk out, non-temporal (see "store policies").
template <typename T, int W>
__device__ __forceinline__ void store_chunk(Vec<T, W>* dst, const Vec<T, W>& v) {
    static_assert((W & (W - 1)) == 0, "a 3-vector type is padded to 4 elements");
    typedef T NV __attribute__((ext_vector_type(W)));
    NV x;
#pragma unroll
    for (int k = 0; k < W; ++k) x[k] = v.v[k];
    __builtin_nontemporal_store(x, reinterpret_cast<NV*>(dst));
}

template <typename T>
__device__ __forceinline__ void flush_tile(const T* __restrict__ tile, T* __restrict__ dst, int count,
                                           int lane) {
    constexpr int W = 16 / sizeof(T);
    using V = Vec<T, W>;
    if (__builtin_expect((reinterpret_cast<uintptr_t>(dst) & 15u) == 0, 1)) {
        const int nv = count / W;
        for (int i = lane; i < nv; i += 64)
            store_chunk<T, W>(reinterpret_cast<V*>(dst) + i, reinterpret_cast<const V*>(tile)[i]);
        for (int i = nv * W + lane; i < count; i += 64) dst[i] = tile[i];
    } else {
        for (int i = lane; i < count; i += 64) dst[i] = tile[i];
    }
}

// flush_tile() with the trip count known at compile time (packed shapes: at most CHUNKS 16-byte
// chunks per lane): ALL LDS reads are issued before the first store.  The rolled loop above reads,
// waits for LDS, stores, and only then reads the next chunk -- four LDS round trips back to back at
// the very end of every wave.
template <typename T, int CHUNKS>
__device__ __forceinline__ void flush_tile_unrolled(const T* __restrict__ tile, T* __restrict__ dst, int count,
                                                    int lane) {
    constexpr int W = 16 / sizeof(T);
    using V = Vec<T, W>;
    if (__builtin_expect((reinterpret_cast<uintptr_t>(dst) & 15u) == 0, 1)) {
        const int nv = count / W;
        V buf[CHUNKS];
#pragma unroll
        for (int u = 0; u < CHUNKS; ++u) {
            const int i = lane + 64 * u;
            if (i < nv) buf[u] = reinterpret_cast<const V*>(tile)[i];
        }
#pragma unroll
        for (int u = 0; u < CHUNKS; ++u) {
            const int i = lane + 64 * u;
            if (i < nv) store_chunk<T, W>(reinterpret_cast<V*>(dst) + i, buf[u]);
        }
        for (int i = nv * W + lane; i < count; i += 64) dst[i] = tile[i];
    } else {
        for (int i = lane; i < count; i += 64) dst[i] = tile[i];
    }
}

// Re-flush the values [first, last) of the tile (one env's row after its reset) with exactly the
// chunk -> lane mapping of flush_tile().
template <typename T>
__device__ __forceinline__ void flush_rows(const T* __restrict__ tile, T* __restrict__ dst, int count,
                                           int first, int last, int lane) {
    constexpr int W = 16 / sizeof(T);
    using V = Vec<T, W>;
    if (__builtin_expect((reinterpret_cast<uintptr_t>(dst) & 15u) == 0, 1)) {
        const int nv = count / W;
        const int c0 = first / W, c1 = (last + W - 1) / W;           // chunks touching the row
        for (int c = c0 + ((lane - c0) & 63); c < c1 && c < nv; c += 64)
            reinterpret_cast<V*>(dst)[c] = reinterpret_cast<const V*>(tile)[c];
        for (int i = nv * W + lane; i < last; i += 64)               // scalar tail of the tile
            if (i >= first) dst[i] = tile[i];
    } else {
        for (int i = first + ((lane - first) & 63); i < last; i += 64) dst[i] = tile[i];
    }
}

// ---- the policy inside the rollout (testing_main.py:69-105's loop in one launch) -----------
// SB3 1.1.0 MlpPolicy actor: obs -> Linear(D,64) tanh -> Linear(64,64) tanh -> Linear(64,1), the
// deterministic action = clip(mean, -1, 1) (policies.py predict()).  One lane per env (G == 1
// shapes): the lane keeps its 64 + 64 hidden activations in registers as 32 + 32 float2 accumulators;
// the weights are wave-uniform, so they stream through SGPRs from the constant address space
// (s_load_dwordx16) straight into v_pk_fma_f32's scalar operand -- 2 368 packed FMAs per env-step
// at D = 8, no LDS, no vector loads.  Weights arrive TRANSPOSED ([in][out], row-major) so that one
// input's 64 outgoing weights are contiguous.  float32 math in both builds (the reference's
// policy.predict() runs its float32 torch module on float32-cast observations).
struct PolicyW {
    const float *w1t, *b1, *w2t, *b2, *w3, *b3;      // actor: [D][64], [64], [64][64], [64], [64], [1]
    void* actions_out;                               // T[n_steps][E]: the action each step took
    const void* obs_in;                              // T[E][D]: the observation the first action is taken on
    // SAMPLE (the collector of a PPO iteration, SB3 collect_rollouts): the value net (same layout), the
    // state-independent log-std, per-step value / log-probability outputs, the key and step of the noise stream
    const float *v1t, *vb1, *v2t, *vb2, *v3, *vb3;
    const float* log_std;                            // [1]
    void *values_out, *logp_out;                     // T[n_steps][E]
    uint32_t nk0, nk1, noise_step;
};
constexpr int kPolicyHidden = 64;

// tanh(x) = 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32: abs error < 2e-7 over the reals
// (saturates cleanly: exp -> inf gives 1, exp -> 0 gives -1).
__device__ __forceinline__ float tanh_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);          // 2 log2(e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// One SB3 MlpPolicy head, obs -> Linear(D,64) tanh -> Linear(64,64) tanh -> Linear(64,1): the actor's mean or the
// critic's value, by the weights handed in.
template <int D>
__device__ __forceinline__ float policy_mlp(const float* w1t_, const float* b1_, const float* w2t_, const float* b2_,
                                            const float* w3_, const float* b3_, const float (&x)[D]) {
    constexpr int H2 = kPolicyHidden / 2;
    const F2 ACAS2D_AS4* w1 = (const F2 ACAS2D_AS4*)w1t_;
    const F2 ACAS2D_AS4* b1 = (const F2 ACAS2D_AS4*)b1_;
    const F2 ACAS2D_AS4* w2 = (const F2 ACAS2D_AS4*)w2t_;
    const F2 ACAS2D_AS4* b2 = (const F2 ACAS2D_AS4*)b2_;
    const F2 ACAS2D_AS4* w3 = (const F2 ACAS2D_AS4*)w3_;
    const float ACAS2D_AS4* b3 = (const float ACAS2D_AS4*)b3_;
    F2 h[H2], g[H2];
#pragma unroll
    for (int i = 0; i < H2; ++i) h[i] = b1[i];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const F2 xk = F2{x[k], x[k]};
#pragma unroll
        for (int i = 0; i < H2; ++i) h[i] = __builtin_elementwise_fma(w1[k * H2 + i], xk, h[i]);
    }
#pragma unroll
    for (int i = 0; i < H2; ++i) { h[i] = F2{tanh_hw(h[i].x), tanh_hw(h[i].y)}; g[i] = b2[i]; }
#pragma unroll
    for (int k = 0; k < kPolicyHidden; ++k) {
        const float hk = (k & 1) ? h[k / 2].y : h[k / 2].x;
        const F2 xk = F2{hk, hk};
#pragma unroll
        for (int i = 0; i < H2; ++i) g[i] = __builtin_elementwise_fma(w2[k * H2 + i], xk, g[i]);
    }
    F2 acc = F2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < H2; ++i) acc = __builtin_elementwise_fma(w3[i], F2{tanh_hw(g[i].x), tanh_hw(g[i].y)}, acc);
    return (acc.x + acc.y) + b3[0];
}
// the deterministic action = clip(mean, -1, 1) (SB3 policies.py predict())
template <int D>
__device__ __forceinline__ float policy_action(const PolicyW& pw, const float (&x)[D]) {
    return fminf(fmaxf(policy_mlp<D>(pw.w1t, pw.b1, pw.w2t, pw.b2, pw.w3, pw.b3, x), -1.0f), 1.0f);
}

// ---- kernels ------------------------------------------------------------------------------------------
// ACAS2DEnv.step(), environment.py:29-42 -- and, with ROLLOUT, n_steps of them fused in one launch:
// the state stays in registers, step t reads actions[t][E] and writes obs[t][E][D], reward[t][E],
// done[t][E], outcome[t][E] (and the optional auto-reset side channels [t][E]...), finished envs
// are reset on the fly (ROLLOUT implies AUTO_RESET semantics and a packed shape).  The per-step
// arithmetic is this same code, so rollout(T) == T x step() bit for bit.
// With POLICY (rollout, one lane per env) the action of every step comes from policy_action() on the
// previous observation instead of from actions[t][E], which becomes an output.
// With SAMPLE on top (the collector of a PPO iteration, SB3's collect_rollouts) the action is drawn: mean + exp(log_std)
// eps, eps ~ N(0, 1) from a Philox block per env and step (Box-Muller); the raw action, the critic's value of the
// observation and the log-probability of the draw are stored per step, the env is stepped with the clipped action, and
// a non-finite observation entry reaches the networks as 0 (the reference's NaN d_cpa in exact parallel flight).
template <typename T, int C, int G, bool PACKED, bool AUTO_RESET, bool FAST, bool ROLLOUT, bool POLICY = false,
          bool SAMPLE = false, bool ARENA = false>
__global__ __launch_bounds__(kBlock) void step_kernel(const T* a0, const T* a1, const T* a2, const T* a3, const T* a4,
                                                      const T* a5, int32_t e_n_envs, int32_t tile_elems,
                                                      Params<T> p_arg, StepResetParams<T, ROLLOUT> rp_arg, State<T> s_arg,
                                                      StepIO<T> io_arg, uint32_t k0, uint32_t k1,
                                                      int64_t env_offset, int N_arg, int n_steps, PolicyW pw) {
    static_assert(!ROLLOUT || (AUTO_RESET && PACKED), "rollout: auto-reset semantics, packed shapes");
    static_assert(!POLICY || (ROLLOUT && G == 1), "in-kernel policy: rollout mode, one lane per env");
    static_assert(!SAMPLE || POLICY, "sampling needs the in-kernel policy");
    constexpr int NS = PACKED ? C * G : 0;         // packed shapes: n_traffic is a compile-time constant
    const int N = PACKED ? NS : N_arg;
    constexpr int EPW = 64 / G;                    // envs per wavefront
    // The FIRST 14 dwords of the kernel arguments (a0 .. a5, the env count, the tile size) are preloaded into SGPRs by the
    // command processor (gfx950 kernarg preload, KFLAGS in the Makefile): loads through them leave with the wave's first
    // instructions instead of behind a scalar-load round trip to the argument segment (~0.27 us, all waves of a launch
    // miss together).  The grid size follows from the env count (geometry_for), so nothing of the launch geometry is
    // fetched either.  Six pointers do not name fourteen input arrays, so:
    //   ARENA   (launch_step_impl found the state laid out as consecutive [k][E] rows, as the Python host allocates it)
    //           a0 = own_x (then own_y, own_psi, total_reward, steps), a1 = own_v (then goal_x, goal_y, episode),
    //           a2 = trf_x (then trf_y), a3 = trf_psi (then trf_v), a4 = actions: EVERY load leaves at once;
    //   else    a0 .. a5 = trf_x, trf_y, trf_psi, trf_v, own_x, own_y: the bulk of the bytes leaves at once, the rest
    //           behind the round trip.
    static_assert(!ARENA || (sizeof(T) == 4 && PACKED && AUTO_RESET && !ROLLOUT), "arena launches: the float32 per-step auto-reset kernel");
    // A wavefront issues one instruction per four cycles whatever its kind, and every instruction in front of the first
    // load is on the launch's critical path: env indices are 32-bit here (n_envs < 2^31, geometry_for).
    const uint32_t E32 = (uint32_t)e_n_envs;
    const int lane = threadIdx.x & 63;
    const int j = lane & (G - 1), el = lane / G;   // lane in group, env in wave
    const int wib = wave_in_block();
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr uint32_t kEnvsPerBlock = EPW * kWavesPerBlock;
    // Arena launches come in whole multiples of eight workgroups (launch_step_impl): every wave is full and the XCD remap
    // unconditional -- no bounds check, no clamped load index, no lane guards on the stores, fifteen instructions less
    // in front of the first load.
    constexpr bool FULL = ARENA;
    const uint32_t wave32 = (FULL ? (blockIdx.x & 7u) * (E32 / (kEnvsPerBlock * 8u)) + (blockIdx.x >> 3)
                                  : remap_block_of((E32 + (kEnvsPerBlock - 1u)) / kEnvsPerBlock)) * kWavesPerBlock + (uint32_t)wib;
    const uint32_t e_wave32 = wave32 * EPW;        // first env of this wave (scalar); < 2^31 + 128
    if constexpr (!FULL) { if (e_wave32 >= E32) return; }     // whole wave idle
    const int D = 5 + 3 * N;
    const int n_rows = FULL ? EPW : (int)((E32 - e_wave32) < (uint32_t)EPW ? (E32 - e_wave32) : (uint32_t)EPW);
    const bool active = FULL ? true : el < n_rows; // whole groups are active or not
    const int64_t n_envs = E32, e_wave = e_wave32, wave = wave32;
    (void)wave;
    // per wave: the observation tile, then (HANDOFF) the reset slots (SlotLayout) / one 4N+1-value scratch
    constexpr bool HANDOFF = PACKED && AUTO_RESET;   // finished envs are reset BEFORE the wave's stores
    T* tile = reinterpret_cast<T*>(lds_raw) + wib * tile_elems;
    T* row = tile + el * D;
    T* scratch = HANDOFF ? tile + (EPW * D + 3) / 4 * 4 : nullptr;     // 16-byte aligned within the tile allocation

    ACAS2D_STAMP(0, wave, lane, false);
    ACAS2D_STAMP(1, wave, lane, false);
    int32_t steps = 0;
    uint32_t episode = 0;
    T total = T(0);
    Own<T> o{};
    Traffic<T, C> tr{};
    bool frozen = false;
    T action_next = T(0);                                  // step t+1's action is fetched during step t
    // Packed shapes run loads and arithmetic on EVERY lane of the wave -- the lanes past the last env
    // (last wave only) on a copy of the wave's first env -- and predicate only the stores.  With the
    // loads and their first uses inside `if (active)`, the compiler's waitcnt model kept them pending
    // on the path around it and, vmcnt retiring in order, made every later loop and LDS read of the
    // kernel wait for all stores issued in between (s_waitcnt vmcnt(0) before the tile flush).
    const bool run = PACKED ? true : active;
    const int el_l = (PACKED && !active) ? 0 : el;         // the env a lane LOADS
    // ---- every load of this lane up front, all requests in flight at once; the loads through the preloaded pointers
    // leave in front of the wait for the other arguments ...
    State<T> s_in = s_arg;                                 // (fields nobody reads cost nothing)
    const T* act_in = io_arg.actions;
    if constexpr (ARENA) {
        act_in = a4;                                       // (the state pointers the STORES use: behind the loads, below)
    } else {
        s_in.trf_x = const_cast<T*>(a0); s_in.trf_y = const_cast<T*>(a1); s_in.trf_psi = const_cast<T*>(a2);
        s_in.trf_v = const_cast<T*>(a3); s_in.own_x = const_cast<T*>(a4); s_in.own_y = const_cast<T*>(a5);
    }
    // only the per-step auto-reset launch takes a second state generation (launch_step_impl); elsewhere the offsets are
    // compile-time zeros and cost no registers
    if constexpr (ROLLOUT || !AUTO_RESET) { s_in.w_env = 0; s_in.w_trf = 0; }
    T ox = T(0), oy = T(0);
    if constexpr (ARENA) {
        // Unmodified preloaded bases + 32-bit per-lane byte offsets (launch_step_impl checked that they fit): one VALU add
        // per array instead of four scalar ones.  The player's scalars and the action first, the traffic vectors (the
        // bulk) last: loads return in order, so the player-side arithmetic can start while the vectors are landing.
        if (run) {
            const uint32_t ie = e_wave32 + (uint32_t)el_l;                 // this lane's env
            const auto at = [](const T* base, uint32_t elem) {
                return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(elem * 4u));
            };
            ox = *at(a0, ie); oy = *at(a0, ie + E32);
            o = Own<T>{ox, oy, *at(a0, ie + 2u * E32), *at(a1, ie), T(0), *at(a1, ie + E32), *at(a1, ie + 2u * E32)};
            steps = *reinterpret_cast<const int32_t*>(at(a0, ie + 4u * E32));
            total = *at(a0, ie + 3u * E32);
            episode = *reinterpret_cast<const uint32_t*>(at(a1, ie + 3u * E32));
            action_next = *at(a4, ie);
            using V = Vec<T, C>;
            const uint32_t it = ie * (uint32_t)N + (uint32_t)(j * C), EN32 = E32 * (uint32_t)N;
            tr.psi = *reinterpret_cast<const V*>(at(a3, it));
            tr.v = *reinterpret_cast<const V*>(at(a3, it + EN32));
            tr.x = *reinterpret_cast<const V*>(at(a2, it));
            tr.y = *reinterpret_cast<const V*>(at(a2, it + EN32));
        }
    } else if (run) {
        if constexpr (PACKED)
            tr = load_traffic<T, C>(s_in.trf_psi + e_wave * N, s_in.trf_v + e_wave * N, s_in.trf_x + e_wave * N,
                                    s_in.trf_y + e_wave * N, el_l * N + j * C);
        ox = (s_in.own_x + e_wave)[el_l]; oy = (s_in.own_y + e_wave)[el_l];
    }
    __builtin_amdgcn_sched_barrier(0);                     // ... and stay there: nothing below is scheduled above them
    // everything else the loads (ARENA: the stores) need from the kernel arguments, requested in ONE scalar-load round
    // trip (left alone, hipcc fetches them one by one, each behind its own s_waitcnt)
    if constexpr (ARENA)
        asm volatile("" :: "s"(s_arg.w_env), "s"(s_arg.w_trf), "s"(io_arg.obs), "s"(io_arg.reward), "s"(io_arg.done),
                     "s"(io_arg.outcome) : "memory");
    else
        asm volatile("" :: "s"(s_arg.own_psi), "s"(s_arg.own_v), "s"(s_arg.goal_x), "s"(s_arg.goal_y),
                     "s"(s_arg.steps), "s"(s_arg.total_reward), "s"(s_arg.episode), "s"(io_arg.actions),
                     "s"(s_arg.w_env), "s"(s_arg.w_trf) : "memory");
    if constexpr (ARENA) {
        const int64_t E = n_envs, EN = n_envs * N;
        T* m = const_cast<T*>(a0);
        T* c = const_cast<T*>(a1);
        s_in.own_x = m; s_in.own_y = m + E; s_in.own_psi = m + 2 * E; s_in.total_reward = m + 3 * E;
        s_in.steps = reinterpret_cast<int32_t*>(m + 4 * E);
        s_in.own_v = c; s_in.goal_x = c + E; s_in.goal_y = c + 2 * E; s_in.episode = reinterpret_cast<uint32_t*>(c + 3 * E);
        s_in.trf_x = const_cast<T*>(a2); s_in.trf_y = s_in.trf_x + EN;
        s_in.trf_psi = const_cast<T*>(a3); s_in.trf_v = s_in.trf_psi + EN;
        s_in.trace = nullptr;
    }
    const State<T> s = rebase(s_in, e_wave, N);            // everything below indexes envs by `el`
    StepIO<T> io_in = io_arg;
    io_in.actions = act_in;
    const StepIO<T> io0 = rebase(io_in, e_wave, D);
    if constexpr (!ARENA) {
        if (run) {
            o = Own<T>{ox, oy, s.own_psi[el_l], s.own_v[el_l], T(0), s.goal_x[el_l], s.goal_y[el_l]};
            steps = s.steps[el_l];
            total = s.total_reward[el_l];
            if constexpr (AUTO_RESET) episode = s.episode[el_l];
            else frozen = s.status[el_l] != 0;                             // game.py:243-245
            if constexpr (!POLICY) action_next = io0.actions[el_l];
        }
    }

    // Launch constants into VGPRs only now, AFTER the loads are in flight: pinned() is ~25 v_movs
    // behind a kernarg s_load round trip, which used to sit in front of the first global load.
    // (not for the policy rollout: its MLP needs the 25 registers more than it minds re-fetching
    // launch constants, and has to stay under 256 VGPRs to keep two waves per SIMD)
    // (float32: only the constants the formulation reads, -0.08 us per launch; float64 measured no better that way)
    const Params<T> p = POLICY ? p_arg : (sizeof(T) == 4 ? pinned_for<T, FAST>(p_arg) : pinned(p_arg));
    if constexpr (AUTO_RESET && !ROLLOUT && sizeof(T) == 4) {
        // ... and what the reset of a finished env reads, so that its wave does not start the reset
        // with a scalar-load round trip (the kernel ends with that wave).  Not in the float64 build: its 21 reset
        // constants are 42 SGPRs, and with them pinned hipcc stages the 25 launch constants through one 16-register range
        // in three batches, a scalar round trip each, in front of every wave's arithmetic (float64 FAST 10.8 -> 10.3 us per
        // launch without the request, although a finishing wave then fetches the reset constants when it needs them).
        asm volatile("" :: "s"(rp_arg.own_x0), "s"(rp_arg.own_y0), "s"(rp_arg.own_v), "s"(rp_arg.own_heading0),
                     "s"(rp_arg.own_heading_jitter), "s"(rp_arg.goal_x), "s"(rp_arg.goal_y), "s"(rp_arg.t0_x), "s"(rp_arg.t0_y_base),
                     "s"(rp_arg.t0_y_span), "s"(rp_arg.t0_heading_base), "s"(rp_arg.t0_heading_step), "s"(rp_arg.t0_heading_jitter),
                     "s"(rp_arg.tn_x_max), "s"(rp_arg.tn_y_max), "s"(rp_arg.speed_factor_min), "s"(rp_arg.speed_factor_max),
                     "s"(rp_arg.airspeed), "s"(rp_arg.d_goal0), "s"(rp_arg.h_goal0), "s"(rp_arg.d_dev0),
                     "s"(k0), "s"(k1), "s"(io_arg.term_obs), "s"(io_arg.ep_return), "s"(io_arg.ep_steps), "s"(env_offset));
    }
    const StepResetParams<T, ROLLOUT>& rp = rp_arg;
    TrigCache<T, C> trig;                                  // rollout only (a per-step launch starts cold anyway)
    const int T_steps = ROLLOUT ? n_steps : 1;
    if constexpr (POLICY) {
        // the observation the first action is taken on (reset()'s / the previous step's) into the lane's row
        const T* obs_in = static_cast<const T*>(pw.obs_in) + e_wave * D;
        if (active) { for (int i = 0; i < D; ++i) row[i] = obs_in[el * D + i]; }
    }
    for (int t = 0; t < T_steps; ++t) {
        // outputs of step t: [t][E] / [t][E][D] slices (t == 0 for the per-step launch)
        const int64_t te = ROLLOUT ? (int64_t)t * n_envs : 0;
        const StepIO<T> io{io0.actions + te, io0.obs + te * D, io0.reward + te, io0.done + te, io0.outcome + te,
                           io0.term_obs ? io0.term_obs + te * D : nullptr,
                           io0.ep_return ? io0.ep_return + te : nullptr,
                           io0.ep_steps ? io0.ep_steps + te : nullptr};
        const bool last = !ROLLOUT || t == T_steps - 1;
        uint8_t oc = 0;
        T action = action_next;
        if constexpr (POLICY) {
            constexpr int DP = 5 + 3 * NS;                // compile-time obs width (packed shapes)
            float x[DP];
#pragma unroll
            for (int i = 0; i < DP; ++i) x[i] = (float)row[i];
            // The weights must be RE-READ every step (scalar cache hits): they are loop-invariant, and
            // hoisted out of the step loop hipcc tries to keep all 4 700 of them in SGPRs, spills them
            // to VGPR lanes and reads them back one v_readlane at a time (7x slower).  Laundering the
            // pointers through an empty asm makes the loads depend on the iteration.
            PolicyW pw_t = pw;
            asm volatile("" : "+s"(pw_t.w1t), "+s"(pw_t.b1), "+s"(pw_t.w2t), "+s"(pw_t.b2), "+s"(pw_t.w3), "+s"(pw_t.b3));
            if constexpr (SAMPLE) {
#pragma unroll
                for (int i = 0; i < DP; ++i) x[i] = (x[i] == x[i] && fabsf(x[i]) < __builtin_inff()) ? x[i] : 0.0f;
                const float mean = policy_mlp<DP>(pw_t.w1t, pw_t.b1, pw_t.w2t, pw_t.b2, pw_t.w3, pw_t.b3, x);
                asm volatile("" : "+s"(pw_t.v1t), "+s"(pw_t.vb1), "+s"(pw_t.v2t), "+s"(pw_t.vb2), "+s"(pw_t.v3), "+s"(pw_t.vb3),
                                  "+s"(pw_t.log_std));
                const float value = policy_mlp<DP>(pw_t.v1t, pw_t.vb1, pw_t.v2t, pw_t.vb2, pw_t.v3, pw_t.vb3, x);
                const float log_std = ((const float ACAS2D_AS4*)pw_t.log_std)[0];
                // eps ~ N(0, 1): one Philox block per (global env, noise step + t), Box-Muller on two 24-bit uniforms
                const uint64_t gid = (uint64_t)(env_offset + e_wave + el);
                const U4 w = philox4x32(U4{(uint32_t)gid, (uint32_t)(gid >> 32), pw.noise_step + (uint32_t)t, 0x6e6f6973u},
                                           pw.nk0, pw.nk1);
                const float u1 = u01f(w.x), u2 = u01f(w.y);
                const float eps = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) *    // -2 ln u1
                                  __builtin_amdgcn_cosf(u2);                                                    // cos(2 pi u2)
                const float raw = fmaf(__builtin_amdgcn_exp2f(log_std * 1.4426950408889634f), eps, mean);
                const float logp = fmaf(-0.5f * eps, eps, -log_std) - 0.9189385332046727f;                      // - log sqrt(2 pi)
                if (active) {
                    (static_cast<T*>(pw.actions_out) + e_wave + te)[el] = (T)raw;      // the buffer keeps the RAW action,
                    (static_cast<T*>(pw.values_out) + e_wave + te)[el] = (T)value;     // the env sees the clipped one
                    (static_cast<T*>(pw.logp_out) + e_wave + te)[el] = (T)logp;
                }
                action = (T)fminf(fmaxf(raw, -1.0f), 1.0f);
            } else {
                action = (T)policy_action<DP>(pw_t, x);
                if (active) (static_cast<T*>(pw.actions_out) + e_wave + te)[el] = action;
            }
        } else {
            if (ROLLOUT && active && t + 1 < T_steps) action_next = io.actions[n_envs + el];
        }
        if (run) {
            if (t == 0) ACAS2D_STAMP(2, wave, lane, true);

            // game.py:225 + aircraft.py:16-26 for the player
            o.a_lat = action * p.acc_lat_limit;
            if constexpr (FAST) {
                o.psi = wrap_fast(o.psi + o.a_lat * f_rcp(o.v));        // (a_lat / (v dt)) dt
                T sn, cs;
                f_sincos_rev(o.psi * Const<T>::inv360, &sn, &cs);
                const T vdt = o.v * p.dt;
                o.x = m_fma(vdt, cs, o.x);
                o.y = m_fma(vdt, sn, o.y);
            } else {
                T psi_dot = o.a_lat / (o.v * p.dt);
                o.psi = py_mod360(o.psi + (psi_dot * p.dt));
                T sn, cs;
                m_sincos(deg2rad_ref(o.psi), &sn, &cs);
                o.x = o.x + ((o.v * cs) * p.dt);
                o.y = o.y + ((o.v * sn) * p.dt);
            }
            steps += 1;                                                       // game.py:197
            // `episode` is first USED in the reset loop far below; without this use the compiler waits
            // for its load there with s_waitcnt vmcnt(0) -- i.e. for every store issued since.
            asm volatile("" : "+v"(episode));
            T d_sep = T(0);                               // record rows only
            auto before_traffic = [&](const OwnCtx<T>&) {
                if constexpr (!AUTO_RESET) { if (s.trace) d_sep = minimum_separation<T, C, G, PACKED>(s, o, tr, el, j, N); }
            };
            Seen<T> r = observe<T, C, G, PACKED, FAST>(p, s, o, el, j, N, steps, !frozen, tr, row, last && active,
                                                       ROLLOUT ? &trig : nullptr, before_traffic);

            // game.py:249-292 evaluate()
            T rw = step_reward_5<T, FAST>(p, r.v_closing0, o.psi, r.h_goal, r.d_cpa0, r.d_goal, r.d_dev);
            if constexpr (FAST) rw = rw * m_fma(-(T)steps, p.inv_max_steps, T(1));
            else rw = rw * (T(1) - ((T)steps / (T)p.max_steps));              // :262-263
            if constexpr (!AUTO_RESET) {                                      // :266-276, the record lists
                if (s.trace && j == 0 && active)
                    write_trace<T, FAST>(p, s.trace + el * kTraceWidth, o.psi, d_sep, o.a_lat, r.h_goal, r.d_goal, r.d_dev,
                                         r.v_closing0, r.d_cpa0, rw);
            }
            const bool at_goal = r.d_goal < p.goal_radius;                    // :191-192
            if (r.collided) rw += p.reward_collision;                         // :279-280
            if (at_goal) rw += p.reward_goal;                                 // :283-284
            // game.py:294-314 is_done(): timeout > collision > goal
            oc = (steps > p.max_steps) ? 3 : (r.collided ? 2 : (at_goal ? 1 : 0));
            total = total + rw;                                               // :287
            if (!active) oc = 0;                          // a padding lane never finishes anything
            if (j == 0 && active) {
                io.reward[el] = rw;                       // (plain stores: non-temporal measured the same, 0.6 MB)
                io.done[el] = (uint8_t)(oc != 0);
                io.outcome[el] = oc;
                if constexpr (!HANDOFF) {
                    if (last && (oc == 0 || !AUTO_RESET)) {
                        put_env(s, s.own_x, el, o.x); put_env(s, s.own_y, el, o.y); put_env(s, s.own_psi, el, o.psi);
                        put_env(s, s.steps, el, steps);
                        put_env(s, s.total_reward, el, total);
                        if constexpr (!AUTO_RESET) { if (oc) s.status[el] = oc; }
                    }
                }
            }
        }

        if (t == 0) ACAS2D_STAMP(3, wave, lane, false);
        T* const obs_wave = io.obs;
        if constexpr (HANDOFF) {
            // ---- finished envs (one bit per env: its group's lane 0) are reset by the whole wave NOW,
            // before anything but reward / done / outcome has been stored: the new episode's state
            // reaches the owner lanes through LDS and leaves with the coalesced state stores and the
            // ONE tile flush below.  (Resetting after the flush cost a second store round at the very
            // end of the kernel -- ~12 scattered 4-byte stores per entity lane plus a re-flush of the
            // row: 0.8 us of the 7.7 us launch at 65 536 x 8.)
            bool fresh = false, same_consts = false;
            unsigned long long dm = __ballot(oc != 0 && j == 0);
            constexpr bool SLOTTED = ResetSlots<NS>::SLOTS >= 2;      // N + 1 <= 32: several envs per pass
            if constexpr (SLOTTED) {
                using SL = SlotLayout<T, NS>;
                while (dm) {
                    wave_lds_fence();                     // every row of the tile is complete
                    const unsigned long long taken =
                        wave_reset_slots<T, FAST, NS, G>(p, rp, io, k0, k1, (uint64_t)(env_offset + e_wave), dm, lane,
                                                         episode, tile, scratch);
                    dm &= ~taken;
                    wave_lds_fence();                     // the fresh rows and the slots are complete
                    if ((taken >> (el * G)) & 1ull) {     // my env was reset: its owner group goes on with the new episode
                        const int k_own = __popcll(taken & ((1ull << (el * G)) - 1ull));
                        const T* scr = scratch + k_own * SL::STRIDE;
                        if (j == 0) {
                            if (io.ep_return) io.ep_return[el] = total;
                            if (io.ep_steps) io.ep_steps[el] = steps;
                        }
                        using V = Vec<T, C>;              // the slot's blocks are aligned like the live traffic block
                        tr.x = *reinterpret_cast<const V*>(scr + j * C);
                        tr.y = *reinterpret_cast<const V*>(scr + N + j * C);
                        tr.psi = *reinterpret_cast<const V*>(scr + 2 * N + j * C);
                        tr.v = *reinterpret_cast<const V*>(scr + 3 * N + j * C);
                        // own_v / goal are the configuration's: stored only where an injected state had others
                        same_consts = o.v == (T)rp.own_v && o.gx == (T)rp.goal_x && o.gy == (T)rp.goal_y;
                        o = Own<T>{(T)rp.own_x0, (T)rp.own_y0, scr[SL::OWN_PSI], (T)rp.own_v, T(0), (T)rp.goal_x, (T)rp.goal_y};
                        steps = 1;                                            // environment.py:47
                        total = T(0);
                        episode += 1u;
                        fresh = true;
                        trig.valid = false; trig.dirty = false;   // new headings (stored below)
                        const int i0 = el * N + j * C;
                        put_trf<T, C>(s, s.trf_x, i0, tr.x); put_trf<T, C>(s, s.trf_y, i0, tr.y);
                        *reinterpret_cast<V*>(s.trf_psi + i0) = tr.psi; *reinterpret_cast<V*>(s.trf_v + i0) = tr.v;
                    }
                    wave_lds_fence();                     // the slots are free for the next pass
                }
            }
            while (!SLOTTED && dm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)dm) - 1);   // wave-uniform
                dm &= dm - 1;
                const int el_d = src / G;
                wave_lds_fence();                         // every row of the tile is complete
                wave_reset_env<T, FAST, NS, true>(p, rp, s, io, k0, k1, (uint64_t)(env_offset + e_wave + el_d), el_d, N,
                                                  lane, lane_value(total, src), lane_value(steps, src),
                                                  (uint32_t)lane_value((int)episode, src), tile + el_d * D, scratch);
                wave_lds_fence();                         // the fresh row and the scratch are complete
                if (el == el_d) {                         // the owner group continues with the new episode
                    if (j == 0) {
                        if (io.ep_return) io.ep_return[el] = total;
                        if (io.ep_steps) io.ep_steps[el] = steps;
                    }
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int n = j * C + k;
                        tr.x.v[k] = scratch[n]; tr.y.v[k] = scratch[N + n];
                        tr.psi.v[k] = scratch[2 * N + n]; tr.v.v[k] = scratch[3 * N + n];
                    }
                    o = Own<T>{(T)rp.own_x0, (T)rp.own_y0, scratch[4 * N], (T)rp.own_v, T(0), (T)rp.goal_x, (T)rp.goal_y};
                    steps = 1;                                                // environment.py:47
                    total = T(0);
                    episode += 1u;
                    fresh = true;
                    trig.valid = false; trig.dirty = false;   // new headings (stored below)
                    // the new traffic block: whole 16-byte vectors from the owner lanes
                    using V = Vec<T, C>;
                    const int i0 = el * N + j * C;
                    put_trf<T, C>(s, s.trf_x, i0, tr.x); put_trf<T, C>(s, s.trf_y, i0, tr.y);
                    *reinterpret_cast<V*>(s.trf_psi + i0) = tr.psi; *reinterpret_cast<V*>(s.trf_v + i0) = tr.v;
                }
                wave_lds_fence();                         // scratch is free for the next finished env
            }
            if (active && j == 0) {
                if (fresh) {                              // per-episode constants of the new episode
                    if constexpr (ARENA) {
                        char* c = reinterpret_cast<char*>(const_cast<T*>(a1));
                        const uint32_t ie = e_wave32 + (uint32_t)el;
                        *reinterpret_cast<uint32_t*>(c + (size_t)((ie + 3u * E32) * 4u)) = episode;
                        if (!same_consts) {
                            *reinterpret_cast<T*>(c + (size_t)(ie * 4u)) = o.v;
                            *reinterpret_cast<T*>(c + (size_t)((ie + E32) * 4u)) = o.gx;
                            *reinterpret_cast<T*>(c + (size_t)((ie + 2u * E32) * 4u)) = o.gy;
                        }
                    } else {
                        s.episode[el] = episode;
                        if (!same_consts) { s.own_v[el] = o.v; s.goal_x[el] = o.gx; s.goal_y[el] = o.gy; }
                    }
                }
                if (last || fresh) {
                    if constexpr (ARENA) {
                        // ONE base (the block's write generation) and the 32-bit per-lane offsets the loads used: hipcc
                        // otherwise re-derives five 64-bit pointers in front of these stores, ~45 scalar instructions at
                        // the end of every wave
                        char* mw = reinterpret_cast<char*>(const_cast<T*>(a0)) + (int64_t)s.w_env * 4;
                        const uint32_t ie = e_wave32 + (uint32_t)el;
                        __builtin_nontemporal_store(o.x, reinterpret_cast<T*>(mw + (size_t)(ie * 4u)));
                        __builtin_nontemporal_store(o.y, reinterpret_cast<T*>(mw + (size_t)((ie + E32) * 4u)));
                        __builtin_nontemporal_store(o.psi, reinterpret_cast<T*>(mw + (size_t)((ie + 2u * E32) * 4u)));
                        __builtin_nontemporal_store(steps, reinterpret_cast<int32_t*>(mw + (size_t)((ie + 4u * E32) * 4u)));
                        __builtin_nontemporal_store(total, reinterpret_cast<T*>(mw + (size_t)((ie + 3u * E32) * 4u)));
                    } else {
                        put_env(s, s.own_x, el, o.x); put_env(s, s.own_y, el, o.y); put_env(s, s.own_psi, el, o.psi);
                        put_env(s, s.steps, el, steps);
                        put_env(s, s.total_reward, el, total);
                    }
                }
            }
        }
        // Flush the tile (generic walk: now, the stores drain while finished envs are reset below).
        wave_lds_fence();
        if constexpr (PACKED) {
            constexpr int kChunks = (EPW * (5 + 3 * NS) * (int)sizeof(T) / 16 + 63) / 64;
            flush_tile_unrolled<T, kChunks>(tile, obs_wave, n_rows * D, lane);
        } else {
            flush_tile<T>(tile, obs_wave, n_rows * D, lane);
        }
        if (t == 0) ACAS2D_STAMP(4, wave, lane, false);
        if constexpr (AUTO_RESET && !HANDOFF) {
            // ---- generic walk: finished envs are reset by the whole wave, entity lanes store the new
            // state themselves and the row is flushed again ----
            unsigned long long dm = __ballot(oc != 0 && j == 0);
            while (dm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)dm) - 1);   // wave-uniform
                dm &= dm - 1;
                const int el_d = src / G;
                wave_reset_env<T, FAST, NS, false>(p, rp, s, io, k0, k1, (uint64_t)(env_offset + e_wave + el_d), el_d, N,
                                                   lane, lane_value(total, src), lane_value(steps, src),
                                                   (uint32_t)lane_value((int)episode, src), tile + el_d * D, nullptr);
                wave_lds_fence();                         // the fresh row is complete
                flush_rows<T>(tile, obs_wave, n_rows * D, el_d * D, (el_d + 1) * D, lane);
            }
        }
        if constexpr (ROLLOUT) wave_lds_fence();          // tile reads precede the next step's row writes
    }
    ACAS2D_STAMP(5, wave, lane, false);
    ACAS2D_STAMP(6, wave, lane, true);
    ACAS2D_STAMP(7, wave, lane, false);
}

// ACAS2DEnv.reset(), environment.py:44-48 (do_init != 0: fresh episodes; == 0: keep the injected state).
template <typename T, int C, int G, bool PACKED, bool FAST>
__global__ __launch_bounds__(kBlock) void reset_kernel(Params<T> p_arg, StepResetParams<T, true> rp, State<T> s_arg,
                                                       const uint8_t* __restrict__ mask, T* obs,
                                                       int do_init, uint32_t k0, uint32_t k1,
                                                       int64_t env_offset, int64_t n_envs, int N,
                                                       int tile_elems) {
    constexpr int EPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int j = lane & (G - 1), el = lane / G;
    const int wib = wave_in_block();
    const int64_t e_wave = (remap_block() * kWavesPerBlock + wib) * EPW;
    if (e_wave >= n_envs) return;                          // whole wave idle
    const Params<T> p = pinned(p_arg);
    const int D = 5 + 3 * N;
    const State<T> s = rebase(s_arg, e_wave, N);
    const bool selected = e_wave + el < n_envs && (!mask || mask[e_wave + el]);
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* row = reinterpret_cast<T*>(lds_raw) + wib * tile_elems + el * D;
    if (selected) {
        Own<T> o;
        Traffic<T, C> tr;
        int32_t steps;
        if (do_init) {
            o = reset_env<T, C, G, PACKED>(rp, s, k0, k1, (uint64_t)(env_offset + e_wave + el), s.episode[el], el, j, N, tr);
            steps = 0;
        } else {
            if constexpr (PACKED) tr = load_traffic<T, C>(s, el * N + j * C);
            o = Own<T>{s.own_x[el], s.own_y[el], s.own_psi[el], s.own_v[el], T(0), s.goal_x[el], s.goal_y[el]};
            steps = s.steps[el];
        }
        if (obs) {
            steps += 1;
            T d_sep = T(0);
            auto sep = [&](const OwnCtx<T>&) { if (s.trace) d_sep = minimum_separation<T, C, G, PACKED>(s, o, tr, el, j, N); };
            const Seen<T> r = observe<T, C, G, PACKED, FAST>(p, s, o, el, j, N, steps, false, tr, row, true, nullptr, sep);
            // FAST, fresh episodes: the same goal terms as the in-step reset -- own_context_fresh()
            const bool fresh = FAST && do_init;
            const T d_goal = fresh ? (T)rp.d_goal0 : r.d_goal, h_goal = fresh ? (T)rp.h_goal0 : r.h_goal,
                    d_dev = fresh ? (T)rp.d_dev0 : r.d_dev;
            if (fresh && j == 0) {                        // put_own_obs(), entries 2..4
                row[2] = d_dev * p.inv_d_dev_max; row[3] = d_goal * p.inv_d_goal_max; row[4] = h_goal * Const<T>::inv360;
            }
            if (s.trace && j == 0)                        // game.py:132-160: the records' first entries
                write_trace<T, FAST>(p, s.trace + el * kTraceWidth, o.psi, d_sep, T(0), h_goal, d_goal, d_dev, r.v_closing0,
                                     r.d_cpa0, step_reward_5<T, FAST>(p, r.v_closing0, o.psi, h_goal, r.d_cpa0, d_goal, d_dev));
            wave_lds_fence();
            T* dst = obs + (e_wave + el) * D;             // masked rows are not contiguous: per-row copy
            for (int i = j; i < D; i += G) dst[i] = row[i];
        }
        if (j == 0) {
            s.steps[el] = steps;
            s.total_reward[el] = T(0);
            s.status[el] = 0;
        }
    }
}

// ---- host-side launchers (instantiated per element type in acas2d_f32.hip / acas2d_f64.hip) ----
struct Shape { int C, G; bool packed; };
Shape choose_shape(int n_traffic, int elem_size);
template <typename T>
int launch_step(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dState* st_out, const Acas2dStepIO* io, uint32_t flags,
                uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream);
template <typename T>
int launch_rollout(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io, int32_t n_steps,
                   uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream);
template <typename T>
int launch_rollout_policy(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io,
                          const Acas2dPolicy* pol, const void* obs_in, int32_t n_steps, uint64_t seed,
                          int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream);
template <typename T>
int launch_collect(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io, const Acas2dActorCritic* ac,
                   const void* obs_in, int32_t n_steps, uint64_t seed, int64_t env_offset, int64_t n_envs,
                   int32_t n_traffic, hipStream_t stream);
template <typename T>
int launch_reset(const Acas2dConfig* cfg, const Acas2dState* st, const uint8_t* mask, void* obs,
                 int32_t do_init, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                 hipStream_t stream);

void set_error(const char* fmt, ...);

}  // namespace acas2d
