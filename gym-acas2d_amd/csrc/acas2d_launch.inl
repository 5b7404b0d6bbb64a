// acas2d_launch.inl -- host-side launchers, included by acas2d_f32.hip and acas2d_f64.hip which
// define ACAS2D_PACKED_SHAPES(X) (the (C, G) pairs to instantiate for their element type) and
// kFast, then instantiate launch_step<T> / launch_reset<T>.  (Two translation units: each element
// type is compiled with its own flags.  Both use -ffp-contract=off today, so only the fma()s written
// in the source fuse.)
#include <stdlib.h>
#include <cmath>

#include "acas2d_kernels.hpp"

namespace acas2d {

template <typename T>
static Params<T> make_params(const Acas2dConfig& c) {
    Params<T> p;
    p.dt = (T)c.dt; p.acc_lat_limit = (T)c.acc_lat_limit; p.collision_dist = (T)c.collision_dist;
    p.goal_radius = (T)c.goal_radius; p.safe_distance = (T)c.safe_distance;
    p.d_goal_max = (T)c.d_goal_max; p.d_dev_max = (T)c.d_dev_max; p.d_sep_max = (T)c.d_sep_max;
    p.d_cpa_max = (T)c.d_cpa_max; p.v_closing_max = (T)c.v_closing_max;
    p.rw_d_goal_max = (T)c.rw_d_goal_max; p.rw_d_dev_max = (T)c.rw_d_dev_max;
    p.reward_goal = (T)c.reward_goal; p.reward_collision = (T)c.reward_collision;
    p.inv_dt = (T)(1.0 / c.dt); p.inv_d_goal_max = (T)(1.0 / c.d_goal_max);
    p.inv_d_dev_max = (T)(1.0 / c.d_dev_max); p.inv_d_sep_max = (T)(1.0 / c.d_sep_max);
    p.inv_d_cpa_max = (T)(1.0 / c.d_cpa_max); p.inv_v_closing_max = (T)(1.0 / c.v_closing_max);
    p.inv_rw_d_goal_max = (T)(1.0 / c.rw_d_goal_max); p.inv_rw_d_dev_max = (T)(1.0 / c.rw_d_dev_max);
    p.inv_safe_distance = (T)(1.0 / c.safe_distance); p.inv_max_steps = (T)(1.0 / (double)c.max_steps);
    p.max_steps = c.max_steps;
    return p;
}

template <typename R, typename GT = R>
static ResetParamsT<R, GT> make_reset_params(const Acas2dConfig& c) {
    // the goal terms of a fresh episode (own_context_fresh()): game.py:168-180 at the start position
    const double gdx = c.goal_x - c.own_x0, gdy = c.goal_y - c.own_y0;
    double bearing = std::atan2(gdy, gdx);
    if (bearing < 0) bearing += 6.283185307179586476925;
    else if (bearing == 0) bearing = 0;
    return ResetParamsT<R, GT>{(R)c.own_x0, (R)c.own_y0, (R)c.own_v, (R)c.own_heading0, (R)c.own_heading_jitter,
                           (R)c.goal_x, (R)c.goal_y, (R)c.t0_x, (R)c.t0_y_base, (R)c.t0_y_span,
                           (R)c.t0_heading_base, (R)c.t0_heading_step, (R)c.t0_heading_jitter, (R)c.tn_x_max,
                           (R)c.tn_y_max, (R)c.speed_factor_min, (R)c.speed_factor_max, (R)c.airspeed,
                           (GT)std::sqrt(std::fma(gdy, gdy, gdx * gdx)), (GT)(bearing * 57.29577951308232087680), (GT)gdy};
}

template <typename T>
static State<T> make_state(const Acas2dState& s) {
    return State<T>{(T*)s.own_x, (T*)s.own_y, (T*)s.own_psi, (T*)s.own_v, (T*)s.goal_x, (T*)s.goal_y,
                    (T*)s.trf_x, (T*)s.trf_y, (T*)s.trf_psi, (T*)s.trf_v, s.steps,
                    (T*)s.total_reward, s.status, s.episode, (T*)s.trace, 0, 0};
}

static bool state_complete(const Acas2dState* s) {
    return s && s->own_x && s->own_y && s->own_psi && s->own_v && s->goal_x && s->goal_y && s->trf_x &&
           s->trf_y && s->trf_psi && s->trf_v && s->steps && s->total_reward && s->status && s->episode;
}

// Double-buffered state (acas2d_step_* with a state_out): the element offsets of `out`'s per-step arrays from
// `st`'s -- own_x, own_y, own_psi, steps, total_reward share one (w_env), trf_x and trf_y another (w_trf); e.g.
// every such array allocated as [2][E] / [2][E][N] and the two structs pointing at its two halves.  Everything
// else must be the same buffer in both structs (those arrays change at a reset only, in place).
template <typename T>
static int write_offsets(const Acas2dState* st, const Acas2dState* out, bool auto_reset, int64_t n_envs, int n_traffic,
                         int32_t* w_env, int32_t* w_trf) {
    *w_env = 0; *w_trf = 0;
    if (!out || out == st) return ACAS2D_OK;
    if (!state_complete(out)) { set_error("acas2d_step: state_out has a NULL buffer"); return ACAS2D_EINVAL; }
    if (out->own_v != st->own_v || out->goal_x != st->goal_x || out->goal_y != st->goal_y || out->trf_psi != st->trf_psi ||
        out->trf_v != st->trf_v || out->status != st->status || out->episode != st->episode || out->trace != st->trace) {
        set_error("acas2d_step: state_out must share own_v, goal_x, goal_y, trf_psi, trf_v, status, episode and trace with state "
                  "(only own_x, own_y, own_psi, steps, total_reward, trf_x, trf_y are double-buffered)");
        return ACAS2D_EINVAL;
    }
    const int64_t d = (const T*)out->own_x - (const T*)st->own_x, dt = (const T*)out->trf_x - (const T*)st->trf_x;
    if ((const T*)out->own_y - (const T*)st->own_y != d || (const T*)out->own_psi - (const T*)st->own_psi != d ||
        (const T*)out->total_reward - (const T*)st->total_reward != d || out->steps - st->steps != d ||
        (const T*)out->trf_y - (const T*)st->trf_y != dt) {
        set_error("acas2d_step: state_out's own_x, own_y, own_psi, steps, total_reward must lie at ONE element offset from "
                  "state's, trf_x and trf_y at one (e.g. each array allocated [2][E] / [2][E][N])");
        return ACAS2D_EINVAL;
    }
    if (d == 0 && dt == 0) return ACAS2D_OK;
    if (!auto_reset) {
        set_error("acas2d_step: a separate state_out needs ACAS2D_AUTO_RESET (the latching step leaves frozen traffic unwritten)");
        return ACAS2D_EINVAL;
    }
    const int64_t lim = 0x7fffffffLL - n_envs * (int64_t)n_traffic - 64;
    const int64_t ad = d < 0 ? -d : d, adt = dt < 0 ? -dt : dt;
    if ((ad != 0 && ad < n_envs) || (adt != 0 && adt < n_envs * (int64_t)n_traffic) || ad > lim || adt > lim) {   // (either group may stay in place)
        set_error("acas2d_step: state_out overlaps state, or lies more than 2^31 elements away");
        return ACAS2D_EINVAL;
    }
    *w_env = (int32_t)d; *w_trf = (int32_t)dt;
    return ACAS2D_OK;
}

static int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(err)); return ACAS2D_EHIP; }
    return ACAS2D_OK;
}

// Launch geometry for a shape: one env per G lanes, 64 / G envs per wavefront, 4 wavefronts per
// workgroup, one LDS observation tile per wavefront.
struct Geometry { unsigned grid, block; int tile_elems; size_t lds_bytes; };

template <typename T>
static int geometry_for(const Shape& sh, int64_t n_envs, int n_traffic, Geometry* g) {
    const int64_t epw = 64 / sh.G, envs_per_block = epw * kWavesPerBlock;
    const int64_t blocks = (n_envs + envs_per_block - 1) / envs_per_block;
    if (blocks > 0x7fffffffLL || n_envs > 0x7fffffffLL) { set_error("n_envs = %lld exceeds the launch limit", (long long)n_envs); return ACAS2D_EINVAL; }
    const int W = 16 / (int)sizeof(T);
    // per wave: the observation tile, then the reset slots (SlotLayout<T, N> in the kernels; packed shapes with
    // N + 1 <= 32) or one 4N+1-value hand-off scratch (the other packed shapes), everything 16-byte aligned
    const int64_t tile = (epw * (5 + 3 * (int64_t)n_traffic) + 3) / 4 * 4;
    int64_t scratch = 0;
    if (sh.packed && n_traffic + 1 <= 32) {
        int stride = 2; while (stride < n_traffic + 1) stride *= 2;
        scratch = (64 / stride) * ((4 * (int64_t)n_traffic + 1 + W - 1) / W * W);
    } else if (sh.packed) {
        scratch = 4 * (int64_t)n_traffic + 1;
    }
    const int64_t elems = ((tile + scratch + 3) / 4) * 4;
    const int64_t bytes = elems * kWavesPerBlock * (int64_t)sizeof(T);
    if (bytes > 64 * 1024) {
        set_error("n_traffic = %d needs a %lld-byte LDS observation tile per workgroup (limit 65536)", n_traffic, (long long)bytes);
        return ACAS2D_EINVAL;
    }
    g->grid = (unsigned)blocks; g->block = (unsigned)kBlock; g->tile_elems = (int)elems; g->lds_bytes = (size_t)bytes;
    return ACAS2D_OK;
}

// The step kernel's first 14 argument dwords are what the command processor preloads into SGPRs (see step_kernel):
// six pointers, the env count, the tile size.
#define ACAS2D_EARLY_ARGS(s, n_envs, g) \
    (s).trf_x, (s).trf_y, (s).trf_psi, (s).trf_v, (s).own_x, (s).own_y, (int32_t)(n_envs), (int32_t)(g).tile_elems

// "Arena" layout of a float32 state: the per-env arrays a step reads are consecutive [k][E] rows -- own_x, own_y,
// own_psi, total_reward, steps / own_v, goal_x, goal_y, episode -- and so are the traffic arrays -- trf_x, trf_y /
// trf_psi, trf_v ([k][E][N]).  Five preloaded base pointers (the four blocks and the actions) then name every input of
// the step, and ALL of a wavefront's loads leave before its first scalar-load round trip (step_kernel<..., ARENA>).
// `ACAS2DVecEnv` allocates its float32 state this way.  The kernel also assumes full waves in whole multiples of eight
// workgroups (n_envs a multiple of 1 024 at n_traffic = 8); any other layout or size takes the general kernel, same results.
template <typename T>
static bool arena_layout(const State<T>& s, int64_t E, int N, int G) {
    if (sizeof(T) != 4 || getenv("ACAS2D_NO_ARENA")) return false;
    if (E % ((64 / G) * kWavesPerBlock * 8) != 0) return false;            // whole multiples of eight workgroups (full waves)
    const int64_t EN = E * N;
    if (8 * EN >= (1LL << 32) || 20 * E >= (1LL << 32)) return false;      // the kernel's 32-bit byte offsets
    return s.own_y == s.own_x + E && s.own_psi == s.own_x + 2 * E && s.total_reward == s.own_x + 3 * E &&
           (const void*)s.steps == (const void*)(s.own_x + 4 * E) &&
           s.goal_x == s.own_v + E && s.goal_y == s.own_v + 2 * E && (const void*)s.episode == (const void*)(s.own_v + 3 * E) &&
           s.trf_y == s.trf_x + EN && s.trf_v == s.trf_psi + EN;
}

template <typename T, bool FAST, int C, int G, bool PACKED>
static void step_shape(bool auto_reset, const Geometry& g, hipStream_t stream, const Params<T>& p,
                       const ResetParamsT<T>& rp, const State<T>& s, const StepIO<T>& io, uint32_t k0, uint32_t k1,
                       int64_t env_offset, int64_t n_envs, int N) {
    if constexpr (PACKED && sizeof(T) == 4) {
        if (auto_reset && arena_layout<T>(s, n_envs, N, G)) {
            hipLaunchKernelGGL((step_kernel<T, C, G, true, true, FAST, false, false, false, true>), dim3(g.grid), dim3(g.block),
                               g.lds_bytes, stream, (const T*)s.own_x, (const T*)s.own_v, (const T*)s.trf_x, (const T*)s.trf_psi,
                               io.actions, (const T*)nullptr, (int32_t)n_envs, (int32_t)g.tile_elems,
                               p, rp, s, io, k0, k1, env_offset, N, 1, PolicyW{});
            return;
        }
    }
    if (auto_reset)
        hipLaunchKernelGGL((step_kernel<T, C, G, PACKED, true, FAST, false>), dim3(g.grid), dim3(g.block), g.lds_bytes, stream,
                           ACAS2D_EARLY_ARGS(s, n_envs, g), p, rp, s, io, k0, k1, env_offset, N, 1, PolicyW{});
    else
        hipLaunchKernelGGL((step_kernel<T, C, G, PACKED, false, FAST, false>), dim3(g.grid), dim3(g.block), g.lds_bytes, stream,
                           ACAS2D_EARLY_ARGS(s, n_envs, g), p, rp, s, io, k0, k1, env_offset, N, 1, PolicyW{});
}

template <typename T, bool FAST, int C, int G>
static void rollout_shape(const Geometry& g, hipStream_t stream, const Params<T>& p, const StepResetParams<T, true>& rp,
                          const State<T>& s, const StepIO<T>& io, uint32_t k0, uint32_t k1, int64_t env_offset,
                          int64_t n_envs, int N, int n_steps) {
    hipLaunchKernelGGL((step_kernel<T, C, G, true, true, FAST, true>), dim3(g.grid), dim3(kBlock), g.lds_bytes, stream,
                       ACAS2D_EARLY_ARGS(s, n_envs, g), p, rp, s, io, k0, k1, env_offset, N, n_steps, PolicyW{});
}

// the same with the SB3 actor evaluated in the kernel (thread-per-env shapes only)
template <typename T, bool FAST, int C>
static void policy_shape(const Geometry& g, hipStream_t stream, const Params<T>& p, const StepResetParams<T, true>& rp,
                         const State<T>& s, const StepIO<T>& io, uint32_t k0, uint32_t k1, int64_t env_offset,
                         int64_t n_envs, int N, int n_steps, const PolicyW& pw, bool sample) {
    if (sample)
        hipLaunchKernelGGL((step_kernel<T, C, 1, true, true, FAST, true, true, true>), dim3(g.grid), dim3(kBlock), g.lds_bytes,
                           stream, ACAS2D_EARLY_ARGS(s, n_envs, g), p, rp, s, io, k0, k1, env_offset, N, n_steps, pw);
    else
        hipLaunchKernelGGL((step_kernel<T, C, 1, true, true, FAST, true, true>), dim3(g.grid), dim3(kBlock), g.lds_bytes,
                           stream, ACAS2D_EARLY_ARGS(s, n_envs, g), p, rp, s, io, k0, k1, env_offset, N, n_steps, pw);
}

template <typename T, bool FAST, int C, int G, bool PACKED>
static void reset_shape(const Geometry& g, hipStream_t stream, const Params<T>& p, const StepResetParams<T, true>& rp,
                        const State<T>& s, const uint8_t* mask, T* obs, int do_init, uint32_t k0, uint32_t k1,
                        int64_t env_offset, int64_t n_envs, int N) {
    hipLaunchKernelGGL((reset_kernel<T, C, G, PACKED, FAST>), dim3(g.grid), dim3(kBlock),
                       g.lds_bytes, stream, p, rp, s, mask, obs, do_init, k0, k1, env_offset, n_envs, N, g.tile_elems);
}

static bool shape_instantiated(const Shape& sh) {
    if (!sh.packed) return sh.C == 1 && (sh.G == 1 || sh.G == 4 || sh.G == 16 || sh.G == 64);
#define X(C_, G_) if (sh.C == C_ && sh.G == G_) return true;
    ACAS2D_PACKED_SHAPES(X)
#undef X
    return false;
}

// Shape for (n_traffic, element type): the tuned default of choose_shape(), or the override
// ACAS2D_SHAPE="C,G" (packed, needs C*G == n_traffic) / "generic,G" from the environment.
template <typename T>
static int resolve_shape(int n_traffic, Shape* out) {
    Shape sh = choose_shape(n_traffic, (int)sizeof(T));
    if (const char* ov = getenv("ACAS2D_SHAPE")) {
        int a = 0, b = 0;
        if (sscanf(ov, "generic,%d", &b) == 1) sh = Shape{1, b, false};
        else if (sscanf(ov, "%d,%d", &a, &b) == 2) sh = Shape{a, b, true};
        if (sh.packed && sh.C * sh.G != n_traffic) { set_error("ACAS2D_SHAPE=%s does not tile n_traffic=%d", ov, n_traffic); return ACAS2D_EINVAL; }
    }
    if (!shape_instantiated(sh)) {
        if (sh.packed) sh = Shape{1, n_traffic >= 64 ? 64 : (n_traffic >= 16 ? 16 : (n_traffic >= 4 ? 4 : 1)), false};
        if (!shape_instantiated(sh)) { set_error("no kernel for shape C=%d G=%d", sh.C, sh.G); return ACAS2D_EINVAL; }
    }
    *out = sh;
    return ACAS2D_OK;
}

template <typename T, bool FAST>
static int launch_step_impl(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dState* st_out, const Acas2dStepIO* io_,
                uint32_t flags, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream) {
    if (!cfg || !io_) { set_error("acas2d_step: NULL cfg / io"); return ACAS2D_EINVAL; }
    if (!state_complete(st)) { set_error("acas2d_step: NULL state or a NULL state buffer"); return ACAS2D_EINVAL; }
    if (!io_->actions || !io_->obs || !io_->reward || !io_->done || !io_->outcome) {
        set_error("acas2d_step: actions, obs, reward, done and outcome are required"); return ACAS2D_EINVAL; }
    if (n_traffic < 1) { set_error("acas2d_step: n_traffic = %d (the reference needs traffic[0], game.py:254)", n_traffic); return ACAS2D_EINVAL; }
    if (n_envs < 0 || env_offset < 0) { set_error("acas2d_step: negative n_envs / env_offset"); return ACAS2D_EINVAL; }
    if (n_envs == 0) return ACAS2D_OK;
    Shape sh;
    if (int rc = resolve_shape<T>(n_traffic, &sh)) return rc;
    const bool ar = (flags & ACAS2D_AUTO_RESET) != 0;
    Geometry g;
    if (int rc = geometry_for<T>(sh, n_envs, n_traffic, &g)) return rc;
    const Params<T> p = make_params<T>(*cfg);
    const ResetParamsT<T> rp = make_reset_params<T>(*cfg);
    State<T> s = make_state<T>(*st);
    if (int rc = write_offsets<T>(st, st_out, ar, n_envs, n_traffic, &s.w_env, &s.w_trf)) return rc;
    const StepIO<T> io{(const T*)io_->actions, (T*)io_->obs, (T*)io_->reward, io_->done, io_->outcome,
                       (T*)io_->term_obs, (T*)io_->ep_return, io_->ep_steps};
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (sh.packed) {
#define X(C_, G_) if (sh.C == C_ && sh.G == G_) step_shape<T, FAST, C_, G_, true>(ar, g, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic);
        ACAS2D_PACKED_SHAPES(X)
#undef X
    } else {
        switch (sh.G) {
            case 1:  step_shape<T, FAST, 1, 1, false>(ar, g, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
            case 4:  step_shape<T, FAST, 1, 4, false>(ar, g, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
            case 16: step_shape<T, FAST, 1, 16, false>(ar, g, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
            default: step_shape<T, FAST, 1, 64, false>(ar, g, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
        }
    }
    return check_launch("acas2d_step launch");
}

template <typename T, bool FAST>
static int launch_rollout_impl(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io_, int32_t n_steps,
                   uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream) {
    if (!cfg || !io_) { set_error("acas2d_rollout: NULL cfg / io"); return ACAS2D_EINVAL; }
    if (!state_complete(st)) { set_error("acas2d_rollout: NULL state or a NULL state buffer"); return ACAS2D_EINVAL; }
    if (!io_->actions || !io_->obs || !io_->reward || !io_->done || !io_->outcome) {
        set_error("acas2d_rollout: actions, obs, reward, done and outcome are required"); return ACAS2D_EINVAL; }
    if (n_traffic < 1 || n_steps < 1) { set_error("acas2d_rollout: n_traffic = %d, n_steps = %d", n_traffic, n_steps); return ACAS2D_EINVAL; }
    if (n_envs < 0 || env_offset < 0) { set_error("acas2d_rollout: negative n_envs / env_offset"); return ACAS2D_EINVAL; }
    if (n_envs == 0) return ACAS2D_OK;
    Shape sh;
    if (int rc = resolve_shape<T>(n_traffic, &sh)) return rc;
    if (!sh.packed) {
        set_error("acas2d_rollout: n_traffic = %d has no packed work shape for this element type "
                  "(needs n_traffic in {1,2,3} or a multiple of %d tiling a wave); use acas2d_step", n_traffic,
                  16 / (int)sizeof(T));
        return ACAS2D_EINVAL;
    }
    Geometry g;
    if (int rc = geometry_for<T>(sh, n_envs, n_traffic, &g)) return rc;
    const Params<T> p = make_params<T>(*cfg);
    const StepResetParams<T, true> rp = make_reset_params<double, T>(*cfg);
    const State<T> s = make_state<T>(*st);
    const StepIO<T> io{(const T*)io_->actions, (T*)io_->obs, (T*)io_->reward, io_->done, io_->outcome,
                       (T*)io_->term_obs, (T*)io_->ep_return, io_->ep_steps};
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#define X(C_, G_) if (sh.C == C_ && sh.G == G_) rollout_shape<T, FAST, C_, G_>(g, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic, n_steps);
    ACAS2D_PACKED_SHAPES(X)
#undef X
    return check_launch("acas2d_rollout launch");
}

template <typename T, bool FAST>
static int launch_rollout_policy_impl(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io_,
                          const Acas2dPolicy* pol, const void* obs_in, int32_t n_steps, uint64_t seed,
                          int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream,
                          const Acas2dActorCritic* ac = nullptr) {
    if (!cfg || !io_ || !pol) { set_error("acas2d_rollout_policy: NULL cfg / io / policy"); return ACAS2D_EINVAL; }
    if (!state_complete(st)) { set_error("acas2d_rollout_policy: NULL state or a NULL state buffer"); return ACAS2D_EINVAL; }
    if (!io_->actions || !io_->obs || !io_->reward || !io_->done || !io_->outcome || !obs_in) {
        set_error("acas2d_rollout_policy: obs_in, actions (output), obs, reward, done and outcome are required"); return ACAS2D_EINVAL; }
    if (!pol->w1t || !pol->b1 || !pol->w2t || !pol->b2 || !pol->w3 || !pol->b3 || pol->hidden != kPolicyHidden) {
        set_error("acas2d_rollout_policy: six weight buffers and hidden == %d are required (got hidden = %d)", kPolicyHidden, pol->hidden);
        return ACAS2D_EINVAL; }
    if (n_traffic < 1 || n_steps < 1) { set_error("acas2d_rollout_policy: n_traffic = %d, n_steps = %d", n_traffic, n_steps); return ACAS2D_EINVAL; }
    if (n_envs < 0 || env_offset < 0) { set_error("acas2d_rollout_policy: negative n_envs / env_offset"); return ACAS2D_EINVAL; }
    if (n_envs == 0) return ACAS2D_OK;
    const Shape sh{n_traffic, 1, true};                      // one lane per env, its traffic as one vector
    bool ok = false;
#define X(C_, G_) if (G_ == 1 && C_ == n_traffic) ok = true;
    ACAS2D_PACKED_SHAPES(X)
#undef X
    if (!ok) {
        set_error("acas2d_rollout_policy: n_traffic = %d has no thread-per-env shape for this element type", n_traffic);
        return ACAS2D_EINVAL;
    }
    Geometry g;
    if (int rc = geometry_for<T>(sh, n_envs, n_traffic, &g)) return rc;
    const Params<T> p = make_params<T>(*cfg);
    const StepResetParams<T, true> rp = make_reset_params<double, T>(*cfg);
    const State<T> s = make_state<T>(*st);
    const StepIO<T> io{(const T*)io_->actions, (T*)io_->obs, (T*)io_->reward, io_->done, io_->outcome,
                       (T*)io_->term_obs, (T*)io_->ep_return, io_->ep_steps};
    PolicyW pw{(const float*)pol->w1t, (const float*)pol->b1, (const float*)pol->w2t, (const float*)pol->b2,
               (const float*)pol->w3, (const float*)pol->b3, const_cast<void*>(io_->actions), obs_in,
               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u};
    if (ac) {
        if (!ac->v1t || !ac->vb1 || !ac->v2t || !ac->vb2 || !ac->v3 || !ac->vb3 || !ac->log_std || !ac->values || !ac->logp) {
            set_error("acas2d_collect: the value net, log_std, values and logp are required"); return ACAS2D_EINVAL; }
        pw.v1t = (const float*)ac->v1t; pw.vb1 = (const float*)ac->vb1; pw.v2t = (const float*)ac->v2t;
        pw.vb2 = (const float*)ac->vb2; pw.v3 = (const float*)ac->v3; pw.vb3 = (const float*)ac->vb3;
        pw.log_std = (const float*)ac->log_std; pw.values_out = ac->values; pw.logp_out = ac->logp;
        pw.nk0 = (uint32_t)ac->noise_seed; pw.nk1 = (uint32_t)(ac->noise_seed >> 32); pw.noise_step = ac->noise_step;
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#define X(C_, G_) if constexpr (G_ == 1) { if (C_ == n_traffic) policy_shape<T, FAST, C_>(g, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic, n_steps, pw, ac != nullptr); }
    ACAS2D_PACKED_SHAPES(X)
#undef X
    return check_launch("acas2d_rollout_policy launch");
}

template <typename T, bool FAST>
static int launch_reset_impl(const Acas2dConfig* cfg, const Acas2dState* st, const uint8_t* mask, void* obs,
                 int32_t do_init, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                 hipStream_t stream) {
    if (!cfg) { set_error("acas2d_reset: NULL cfg"); return ACAS2D_EINVAL; }
    if (!state_complete(st)) { set_error("acas2d_reset: NULL state or a NULL state buffer"); return ACAS2D_EINVAL; }
    if (n_traffic < 1) { set_error("acas2d_reset: n_traffic = %d", n_traffic); return ACAS2D_EINVAL; }
    if (n_envs < 0 || env_offset < 0) { set_error("acas2d_reset: negative n_envs / env_offset"); return ACAS2D_EINVAL; }
    if (n_envs == 0) return ACAS2D_OK;
    Shape sh;
    if (int rc = resolve_shape<T>(n_traffic, &sh)) return rc;
    Geometry g;
    if (int rc = geometry_for<T>(sh, n_envs, n_traffic, &g)) return rc;
    const Params<T> p = make_params<T>(*cfg);
    const StepResetParams<T, true> rp = make_reset_params<double, T>(*cfg);
    const State<T> s = make_state<T>(*st);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (do_init < 0) { set_error("acas2d_reset: do_init = %d", do_init); return ACAS2D_EINVAL; }
    if (sh.packed) {
#define X(C_, G_) if (sh.C == C_ && sh.G == G_) reset_shape<T, FAST, C_, G_, true>(g, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic);
        ACAS2D_PACKED_SHAPES(X)
#undef X
    } else {
        switch (sh.G) {
            case 1:  reset_shape<T, FAST, 1, 1, false>(g, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
            case 4:  reset_shape<T, FAST, 1, 4, false>(g, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
            case 16: reset_shape<T, FAST, 1, 16, false>(g, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
            default: reset_shape<T, FAST, 1, 64, false>(g, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
        }
    }
    return check_launch("acas2d_reset launch");
}

// Formulation: the element type's own -- FAST for float32, EXACT for float64 -- unless the
// configuration asks for ACAS2D_MATH_FAST, which gives the float64 entry points the algebraic formulation in
// float64 arithmetic (float32 has no other).  Chosen per call; nothing is cached between calls.
template <typename T>
static bool fast_math(const Acas2dConfig* cfg) { return kFast || (cfg && cfg->math == ACAS2D_MATH_FAST); }

template <typename T>
int launch_step(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dState* st_out, const Acas2dStepIO* io, uint32_t flags,
                uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream) {
    return fast_math<T>(cfg) ? launch_step_impl<T, true>(cfg, st, st_out, io, flags, seed, env_offset, n_envs, n_traffic, stream)
                             : launch_step_impl<T, kFast>(cfg, st, st_out, io, flags, seed, env_offset, n_envs, n_traffic, stream);
}
template <typename T>
int launch_rollout(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io, int32_t n_steps,
                   uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream) {
    return fast_math<T>(cfg) ? launch_rollout_impl<T, true>(cfg, st, io, n_steps, seed, env_offset, n_envs, n_traffic, stream)
                             : launch_rollout_impl<T, kFast>(cfg, st, io, n_steps, seed, env_offset, n_envs, n_traffic, stream);
}
template <typename T>
int launch_rollout_policy(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io,
                          const Acas2dPolicy* pol, const void* obs_in, int32_t n_steps, uint64_t seed,
                          int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream) {
    return fast_math<T>(cfg)
               ? launch_rollout_policy_impl<T, true>(cfg, st, io, pol, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, stream)
               : launch_rollout_policy_impl<T, kFast>(cfg, st, io, pol, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, stream);
}
template <typename T>
int launch_collect(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io, const Acas2dActorCritic* ac,
                   const void* obs_in, int32_t n_steps, uint64_t seed, int64_t env_offset, int64_t n_envs,
                   int32_t n_traffic, hipStream_t stream) {
    if (!ac) { set_error("acas2d_collect: NULL actor-critic"); return ACAS2D_EINVAL; }
    return fast_math<T>(cfg)
               ? launch_rollout_policy_impl<T, true>(cfg, st, io, &ac->actor, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, stream, ac)
               : launch_rollout_policy_impl<T, kFast>(cfg, st, io, &ac->actor, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, stream, ac);
}
template <typename T>
int launch_reset(const Acas2dConfig* cfg, const Acas2dState* st, const uint8_t* mask, void* obs,
                 int32_t do_init, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                 hipStream_t stream) {
    return fast_math<T>(cfg) ? launch_reset_impl<T, true>(cfg, st, mask, obs, do_init, seed, env_offset, n_envs, n_traffic, stream)
                             : launch_reset_impl<T, kFast>(cfg, st, mask, obs, do_init, seed, env_offset, n_envs, n_traffic, stream);
}

// 1 when acas2d_step_* (auto-reset) would take the kernel whose loads all go through preloaded base pointers for this
// state: float32, a packed work shape, the consecutive layout of arena_layout()
template <typename T>
int state_consecutive(const Acas2dState* st, int64_t n_envs, int32_t n_traffic) {
    if (!state_complete(st) || n_envs <= 0 || n_traffic < 1) return 0;
    Shape sh;
    if (resolve_shape<T>(n_traffic, &sh) != ACAS2D_OK || !sh.packed) return 0;
    return arena_layout<T>(make_state<T>(*st), n_envs, n_traffic, sh.G) ? 1 : 0;
}

template <typename T>
int shape_geometry(int64_t n_envs, int32_t n_traffic, int32_t* lanes, int32_t* per_lane, int64_t* grid) {
    Shape sh;
    if (int rc = resolve_shape<T>(n_traffic, &sh)) return rc;
    Geometry g;
    if (int rc = geometry_for<T>(sh, n_envs, n_traffic, &g)) return rc;
    *lanes = sh.G; *per_lane = sh.packed ? sh.C : -1; *grid = g.grid;
    return ACAS2D_OK;
}

}  // namespace acas2d
