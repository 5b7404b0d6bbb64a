// acas2d_launch.inl -- host-side launchers, included by acas2d_f32.hip and acas2d_f64.hip which
// then instantiate launch_step<T> / launch_reset<T> for their element type.  (Two translation
// units so that the float64 parity build can be compiled with -ffp-contract=off while the
// float32 throughput build keeps fused multiply-adds.)
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
    p.max_steps = c.max_steps;
    return p;
}

static ResetParams make_reset_params(const Acas2dConfig& c) {
    return ResetParams{c.own_x0, c.own_y0, c.own_v, c.own_heading0, c.own_heading_jitter, c.goal_x,
                       c.goal_y, c.t0_x, c.t0_y_base, c.t0_y_span, c.t0_heading_base,
                       c.t0_heading_step, c.t0_heading_jitter, c.tn_x_max, c.tn_y_max,
                       c.speed_factor_min, c.speed_factor_max, c.airspeed};
}

template <typename T>
static State<T> make_state(const Acas2dState& s) {
    return State<T>{(T*)s.own_x, (T*)s.own_y, (T*)s.own_psi, (T*)s.own_v, (T*)s.goal_x, (T*)s.goal_y,
                    (T*)s.trf_x, (T*)s.trf_y, (T*)s.trf_psi, (T*)s.trf_v, s.steps,
                    (T*)s.total_reward, s.status, s.episode};
}

static bool state_complete(const Acas2dState* s) {
    return s && s->own_x && s->own_y && s->own_psi && s->own_v && s->goal_x && s->goal_y && s->trf_x &&
           s->trf_y && s->trf_psi && s->trf_v && s->steps && s->total_reward && s->status && s->episode;
}

static int grid_for(int64_t n_envs, int G, unsigned* grid) {
    const int64_t blocks = (n_envs * G + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) { set_error("n_envs * lanes_per_env = %lld * %d exceeds the grid limit", (long long)n_envs, G); return ACAS2D_EINVAL; }
    *grid = (unsigned)blocks;
    return ACAS2D_OK;
}

static int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(err)); return ACAS2D_EHIP; }
    return ACAS2D_OK;
}

template <typename T, int G>
static void step_g(bool auto_reset, unsigned grid, hipStream_t stream, const Params<T>& p,
                   const ResetParams& rp, const State<T>& s, const StepIO<T>& io, uint32_t k0, uint32_t k1,
                   int64_t env_offset, int64_t n_envs, int N) {
    if (auto_reset)
        hipLaunchKernelGGL((step_kernel<T, G, true>), dim3(grid), dim3(kBlock), 0, stream, p, rp, s, io, k0, k1, env_offset, n_envs, N);
    else
        hipLaunchKernelGGL((step_kernel<T, G, false>), dim3(grid), dim3(kBlock), 0, stream, p, rp, s, io, k0, k1, env_offset, n_envs, N);
}

template <typename T>
int launch_step(const Acas2dConfig* cfg, const Acas2dState* st, const Acas2dStepIO* io_, uint32_t flags,
                uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, hipStream_t stream) {
    if (!cfg || !io_) { set_error("acas2d_step: NULL cfg / io"); return ACAS2D_EINVAL; }
    if (!state_complete(st)) { set_error("acas2d_step: NULL state or a NULL state buffer"); return ACAS2D_EINVAL; }
    if (!io_->actions || !io_->obs || !io_->reward || !io_->done || !io_->outcome) {
        set_error("acas2d_step: actions, obs, reward, done and outcome are required"); return ACAS2D_EINVAL; }
    if (n_traffic < 1) { set_error("acas2d_step: n_traffic = %d (the reference needs traffic[0], game.py:254)", n_traffic); return ACAS2D_EINVAL; }
    if (n_envs < 0 || env_offset < 0) { set_error("acas2d_step: negative n_envs / env_offset"); return ACAS2D_EINVAL; }
    if (n_envs == 0) return ACAS2D_OK;
    const int G = lanes_per_env(n_traffic);
    unsigned grid;
    if (int rc = grid_for(n_envs, G, &grid)) return rc;
    const Params<T> p = make_params<T>(*cfg);
    const ResetParams rp = make_reset_params(*cfg);
    const State<T> s = make_state<T>(*st);
    const StepIO<T> io{(const T*)io_->actions, (T*)io_->obs, (T*)io_->reward, io_->done, io_->outcome,
                       (T*)io_->term_obs, (T*)io_->ep_return, io_->ep_steps};
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const bool ar = (flags & ACAS2D_AUTO_RESET) != 0;
    switch (G) {
        case 1:  step_g<T, 1>(ar, grid, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
        case 2:  step_g<T, 2>(ar, grid, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
        case 4:  step_g<T, 4>(ar, grid, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
        case 8:  step_g<T, 8>(ar, grid, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
        case 16: step_g<T, 16>(ar, grid, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
        case 32: step_g<T, 32>(ar, grid, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
        default: step_g<T, 64>(ar, grid, stream, p, rp, s, io, k0, k1, env_offset, n_envs, n_traffic); break;
    }
    return check_launch("acas2d_step launch");
}

template <typename T, int G>
static void reset_g(unsigned grid, hipStream_t stream, const Params<T>& p, const ResetParams& rp,
                    const State<T>& s, const uint8_t* mask, T* obs, int do_init, uint32_t k0, uint32_t k1,
                    int64_t env_offset, int64_t n_envs, int N) {
    hipLaunchKernelGGL((reset_kernel<T, G>), dim3(grid), dim3(kBlock), 0, stream, p, rp, s, mask, obs, do_init, k0, k1, env_offset, n_envs, N);
}

template <typename T>
int launch_reset(const Acas2dConfig* cfg, const Acas2dState* st, const uint8_t* mask, void* obs,
                 int32_t do_init, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                 hipStream_t stream) {
    if (!cfg) { set_error("acas2d_reset: NULL cfg"); return ACAS2D_EINVAL; }
    if (!state_complete(st)) { set_error("acas2d_reset: NULL state or a NULL state buffer"); return ACAS2D_EINVAL; }
    if (n_traffic < 1) { set_error("acas2d_reset: n_traffic = %d", n_traffic); return ACAS2D_EINVAL; }
    if (n_envs < 0 || env_offset < 0) { set_error("acas2d_reset: negative n_envs / env_offset"); return ACAS2D_EINVAL; }
    if (n_envs == 0) return ACAS2D_OK;
    const int G = lanes_per_env(n_traffic);
    unsigned grid;
    if (int rc = grid_for(n_envs, G, &grid)) return rc;
    const Params<T> p = make_params<T>(*cfg);
    const ResetParams rp = make_reset_params(*cfg);
    const State<T> s = make_state<T>(*st);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    switch (G) {
        case 1:  reset_g<T, 1>(grid, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
        case 2:  reset_g<T, 2>(grid, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
        case 4:  reset_g<T, 4>(grid, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
        case 8:  reset_g<T, 8>(grid, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
        case 16: reset_g<T, 16>(grid, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
        case 32: reset_g<T, 32>(grid, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
        default: reset_g<T, 64>(grid, stream, p, rp, s, mask, (T*)obs, do_init, k0, k1, env_offset, n_envs, n_traffic); break;
    }
    return check_launch("acas2d_reset launch");
}

}  // namespace acas2d
