// acas2d_api.hip -- the extern "C" boundary declared in include/acas2d.h.
#include <stdarg.h>
#include <stdio.h>

#include "acas2d_kernels.hpp"

namespace acas2d {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// Default work shape for (n_traffic, element size): packed (C traffic per lane as one 16-byte
// vector, G = N / C lanes per env) where N tiles that way, else the generic strided walk with
// G in {1, 4, 16, 64} lanes per env.  Tuned on MI355X.
Shape choose_shape(int n, int elem_size) {
    const int c16 = 16 / elem_size;                       // values per 16-byte vector
    if (n == 1) return Shape{1, 1, true};
    if (n == 2) return Shape{2, 1, true};
    if (n == 3) return Shape{3, 1, true};
    // float64: 4 aircraft per lane (two 16-byte accesses per field) where N allows -- half the lanes per env, so
    // half the redundant player-side arithmetic: 20.7 vs 25.0 us per launch at 65 536 x 8 (EXACT), 12.4 vs 15.0 (FAST)
    if (elem_size == 8 && n % 4 == 0) {
        const int g = n / 4;
        if (g <= 64 && (g & (g - 1)) == 0) return Shape{4, g, true};
    }
    if (n % c16 == 0) {
        const int g = n / c16;
        if (g <= 64 && (g & (g - 1)) == 0) return Shape{c16, g, true};
    }
    return Shape{1, n >= 64 ? 64 : (n >= 16 ? 16 : (n >= 4 ? 4 : 1)), false};
}

template <typename T>
int shape_geometry(int64_t n_envs, int32_t n_traffic, int32_t* lanes, int32_t* per_lane, int64_t* grid);
template <typename T>
int state_consecutive(const Acas2dState* st, int64_t n_envs, int32_t n_traffic);

}  // namespace acas2d

using namespace acas2d;

extern "C" {

int acas2d_abi_version(void) { return ACAS2D_ABI_VERSION; }
size_t acas2d_config_size(void) { return sizeof(Acas2dConfig); }
size_t acas2d_state_size(void) { return sizeof(Acas2dState); }
const char* acas2d_last_error(void) { return g_error; }

int acas2d_step_f32(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dState* state_out, const Acas2dStepIO* io,
                    uint32_t flags, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_step<float>(cfg, state, state_out, io, flags, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_step_f64(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dState* state_out, const Acas2dStepIO* io,
                    uint32_t flags, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_step<double>(cfg, state, state_out, io, flags, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_rollout_f32(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io, int32_t n_steps,
                       uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_rollout<float>(cfg, state, io, n_steps, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_rollout_f64(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io, int32_t n_steps,
                       uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_rollout<double>(cfg, state, io, n_steps, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_rollout_policy_f32(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io,
                              const Acas2dPolicy* policy, const void* obs_in, int32_t n_steps, uint64_t seed,
                              int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_rollout_policy<float>(cfg, state, io, policy, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_rollout_policy_f64(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io,
                              const Acas2dPolicy* policy, const void* obs_in, int32_t n_steps, uint64_t seed,
                              int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_rollout_policy<double>(cfg, state, io, policy, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_collect_f32(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io, const Acas2dActorCritic* ac,
                       const void* obs_in, int32_t n_steps, uint64_t seed, int64_t env_offset, int64_t n_envs,
                       int32_t n_traffic, void* stream) {
    return launch_collect<float>(cfg, state, io, ac, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_collect_f64(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io, const Acas2dActorCritic* ac,
                       const void* obs_in, int32_t n_steps, uint64_t seed, int64_t env_offset, int64_t n_envs,
                       int32_t n_traffic, void* stream) {
    return launch_collect<double>(cfg, state, io, ac, obs_in, n_steps, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_reset_f32(const Acas2dConfig* cfg, const Acas2dState* state, const uint8_t* mask, void* obs,
                     int32_t do_init, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                     void* stream) {
    return launch_reset<float>(cfg, state, mask, obs, do_init, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_reset_f64(const Acas2dConfig* cfg, const Acas2dState* state, const uint8_t* mask, void* obs,
                     int32_t do_init, uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic,
                     void* stream) {
    return launch_reset<double>(cfg, state, mask, obs, do_init, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}

int acas2d_state_is_consecutive(const Acas2dState* state, int64_t n_envs, int32_t n_traffic, int32_t elem_size) {
    return elem_size == 4 ? state_consecutive<float>(state, n_envs, n_traffic)
                          : (elem_size == 8 ? state_consecutive<double>(state, n_envs, n_traffic) : 0);
}
int acas2d_launch_geometry(int64_t n_envs, int32_t n_traffic, int32_t elem_size, int32_t* lanes_per_env,
                           int32_t* traffic_per_lane, int32_t* block_threads, int64_t* grid_blocks) {
    if (n_traffic < 1 || n_envs < 0 || (elem_size != 4 && elem_size != 8)) {
        set_error("acas2d_launch_geometry: bad sizes"); return ACAS2D_EINVAL; }
    int32_t g = 0, c = 0;
    int64_t grid = 0;
    const int rc = elem_size == 4 ? shape_geometry<float>(n_envs, n_traffic, &g, &c, &grid)
                                  : shape_geometry<double>(n_envs, n_traffic, &g, &c, &grid);
    if (rc) return rc;
    if (lanes_per_env) *lanes_per_env = g;
    if (traffic_per_lane) *traffic_per_lane = c;
    if (block_threads) *block_threads = kBlock;
    if (grid_blocks) *grid_blocks = grid;
    return ACAS2D_OK;
}

}  // extern "C"
