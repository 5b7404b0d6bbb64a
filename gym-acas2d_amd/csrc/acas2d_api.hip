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

// lanes per env: the power of two >= n_traffic, capped at one wave (64); more than 64 traffic
// aircraft are walked in strides of 64 by the same lanes.
int lanes_per_env(int n_traffic) {
    int g = 1;
    while (g < n_traffic && g < 64) g <<= 1;
    return g;
}

}  // namespace acas2d

using namespace acas2d;

extern "C" {

int acas2d_abi_version(void) { return ACAS2D_ABI_VERSION; }
size_t acas2d_config_size(void) { return sizeof(Acas2dConfig); }
const char* acas2d_last_error(void) { return g_error; }

int acas2d_step_f32(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io, uint32_t flags,
                    uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_step<float>(cfg, state, io, flags, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
}
int acas2d_step_f64(const Acas2dConfig* cfg, const Acas2dState* state, const Acas2dStepIO* io, uint32_t flags,
                    uint64_t seed, int64_t env_offset, int64_t n_envs, int32_t n_traffic, void* stream) {
    return launch_step<double>(cfg, state, io, flags, seed, env_offset, n_envs, n_traffic, (hipStream_t)stream);
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

int acas2d_launch_geometry(int64_t n_envs, int32_t n_traffic, int32_t* lanes, int32_t* block_threads,
                           int64_t* grid_blocks) {
    if (n_traffic < 1 || n_envs < 0) { set_error("acas2d_launch_geometry: bad sizes"); return ACAS2D_EINVAL; }
    const int g = lanes_per_env(n_traffic);
    if (lanes) *lanes = g;
    if (block_threads) *block_threads = kBlock;
    if (grid_blocks) *grid_blocks = (n_envs * g + kBlock - 1) / kBlock;
    return ACAS2D_OK;
}

}  // extern "C"
