// Using the engine from plain C/C++ through include/acas2d.h -- no Python, no torch.
//
// The reference's baseline_main.py:39-61 (constant action 0 until every episode is over, here with
// VecEnv auto-reset) on E envs: hipMalloc the struct-of-arrays state, acas2d_reset_f32 draws the
// episodes (game.py:80-116), acas2d_step_f32 advances them.  Prints one line the GPU test compares
// with ACAS2DVecEnv doing the same:  E N steps  finished  sum(reward)  checksum(obs)
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/c_abi_example.cpp \
//         -Lgym-acas2d_amd/csrc -lacas2d_hip -Wl,-rpath,gym-acas2d_amd/csrc -o c_abi_example
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "acas2d.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ACAS(x) do { int rc_ = (x); if (rc_ != ACAS2D_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, acas2d_last_error()); return 3; } } while (0)

// gym_ACAS2D/settings.py:1-54 with the normalisers of game.py:120-128 / rewards.py:22-23,46-47
static Acas2dConfig default_config() {
    const double W = 1600, H = 1000, FPS = 100, SIZE = 24, AIRSPEED = 200, MAX_STEPS = 1000;
    const double CR = 2 * SIZE, GR = 6 * SIZE, step_len = AIRSPEED / FPS * MAX_STEPS;
    Acas2dConfig c = {};
    c.dt = 1 / FPS; c.acc_lat_limit = 20 * 9.80665; c.max_steps = (int32_t)MAX_STEPS;
    c.collision_dist = 2 * CR; c.goal_radius = GR; c.safe_distance = 4 * CR;
    c.own_x0 = CR; c.own_y0 = H / 2; c.own_v = AIRSPEED; c.own_heading0 = 0; c.own_heading_jitter = 3;
    c.goal_x = W - GR; c.goal_y = H / 2;
    const double d0 = c.goal_x - c.own_x0, diag = __builtin_sqrt(W * W + H * H);
    c.d_goal_max = d0 + step_len; c.d_dev_max = step_len; c.d_sep_max = diag + 2 * step_len;
    c.d_cpa_max = diag; c.v_closing_max = 2 * AIRSPEED;
    c.rw_d_goal_max = (W - GR - 2 * SIZE) + step_len; c.rw_d_dev_max = (W - GR - 2 * SIZE) / 2;
    c.reward_goal = 1000; c.reward_collision = -1000;
    c.t0_x = W - CR; c.t0_y_base = CR; c.t0_y_span = H - 2 * CR;
    c.t0_heading_base = 145; c.t0_heading_step = 70; c.t0_heading_jitter = 15;
    c.tn_x_max = W - SIZE; c.tn_y_max = 3 * H / 5;
    c.speed_factor_min = 1; c.speed_factor_max = 1; c.airspeed = AIRSPEED;
    return c;
}

int main(int argc, char** argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 4096;
    const int32_t N = argc > 2 ? atoi(argv[2]) : 3;
    const int steps = argc > 3 ? atoi(argv[3]) : 500;
    const int D = 5 + 3 * N;
    if ((size_t)acas2d_config_size() != sizeof(Acas2dConfig) || acas2d_abi_version() != ACAS2D_ABI_VERSION) {
        fprintf(stderr, "header / library mismatch\n"); return 1; }
    const Acas2dConfig cfg = default_config();

    // one zeroed allocation per array (the library allocates nothing and keeps no state)
    auto dmalloc = [](size_t bytes, void** p) { hipError_t e = hipMalloc(p, bytes); return e == hipSuccess ? hipMemset(*p, 0, bytes) : e; };
    Acas2dState st = {};
    void** f_e[] = {&st.own_v, &st.goal_x, &st.goal_y};
    for (void** p : f_e) HIP(dmalloc(E * sizeof(float), p));
    void** f_en[] = {&st.trf_psi, &st.trf_v};
    for (void** p : f_en) HIP(dmalloc(E * N * sizeof(float), p));
    HIP(dmalloc(E, (void**)&st.status)); HIP(dmalloc(E * 4, (void**)&st.episode));
    // the arrays a step rewrites for every env are double-buffered: [2][E] / [2][E][N] each, `st` on the first halves,
    // `st2` on the second (acas2d.h, acas2d_step_*'s state_out); every other buffer is shared
    void** g_e[] = {&st.own_x, &st.own_y, &st.own_psi, &st.total_reward, (void**)&st.steps};      // 4-byte elements all
    for (void** p : g_e) HIP(dmalloc(2 * E * sizeof(float), p));
    void** g_en[] = {&st.trf_x, &st.trf_y};
    for (void** p : g_en) HIP(dmalloc(2 * E * N * sizeof(float), p));
    Acas2dState st2 = st;
    st2.own_x = (float*)st.own_x + E; st2.own_y = (float*)st.own_y + E; st2.own_psi = (float*)st.own_psi + E;
    st2.total_reward = (float*)st.total_reward + E; st2.steps = st.steps + E;
    st2.trf_x = (float*)st.trf_x + E * N; st2.trf_y = (float*)st.trf_y + E * N;
    const Acas2dState* gen[2] = {&st, &st2};
    Acas2dStepIO io = {};
    void *actions, *obs, *reward;
    HIP(dmalloc(E * sizeof(float), &actions)); HIP(dmalloc(E * D * sizeof(float), &obs)); HIP(dmalloc(E * sizeof(float), &reward));
    HIP(dmalloc(E, (void**)&io.done)); HIP(dmalloc(E, (void**)&io.outcome));
    io.actions = actions; io.obs = obs; io.reward = reward;       // side channels (term_obs, ...) left NULL

    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    const uint64_t seed = 13;
    ACAS(acas2d_reset_f32(&cfg, &st, /*mask*/ nullptr, obs, /*do_init*/ 1, seed, /*env_offset*/ 0, E, N, stream));
    std::vector<uint8_t> done(E);
    std::vector<float> rew(E), o((size_t)E * D);
    long long finished = 0;
    double reward_sum = 0;
    for (int t = 0; t < steps; ++t) {
        ACAS(acas2d_step_f32(&cfg, gen[t & 1], gen[(t + 1) & 1], &io, ACAS2D_AUTO_RESET, seed, 0, E, N, stream));   // read one generation, write the other
        HIP(hipMemcpyAsync(done.data(), io.done, E, hipMemcpyDeviceToHost, stream));
        HIP(hipMemcpyAsync(rew.data(), reward, E * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP(hipStreamSynchronize(stream));
        for (int64_t e = 0; e < E; ++e) { finished += done[e]; reward_sum += rew[e]; }
    }
    HIP(hipMemcpy(o.data(), obs, o.size() * sizeof(float), hipMemcpyDeviceToHost));
    double checksum = 0;
    for (size_t i = 0; i < o.size(); ++i) checksum += (double)o[i] * (double)(1 + i % 7);
    printf("%lld %d %d %lld %.9e %.9e\n", (long long)E, N, steps, finished, reward_sum, checksum);
    return 0;
}
