// acas2d_ppo.hip -- one PPO minibatch update of the SB3 MlpPolicy actor-critic as TWO hand-written launches.
//
// What it replaces: the body of the minibatch loop of SB3 1.1.0's PPO.train() as training_main.py:44-52 runs it
// (`PPO('MlpPolicy', env).learn()`): forward of the separate 2 x 64 tanh actor / critic on the minibatch, the clipped
// surrogate + value + entropy loss, backward, clip_grad_norm_, Adam -- ~60 library kernels per minibatch when it runs
// as torch ops (gym-acas2d_amd/ppo.py, ppo_loss()), and what bounds a PPO iteration on the device-resident env.
//
//   ppo_grad_kernel    one wave per 64 samples and network (blockIdx.y: 0 actor, 1 critic), one lane per sample.
//                      The lane gathers its sample by the minibatch's index buffer, runs the network forward with the
//                      (wave-uniform) weights coming through scalar loads, derives d loss / d output from the PPO loss
//                      (the advantage statistics of the WHOLE minibatch are recomputed by every actor wave: 2 B loads
//                      per lane, no extra launch, no grid sync), and back-propagates to the pre-activations.  The
//                      per-sample vectors (h1, h2, dz1, dz2) live in LDS, row stride 65 so that "every lane writes its
//                      own row's element i" and "every lane reads column t of row s" are both conflict-free; the
//                      weight gradients are then sums over the 64 samples of outer products, taken by thread t for
//                      row t of each weight matrix, and added to the global gradient with float atomics.
//   ppo_apply_kernel   one workgroup: the global gradient norm, torch.nn.utils.clip_grad_norm_'s coefficient, Adam
//                      (torch.optim.Adam's bias-corrected form) on the 13 parameter tensors in place, gradient zeroed
//                      for the next minibatch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acas2d.h"

namespace acas2d {
void set_error(const char* fmt, ...);

namespace {

constexpr int kH = 64;               // hidden width of SB3's MlpPolicy
constexpr int kRow = 65;             // LDS row stride of a per-sample 64-vector (conflict-free rows AND columns)
#define ACAS2D_C4 __attribute__((address_space(4)))

// gradient / moment block of one network, in floats: w1 [64][D], b1 [64], w2 [64][64], b2 [64], w3 [64], b3 [1]
__host__ __device__ constexpr int net_size(int D) { return kH * D + kH + kH * kH + kH + kH + 1; }
__host__ __device__ constexpr int off_b1(int D) { return kH * D; }
__host__ __device__ constexpr int off_w2(int D) { return kH * D + kH; }
__host__ __device__ constexpr int off_b2(int D) { return kH * D + kH + kH * kH; }
__host__ __device__ constexpr int off_w3(int D) { return kH * D + kH + kH * kH + kH; }
__host__ __device__ constexpr int off_b3(int D) { return kH * D + kH + kH * kH + kH + kH; }

struct NetW { const float *w1, *b1, *w2, *b2, *w3, *b3; };      // torch layouts: [out][in]

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int D>
__global__ __launch_bounds__(64) void ppo_grad_kernel(NetW actor, NetW critic, const float* log_std_p, const float* obs,
                                                      const float* act, const float* old_logp, const float* adv,
                                                      const float* ret, const int64_t* idx, int B, float clip_range,
                                                      float vf_coef, float* grad, float* stats) {
    extern __shared__ float lds[];
    float* l_h1 = lds;                       // [64][65]
    float* l_h2 = l_h1 + 64 * kRow;
    float* l_dz1 = l_h2 + 64 * kRow;
    float* l_dz2 = l_dz1 + 64 * kRow;
    float* l_x = l_dz2 + 64 * kRow;          // [64][D + 1]
    float* l_do = l_x + 64 * (D + 1);        // [64]
    const int lane = threadIdx.x;
    const bool is_actor = blockIdx.y == 0;
    const NetW net = is_actor ? actor : critic;
    const int row = blockIdx.x * 64 + lane;
    const bool live = row < B;
    const int64_t s = idx[live ? row : 0];

    // ---- the minibatch's advantage statistics (SB3 normalises per minibatch; torch.std is Bessel-corrected)
    float a_mean = 0.0f, a_std = 1.0f;
    if (is_actor) {
        float sum = 0.0f;
        for (int i = lane; i < B; i += 64) sum += adv[idx[i]];
        a_mean = wave_sum(sum) / (float)B;
        float sq = 0.0f;
        for (int i = lane; i < B; i += 64) { const float d = adv[idx[i]] - a_mean; sq = fmaf(d, d, sq); }
        a_std = sqrtf(wave_sum(sq) / (float)(B > 1 ? B - 1 : 1));
    }

    // ---- forward: obs -> Linear(D, 64) tanh -> Linear(64, 64) tanh -> Linear(64, 1), weights by scalar loads
    float x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { x[k] = obs[s * D + k]; l_x[lane * (D + 1) + k] = x[k]; }
    const float ACAS2D_C4* w1 = (const float ACAS2D_C4*)net.w1;
    const float ACAS2D_C4* b1 = (const float ACAS2D_C4*)net.b1;
    const float ACAS2D_C4* w2 = (const float ACAS2D_C4*)net.w2;
    const float ACAS2D_C4* b2 = (const float ACAS2D_C4*)net.b2;
    const float ACAS2D_C4* w3 = (const float ACAS2D_C4*)net.w3;
    const float ACAS2D_C4* b3 = (const float ACAS2D_C4*)net.b3;
    for (int i = 0; i < kH; ++i) {
        float z = b1[i];
#pragma unroll
        for (int k = 0; k < D; ++k) z = fmaf(w1[i * D + k], x[k], z);
        l_h1[lane * kRow + i] = tanhf(z);
    }
    float h1[kH];
#pragma unroll
    for (int k = 0; k < kH; ++k) h1[k] = l_h1[lane * kRow + k];
    float out = b3[0];
    for (int i = 0; i < kH; ++i) {
        float z = b2[i];
#pragma unroll
        for (int k = 0; k < kH; ++k) z = fmaf(w2[i * kH + k], h1[k], z);
        const float h2 = tanhf(z);
        l_h2[lane * kRow + i] = h2;
        out = fmaf(w3[i], h2, out);
    }

    // ---- d loss / d output (SB3 PPO.train(): clipped surrogate on minibatch-normalised advantages, MSE value loss)
    float dout = 0.0f, dls = 0.0f, pg_s = 0.0f, vf_s = 0.0f;
    if (live) {
        if (is_actor) {
            const float ls = log_std_p[0], inv_var = expf(-2.0f * ls);
            const float diff = act[s] - out;
            const float logp = -0.5f * diff * diff * inv_var - ls - 0.9189385332046727f;
            const float a = (adv[s] - a_mean) / (a_std + 1e-8f);
            const float ratio = expf(logp - old_logp[s]);
            const float surr1 = a * ratio, surr2 = a * fminf(fmaxf(ratio, 1.0f - clip_range), 1.0f + clip_range);
            pg_s = -fminf(surr1, surr2) / (float)B;
            const float dlogp = (surr1 <= surr2) ? -(a * ratio) / (float)B : 0.0f;     // torch.min: ties go to the first operand
            dout = dlogp * diff * inv_var;                       // d logp / d mean
            dls = dlogp * (diff * diff * inv_var - 1.0f);        // d logp / d log_std
        } else {
            const float e = out - ret[s];
            vf_s = e * e / (float)B;
            dout = vf_coef * 2.0f * e / (float)B;
        }
    }
    l_do[lane] = dout;

    // ---- backward to the pre-activations: dz2 = dout w3 (1 - h2^2), dh1 = W2^T dz2, dz1 = dh1 (1 - h1^2)
    float dh1[kH];
#pragma unroll
    for (int k = 0; k < kH; ++k) dh1[k] = 0.0f;
    for (int i = 0; i < kH; ++i) {
        const float h2 = l_h2[lane * kRow + i];
        const float dz2 = dout * w3[i] * (1.0f - h2 * h2);
        l_dz2[lane * kRow + i] = dz2;
#pragma unroll
        for (int k = 0; k < kH; ++k) dh1[k] = fmaf(w2[i * kH + k], dz2, dh1[k]);
    }
#pragma unroll
    for (int k = 0; k < kH; ++k) l_dz1[lane * kRow + k] = dh1[k] * (1.0f - h1[k] * h1[k]);
    __syncthreads();

    // ---- weight gradients: thread t takes row t of every weight matrix, summed over the wave's 64 samples
    float* g = grad + (is_actor ? 0 : net_size(D));
    const int t = lane;
    {
        float acc[kH];
#pragma unroll
        for (int j = 0; j < kH; ++j) acc[j] = 0.0f;
        float bsum = 0.0f;
        for (int q = 0; q < 64; ++q) {
            const float dz = l_dz2[q * kRow + t];
            bsum += dz;
#pragma unroll
            for (int j = 0; j < kH; ++j) acc[j] = fmaf(dz, l_h1[q * kRow + j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < kH; ++j) atomicAdd(g + off_w2(D) + t * kH + j, acc[j]);
        atomicAdd(g + off_b2(D) + t, bsum);
    }
    {
        float acc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) acc[k] = 0.0f;
        float bsum = 0.0f, w3sum = 0.0f;
        for (int q = 0; q < 64; ++q) {
            const float dz = l_dz1[q * kRow + t];
            bsum += dz;
            w3sum = fmaf(l_do[q], l_h2[q * kRow + t], w3sum);
#pragma unroll
            for (int k = 0; k < D; ++k) acc[k] = fmaf(dz, l_x[q * (D + 1) + k], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) atomicAdd(g + t * D + k, acc[k]);
        atomicAdd(g + off_b1(D) + t, bsum);
        atomicAdd(g + off_w3(D) + t, w3sum);
    }
    const float dsum = wave_sum(dout), lsum = wave_sum(dls), pgsum = wave_sum(pg_s), vfsum = wave_sum(vf_s);
    if (lane == 0) {
        atomicAdd(g + off_b3(D), dsum);
        if (is_actor) { atomicAdd(grad + 2 * net_size(D), lsum); atomicAdd(stats + 0, pgsum); }
        else atomicAdd(stats + 1, vfsum);
    }
}

struct Segment { float* p; int offset, count; };
struct Segments { Segment s[13]; };

// clip_grad_norm_ + Adam for all parameters, in place; the gradient and the statistics accumulate for ONE minibatch
__global__ __launch_bounds__(1024) void ppo_apply_kernel(Segments seg, int total, float* grad, float* m, float* v,
                                                         int32_t* step, float ent_coef, float max_norm, float lr,
                                                         float beta1, float beta2, float eps, float* stats) {
    __shared__ float red[16];
    __shared__ float coef_s;
    const int tid = threadIdx.x;
    if (tid == 0) grad[total - 1] -= ent_coef;               // d(ent_coef * -entropy) / d log_std (the last entry)
    __syncthreads();
    float sq = 0.0f;
    for (int i = tid; i < total; i += 1024) sq = fmaf(grad[i], grad[i], sq);
    sq = wave_sum(sq);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.0f;
        for (int i = 0; i < 16; ++i) tot += red[i];
        const float norm = sqrtf(tot);
        coef_s = fminf(1.0f, max_norm / (norm + 1e-6f));      // torch.nn.utils.clip_grad_norm_
        stats[2] = norm;
        stats[4] = stats[0]; stats[5] = stats[1];             // the minibatch's policy / value loss, for the log
        stats[0] = 0.0f; stats[1] = 0.0f;
    }
    __syncthreads();
    const float coef = coef_s;
    const int tstep = step[0] + 1;
    const float bc1 = 1.0f - powf(beta1, (float)tstep), bc2 = 1.0f - powf(beta2, (float)tstep);
    for (int k = 0; k < 13; ++k) {
        const Segment sg = seg.s[k];
        for (int i = tid; i < sg.count; i += 1024) {
            const int gi = sg.offset + i;
            const float gr = grad[gi] * coef;
            const float mm = fmaf(beta1, m[gi], (1.0f - beta1) * gr);
            const float vv = fmaf(beta2, v[gi], (1.0f - beta2) * gr * gr);
            m[gi] = mm; v[gi] = vv;
            sg.p[i] -= (lr / bc1) * mm / (sqrtf(vv) / sqrtf(bc2) + eps);       // torch.optim.Adam
            grad[gi] = 0.0f;
        }
    }
    __syncthreads();
    if (tid == 0) step[0] = tstep;
}

// Dynamic LDS of ppo_grad_kernel<D>: 4 x 64 per-sample vectors with row stride 65, the observations, one scratch row
// (69.6 - 74.8 KB for D = 8 ... 29).  More than the 64 KB a HIP launch gets without asking: the attribute below raises
// the kernel's limit, and the size is checked against what the device reports (gfx950: 160 KB per workgroup).
template <int D>
int launch_grad(const Acas2dPpoUpdate& u, hipStream_t stream) {
    const NetW a{(const float*)u.actor_w1, (const float*)u.actor_b1, (const float*)u.actor_w2, (const float*)u.actor_b2,
                 (const float*)u.actor_w3, (const float*)u.actor_b3};
    const NetW c{(const float*)u.critic_w1, (const float*)u.critic_b1, (const float*)u.critic_w2, (const float*)u.critic_b2,
                 (const float*)u.critic_w3, (const float*)u.critic_b3};
    const size_t lds_bytes = (size_t)(4 * 64 * kRow + 64 * (D + 1) + 64) * sizeof(float);
    static int lds_limit = -1;                               // per instantiation, set once
    if (lds_limit < 0) {
        int dev = 0, optin = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) {
            set_error("acas2d_ppo_update: cannot query the device's LDS size"); return ACAS2D_EHIP; }
        if ((size_t)optin >= lds_bytes)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_grad_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipGetLastError();
        lds_limit = optin;
    }
    if ((size_t)lds_limit < lds_bytes) {
        set_error("acas2d_ppo_update: the gradient kernel needs %zu bytes of LDS per workgroup, this device offers %d "
                  "(built for gfx950's 160 KB)", lds_bytes, lds_limit);
        return ACAS2D_EINVAL;
    }
    hipLaunchKernelGGL((ppo_grad_kernel<D>), dim3((unsigned)((u.n_rows + 63) / 64), 2), dim3(64), lds_bytes, stream, a, c,
                       (const float*)u.log_std, (const float*)u.obs, (const float*)u.act, (const float*)u.old_logp,
                       (const float*)u.adv, (const float*)u.ret, (const int64_t*)u.idx, u.n_rows, u.clip_range, u.vf_coef,
                       (float*)u.grad, (float*)u.stats);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { set_error("acas2d_ppo_update gradient launch: %s", hipGetErrorString(err)); return ACAS2D_EHIP; }
    return ACAS2D_OK;
}

}  // namespace
}  // namespace acas2d

using namespace acas2d;

extern "C" int acas2d_ppo_workspace_floats(int32_t obs_dim) { return 2 * net_size(obs_dim) + 1; }

extern "C" int acas2d_ppo_update_f32(const Acas2dPpoUpdate* u, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!u) { set_error("acas2d_ppo_update: NULL argument"); return ACAS2D_EINVAL; }
    const void* need[] = {u->actor_w1, u->actor_b1, u->actor_w2, u->actor_b2, u->actor_w3, u->actor_b3, u->critic_w1, u->critic_b1,
                          u->critic_w2, u->critic_b2, u->critic_w3, u->critic_b3, u->log_std, u->obs, u->act, u->old_logp, u->adv,
                          u->ret, u->idx, u->grad, u->adam_m, u->adam_v, u->adam_step, u->stats};
    for (const void* p : need) if (!p) { set_error("acas2d_ppo_update: every pointer is required"); return ACAS2D_EINVAL; }
    if (u->n_rows < 2) { set_error("acas2d_ppo_update: n_rows = %d (the advantage normalisation needs 2)", u->n_rows); return ACAS2D_EINVAL; }
    const int D = u->obs_dim;
    int rc;
    switch (D) {
        case 8: rc = launch_grad<8>(*u, stream); break;
        case 11: rc = launch_grad<11>(*u, stream); break;
        case 14: rc = launch_grad<14>(*u, stream); break;
        case 17: rc = launch_grad<17>(*u, stream); break;
        case 29: rc = launch_grad<29>(*u, stream); break;
        default: set_error("acas2d_ppo_update: obs_dim = %d (built for n_traffic in {1, 2, 3, 4, 8})", D); return ACAS2D_EINVAL;
    }
    if (rc != ACAS2D_OK) return rc;                      // (a failed gradient launch must not read as a zero gradient)
    if (u->max_grad_norm < 0.0f) return ACAS2D_OK;      // tests: the raw gradient stays in `grad`, nothing is applied
    Segments seg;
    float* ptrs[13] = {(float*)u->actor_w1, (float*)u->actor_b1, (float*)u->actor_w2, (float*)u->actor_b2, (float*)u->actor_w3,
                       (float*)u->actor_b3, (float*)u->critic_w1, (float*)u->critic_b1, (float*)u->critic_w2, (float*)u->critic_b2,
                       (float*)u->critic_w3, (float*)u->critic_b3, (float*)u->log_std};
    const int cnt[6] = {kH * D, kH, kH * kH, kH, kH, 1};
    int off = 0;
    for (int k = 0; k < 12; ++k) { seg.s[k] = Segment{ptrs[k], off, cnt[k % 6]}; off += cnt[k % 6]; }
    seg.s[12] = Segment{ptrs[12], off, 1};
    hipLaunchKernelGGL(ppo_apply_kernel, dim3(1), dim3(1024), 0, stream, seg, off + 1, (float*)u->grad, (float*)u->adam_m,
                       (float*)u->adam_v, (int32_t*)u->adam_step, u->ent_coef, u->max_grad_norm, u->learning_rate, u->beta1,
                       u->beta2, u->adam_eps, (float*)u->stats);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { set_error("acas2d_ppo_update launch: %s", hipGetErrorString(err)); return ACAS2D_EHIP; }
    return ACAS2D_OK;
}
