// probe_math.hip -- accuracy of gfx950 hardware transcendentals vs float64, to decide which
// the float32 step kernel may use under its 1e-5 tolerance.  Build: hipcc --offload-arch=gfx950
// -O2 probe_math.hip -o bin/probe_math ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

__global__ void k(const float* x, float* o_sin, float* o_cos, float* o_rsq, float* o_rcp, float* o_sqrt, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    o_sin[i] = __builtin_amdgcn_sinf(v);    // v_sin_f32: sin(2 pi v)
    o_cos[i] = __builtin_amdgcn_cosf(v);    // v_cos_f32
    float w = 1.0f + v * 1000.0f;
    o_rsq[i] = __builtin_amdgcn_rsqf(w);
    o_rcp[i] = __builtin_amdgcn_rcpf(w);
    o_sqrt[i] = __builtin_amdgcn_sqrtf(w);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n), rq(n), rc(n), sq(n);
    for (int i = 0; i < n; ++i) x[i] = (float)((double)i / (n - 1) * 1.02 - 0.01);   // [-0.01, 1.01] revolutions
    float *dx, *d[5];
    hipMalloc(&dx, n * 4);
    for (auto& p : d) hipMalloc(&p, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], d[4], n);
    hipMemcpy(s.data(), d[0], n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d[1], n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rq.data(), d[2], n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rc.data(), d[3], n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(sq.data(), d[4], n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, erq = 0, erc = 0, esq = 0, es_rel_small = 0;
    for (int i = 0; i < n; ++i) {
        double a = 2.0 * M_PI * (double)x[i];
        es = fmax(es, fabs(s[i] - sin(a)));
        ec = fmax(ec, fabs(c[i] - cos(a)));
        if (fabs(x[i]) < 0.01 && x[i] != 0) es_rel_small = fmax(es_rel_small, fabs(s[i] - sin(a)) / fabs(sin(a)));
        double w = (double)(1.0f + x[i] * 1000.0f);
        erq = fmax(erq, fabs(rq[i] * sqrt(w) - 1.0));
        erc = fmax(erc, fabs(rc[i] * w - 1.0));
        esq = fmax(esq, fabs(sq[i] / sqrt(w) - 1.0));
    }
    printf("v_sin_f32 max abs err %.3e (rel err near 0: %.3e)\nv_cos_f32 max abs err %.3e\n", es, es_rel_small, ec);
    printf("v_rsq_f32 max rel err %.3e\nv_rcp_f32 max rel err %.3e\nv_sqrt_f32 max rel err %.3e\n", erq, erc, esq);
    return 0;
}
