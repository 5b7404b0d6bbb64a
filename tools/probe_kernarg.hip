// probe_kernarg.hip -- what do large by-value kernel arguments cost per launch?  hipGraph of 200
// dependent launches (512 blocks x 256 threads) of: K0 no args; K1 one pointer; K2 a 464-byte
// by-value block of which only the LAST word is read; K3 the same block, every word read
// (serialised s_load chains as hipcc emits them); K4 every word read through ONE vector load of
// the block from device memory instead of the kernarg segment.
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { unsigned w[116]; };
__global__ void k0() {}
__global__ void k1(unsigned* out) { if (out && threadIdx.x == 1000) out[0] = 1; }
__global__ void k2(Big b, unsigned* out) { if (b.w[115] == 0xdeadbeefu) out[0] = 1; }
__global__ void k3(Big b, unsigned* out) {
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 116; ++i) x ^= b.w[i] * (i + 1);
    if (x == 0xdeadbeefu) out[0] = x;
}
__global__ void k4(const Big* __restrict__ b, unsigned* out) {
    const uint4* p = reinterpret_cast<const uint4*>(b);
    unsigned lane = threadIdx.x & 63;
    uint4 v = lane < 29 ? p[lane] : uint4{0, 0, 0, 0};      // one wave-instruction fetches the block
    unsigned x = v.x ^ v.y * 3 ^ v.z * 5 ^ v.w * 7;
    for (int m = 1; m < 64; m <<= 1) x ^= __shfl_xor(x, m, 64);
    if (x == 0xdeadbeefu) out[0] = x;
}
template <typename F> static float run(const char* name, F launch) {
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) launch(s);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-52s %.2f us/launch\n", name, ms * 1e3f / (200 * 20));
    return ms;
}
int main() {
    unsigned* out; hipMalloc(&out, 64); Big hb{}; for (int i = 0; i < 116; ++i) hb.w[i] = i * 2654435761u;
    Big* db; hipMalloc(&db, sizeof(Big)); hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice);
    dim3 g(512), b(256);
    run("K0 no arguments", [&](hipStream_t s) { hipLaunchKernelGGL(k0, g, b, 0, s); });
    run("K1 one pointer", [&](hipStream_t s) { hipLaunchKernelGGL(k1, g, b, 0, s, out); });
    run("K2 464-byte by-value block, last word read", [&](hipStream_t s) { hipLaunchKernelGGL(k2, g, b, 0, s, hb, out); });
    run("K3 464-byte by-value block, every word read", [&](hipStream_t s) { hipLaunchKernelGGL(k3, g, b, 0, s, hb, out); });
    run("K4 block in device memory, one vector load", [&](hipStream_t s) { hipLaunchKernelGGL(k4, g, b, 0, s, db, out); });
    return 0;
}
