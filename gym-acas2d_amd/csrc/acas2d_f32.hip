// float32 instantiation (throughput mode, FAST formulation).  Built with -ffp-contract=off like the
// float64 one: only the fma()s written in the source fuse, so every instantiation (work shape, rollout,
// policy variant) rounds alike -- measured free (7.13 us either way at 65 536 x 8).
#define ACAS2D_PACKED_SHAPES(X) X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(8, 1) X(4, 2) X(2, 4) X(4, 4) X(4, 8) X(4, 16) X(8, 8) X(2, 32)
namespace acas2d {
constexpr bool kFast = true;
}
#include "acas2d_launch.inl"
namespace acas2d {
template int launch_step<float>(const Acas2dConfig*, const Acas2dState*, const Acas2dState*, const Acas2dStepIO*, uint32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_rollout<float>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_rollout_policy<float>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, const Acas2dPolicy*, const void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_collect<float>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, const Acas2dActorCritic*, const void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_reset<float>(const Acas2dConfig*, const Acas2dState*, const uint8_t*, void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int shape_geometry<float>(int64_t, int32_t, int32_t*, int32_t*, int64_t*);
template int state_consecutive<float>(const Acas2dState*, int64_t, int32_t);
}

#ifdef ACAS2D_STAMPS
extern "C" int acas2d_debug_set_stamps_f32(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(acas2d::g_stamps), &buf, sizeof(buf));
}
#endif
