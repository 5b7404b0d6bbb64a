// float64 instantiation (parity mode, EXACT formulation).  Built with -ffp-contract=off: the only
// fused multiply-adds are the explicit ones that mirror the reference's OpenBLAS ddot
// (kinematics.py:11,77).
#define ACAS2D_PACKED_SHAPES(X) X(1, 1) X(3, 1) X(2, 1) X(4, 1) X(2, 4) X(4, 2) X(4, 4) X(4, 8) X(2, 32) X(4, 16)
namespace acas2d {
constexpr bool kFast = false;
}
#include "acas2d_launch.inl"
namespace acas2d {
template int launch_step<double>(const Acas2dConfig*, const Acas2dState*, const Acas2dState*, const Acas2dStepIO*, uint32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_rollout<double>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_rollout_policy<double>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, const Acas2dPolicy*, const void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_collect<double>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, const Acas2dActorCritic*, const void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_reset<double>(const Acas2dConfig*, const Acas2dState*, const uint8_t*, void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int shape_geometry<double>(int64_t, int32_t, int32_t*, int32_t*, int64_t*);
template int state_consecutive<double>(const Acas2dState*, int64_t, int32_t);
}
