// float64 instantiation (parity mode).  Built with -ffp-contract=off: the only fused
// multiply-adds are the explicit ones that mirror the reference's OpenBLAS ddot
// (kinematics.py:11,77).
#include "acas2d_launch.inl"
namespace acas2d {
template int launch_step<double>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, uint32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_reset<double>(const Acas2dConfig*, const Acas2dState*, const uint8_t*, void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
}
