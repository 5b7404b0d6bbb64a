// float32 instantiation (throughput mode).  Built with the default -ffp-contract=fast.
#include "acas2d_launch.inl"
namespace acas2d {
template int launch_step<float>(const Acas2dConfig*, const Acas2dState*, const Acas2dStepIO*, uint32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
template int launch_reset<float>(const Acas2dConfig*, const Acas2dState*, const uint8_t*, void*, int32_t, uint64_t, int64_t, int64_t, int32_t, hipStream_t);
}
