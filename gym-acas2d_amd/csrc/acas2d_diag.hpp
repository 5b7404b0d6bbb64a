// acas2d_diag.hpp -- diagnostic hooks of the step kernels, kept out of the product source.
//
// In-kernel stamps: DIAGNOSTIC build only (tools/diag_stamps.py builds libacas2d_hip_diag.so with
// -DACAS2D_STAMPS).  In the product build ACAS2D_STAMP() is empty and no stamp executes.  Stamp values
// go to a buffer of their own that no other code reads; no output is computed from them.
#pragma once

#ifdef ACAS2D_STAMPS
// (included inside namespace acas2d)
static __device__ unsigned long long* g_stamps = nullptr;   // [n_waves][16], set by acas2d_debug_set_stamps_*
#define ACAS2D_STAMP(k, wave_id, lane_id, drain)                                              \
    do {                                                                                      \
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
        if ((lane_id) == 0 && acas2d::g_stamps)                                               \
            acas2d::g_stamps[(wave_id) * 16 + (k)] = ((k) == 0 || (k) == 7) ? __builtin_amdgcn_s_memrealtime() \
                                                                            : __builtin_amdgcn_s_memtime();    \
    } while (0)
#else
#define ACAS2D_STAMP(k, wave_id, lane_id, drain) do { } while (0)
#endif
