// coop_wave_dev.hpp — the GPU wave primitives behind coop_dev.hpp's policy parameter: one 64-lane wavefront per
// workgroup (so a workgroup barrier is a wave barrier and LDS exchange needs nothing stronger).
#pragma once
#include <hip/hip_runtime.h>
#include "spatial_dev.hpp"

namespace NBL_NS {



struct DevWave {
  DEV int lane() const { return (int)(threadIdx.x & 63u); }
  // Every DevWave kernel runs ONE wavefront per workgroup, and a wave's LDS instructions execute in order: a write is visible to the
  // wave's later reads without waiting.  The "barrier" therefore only has to stop the COMPILER from moving LDS accesses across it;
  // __syncthreads() would also drain the LDS queue (s_waitcnt lgkmcnt(0)) and stall the loads already in flight behind it.
  DEV void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // max over the wavefront.  DPP row shifts (register crossbar, a few cycles each) instead of six ds_bpermute round trips:
  // after row_shr 1, 2, 4, 8 the last lane of every 16-lane row holds the row maximum; four readlanes finish.
  DEV double maxAll(double v) const {
#define NBL_DPP_MAX_STEP(CTRL)                                                                             \
    {                                                                                                      \
      const int lo = __double2loint(v), hi = __double2hiint(v);                                            \
      const int tlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);                          \
      const int thi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);                          \
      v = fmax(v, __hiloint2double(thi, tlo));                                                             \
    }
    NBL_DPP_MAX_STEP(0x111) NBL_DPP_MAX_STEP(0x112) NBL_DPP_MAX_STEP(0x114) NBL_DPP_MAX_STEP(0x118)
#undef NBL_DPP_MAX_STEP
    return fmax(fmax(bcast(v, 15), bcast(v, 31)), fmax(bcast(v, 47), bcast(v, 63)));
  }
  // min over the lanes that can hold a row (lanes >= ROWS must hold +inf or a NaN; NaNs lose, like in maxAll's fmax).  Builds of at most 32
  // rows: two DPP rows instead of four, and v_min_f64 itself - fmin() makes the compiler quiet both operands first (a v_max_f64 v, v, v
  // each: the reduction was 40 instructions, it is 19).
  template <int ROWS>
  DEV double minRows(double v) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (ROWS <= 32) {
#define NBL_DPP_MIN_STEP(CTRL)                                                                             \
      {                                                                                                    \
        const int lo = __double2loint(v), hi = __double2hiint(v);                                          \
        const int tlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);                        \
        const int thi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);                        \
        const double t = __hiloint2double(thi, tlo);                                                       \
        asm("v_min_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(t));                                            \
      }
      NBL_DPP_MIN_STEP(0x111) NBL_DPP_MIN_STEP(0x112) NBL_DPP_MIN_STEP(0x114) NBL_DPP_MIN_STEP(0x118)
#undef NBL_DPP_MIN_STEP
      const double a = bcast(v, 15), b = bcast(v, 31);
      double r;
      asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      return r;
    }
#endif
    return -maxAll(-v);
  }
  DEV uint64_t ballot(bool p) const { return (uint64_t)__ballot(p ? 1 : 0); }
  DEV double shfl(double v, int src) const { return __shfl(v, src & 63); }
  DEV int shflI(int v, int src) const { return __shfl(v, src & 63); }
  // value of lane `src`, src WAVE-UNIFORM: v_readlane (a few cycles) instead of ds_bpermute (an LDS round trip)
  DEV int bcastI(int v, int src) const { return __builtin_amdgcn_readlane(v, src); }
  DEV double bcast(double v, int src) const {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
  }
};

// The same primitives for kernels whose workgroups hold SEVERAL independent wavefronts (k_contact_cascade_stages: one stage per
// wavefront): "sync" orders the wave's own LDS traffic (a wave's LDS instructions execute in order; the fence only stops the
// compiler from moving accesses across it) and never waits for the other wavefronts of the group.
struct DevWaveInGroup : DevWave {};

// workgroup -> world: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a contiguous range of worlds
// (neighbouring worlds share the cache lines of the lane-interleaved rows of the saved record, which then meet in one L2)
DEV int64_t coopWorld(uint32_t bid, uint32_t nblk) {
  if ((nblk & 7u) == 0u) return (int64_t)(bid & 7u) * (nblk >> 3) + (bid >> 3);
  return (int64_t)bid;
}

}  // namespace NBL_NS
