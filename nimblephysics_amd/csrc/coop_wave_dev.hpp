// coop_wave_dev.hpp — the GPU wave primitives behind coop_dev.hpp's policy parameter: one 64-lane wavefront per
// workgroup (so a workgroup barrier is a wave barrier and LDS exchange needs nothing stronger).
#pragma once
#include <hip/hip_runtime.h>
#include "spatial_dev.hpp"

namespace nbl {

struct DevWave {
  DEV int lane() const { return (int)(threadIdx.x & 63u); }
  DEV void sync() const { __syncthreads(); }
  DEV double maxAll(double v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
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

// workgroup -> world: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a contiguous range of worlds
// (neighbouring worlds share the cache lines of the lane-interleaved rows of the saved record, which then meet in one L2)
DEV int64_t coopWorld(uint32_t bid, uint32_t nblk) {
  if ((nblk & 7u) == 0u) return (int64_t)(bid & 7u) * (nblk >> 3) + (bid >> 3);
  return (int64_t)bid;
}

}  // namespace nbl
