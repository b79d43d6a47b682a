// lcp_dev.hpp — the few definitions the contact kernels share: the row budget of the LCP, the row classes of
// ConstrainedGroupGradientMatrices::constructMatrices (CGGM.cpp:535-713) and the lane-interleaved view of HBM rows
// (element e of world b at base[e * B + b]).  The dense algebra of the LCP lives in coop_dev.hpp (one world per wavefront).
#pragma once
#include "spatial_dev.hpp"

namespace NBL_NS {

#ifndef NBL_MAXC
#define NBL_MAXC 8
#endif
constexpr int MAXR = 3 * NBL_MAXC;  // LCP rows per world handled by the device path (8 frictional contacts: 24; the 16-contact build: 48)

// one bit per LCP row of a world (ballots restricted to the row lanes)
#if NBL_MAXC > 8
typedef uint64_t RowMask;
#else
typedef uint32_t RowMask;
#endif
DEV int rmPop(RowMask m) { return __builtin_popcountll((unsigned long long)m); }
DEV int rmCtz(RowMask m) { return __builtin_ctzll((unsigned long long)m); }
constexpr RowMask RM1 = 1;

struct LaneMem {
  double* base;
  int64_t B, b;
  DEV double& at(int idx) const { return base[(int64_t)idx * B + b]; }
};

constexpr int RC_NOT_CLAMPING = 0, RC_CLAMPING = 1, RC_UPPER_BOUND = 2;

}  // namespace NBL_NS
