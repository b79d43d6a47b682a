// lcp_dev.hpp — the few definitions the contact kernels share: the row budget of the LCP, the row classes of
// ConstrainedGroupGradientMatrices::constructMatrices (CGGM.cpp:535-713) and the lane-interleaved view of HBM rows
// (element e of world b at base[e * B + b]).  The dense algebra of the LCP lives in coop_dev.hpp (one world per wavefront).
#pragma once
#include "spatial_dev.hpp"

namespace nbl {

constexpr int MAXR = 24;  // LCP rows per world handled by the device path (8 frictional contacts)

struct LaneMem {
  double* base;
  int64_t B, b;
  DEV double& at(int idx) const { return base[(int64_t)idx * B + b]; }
};

constexpr int RC_NOT_CLAMPING = 0, RC_CLAMPING = 1, RC_UPPER_BOUND = 2;

}  // namespace nbl
