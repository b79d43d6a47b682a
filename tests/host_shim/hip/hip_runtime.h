// tests/host_shim/hip/hip_runtime.h — TEST-ONLY stand-in so that the device headers of the product
// (nimblephysics_amd/csrc/*_dev.hpp) can be compiled with g++ and unit-tested on the host against the
// reference's own Dantzig solver.  Never on the product's include path.
#pragma once
#include <math.h>
#include <stdint.h>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
