// oracle/ref_pgs_prelude.hpp - TEST INFRASTRUCTURE.  What PgsBoxedLcpSolver::solve (dart/constraint/PgsBoxedLcpSolver.cpp:79-268) needs around
// it to compile on its own: the scalar type, the class shell of dart/constraint/PgsBoxedLcpSolver.hpp:44-97 (Option with its defaults, the
// order cache) and the vendored ODE helpers it calls (compiled from the reference's odelcpsolver sources into the same shared object).
// oracle/ref_build.py reads the function from the reference's file at build time and compiles it between this file and
// ref_pgs_epilogue.hpp; nothing of the reference is stored here.
#include <cassert>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "dart/external/odelcpsolver/matrix.h"
#include "dart/external/odelcpsolver/misc.h"

typedef double s_t;
using std::abs;
using std::isnan;

namespace dart {
namespace constraint {
class PgsBoxedLcpSolver {
public:
  struct Option {
    int mMaxIteration;
    s_t mDeltaXThreshold;
    s_t mRelativeDeltaXTolerance;
    s_t mEpsilonForDivision;
    bool mRandomizeConstraintOrder;
    Option(int maxIteration = 30, s_t deltaXTolerance = 1e-6, s_t relativeDeltaXTolerance = 1e-3, s_t epsilonForDivision = 1e-9,
           bool randomizeConstraintOrder = false)
      : mMaxIteration(maxIteration), mDeltaXThreshold(deltaXTolerance), mRelativeDeltaXTolerance(relativeDeltaXTolerance),
        mEpsilonForDivision(epsilonForDivision), mRandomizeConstraintOrder(randomizeConstraintOrder) {}
  };
  bool solve(int n, s_t* A, s_t* x, s_t* b, int nub, s_t* lo, s_t* hi, int* findex, bool earlyTermination);
  Option mOption;
  mutable std::vector<int> mCacheOrder;
};
