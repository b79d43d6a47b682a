// oracle/ref_shim.cpp — C entry point around the REFERENCE's vendored Dantzig solver.
// Compiled together with /root/reference/dart/external/odelcpsolver/*.cpp (from where they lie, never
// copied) into oracle/_ref/libodelcp_ref.so by oracle/ref_build.py.  Mirrors
// DantzigBoxedLcpSolver::solve (dart/constraint/DantzigBoxedLcpSolver.cpp:54-107): nub = 0, w = nullptr,
// any exception -> failure.
bool dSolveLCP(int n, double* A, double* x, double* b, double* w, int nub, double* lo, double* hi, int* findex,
               bool earlyTermination);

extern "C" int nbo_ref_dantzig(int n, double* A, double* x, double* b, double* lo, double* hi, int* findex, int early) {
  try {
    return dSolveLCP(n, A, x, b, nullptr, 0, lo, hi, findex, early != 0) ? 1 : 0;
  } catch (...) {
    return 0;
  }
}
