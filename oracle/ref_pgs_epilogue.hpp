}  // namespace constraint
}  // namespace dart

// oracle/ref_pgs_epilogue.hpp - TEST INFRASTRUCTURE.  C entry point around the reference's PgsBoxedLcpSolver::solve: dense n x n row-major
// A in, padded to the solver's row stride dPAD(n) (as BoxedLcpConstraintSolver.cpp:209-215 lays it out); A, x and b are updated in place
// like the reference does (the normalisation of A and b survives the call).
extern "C" int nbo_ref_pgs(int n, double* A, double* x, double* b, double* lo, double* hi, int* findex, int maxIteration, double deltaX,
                           double relTol, double epsDiv) {
  const int nskip = dPAD(n);
  std::vector<double> Ap((size_t)n * nskip, 0.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ap[(size_t)i * nskip + j] = A[(size_t)i * n + j];
  dart::constraint::PgsBoxedLcpSolver solver;
  solver.mOption = dart::constraint::PgsBoxedLcpSolver::Option(maxIteration, deltaX, relTol, epsDiv, false);
  const bool ok = solver.solve(n, Ap.data(), x, b, 0, lo, hi, findex, false);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) A[(size_t)i * n + j] = Ap[(size_t)i * nskip + j];
  return ok ? 1 : 0;
}
