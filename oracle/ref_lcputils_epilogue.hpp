// oracle/ref_lcputils_epilogue.hpp - TEST INFRASTRUCTURE: C entry points around the reference's LCPUtils functions compiled by
// oracle/ref_build.py::build_lcputils (row-major n x n matrices in and out).
namespace {
using namespace dart::constraint;
void loadProblem(int n, const double* A, const double* x, const double* b, const double* hi, const double* lo, const int* findex,
                 Eigen::MatrixXs& mA, Eigen::VectorXs& mX, Eigen::VectorXs& mB, Eigen::VectorXs& mHi, Eigen::VectorXs& mLo, Eigen::VectorXi& mF) {
  mA = Eigen::MatrixXs::Zero(n, n); mX = Eigen::VectorXs::Zero(n); mB = Eigen::VectorXs::Zero(n); mHi = Eigen::VectorXs::Zero(n);
  mLo = Eigen::VectorXs::Zero(n); mF = Eigen::VectorXi::Zero(n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) mA(i, j) = A[i * n + j];
    mX(i) = x[i]; mB(i) = b[i]; mHi(i) = hi[i]; mLo(i) = lo[i]; mF(i) = findex[i];
  }
}
// the reduced problem and mapOut (nOrig x nr, row-major) back into caller arrays of the ORIGINAL size; returns the reduced size
int storeProblem(int nOrig, const Eigen::MatrixXs& mA, const Eigen::VectorXs& mX, const Eigen::VectorXs& mB, const Eigen::VectorXs& mHi,
                 const Eigen::VectorXs& mLo, const Eigen::VectorXi& mF, const Eigen::MatrixXs& mapOut, double* A, double* x, double* b, double* hi,
                 double* lo, int* findex, double* map) {
  const int nr = mA.cols();
  for (int i = 0; i < nr; i++) {
    for (int j = 0; j < nr; j++) A[i * nr + j] = mA(i, j);
    x[i] = mX(i); b[i] = mB(i); hi[i] = mHi(i); lo[i] = mLo(i); findex[i] = mF(i);
  }
  for (int i = 0; i < nOrig; i++) for (int j = 0; j < nr; j++) map[i * nr + j] = mapOut(i, j);
  return nr;
}
}  // namespace

extern "C" {
int ref_lcp_valid(int n, const double* A, const double* x, const double* b, const double* hi, const double* lo, const int* findex, int ignoreFriction) {
  Eigen::MatrixXs mA; Eigen::VectorXs mX, mB, mHi, mLo; Eigen::VectorXi mF;
  loadProblem(n, A, x, b, hi, lo, findex, mA, mX, mB, mHi, mLo, mF);
  return LCPUtils::isLCPSolutionValid(mA, mX, mB, mHi, mLo, mF, ignoreFriction != 0) ? 1 : 0;
}
// in place on arrays of the original size n (the first nr x nr / nr entries are the reduced problem on return); map: n x nr row-major
int ref_lcp_reduce(int n, double* A, double* x, double* b, double* hi, double* lo, int* findex, double* map) {
  Eigen::MatrixXs mA; Eigen::VectorXs mX, mB, mHi, mLo; Eigen::VectorXi mF;
  loadProblem(n, A, x, b, hi, lo, findex, mA, mX, mB, mHi, mLo, mF);
  const Eigen::MatrixXs mapOut = LCPUtils::reduce(mA, mX, mB, mHi, mLo, mF);
  return storeProblem(n, mA, mX, mB, mHi, mLo, mF, mapOut, A, x, b, hi, lo, findex, map);
}
int ref_lcp_remove_friction(int n, double* A, double* x, double* b, double* hi, double* lo, int* findex, double* map) {
  Eigen::MatrixXs mA; Eigen::VectorXs mX, mB, mHi, mLo; Eigen::VectorXi mF;
  loadProblem(n, A, x, b, hi, lo, findex, mA, mX, mB, mHi, mLo, mF);
  const Eigen::MatrixXs mapOut = LCPUtils::removeFriction(mA, mX, mB, mHi, mLo, mF);
  return storeProblem(n, mA, mX, mB, mHi, mLo, mF, mapOut, A, x, b, hi, lo, findex, map);
}
}
