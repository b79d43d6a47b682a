// oracle/ref_lcputils_prelude.hpp - TEST INFRASTRUCTURE.  What LCPUtils::isLCPSolutionValid / reduce / removeFriction / mergeLCPColumns /
// dropLCPColumn (dart/constraint/LCPUtils.cpp:12-80, 144-247, 346-549) need around them to compile on their own: the scalar type, a small
// DYNAMIC matrix / vector class under Eigen's names (own code - Eigen itself is not on this machine; these functions use no decomposition:
// element access, column / row views, a matrix-vector product, squaredNorm, Zero / Identity) and the class shell of
// dart/constraint/LCPUtils.hpp.  oracle/ref_build.py reads the functions from the reference's file at build time and compiles them between
// this file and ref_lcputils_epilogue.hpp (C entry points); nothing of the reference is stored here.
// Products and norms are coefficient-based sums in index order (what Eigen's column-major kernels do per coefficient without FMA); the
// functions only COMPARE such sums with thresholds (1e-4 column distance, 1e-5 validity), every number they hand back is a copy of an
// input or an input times 2, so the results do not depend on the summation order except exactly at a threshold.
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <math.h>
#include <vector>

typedef double s_t;
using std::abs;

namespace Eigen {
template <class T>
struct DynVec;
template <class T>
struct DynMat;

// the difference of two columns, as far as the functions use it: squaredNorm()
template <class T>
struct ColDiff {
  std::vector<T> d;
  T squaredNorm() const { T s = 0; for (const T& x : d) s += x * x; return s; }
};
// a column / row of a matrix: assignment, += a column, *= a scalar, difference of two columns
template <class T>
struct ColView {
  DynMat<T>* m; int j;
  ColView& operator=(const ColView& o) { for (int i = 0; i < m->rows(); i++) (*m)(i, j) = (*o.m)(i, o.j); return *this; }
  ColView& operator+=(const ColView& o) { for (int i = 0; i < m->rows(); i++) (*m)(i, j) += (*o.m)(i, o.j); return *this; }
  ColView& operator*=(T s) { for (int i = 0; i < m->rows(); i++) (*m)(i, j) *= s; return *this; }
  ColDiff<T> operator-(const ColView& o) const { ColDiff<T> r; r.d.resize(m->rows()); for (int i = 0; i < m->rows(); i++) r.d[i] = (*m)(i, j) - (*o.m)(i, o.j); return r; }
};
template <class T>
struct RowView {
  DynMat<T>* m; int i;
  RowView& operator=(const RowView& o) { for (int j = 0; j < m->cols(); j++) (*m)(i, j) = (*o.m)(o.i, j); return *this; }
};

template <class T>
struct DynVec {
  std::vector<T> d;
  DynVec() {}
  explicit DynVec(int n) : d(n) {}
  static DynVec Zero(int n) { DynVec v; v.d.assign(n, T(0)); return v; }
  int size() const { return (int)d.size(); }
  T& operator()(int i) { return d[i]; }
  const T& operator()(int i) const { return d[i]; }
  DynVec operator-(const DynVec& o) const { DynVec r(size()); for (int i = 0; i < size(); i++) r.d[i] = d[i] - o.d[i]; return r; }
};

template <class T>
struct DynMat {   // column-major like Eigen's default
  int r = 0, c = 0;
  std::vector<T> d;
  DynMat() {}
  DynMat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_) {}
  static DynMat Zero(int r_, int c_) { DynMat m(r_, c_); std::fill(m.d.begin(), m.d.end(), T(0)); return m; }
  static DynMat Identity(int r_, int c_) { DynMat m = Zero(r_, c_); for (int i = 0; i < r_ && i < c_; i++) m(i, i) = T(1); return m; }
  int rows() const { return r; }
  int cols() const { return c; }
  T& operator()(int i, int j) { return d[(size_t)j * r + i]; }
  const T& operator()(int i, int j) const { return d[(size_t)j * r + i]; }
  ColView<T> col(int j) { return ColView<T>{this, j}; }
  ColView<T> col(int j) const { return ColView<T>{const_cast<DynMat*>(this), j}; }
  RowView<T> row(int i) { return RowView<T>{this, i}; }
  RowView<T> row(int i) const { return RowView<T>{const_cast<DynMat*>(this), i}; }
  // y_i = sum_j A_ij x_j, j ascending from an accumulator at zero (Eigen's column-major GEMV adds the columns in that order to a zeroed result)
  DynVec<T> operator*(const DynVec<T>& x) const {
    DynVec<T> y = DynVec<T>::Zero(r);
    for (int j = 0; j < c; j++) { const T xj = x(j); for (int i = 0; i < r; i++) y(i) += (*this)(i, j) * xj; }
    return y;
  }
};
typedef DynMat<s_t> MatrixXs;
typedef DynVec<s_t> VectorXs;
typedef DynVec<int> VectorXi;
}  // namespace Eigen

namespace dart {
namespace constraint {
class LCPUtils {   // dart/constraint/LCPUtils.hpp: the static functions compiled here
public:
  static bool isLCPSolutionValid(const Eigen::MatrixXs& mA, const Eigen::VectorXs& mX, const Eigen::VectorXs& mB, const Eigen::VectorXs& mHi,
                                 const Eigen::VectorXs& mLo, const Eigen::VectorXi& mFIndex, bool ignoreFrictionIndices);
  static Eigen::MatrixXs reduce(Eigen::MatrixXs& A, Eigen::VectorXs& X, Eigen::VectorXs& b, Eigen::VectorXs& hi, Eigen::VectorXs& lo, Eigen::VectorXi& fIndex);
  static Eigen::MatrixXs removeFriction(Eigen::MatrixXs& A, Eigen::VectorXs& X, Eigen::VectorXs& b, Eigen::VectorXs& hi, Eigen::VectorXs& lo, Eigen::VectorXi& fIndex);
  static void mergeLCPColumns(int colA, int colB, Eigen::MatrixXs& A, Eigen::VectorXs& X, Eigen::VectorXs& b, Eigen::VectorXs& hi, Eigen::VectorXs& lo,
                              Eigen::VectorXi& fIndex, Eigen::MatrixXs& mapOut);
  static void dropLCPColumn(int col, Eigen::MatrixXs& A, Eigen::VectorXs& X, Eigen::VectorXs& b, Eigen::VectorXs& hi, Eigen::VectorXs& lo,
                            Eigen::VectorXi& fIndex, Eigen::MatrixXs& mapOut);
};
}  // namespace constraint
}  // namespace dart
