// oracle/ref_geometry_prelude.hpp - TEST INFRASTRUCTURE.  What the reference's own spatial-algebra primitives (dart/math/Geometry.cpp:
// expMapRot, expMapJac, logMap, AdT, AdR, AdTAngular, AdTLinear, AdInvT, AdInvRLinear, ad, dAdT, dAdInvT, dAdInvR, dad, transformInertia,
// expMap, expMapDart, expAngular, makeSkewSymmetric, eulerXYZToMatrix, eulerZYXToMatrix) and its flat-array articulated-body
// algorithm (dart/dynamics/SimpleFeatherstone.{hpp,cpp}: forwardDynamics) need in order to compile WITHOUT Eigen and without the rest of
// DART: a small fixed-size matrix class with the handful of Eigen operations those functions use, under Eigen's names.
// oracle/ref_build.py concatenates this file, the function bodies read from /root/reference at build time (nothing of them is stored in
// this repo) and ref_geometry_epilogue.hpp into oracle/_ref/, and compiles libgeometry_ref.so from it.
//
// Arithmetic: every expression is evaluated eagerly, coefficient by coefficient, in the order it is written; inner products sum their
// terms in index order.  Eigen evaluates the same coefficient-wise expressions lazily with the same operations per coefficient, so
// element-wise code (expMap, logMap, transformInertia, the Taylor branches) is reproduced exactly; for 3- and 6-term inner products
// Eigen is free to associate the sum differently (unrolled reductions, packets), which is why the tests that go through matrix products
// compare to a few ulps instead of bit for bit.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <iostream>
#include <memory>
#include <vector>

typedef double s_t;
using std::abs;
using std::acos;
using std::cos;
using std::sin;
using std::sqrt;

namespace Eigen {

enum { StrictlyLower = 1 };

template <int R, int C>
struct Mat;
typedef Mat<3, 1> Vector3s;
typedef Mat<6, 1> Vector6s;
typedef Mat<3, 3> Matrix3s;
typedef Mat<6, 6> Matrix6s;
typedef Mat<6, 6> MatrixXs;   // the one dynamic matrix of the compiled range (SimpleFeatherstone: `MatrixXs PI = articulatedInertia`) is 6 x 6
typedef Mat<6, 1> VectorXs;   // ... and the one dynamic vector (FreeJoint::integratePositionsExplicit: positions / velocities of a free joint) has 6 entries

// a writable 3-segment of a 6-vector / of an isometry's translation column
struct Seg3 {
  double* p;
  int stride;
  double& at(int i) const { return p[i * stride]; }
  Seg3& noalias() { return *this; }
  inline Seg3& operator=(const Vector3s& v);
  inline Seg3& operator=(const Seg3& v);
  inline Seg3& operator+=(const Vector3s& v);
  inline operator Vector3s() const;
  inline Vector3s cross(const Vector3s& o) const;
};

template <int R, int C>
struct Mat {
  double m[R][C];
  Mat() {
    for (int i = 0; i < R; i++)
      for (int j = 0; j < C; j++) m[i][j] = 0;
  }
  Mat(double x, double y, double z) {
    static_assert(R == 3 && C == 1, "3-vector constructor");
    m[0][0] = x; m[1][0] = y; m[2][0] = z;
  }
  static Mat Zero() { return Mat(); }
  static Mat Identity() {
    Mat r;
    for (int i = 0; i < (R < C ? R : C); i++) r.m[i][i] = 1;
    return r;
  }
  static Mat Unit(int i) { Mat r; r.m[i][0] = 1; return r; }
  static Mat UnitX() { return Unit(0); }
  static Mat UnitY() { return Unit(1); }
  static Mat UnitZ() { return Unit(2); }
  void normalize() { const double n = norm(); for (int i = 0; i < R; i++) m[i][0] /= n; }     // Eigen divides by the norm (no reciprocal)
  Mat normalized() const { Mat r = *this; r.normalize(); return r; }
  struct ColRef {
    Mat* t; int j;
    void operator=(const Mat<R, 1>& v) { for (int i = 0; i < R; i++) t->m[i][j] = v.m[i][0]; }
  };
  ColRef col(int j) { return ColRef{this, j}; }
  double& operator()(int i, int j) { return m[i][j]; }
  double operator()(int i, int j) const { return m[i][j]; }
  double& operator()(int i) { return m[i][0]; }
  double operator()(int i) const { return m[i][0]; }
  double& operator[](int i) { return m[i][0]; }
  double operator[](int i) const { return m[i][0]; }
  Mat& noalias() { return *this; }
  void setZero() { *this = Mat(); }
  Mat operator+(const Mat& o) const { Mat r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.m[i][j] = m[i][j] + o.m[i][j]; return r; }
  Mat operator-(const Mat& o) const { Mat r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.m[i][j] = m[i][j] - o.m[i][j]; return r; }
  Mat operator-() const { Mat r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.m[i][j] = -m[i][j]; return r; }
  Mat operator*(double s) const { Mat r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.m[i][j] = m[i][j] * s; return r; }
  Mat operator/(double s) const { Mat r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.m[i][j] = m[i][j] / s; return r; }
  Mat& operator+=(const Mat& o) { *this = *this + o; return *this; }
  Mat& operator-=(const Mat& o) { *this = *this - o; return *this; }
  Mat& operator*=(double s) { *this = *this * s; return *this; }
  Mat& operator/=(double s) { *this = *this / s; return *this; }
  bool operator!=(const Mat& o) const { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) if (m[i][j] != o.m[i][j]) return true; return false; }
  template <int K>
  Mat<R, K> operator*(const Mat<C, K>& o) const {
    Mat<R, K> r;
    for (int i = 0; i < R; i++)
      for (int j = 0; j < K; j++) {
        double s = m[i][0] * o.m[0][j];
        for (int k = 1; k < C; k++) s += m[i][k] * o.m[k][j];
        r.m[i][j] = s;
      }
    return r;
  }
  Mat<C, R> transpose() const { Mat<C, R> r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.m[j][i] = m[i][j]; return r; }
  double value() const { static_assert(R == 1 && C == 1, "value() of a 1 x 1"); return m[0][0]; }
  double dot(const Mat& o) const {
    static_assert(C == 1, "dot of vectors");
    double s = m[0][0] * o.m[0][0];
    for (int i = 1; i < R; i++) s += m[i][0] * o.m[i][0];
    return s;
  }
  double squaredNorm() const { return dot(*this); }
  double norm() const { return std::sqrt(squaredNorm()); }
  Vector3s cross(const Vector3s& o) const {
    static_assert(R == 3 && C == 1, "cross of 3-vectors");
    return Vector3s(m[1][0] * o.m[2][0] - m[2][0] * o.m[1][0], m[2][0] * o.m[0][0] - m[0][0] * o.m[2][0], m[0][0] * o.m[1][0] - m[1][0] * o.m[0][0]);
  }
  // 3-segments of a 6-vector: writable on a non-const vector, a copy on a const one
  template <int N> Seg3 head() { static_assert(N == 3 && C == 1 && R >= 3, "head<3>"); return Seg3{&m[0][0], 1}; }
  template <int N> Seg3 tail() { static_assert(N == 3 && C == 1 && R >= 3, "tail<3>"); return Seg3{&m[R - 3][0], 1}; }
  template <int N> Vector3s head() const { return Vector3s(m[0][0], m[1][0], m[2][0]); }
  template <int N> Vector3s tail() const { return Vector3s(m[R - 3][0], m[R - 2][0], m[R - 1][0]); }
  // `ret.triangularView<Eigen::StrictlyLower>() = ret.transpose();`
  struct LowerView {
    Mat* t;
    void operator=(const Mat& o) { for (int i = 0; i < R; i++) for (int j = 0; j < i; j++) t->m[i][j] = o.m[i][j]; }
  };
  template <int Mode> LowerView triangularView() { static_assert(Mode == StrictlyLower, "strictly lower only"); return LowerView{this}; }
};
template <int R, int C>
inline Mat<R, C> operator*(double s, const Mat<R, C>& a) { Mat<R, C> r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r.m[i][j] = s * a.m[i][j]; return r; }

inline Seg3& Seg3::operator=(const Vector3s& v) { for (int i = 0; i < 3; i++) at(i) = v[i]; return *this; }
inline Seg3& Seg3::operator=(const Seg3& v) { const Vector3s t = v; return *this = t; }
inline Seg3& Seg3::operator+=(const Vector3s& v) { for (int i = 0; i < 3; i++) at(i) = at(i) + v[i]; return *this; }
inline Seg3::operator Vector3s() const { return Vector3s(at(0), at(1), at(2)); }
inline Vector3s Seg3::cross(const Vector3s& o) const { return Vector3s(*this).cross(o); }

// Eigen::Transform<s_t, 3, Isometry>
struct Isometry3s {
  double m[4][4];
  Isometry3s() : m{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}} {}
  static Isometry3s Identity() { return Isometry3s(); }
  double operator()(int r, int c) const { return m[r][c]; }
  double& operator()(int r, int c) { return m[r][c]; }
  Matrix3s linear() const { Matrix3s r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j]; return r; }
  struct Lin3 {      // `tf.linear() = R;`
    Isometry3s* t;
    void operator=(const Matrix3s& R) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t->m[i][j] = R.m[i][j]; }
    operator Matrix3s() const { return static_cast<const Isometry3s*>(t)->linear(); }
  };
  Lin3 linear() { return Lin3{this}; }
  Isometry3s(const Isometry3s&) = default;
  Isometry3s& operator=(const Isometry3s&) = default;
  Vector3s translation() const { return Vector3s(m[0][3], m[1][3], m[2][3]); }
  Seg3 translation() { return Seg3{&m[0][3], 4}; }
  // product of two isometries: the affine 3 x 4 parts, R = R1 R2, p = R1 p2 + p1
  Isometry3s operator*(const Isometry3s& o) const {
    Isometry3s r;
    const Matrix3s R = linear() * o.linear();
    const Vector3s p = linear() * o.translation() + translation();
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) r.m[i][j] = R.m[i][j]; r.m[i][3] = p[i]; }
    return r;
  }
  Vector3s operator*(const Vector3s& x) const { return linear() * x + translation(); }
  Isometry3s inverse() const {     // Transform<Isometry>::inverse(): R^T, -R^T p
    Isometry3s r;
    const Matrix3s Rt = linear().transpose();
    const Vector3s t = Rt * translation();
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) r.m[i][j] = Rt.m[i][j]; r.m[i][3] = -t[i]; }
    return r;
  }
};
}  // namespace Eigen

#define DART_EPSILON (1.0E-6)    // dart/math/MathTypes.hpp

namespace dart {
namespace math {
typedef Eigen::Matrix6s Inertia;   // dart/math/MathTypes.hpp
namespace constantsd {
inline double pi() { return 3.141592653589793238462643383279502884; }
}
using std::max;
using std::min;
// forward declarations (dart/math/Geometry.hpp order differs from the order of definition)
void dLineClosestApproach(const Eigen::Vector3s& pa, const Eigen::Vector3s& ua, const Eigen::Vector3s& pb, const Eigen::Vector3s& ub, s_t* alpha, s_t* beta);
Eigen::Matrix3s makeSkewSymmetric(const Eigen::Vector3s& _v);
Eigen::Vector3s logMap(const Eigen::Matrix3s& _R);
Eigen::Matrix3s expMapRot(const Eigen::Vector3s& _q);
Eigen::Isometry3s expAngular(const Eigen::Vector3s& _s);
Eigen::Isometry3s expMap(const Eigen::Vector6s& _S);
Eigen::Vector6s AdInvT(const Eigen::Isometry3s& _T, const Eigen::Vector6s& _V);
Eigen::Vector6s ad(const Eigen::Vector6s& _X, const Eigen::Vector6s& _Y);
Eigen::Vector6s dad(const Eigen::Vector6s& _s, const Eigen::Vector6s& _t);
Eigen::Vector6s dAdInvT(const Eigen::Isometry3s& _T, const Eigen::Vector6s& _F);
Inertia transformInertia(const Eigen::Isometry3s& _T, const Inertia& _I);
}  // namespace math
}  // namespace dart
