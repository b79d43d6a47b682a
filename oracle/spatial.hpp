// oracle/spatial.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Scalar fp64 restatement of the spatial-algebra primitives of the reference
// (dart/math/Geometry.cpp).  6-vectors are [omega(0:3); v(3:6)] exactly as in
// the reference (Geometry.cpp:1300-1311).  Nothing in the shipped product may
// include this file.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace nbo {

typedef double s_t;

struct Vec3 {
  s_t v[3];
  s_t& operator[](int i) { return v[i]; }
  const s_t& operator[](int i) const { return v[i]; }
};
inline Vec3 mk3(s_t a, s_t b, s_t c) { Vec3 r; r.v[0] = a; r.v[1] = b; r.v[2] = c; return r; }
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return mk3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return mk3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3 operator-(const Vec3& a) { return mk3(-a[0], -a[1], -a[2]); }
inline Vec3 operator*(s_t s, const Vec3& a) { return mk3(s * a[0], s * a[1], s * a[2]); }
inline Vec3 operator*(const Vec3& a, s_t s) { return mk3(s * a[0], s * a[1], s * a[2]); }
inline s_t dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return mk3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline s_t norm(const Vec3& a) { return std::sqrt(dot(a, a)); }

// 3x3, row-major
struct Mat3 {
  s_t m[9];
  s_t& operator()(int r, int c) { return m[3 * r + c]; }
  const s_t& operator()(int r, int c) const { return m[3 * r + c]; }
};
inline Mat3 zero3() { Mat3 r; std::memset(r.m, 0, sizeof(r.m)); return r; }
inline Mat3 eye3() { Mat3 r = zero3(); r(0, 0) = r(1, 1) = r(2, 2) = 1; return r; }
inline Mat3 operator*(const Mat3& a, const Mat3& b) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
  return r;
}
inline Vec3 operator*(const Mat3& a, const Vec3& x) {
  return mk3(a(0, 0) * x[0] + a(0, 1) * x[1] + a(0, 2) * x[2], a(1, 0) * x[0] + a(1, 1) * x[1] + a(1, 2) * x[2],
             a(2, 0) * x[0] + a(2, 1) * x[1] + a(2, 2) * x[2]);
}
inline Mat3 operator+(const Mat3& a, const Mat3& b) { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
inline Mat3 operator-(const Mat3& a, const Mat3& b) { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] - b.m[i]; return r; }
inline Mat3 operator*(s_t s, const Mat3& a) { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = s * a.m[i]; return r; }
inline Mat3 transpose(const Mat3& a) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(j, i);
  return r;
}
inline Vec3 tmul(const Mat3& a, const Vec3& x) {  // a^T x
  return mk3(a(0, 0) * x[0] + a(1, 0) * x[1] + a(2, 0) * x[2], a(0, 1) * x[0] + a(1, 1) * x[1] + a(2, 1) * x[2],
             a(0, 2) * x[0] + a(1, 2) * x[1] + a(2, 2) * x[2]);
}
// math::makeSkewSymmetric
inline Mat3 skew(const Vec3& a) {
  Mat3 r = zero3();
  r(0, 1) = -a[2]; r(0, 2) = a[1];
  r(1, 0) = a[2];  r(1, 2) = -a[0];
  r(2, 0) = -a[1]; r(2, 1) = a[0];
  return r;
}

// Eigen::Isometry3s
struct Iso {
  Mat3 R;
  Vec3 p;
};
inline Iso isoIdentity() { Iso t; t.R = eye3(); t.p = mk3(0, 0, 0); return t; }
inline Iso operator*(const Iso& a, const Iso& b) { Iso r; r.R = a.R * b.R; r.p = a.R * b.p + a.p; return r; }
inline Iso inverse(const Iso& a) { Iso r; r.R = transpose(a.R); r.p = -(tmul(a.R, a.p)); return r; }
inline Vec3 apply(const Iso& a, const Vec3& x) { return a.R * x + a.p; }

struct Vec6 {
  s_t v[6];
  s_t& operator[](int i) { return v[i]; }
  const s_t& operator[](int i) const { return v[i]; }
};
inline Vec6 zero6() { Vec6 r; for (int i = 0; i < 6; i++) r.v[i] = 0; return r; }
inline Vec6 mk6(const Vec3& w, const Vec3& v) { Vec6 r; for (int i = 0; i < 3; i++) { r.v[i] = w[i]; r.v[3 + i] = v[i]; } return r; }
inline Vec3 head(const Vec6& a) { return mk3(a[0], a[1], a[2]); }
inline Vec3 tail(const Vec6& a) { return mk3(a[3], a[4], a[5]); }
inline Vec6 operator+(const Vec6& a, const Vec6& b) { Vec6 r; for (int i = 0; i < 6; i++) r.v[i] = a[i] + b[i]; return r; }
inline Vec6 operator-(const Vec6& a, const Vec6& b) { Vec6 r; for (int i = 0; i < 6; i++) r.v[i] = a[i] - b[i]; return r; }
inline Vec6 operator-(const Vec6& a) { Vec6 r; for (int i = 0; i < 6; i++) r.v[i] = -a[i]; return r; }
inline Vec6 operator*(s_t s, const Vec6& a) { Vec6 r; for (int i = 0; i < 6; i++) r.v[i] = s * a[i]; return r; }
inline Vec6 operator*(const Vec6& a, s_t s) { return s * a; }
inline s_t dot(const Vec6& a, const Vec6& b) { s_t s = 0; for (int i = 0; i < 6; i++) s += a[i] * b[i]; return s; }

// 6x6 row-major
struct Mat6 {
  s_t m[36];
  s_t& operator()(int r, int c) { return m[6 * r + c]; }
  const s_t& operator()(int r, int c) const { return m[6 * r + c]; }
};
inline Mat6 zero66() { Mat6 r; std::memset(r.m, 0, sizeof(r.m)); return r; }
inline Vec6 operator*(const Mat6& a, const Vec6& x) {
  Vec6 r;
  for (int i = 0; i < 6; i++) { s_t s = 0; for (int j = 0; j < 6; j++) s += a(i, j) * x[j]; r.v[i] = s; }
  return r;
}
inline Mat6 operator*(const Mat6& a, const Mat6& b) {
  Mat6 r;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { s_t s = 0; for (int k = 0; k < 6; k++) s += a(i, k) * b(k, j); r(i, j) = s; }
  return r;
}
inline Mat6 operator+(const Mat6& a, const Mat6& b) { Mat6 r; for (int i = 0; i < 36; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
inline Mat6 operator-(const Mat6& a, const Mat6& b) { Mat6 r; for (int i = 0; i < 36; i++) r.m[i] = a.m[i] - b.m[i]; return r; }
inline Mat6 transpose(const Mat6& a) {
  Mat6 r;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) r(i, j) = a(j, i);
  return r;
}

// ---- Geometry.cpp:1300  AdT:  res = T * s * Inv(T) -------------------------
inline Vec6 AdT(const Iso& T, const Vec6& V) {
  Vec3 w = T.R * head(V);
  Vec3 v = T.R * tail(V) + cross(T.p, w);
  return mk6(w, v);
}
// ---- Geometry.cpp:1437  AdInvT: res = Inv(T)*s*T ---------------------------
inline Vec6 AdInvT(const Iso& T, const Vec6& V) {
  Vec3 w = tmul(T.R, head(V));
  Vec3 v = tmul(T.R, tail(V) + cross(head(V), T.p));
  return mk6(w, v);
}
// ---- Geometry.cpp:1461  AdInvRLinear ---------------------------------------
inline Vec6 AdInvRLinear(const Iso& T, const Vec3& v) { return mk6(mk3(0, 0, 0), tmul(T.R, v)); }
// ---- Geometry.cpp:1469  ad -------------------------------------------------
inline Vec6 ad(const Vec6& X, const Vec6& Y) {
  return mk6(cross(head(X), head(Y)), cross(head(X), tail(Y)) + cross(tail(X), head(Y)));
}
// ---- Geometry.cpp:1504  dAdT -----------------------------------------------
inline Vec6 dAdT(const Iso& T, const Vec6& F) {
  return mk6(tmul(T.R, head(F) + cross(tail(F), T.p)), tmul(T.R, tail(F)));
}
// ---- Geometry.cpp:1530  dAdInvT --------------------------------------------
inline Vec6 dAdInvT(const Iso& T, const Vec6& F) {
  Vec3 f = T.R * tail(F);
  Vec3 m = T.R * head(F) + cross(T.p, f);
  return mk6(m, f);
}
// ---- Geometry.cpp:3506  dad -------------------------------------------------
inline Vec6 dad(const Vec6& s, const Vec6& t) {
  return mk6(cross(head(t), head(s)) + cross(tail(t), tail(s)), cross(tail(t), head(s)));
}
// getAdTMatrix (Geometry.cpp:1313): 6x6 matrix of AdT
inline Mat6 AdTMatrix(const Iso& T) {
  Mat6 r = zero66();
  Mat3 pR = skew(T.p) * T.R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      r(i, j) = T.R(i, j);
      r(3 + i, 3 + j) = T.R(i, j);
      r(3 + i, j) = pR(i, j);
    }
  return r;
}
// ---- Geometry.cpp:3515 transformInertia(T, I) = AdT(T)^T * I * AdT(T) ------
// (the reference uses an unrolled closed form with the same value; this oracle forms the
// congruence explicitly — see tests/test_oracle_props.py::test_transform_inertia_is_congruence)
inline Mat6 transformInertia(const Iso& T, const Mat6& I) {
  Mat6 A = AdTMatrix(T);
  return transpose(A) * I * A;
}

// ---- Geometry.cpp:539 expMapRot (note the 2nd-order Taylor branch below 1e-3) ----
inline Mat3 expMapRot(const Vec3& q) {
  s_t theta = norm(q);
  Mat3 qss = skew(q);
  Mat3 qss2 = qss * qss;
  if (theta < 1.0e-3) return eye3() + qss + 0.5 * qss2;
  return eye3() + (std::sin(theta) / theta) * qss + ((1 - std::cos(theta)) / (theta * theta)) * qss2;
}
// ---- Geometry.cpp:556 expMapJac --------------------------------------------
inline Mat3 expMapJac(const Vec3& q) {
  s_t theta = norm(q);
  Mat3 qss = skew(q);
  Mat3 qss2 = qss * qss;
  if (theta < 1.0e-3) return eye3() + 0.5 * qss + (1.0 / 6.0) * qss2;
  return eye3() + ((1 - std::cos(theta)) / (theta * theta)) * qss +
         ((theta - std::sin(theta)) / (theta * theta * theta)) * qss2;
}
// ---- Geometry.cpp:720 logMap(R) ---------------------------------------------
inline Vec3 logMap(const Mat3& R) {
  const s_t pi = 3.14159265358979323846;
  const s_t DART_EPSILON = 1e-6;
  s_t c = 0.5 * (R(0, 0) + R(1, 1) + R(2, 2) - 1.0);
  c = std::fmax(std::fmin(c, 1.0), -1.0);
  s_t theta = std::acos(c);
  if (theta > pi - DART_EPSILON) {
    s_t delta = 0.5 + 0.125 * (pi - theta) * (pi - theta);
    s_t a = theta * std::sqrt(1.0 + (R(0, 0) - 1.0) * delta);
    s_t b = theta * std::sqrt(1.0 + (R(1, 1) - 1.0) * delta);
    s_t d = theta * std::sqrt(1.0 + (R(2, 2) - 1.0) * delta);
    return mk3(R(2, 1) > R(1, 2) ? a : -a, R(0, 2) > R(2, 0) ? b : -b, R(1, 0) > R(0, 1) ? d : -d);
  }
  s_t alpha;
  if (theta > DART_EPSILON) alpha = 0.5 * theta / std::sin(theta);
  else alpha = 0.5 + (1.0 / 12.0) * theta * theta;
  return mk3(alpha * (R(2, 1) - R(1, 2)), alpha * (R(0, 2) - R(2, 0)), alpha * (R(1, 0) - R(0, 1)));
}
// ---- Geometry.cpp:3414 expAngular(s): exact Rodrigues, Taylor only below 1e-6 ----
inline Mat3 expAngular(const Vec3& s) {
  s_t s2[3] = {s[0] * s[0], s[1] * s[1], s[2] * s[2]};
  s_t s3[3] = {s[0] * s[1], s[1] * s[2], s[2] * s[0]};
  s_t theta = std::sqrt(s2[0] + s2[1] + s2[2]);
  s_t cos_t = std::cos(theta), alpha, beta;
  if (theta > 1e-6) { alpha = std::sin(theta) / theta; beta = (1.0 - cos_t) / theta / theta; }
  else { alpha = 1.0 - theta * theta / 6.0; beta = 0.5 - theta * theta / 24.0; }
  Mat3 r;
  r(0, 0) = beta * s2[0] + cos_t;     r(1, 0) = beta * s3[0] + alpha * s[2]; r(2, 0) = beta * s3[2] - alpha * s[1];
  r(0, 1) = beta * s3[0] - alpha * s[2]; r(1, 1) = beta * s2[1] + cos_t;     r(2, 1) = beta * s3[1] + alpha * s[0];
  r(0, 2) = beta * s3[2] + alpha * s[1]; r(1, 2) = beta * s3[1] - alpha * s[0]; r(2, 2) = beta * s2[2] + cos_t;
  return r;
}

// ---- small dense helpers (stand-in for Eigen::MatrixXs, row-major) ----------
struct MatX {
  int r, c;
  std::vector<s_t> d;
  MatX() : r(0), c(0) {}
  MatX(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
  s_t& operator()(int i, int j) { return d[(size_t)i * c + j]; }
  const s_t& operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};
typedef std::vector<s_t> VecX;
inline MatX matmul(const MatX& a, const MatX& b) {
  MatX o(a.r, b.c);
  for (int i = 0; i < a.r; i++)
    for (int k = 0; k < a.c; k++) {
      s_t aik = a(i, k);
      if (aik == 0) continue;
      for (int j = 0; j < b.c; j++) o(i, j) += aik * b(k, j);
    }
  return o;
}
inline VecX matvec(const MatX& a, const VecX& x) {
  VecX o(a.r, 0.0);
  for (int i = 0; i < a.r; i++) { s_t s = 0; for (int j = 0; j < a.c; j++) s += a(i, j) * x[j]; o[i] = s; }
  return o;
}
inline VecX matTvec(const MatX& a, const VecX& x) {
  VecX o(a.c, 0.0);
  for (int i = 0; i < a.r; i++) for (int j = 0; j < a.c; j++) o[j] += a(i, j) * x[i];
  return o;
}
inline MatX transposeX(const MatX& a) {
  MatX o(a.c, a.r);
  for (int i = 0; i < a.r; i++) for (int j = 0; j < a.c; j++) o(j, i) = a(i, j);
  return o;
}
inline MatX identityX(int n) { MatX o(n, n); for (int i = 0; i < n; i++) o(i, i) = 1; return o; }

// SPD inverse via Cholesky (stand-in for Eigen .llt()/.ldlt() solves with Identity,
// Skeleton.cpp:12605, ConfigurationSpace.hpp:53)
inline bool spdInverse(const MatX& A, MatX& Ainv) {
  int n = A.r;
  MatX L(n, n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      s_t s = A(i, j);
      for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k);
      if (i == j) { if (s <= 0) return false; L(i, i) = std::sqrt(s); }
      else L(i, j) = s / L(j, j);
    }
  Ainv = MatX(n, n);
  for (int col = 0; col < n; col++) {
    VecX y(n, 0.0);
    for (int i = 0; i < n; i++) {
      s_t s = (i == col) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= L(i, k) * y[k];
      y[i] = s / L(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
      s_t s = y[i];
      for (int k = i + 1; k < n; k++) s -= L(k, i) * Ainv(k, col);
      Ainv(i, col) = s / L(i, i);
    }
  }
  return true;
}

}  // namespace nbo
