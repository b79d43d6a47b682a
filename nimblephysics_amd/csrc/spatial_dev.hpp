// spatial_dev.hpp — device-side spatial algebra for the one-world-per-lane kernels (gfx950).
//
// Everything is a small struct of fp64 scalars that lives in VGPRs; all loops have compile-time
// trip counts so hipcc fully unrolls them.  6-vectors are [omega; v] like the reference
// (dart/math/Geometry.cpp:1300-1311) so results can be compared component by component.
// Symmetric 6x6 matrices are stored packed (21 doubles): index sym6(i,j), i<=j.
#pragma once
#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__

// The device namespace of this instantiation of the library (the library is built several times from one set of sources, see
// abi_variants.h: every build names its own - -DNBL_NS=nbl_c16 - so that their kernels and constants never meet at link time).
#ifndef NBL_NS
#define NBL_NS nbl
#endif
namespace NBL_NS {

struct V3 { double x, y, z; };
struct M3 { double m[9]; };            // row-major
struct T12 { M3 R; V3 p; };            // rigid transform
struct V6 { V3 w, v; };                // twist / wrench  [angular; linear]
struct S6 { double a[21]; };           // symmetric 6x6, packed upper triangle by rows

DEV V3 mk3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
DEV V3 operator*(double s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV double get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

DEV V3 mul(const M3& A, V3 x) {
  return mk3(A.m[0] * x.x + A.m[1] * x.y + A.m[2] * x.z, A.m[3] * x.x + A.m[4] * x.y + A.m[5] * x.z,
             A.m[6] * x.x + A.m[7] * x.y + A.m[8] * x.z);
}
DEV V3 tmul(const M3& A, V3 x) {  // A^T x
  return mk3(A.m[0] * x.x + A.m[3] * x.y + A.m[6] * x.z, A.m[1] * x.x + A.m[4] * x.y + A.m[7] * x.z,
             A.m[2] * x.x + A.m[5] * x.y + A.m[8] * x.z);
}
DEV M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
DEV M3 mulABt(const M3& A, const M3& B) {  // A * B^T
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B.m[3 * j] + A.m[3 * i + 1] * B.m[3 * j + 1] + A.m[3 * i + 2] * B.m[3 * j + 2];
  return C;
}
DEV M3 mulAtB(const M3& A, const M3& B) {  // A^T * B
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[i] * B.m[j] + A.m[3 + i] * B.m[3 + j] + A.m[6 + i] * B.m[6 + j];
  return C;
}
DEV M3 transpose(const M3& A) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * j + i];
  return C;
}
DEV M3 eye3() { M3 r; r.m[0] = 1; r.m[1] = 0; r.m[2] = 0; r.m[3] = 0; r.m[4] = 1; r.m[5] = 0; r.m[6] = 0; r.m[7] = 0; r.m[8] = 1; return r; }
DEV M3 skew(V3 a) {
  M3 r;
  r.m[0] = 0;    r.m[1] = -a.z; r.m[2] = a.y;
  r.m[3] = a.z;  r.m[4] = 0;    r.m[5] = -a.x;
  r.m[6] = -a.y; r.m[7] = a.x;  r.m[8] = 0;
  return r;
}

DEV T12 mulT(const T12& a, const T12& b) { T12 r; r.R = mul(a.R, b.R); r.p = mul(a.R, b.p) + a.p; return r; }
DEV T12 invT(const T12& a) { T12 r; r.R = transpose(a.R); r.p = -tmul(a.R, a.p); return r; }

DEV V6 mk6(V3 w, V3 v) { V6 r; r.w = w; r.v = v; return r; }
DEV V6 zero6() { return mk6(mk3(0, 0, 0), mk3(0, 0, 0)); }
DEV V6 operator+(V6 a, V6 b) { return mk6(a.w + b.w, a.v + b.v); }
DEV V6 operator-(V6 a, V6 b) { return mk6(a.w - b.w, a.v - b.v); }
DEV V6 operator-(V6 a) { return mk6(-a.w, -a.v); }
DEV V6 operator*(double s, V6 a) { return mk6(s * a.w, s * a.v); }
DEV double dot(V6 a, V6 b) { return dot(a.w, b.w) + dot(a.v, b.v); }
DEV double get(const V6& a, int i) { return i < 3 ? get(a.w, i) : get(a.v, i - 3); }
DEV void toArr(const V6& a, double* o) { o[0] = a.w.x; o[1] = a.w.y; o[2] = a.w.z; o[3] = a.v.x; o[4] = a.v.y; o[5] = a.v.z; }
DEV V6 fromArr(const double* o) { return mk6(mk3(o[0], o[1], o[2]), mk3(o[3], o[4], o[5])); }

// Geometry.cpp:1300 AdT, :1437 AdInvT, :1504 dAdT, :1530 dAdInvT, :1469 ad, :3506 dad
DEV V6 AdT(const T12& T, V6 V) { V3 w = mul(T.R, V.w); return mk6(w, mul(T.R, V.v) + cross(T.p, w)); }
DEV V6 AdInvT(const T12& T, V6 V) { return mk6(tmul(T.R, V.w), tmul(T.R, V.v + cross(V.w, T.p))); }
DEV V6 dAdT(const T12& T, V6 F) { return mk6(tmul(T.R, F.w + cross(F.v, T.p)), tmul(T.R, F.v)); }
DEV V6 dAdInvT(const T12& T, V6 F) { V3 f = mul(T.R, F.v); return mk6(mul(T.R, F.w) + cross(T.p, f), f); }
DEV V6 ad(V6 X, V6 Y) { return mk6(cross(X.w, Y.w), cross(X.w, Y.v) + cross(X.v, Y.w)); }
DEV V6 dad(V6 s, V6 t) { return mk6(cross(t.w, s.w) + cross(t.v, s.v), cross(t.v, s.w)); }

// ---- packed symmetric 6x6 ----
DEV constexpr int sym6(int i, int j) { return i <= j ? (i * 6 - (i * (i - 1)) / 2 + (j - i)) : (j * 6 - (j * (j - 1)) / 2 + (i - j)); }
DEV S6 zeroS6() { S6 r;
#pragma unroll
  for (int i = 0; i < 21; i++) r.a[i] = 0; return r; }
DEV V6 mul(const S6& A, V6 x) {
  double xi[6], o[6];
  toArr(x, xi);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) s += A.a[sym6(i, j)] * xi[j];
    o[i] = s;
  }
  return fromArr(o);
}
DEV void addTo(S6& A, const S6& B) {
#pragma unroll
  for (int i = 0; i < 21; i++) A.a[i] += B.a[i];
}
// A -= s * x x^T
DEV void rank1Sub(S6& A, V6 x, double s) {
  double xi[6];
  toArr(x, xi);
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) A.a[sym6(i, j)] -= s * xi[i] * xi[j];
}
// blocks: A = [[TL, TR],[TR^T, BR]]
DEV void blocks(const S6& A, M3& TL, M3& TR, M3& BR) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      TL.m[3 * i + j] = A.a[sym6(i, j)];
      TR.m[3 * i + j] = A.a[sym6(i, 3 + j)];
      BR.m[3 * i + j] = A.a[sym6(3 + i, 3 + j)];
    }
}
DEV S6 fromBlocks(const M3& TL, const M3& TR, const M3& BR) {
  S6 A;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (j >= i) { A.a[sym6(i, j)] = TL.m[3 * i + j]; A.a[sym6(3 + i, 3 + j)] = BR.m[3 * i + j]; }
      A.a[sym6(i, 3 + j)] = TR.m[3 * i + j];
    }
  return A;
}
// X^T A X with X = AdInvT(T) = Ad(T^-1): moves an inertia expressed in the child frame into the
// parent frame (value of transformInertia(T.inverse(), A), Geometry.cpp:3515, used at
// detail/GenericJoint.hpp:2168-2185).  With P = [p]x and primes = blocks rotated by R:
//   BR_new = BR',  TR_new = TR' + P BR',  TL_new = TL' + P TR'^T + (P TR'^T)^T - P BR' P
DEV S6 congruenceToParent(const T12& T, const S6& A) {
  M3 TL, TR, BR;
  blocks(A, TL, TR, BR);
  M3 TLr = mulABt(mul(T.R, TL), T.R);
  M3 TRr = mulABt(mul(T.R, TR), T.R);
  M3 BRr = mulABt(mul(T.R, BR), T.R);
  M3 P = skew(T.p);
  M3 PBR = mul(P, BRr);
  M3 TRn;
#pragma unroll
  for (int i = 0; i < 9; i++) TRn.m[i] = TRr.m[i] + PBR.m[i];
  M3 PTRt = mulABt(P, TRr);   // P * TR'^T
  M3 PBRP = mul(PBR, P);      // P BR' P
  M3 TLn;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) TLn.m[3 * i + j] = TLr.m[3 * i + j] + PTRt.m[3 * i + j] + PTRt.m[3 * j + i] - PBRP.m[3 * i + j];
  return fromBlocks(TLn, TRn, BRr);
}

// ---- exponential / logarithm maps, restating the exact formulas (and branch thresholds) of
//      Geometry.cpp:539 expMapRot, :556 expMapJac, :720 logMap, :3414 expAngular ----
DEV M3 expMapRot(V3 q) {
  double th2 = dot(q, q), theta = sqrt(th2);
  M3 S = skew(q), S2 = mul(S, S), R;
  double A, B;
  if (theta < 1.0e-3) { A = 1.0; B = 0.5; }
  else { double st, ct; sincos(theta, &st, &ct); A = st / theta; B = (1 - ct) / (theta * theta); }   // one range reduction for both
  M3 I = eye3();
#pragma unroll
  for (int i = 0; i < 9; i++) R.m[i] = I.m[i] + A * S.m[i] + B * S2.m[i];
  return R;
}
DEV M3 expMapJac(V3 q) {
  double th2 = dot(q, q), theta = sqrt(th2);
  M3 S = skew(q), S2 = mul(S, S), J;
  double A, B;
  if (theta < 1.0e-3) { A = 0.5; B = 1.0 / 6.0; }
  else { double st, ct; sincos(theta, &st, &ct); A = (1 - ct) / (theta * theta); B = (theta - st) / (theta * theta * theta); }
  M3 I = eye3();
#pragma unroll
  for (int i = 0; i < 9; i++) J.m[i] = I.m[i] + A * S.m[i] + B * S2.m[i];
  return J;
}
DEV V3 logMap(const M3& R) {
  const double pi = 3.14159265358979323846;
  double c = 0.5 * (R.m[0] + R.m[4] + R.m[8] - 1.0);
  c = fmax(fmin(c, 1.0), -1.0);
  double theta = acos(c);
  if (theta > pi - 1e-6) {
    double delta = 0.5 + 0.125 * (pi - theta) * (pi - theta);
    double a = theta * sqrt(1.0 + (R.m[0] - 1.0) * delta);
    double b = theta * sqrt(1.0 + (R.m[4] - 1.0) * delta);
    double d = theta * sqrt(1.0 + (R.m[8] - 1.0) * delta);
    return mk3(R.m[7] > R.m[5] ? a : -a, R.m[2] > R.m[6] ? b : -b, R.m[3] > R.m[1] ? d : -d);
  }
  // sin(acos(c)) = sqrt((1 - c)(1 + c)): both factors are exact near |c| = 1, no second transcendental
  double alpha = (theta > 1e-6) ? 0.5 * theta / sqrt((1.0 - c) * (1.0 + c)) : 0.5 + (1.0 / 12.0) * theta * theta;
  return mk3(alpha * (R.m[7] - R.m[5]), alpha * (R.m[2] - R.m[6]), alpha * (R.m[3] - R.m[1]));
}
DEV M3 expAngular(V3 s) {
  double s2x = s.x * s.x, s2y = s.y * s.y, s2z = s.z * s.z;
  double s3x = s.x * s.y, s3y = s.y * s.z, s3z = s.z * s.x;
  double theta = sqrt(s2x + s2y + s2z), sin_t, cos_t, alpha, beta;
  sincos(theta, &sin_t, &cos_t);
  if (theta > 1e-6) { alpha = sin_t / theta; beta = (1.0 - cos_t) / theta / theta; }
  else { alpha = 1.0 - theta * theta / 6.0; beta = 0.5 - theta * theta / 24.0; }
  M3 r;
  r.m[0] = beta * s2x + cos_t;       r.m[1] = beta * s3x - alpha * s.z; r.m[2] = beta * s3z + alpha * s.y;
  r.m[3] = beta * s3x + alpha * s.z; r.m[4] = beta * s2y + cos_t;       r.m[5] = beta * s3y - alpha * s.x;
  r.m[6] = beta * s3z - alpha * s.y; r.m[7] = beta * s3y + alpha * s.x; r.m[8] = beta * s2z + cos_t;
  return r;
}

// Reverse-mode of R = expMapRot(q): given Rbar (dL/dR) returns dL/dq of the exact formula above
// (including its Taylor branch), so the result matches a finite difference of the forward code.
DEV V3 expMapRot_vjp(V3 q, const M3& Rb) {
  double th2 = dot(q, q), theta = sqrt(th2);
  M3 S = skew(q), S2 = mul(S, S);
  double A, B, dA = 0, dB = 0;
  bool taylor = theta < 1.0e-3;
  if (taylor) { A = 1.0; B = 0.5; }
  else {
    double st, ct;
    sincos(theta, &st, &ct);
    A = st / theta; B = (1 - ct) / th2;
    dA = (theta * ct - st) / th2;
    dB = (theta * st - 2 * (1 - ct)) / (th2 * theta);
  }
  double Ab = 0, Bb = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { Ab += Rb.m[i] * S.m[i]; Bb += Rb.m[i] * S2.m[i]; }
  // Sbar = A*Rb + B*(Rb S^T + S^T Rb)
  M3 RbSt = mulABt(Rb, S), StRb = mulAtB(S, Rb), Sb;
#pragma unroll
  for (int i = 0; i < 9; i++) Sb.m[i] = A * Rb.m[i] + B * (RbSt.m[i] + StRb.m[i]);
  V3 qb = mk3(Sb.m[7] - Sb.m[5], Sb.m[2] - Sb.m[6], Sb.m[3] - Sb.m[1]);
  if (!taylor) {
    double tb = (Ab * dA + Bb * dB) / theta;
    qb = qb + tb * q;
  }
  return qb;
}
// Reverse-mode of r = logMap(R) (regular branch; the theta ~ pi branch falls back to the same
// expression evaluated with its own alpha, which is only reached on a measure-zero set).
DEV M3 logMap_vjp(const M3& R, V3 rb) {
  double c = 0.5 * (R.m[0] + R.m[4] + R.m[8] - 1.0);
  bool clamped = (c >= 1.0) || (c <= -1.0);
  c = fmax(fmin(c, 1.0), -1.0);
  double theta = acos(c);
  V3 w = mk3(R.m[7] - R.m[5], R.m[2] - R.m[6], R.m[3] - R.m[1]);
  double alpha, dalpha;
  if (theta > 1e-6) {
    const double st = sqrt((1.0 - c) * (1.0 + c)), ct = c;   // sin / cos of acos(c)
    alpha = 0.5 * theta / st;
    dalpha = 0.5 * (st - theta * ct) / (st * st);
  } else { alpha = 0.5 + (1.0 / 12.0) * theta * theta; dalpha = (1.0 / 6.0) * theta; }
  double ab = dot(rb, w);
  double cb = 0.0;
  if (!clamped && theta > 1e-6) cb = ab * dalpha * (-1.0 / sqrt(1.0 - c * c));
  else if (!clamped) cb = ab * (-1.0 / 6.0);  // d(alpha)/dc at theta->0: alpha ~ 1/2 + theta^2/12, theta^2 ~ 2(1-c)
  V3 wb = alpha * rb;
  M3 Rb;
  Rb.m[0] = 0.5 * cb; Rb.m[4] = 0.5 * cb; Rb.m[8] = 0.5 * cb;
  Rb.m[7] = wb.x; Rb.m[5] = -wb.x;
  Rb.m[2] = wb.y; Rb.m[6] = -wb.y;
  Rb.m[3] = wb.z; Rb.m[1] = -wb.z;
  return Rb;
}

// 6x6 SPD: LDL^T factor in place on a packed symmetric matrix (lower part in the same packed slots),
// then solve.  (math::inverse<SE3Space> uses Eigen ldlt, ConfigurationSpace.hpp:48-65.)
struct LDL6 { double l[15]; double d[6]; };  // l index: tri(i,j) for i>j
DEV constexpr int tri(int i, int j) { return i * (i - 1) / 2 + j; }
DEV LDL6 ldl6(const S6& A) {
  LDL6 f;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = A.a[sym6(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) dj -= f.l[tri(j, k)] * f.l[tri(j, k)] * f.d[k];
    f.d[j] = dj;
    double inv = 1.0 / dj;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double s = A.a[sym6(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) s -= f.l[tri(i, k)] * f.l[tri(j, k)] * f.d[k];
      f.l[tri(i, j)] = s * inv;
    }
  }
  return f;
}
DEV void ldl6Solve(const LDL6& f, double* x) {  // in place
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int k = 0; k < i; k++) x[i] -= f.l[tri(i, k)] * x[k];
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] /= f.d[i];
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int k = i + 1; k < 6; k++) x[i] -= f.l[tri(k, i)] * x[k];
}

DEV V3 solve3(const M3& A, V3 b) {  // Cramer
  double c00 = A.m[4] * A.m[8] - A.m[5] * A.m[7], c01 = A.m[5] * A.m[6] - A.m[3] * A.m[8], c02 = A.m[3] * A.m[7] - A.m[4] * A.m[6];
  double det = A.m[0] * c00 + A.m[1] * c01 + A.m[2] * c02, id = 1.0 / det;
  double c10 = A.m[2] * A.m[7] - A.m[1] * A.m[8], c11 = A.m[0] * A.m[8] - A.m[2] * A.m[6], c12 = A.m[1] * A.m[6] - A.m[0] * A.m[7];
  double c20 = A.m[1] * A.m[5] - A.m[2] * A.m[4], c21 = A.m[2] * A.m[3] - A.m[0] * A.m[5], c22 = A.m[0] * A.m[4] - A.m[1] * A.m[3];
  return mk3(id * (c00 * b.x + c10 * b.y + c20 * b.z), id * (c01 * b.x + c11 * b.y + c21 * b.z), id * (c02 * b.x + c12 * b.y + c22 * b.z));
}

// VJP of the ball joint's position integration q' = logMap(R(q) R(w dt)) (BallJoint.cpp:333-349; the reference finite-differences it,
// :351-408): posPos^T g and velPos^T g of the 3 x 3 block.
DEV void so3IntegrationVjp(V3 q, V3 w, double dt, V3 g, V3& posT, V3& velT) {
  const M3 R = expMapRot(q), E = expMapRot(dt * w);
  const M3 Rnb = logMap_vjp(mul(R, E), g);
  posT = expMapRot_vjp(q, mulABt(Rnb, E));
  velT = dt * expMapRot_vjp(dt * w, mulAtB(R, Rnb));
}

// The same for the free joint's q' = [logMap(R(r) R(w dt)); p + R(r) vl dt] (FreeJoint.cpp:922-929): posPos^T g and velPos^T g, 6 entries each.
DEV void se3IntegrationVjp(V3 r, V3 w, V3 vl, double dt, V3 grn, V3 gpn, double (&posT)[6], double (&velT)[6]) {
  const M3 R = expMapRot(r), E = expMapRot(dt * w);
  const M3 Rnb = logMap_vjp(mul(R, E), grn);
  M3 Rb = mulABt(Rnb, E);
  const V3 vdt = dt * vl;
  Rb.m[0] += gpn.x * vdt.x; Rb.m[1] += gpn.x * vdt.y; Rb.m[2] += gpn.x * vdt.z;   // p' = p + R vdt
  Rb.m[3] += gpn.y * vdt.x; Rb.m[4] += gpn.y * vdt.y; Rb.m[5] += gpn.y * vdt.z;
  Rb.m[6] += gpn.z * vdt.x; Rb.m[7] += gpn.z * vdt.y; Rb.m[8] += gpn.z * vdt.z;
  const V3 posr = expMapRot_vjp(r, Rb), velw = dt * expMapRot_vjp(dt * w, mulAtB(R, Rnb)), vell = dt * tmul(R, gpn);
  posT[0] = posr.x; posT[1] = posr.y; posT[2] = posr.z; posT[3] = gpn.x; posT[4] = gpn.y; posT[5] = gpn.z;
  velT[0] = velw.x; velT[1] = velw.y; velT[2] = velw.z; velT[3] = vell.x; velT[4] = vell.y; velT[5] = vell.z;
}
DEV double pick3(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

}  // namespace NBL_NS
