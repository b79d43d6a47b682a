// oracle/ref_boxbox_prelude.hpp - TEST INFRASTRUCTURE.  What the reference's own analytic narrow phases (dart/collision/dart/
// DARTCollide.cpp: the ODE-derived dBoxBox and its helpers from `typedef s_t dVector3[4]` to the end of collideBoxBox, and collideBoxSphere /
// collideSphereBox / collideSphereSphere, and collideCapsuleCapsule / collideSphereCapsule / collideCapsuleSphere) need in order to compile WITHOUT Eigen and without the rest of DART: stand-ins for the handful of Eigen::Vector3s / Isometry3s operations that range uses and
// for the collision types it fills in.  oracle/ref_build.py concatenates this file, that line range read from /root/reference at build time
// (nothing of it is stored in this repo) and ref_boxbox_epilogue.hpp into oracle/_ref/, and compiles libdboxbox_ref.so from it.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <vector>

using std::abs;    // the range calls abs() on doubles: make sure the floating-point overloads are the ones found, as in the reference's build
using std::fabs;
using std::sqrt;

typedef double s_t;

namespace Eigen {
struct Vector3s {
  double v[3];
  Vector3s() : v{0, 0, 0} {}
  Vector3s(double x, double y, double z) : v{x, y, z} {}
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  Vector3s operator+(const Vector3s& o) const { return Vector3s(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vector3s operator-(const Vector3s& o) const { return Vector3s(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vector3s operator-() const { return Vector3s(-v[0], -v[1], -v[2]); }
  Vector3s operator*(double s) const { return Vector3s(v[0] * s, v[1] * s, v[2] * s); }
  Vector3s& operator+=(const Vector3s& o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
  Vector3s& operator-=(const Vector3s& o) { v[0] -= o.v[0]; v[1] -= o.v[1]; v[2] -= o.v[2]; return *this; }
  Vector3s& operator*=(double s) { v[0] *= s; v[1] *= s; v[2] *= s; return *this; }
  double dot(const Vector3s& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  static Vector3s UnitZ() { return Vector3s(0, 0, 1); }
  double squaredNorm() const { return dot(*this); }
  double norm() const { return std::sqrt(squaredNorm()); }                      // Eigen: sqrt of the sum of squares, in this order
  void setZero() { v[0] = v[1] = v[2] = 0; }
  void normalize() { const double n = norm(); v[0] /= n; v[1] /= n; v[2] /= n; }   // Eigen divides by the norm (no reciprocal)
  Vector3s normalized() const { Vector3s r = *this; r.normalize(); return r; }
  // `vec << a, b, c;`
  struct Comma {
    Vector3s* t; int i;
    Comma operator,(double x) { t->v[i] = x; return Comma{t, i + 1}; }
  };
  Comma operator<<(double x) { v[0] = x; return Comma{this, 1}; }
};
inline Vector3s operator*(double s, const Vector3s& a) { return a * s; }

struct Matrix3s {
  double m[3][3];
  Vector3s col(int c) const { return Vector3s(m[0][c], m[1][c], m[2][c]); }
  // fixed-size matrix * vector: every coefficient is the sum over the columns in order (Eigen's coefficient-based product)
  Vector3s operator*(const Vector3s& x) const {
    return Vector3s(m[0][0] * x[0] + m[0][1] * x[1] + m[0][2] * x[2], m[1][0] * x[0] + m[1][1] * x[1] + m[1][2] * x[2],
                    m[2][0] * x[0] + m[2][1] * x[1] + m[2][2] * x[2]);
  }
};
struct Isometry3s {
  double m[4][4];
  Isometry3s() : m{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}} {}
  double operator()(int r, int c) const { return m[r][c]; }
  double& operator()(int r, int c) { return m[r][c]; }
  Vector3s translation() const { return Vector3s(m[0][3], m[1][3], m[2][3]); }
  Matrix3s linear() const { Matrix3s r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[i][j]; return r; }
  Vector3s operator*(const Vector3s& x) const { return linear() * x + translation(); }
  Isometry3s inverse() const {     // Transform<Isometry>::inverse(): R^T, -R^T p
    Isometry3s r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i];
    const Vector3s t = r.linear() * translation();
    for (int i = 0; i < 3; i++) r.m[i][3] = -t[i];
    return r;
  }
};
}  // namespace Eigen

namespace dart {
namespace math {
namespace constantsd {
inline double pi() { return 3.141592653589793238462643383279502884; }
}
}  // namespace math
namespace collision {
struct CollisionObject {};
struct CollisionOption { double contactClippingDepth = 0.03; };
enum ContactType { UNSUPPORTED = 0, VERTEX_FACE = 1, FACE_VERTEX = 2, EDGE_EDGE = 3, SPHERE_BOX = 4, BOX_SPHERE = 5, SPHERE_SPHERE = 6,
                   PIPE_SPHERE = 13, SPHERE_PIPE = 14, PIPE_PIPE = 15 };
enum ClipSphereHalfspace { BOTH = 0, TOP = 1, BOTTOM = 2 };
#define DART_COLLISION_EPS 1E-6
struct Contact {
  Eigen::Vector3s point, normal;
  double penetrationDepth = 0;
  CollisionObject* collisionObject1 = nullptr;
  CollisionObject* collisionObject2 = nullptr;
  int type = UNSUPPORTED;
  Eigen::Vector3s edgeAClosestPoint, edgeAFixedPoint, edgeADir, edgeBClosestPoint, edgeBFixedPoint, edgeBDir;
  Eigen::Vector3s sphereCenter, face1Normal, face2Normal, face3Normal, centerA, centerB;
  bool face1Locked = false, face2Locked = false, face3Locked = false;
  double radiusA = 0, radiusB = 0;
  Eigen::Vector3s pipeDir, pipeClosestPoint, pipeFixedPoint;   // capsule contacts (Contact.hpp:186-199)
  double sphereRadius = 0, pipeRadius = 0;
};
struct CollisionResult {
  std::vector<Contact> contacts;
  void addContact(const Contact& c) { contacts.push_back(c); }
};
using namespace math;
