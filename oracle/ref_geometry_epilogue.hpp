// oracle/ref_geometry_epilogue.hpp - TEST INFRASTRUCTURE.  C entry points to the reference's own functions compiled above it
// (oracle/ref_build.py, build_geometry): plain arrays in, plain arrays out.  Conventions of the callers (tests/test_oracle_ref_geometry.py):
// 3 x 3 and 6 x 6 matrices row-major, transforms 12 doubles = R row-major (9) then p (3), 6-vectors [omega; v].

namespace {
Eigen::Isometry3s iso12(const double* t) {
  Eigen::Isometry3s T;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T(i, j) = t[3 * i + j]; T(i, 3) = t[9 + i]; }
  return T;
}
void out12(const Eigen::Isometry3s& T, double* t) { for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) t[3 * i + j] = T(i, j); t[9 + i] = T(i, 3); } }
Eigen::Vector3s v3(const double* x) { return Eigen::Vector3s(x[0], x[1], x[2]); }
Eigen::Vector6s v6(const double* x) { Eigen::Vector6s r; for (int i = 0; i < 6; i++) r[i] = x[i]; return r; }
Eigen::Matrix3s m3(const double* x) { Eigen::Matrix3s r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = x[3 * i + j]; return r; }
Eigen::Matrix6s m6(const double* x) { Eigen::Matrix6s r; for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) r(i, j) = x[6 * i + j]; return r; }
void o3(const Eigen::Vector3s& v, double* x) { for (int i = 0; i < 3; i++) x[i] = v[i]; }
void o6(const Eigen::Vector6s& v, double* x) { for (int i = 0; i < 6; i++) x[i] = v[i]; }
void o33(const Eigen::Matrix3s& v, double* x) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) x[3 * i + j] = v(i, j); }
void o66(const Eigen::Matrix6s& v, double* x) { for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) x[6 * i + j] = v(i, j); }
}  // namespace

extern "C" {
using namespace dart::math;
void ref_expMapRot(const double* q, double* R) { o33(expMapRot(v3(q)), R); }
void ref_expMapJac(const double* q, double* J) { o33(expMapJac(v3(q)), J); }
void ref_logMap(const double* R, double* w) { o3(logMap(m3(R)), w); }
void ref_expMap(const double* S, double* T) { out12(expMap(v6(S)), T); }
void ref_expMapDart(const double* S, double* T) { out12(expMapDart(v6(S)), T); }
void ref_expAngular(const double* s, double* T) { out12(expAngular(v3(s)), T); }
void ref_makeSkewSymmetric(const double* v, double* M) { o33(makeSkewSymmetric(v3(v)), M); }
void ref_eulerXYZToMatrix(const double* a, double* R) { o33(eulerXYZToMatrix(v3(a)), R); }
void ref_eulerZYXToMatrix(const double* a, double* R) { o33(eulerZYXToMatrix(v3(a)), R); }
void ref_AdT(const double* T, const double* V, double* out) { o6(AdT(iso12(T), v6(V)), out); }
void ref_AdR(const double* T, const double* V, double* out) { o6(AdR(iso12(T), v6(V)), out); }
void ref_AdTAngular(const double* T, const double* w, double* out) { o6(AdTAngular(iso12(T), v3(w)), out); }
void ref_AdTLinear(const double* T, const double* v, double* out) { o6(AdTLinear(iso12(T), v3(v)), out); }
void ref_AdInvT(const double* T, const double* V, double* out) { o6(AdInvT(iso12(T), v6(V)), out); }
void ref_AdInvRLinear(const double* T, const double* v, double* out) { o6(AdInvRLinear(iso12(T), v3(v)), out); }
void ref_ad(const double* X, const double* Y, double* out) { o6(ad(v6(X), v6(Y)), out); }
void ref_dad(const double* s, const double* t, double* out) { o6(dad(v6(s), v6(t)), out); }
void ref_dAdT(const double* T, const double* F, double* out) { o6(dAdT(iso12(T), v6(F)), out); }
void ref_dAdInvT(const double* T, const double* F, double* out) { o6(dAdInvT(iso12(T), v6(F)), out); }
void ref_dAdInvR(const double* T, const double* F, double* out) { o6(dAdInvR(iso12(T), v6(F)), out); }
void ref_transformInertia(const double* T, const double* I, double* out) { o66(transformInertia(iso12(T), m6(I)), out); }

// FreeJoint::integratePositionsExplicit: q' = log(T(q) T(v dt)), T(x) = [expMapRot(x[0:3]), x[3:6]]
void ref_free_joint_integrate(const double* q, const double* v, double dt, double* qn) {
  dart::dynamics::FreeJoint j;
  o6(j.integratePositionsExplicit(v6(q), v6(v), dt), qn);
}
// ContactConstraint::getTangentBasisMatrixODE with the default first frictional direction (z): out = t1 (3), t2 (3)
void ref_tangent_basis(const double* n, double* out) {
  dart::constraint::ContactConstraint c;
  const auto T = c.getTangentBasisMatrixODE(v3(n));
  for (int i = 0; i < 3; i++) { out[i] = T(i, 0); out[3 + i] = T(i, 1); }
}

void ref_tangent_basis_gradient(const double* n, const double* g, double* out) {
  dart::constraint::ContactConstraint c;
  const auto T = c.getTangentBasisMatrixODEGradient(v3(n), v3(g));
  for (int i = 0; i < 3; i++) { out[i] = T(i, 0); out[3 + i] = T(i, 1); }
}
// math::getContactPoint / getContactPointGradient (edge-edge contacts): in = aP, aD, bP, bD (getContactPoint) or aP, aPg, aD, aDg, bP, bPg, bD, bDg
void ref_getContactPoint(const double* in, double rA, double rB, double* out) {
  o3(getContactPoint(v3(in), v3(in + 3), v3(in + 6), v3(in + 9), rA, rB), out);
}
void ref_getContactPointGradient(const double* in, double rA, double rB, double* out) {
  o3(getContactPointGradient(v3(in), v3(in + 3), v3(in + 6), v3(in + 9), v3(in + 12), v3(in + 15), v3(in + 18), v3(in + 21), rA, rB), out);
}
// math::closestPointOnLineGradient (capsule contacts): in = pointOnLine, its gradient, lineDirection, its gradient, goalPoint, its gradient
void ref_closestPointOnLineGradient(const double* in, double* out) {
  o3(closestPointOnLineGradient(v3(in), v3(in + 3), v3(in + 6), v3(in + 9), v3(in + 12), v3(in + 15)), out);
}

// dart::dynamics::SimpleFeatherstone::forwardDynamics on a caller-described tree of n one-DOF joints (no gravity, no damping: the
// reference's flat-array ABA has neither): parent[n] (-1 = root), axis [n][6] (the joint's screw axis: position map expMap(axis q) AND
// body-frame Jacobian), transformFromParent / transformFromChildren [n][12], inertia [n][36] (spatial tensor, row-major).
void ref_simple_featherstone(int n, const int* parent, const double* axis, const double* fromParent, const double* fromChildren,
                             const double* inertia, const double* pos, const double* vel, const double* force, double* acc) {
  dart::dynamics::SimpleFeatherstone sf;
  for (int i = 0; i < n; i++) {
    dart::dynamics::JointAndBody& jb = sf.emplaceBack();
    jb.axis = v6(axis + 6 * i);
    jb.transformFromParent = iso12(fromParent + 12 * i);
    jb.transformFromChildren = iso12(fromChildren + 12 * i);
    jb.inertia = m6(inertia + 36 * i);
    jb.parentIndex = parent[i];
  }
  std::vector<double> p(pos, pos + n), v(vel, vel + n), f(force, force + n);
  sf.forwardDynamics(p.data(), v.data(), f.data(), acc);
}
}
