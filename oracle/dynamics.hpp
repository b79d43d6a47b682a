// oracle/dynamics.hpp — TEST INFRASTRUCTURE ONLY.
//
// Scalar restatement of the reference's articulated-body dynamics for the joint
// types on the hot path (Revolute, Prismatic, Free and Ball [identity-Jacobian build], Weld).
// One world at a time, no batching, written to follow the reference's recursion
// order so that each routine can be read next to the file:line it cites.
#pragma once
#include "../include/nimble_amd.h"
#include "spatial.hpp"

namespace nbo {

struct Body {
  int parent, jtype, dofOff, ndof;
  Iso Tpj, Tcj, TcjInv;
  Vec3 axis;
  s_t pitch = 0;   // screw joints: translation per turn
  Mat6 G;     // spatial inertia, Inertia.cpp:1368-1383
  Vec6 S[6];  // constant relative Jacobian columns in the child frame
  std::vector<int> children;
};

struct BoxCollider {
  int body;  // -1 = world-fixed
  Iso T;     // in body frame
  Vec3 size;
  s_t mu;
  int shape;  // NBL_SHAPE_BOX | NBL_SHAPE_SPHERE (radius = size.x)
  s_t restitution = 0;   // BodyNode::getRestitutionCoeff of the owning body
  // the BodyNode the collider belongs to and that node's parent, in the caller's numbering (nbl_model_desc.box_node / box_node_parent: a
  // compound joint expanded into a chain of 1-DOF joints puts massless virtual links between a body and its real parent); -2: not given
  int node = -2, nodeParent = -2;
};

struct Model {
  int nb, n;
  std::vector<Body> bodies;
  VecX damping, spring, rest, posLo, posHi, velLo, velHi, forceLo, forceHi;
  Vec3 gravity;
  s_t dt;
  std::vector<int> actionMap;
  std::vector<BoxCollider> boxes;
  std::vector<int> skeleton;            // per body: the dart Skeleton it belongs to (constrained groups unite skeletons)
  int maxContacts;
  s_t clippingDepth, fallbackCfm;
  bool penetrationCorrection = false;   // World::setPenetrationCorrectionEnabled (off by default)
  std::vector<int> selfCollision;       // per body: bit 0 Skeleton::isEnabledSelfCollisionCheck, bit 1 isEnabledAdjacentBodyCheck
  std::vector<int> limitEnforced;       // per DOF: Joint::isPositionLimitEnforced of its joint (JointAspect.hpp:165: off by default)
  // Test instrument (tools/soak_parity.py), NOT the reference's behaviour: with lcpNoiseUlps = k > 0 every entry of the LCP matrix A (and of b) is
  // multiplied by 1 + j k 2^-52, j in {-1, 0, 1} drawn per entry pair (i, j) / (j, i) from (seed, sample): the A another valid order
  // of the same floating-point sums could have produced.  Worlds whose answer changes under it have no stable reference answer.
  // lcpNoiseAbsolute: the noise is j k 2^-52 max |A| ADDED to every entry instead - the rounding error of entries that are sums with
  // cancellation (what a degenerate A, e.g. redundant joint-limit rows next to a contact, is sensitive to).
  int lcpNoiseUlps = 0;
  bool lcpNoiseAbsolute = false;
  // Second test instrument: lcpForced (one entry per LCP row) replaces the OUTPUT of the solver stages 1-3 of a group whose stage 0 fails -
  // "had Dantzig ended on this solution" -, if isLCPSolutionValid accepts it on the group's A (else the step reports 0x40000000);
  // everything after the solver (registration, row classes, standardisation, impulses, backward pass) is the reference's.
  // lcpAlternateA: A is recomputed as J M^-1 J^T from the dense inverse mass matrix and the joint-space constraint forces - the same
  // matrix through another valid order of floating-point operations than the reference's impulse tests (no random draw involved).
  bool lcpAlternateA = false;
  // lcpNoiseBound = k > 0: every entry of A = J M^-1 J^T and of b = -J v moves by j k 2^-52 x (the sum of the MAGNITUDES of the terms it
  // is the sum of), j in {-1, 0, 1} per entry and solve: the first-order rounding-error bound of those sums, which is what another
  // order of evaluation can move an entry by where the terms cancel (fast bodies with a small relative velocity, light bodies on heavy ones).
  int lcpNoiseBound = 0;
  // Interchange format of the LCP cache with the device (not the reference's): three entries per constraint - a frictionless contact
  // and a joint-limit row use the first, the other two are zero - instead of the reference's 3 / 1 / 1 rows.
  bool lcpCacheSlots = false;
  // Third test instrument (see posJacobiansExact): the position-integration Jacobians of free / ball joints by exact forward-mode
  // differentiation instead of the reference's central differences (FreeJoint.cpp:950-1007, BallJoint.cpp:351-408).
  int exactPosJacobians = 0;            // 1: in extended precision (the instrument); 2: the same formulas in doubles (shows the eps / gap^3 conditioning)
  // Fourth test instrument: pinvNoiseUlps = k > 0 multiplies every entry of the pseudo-inverse Q^+ the BACKWARD pass forms (BackpropSnapshot.cpp:2964-2984)
  // by 1 + j k 2^-52, j in {-1, 0, 1} per entry and world: the Q^+ another pseudo-inverse algorithm of the same accuracy returns.  On a full-rank but
  // ill-conditioned Q the reference's imprecise-inverse branch (||I - Q Q^+||^2 >= 1e-18) adds terms that are Q^+T Q^+ x (I - Q Q^+) b - the round-off of
  // its own pseudo-inverse times cond(Q)^2: a world whose gradient moves under this instrument has no answer that is independent of it.
  int pinvNoiseUlps = 0;
  mutable uint64_t pinvNoiseSample = 0;
  std::vector<s_t> lcpForced;
  bool lcpForcedCfm = false;            // ... as the output of stage 2 (the fallback CFM on the diagonal, PGS) instead of stage 1
  uint64_t lcpNoiseSeed = 0;
  mutable uint64_t lcpNoiseSample = 0;  // counts the solves (nbo_step_batch: starts at the thread's first world index)
};

inline Iso loadIso(const double* t) {
  Iso r;
  for (int i = 0; i < 9; i++) r.R.m[i] = t[i];
  for (int i = 0; i < 3; i++) r.p.v[i] = t[9 + i];
  return r;
}

inline Model buildModel(const nbl_model_desc* d) {
  Model m;
  m.nb = d->n_bodies;
  m.n = d->n_dofs;
  m.bodies.resize(m.nb);
  for (int i = 0; i < m.nb; i++) {
    Body& b = m.bodies[i];
    b.parent = d->parent[i];
    b.jtype = d->joint_type[i];
    b.dofOff = d->dof_offset[i];
    b.ndof = (b.jtype == NBL_JOINT_FREE) ? 6 : (b.jtype == NBL_JOINT_BALL ? 3 : (b.jtype == NBL_JOINT_WELD ? 0 : 1));
    b.pitch = (b.jtype == NBL_JOINT_SCREW) ? (d->pitch ? d->pitch[i] : 0.1) : 0.0;
    b.Tpj = loadIso(d->T_pj + 12 * i);
    b.Tcj = loadIso(d->T_cj + 12 * i);
    b.TcjInv = inverse(b.Tcj);
    b.axis = mk3(d->axis[3 * i], d->axis[3 * i + 1], d->axis[3 * i + 2]);
    // spatial tensor (Inertia.cpp:1368-1383)
    Vec3 c = mk3(d->com[3 * i], d->com[3 * i + 1], d->com[3 * i + 2]);
    s_t mass = d->mass[i];
    const double* I = d->inertia + 6 * i;
    Mat3 Ic;
    Ic(0, 0) = I[0]; Ic(1, 1) = I[1]; Ic(2, 2) = I[2];
    Ic(0, 1) = Ic(1, 0) = I[3]; Ic(0, 2) = Ic(2, 0) = I[4]; Ic(1, 2) = Ic(2, 1) = I[5];
    Mat3 C = skew(c);
    Mat3 TL = Ic + mass * (C * transpose(C));
    b.G = zero66();
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) {
        b.G(r, cc) = TL(r, cc);
        b.G(r, 3 + cc) = mass * C(r, cc);
        b.G(3 + r, cc) = mass * C(cc, r);
      }
    b.G(3, 3) = b.G(4, 4) = b.G(5, 5) = mass;
    // relative Jacobian
    for (int k = 0; k < 6; k++) b.S[k] = zero6();
    if (b.jtype == NBL_JOINT_REVOLUTE) {
      // AdTAngular(T_cj, axis)  RevoluteJoint.cpp:141-152
      b.S[0] = AdT(b.Tcj, mk6(b.axis, mk3(0, 0, 0)));
    } else if (b.jtype == NBL_JOINT_PRISMATIC) {
      // AdTLinear(T_cj, axis)   PrismaticJoint.cpp
      b.S[0] = AdT(b.Tcj, mk6(mk3(0, 0, 0), b.axis));
    } else if (b.jtype == NBL_JOINT_FREE) {
      // getAdTMatrix(T_cj)      FreeJoint.cpp:1049-1056 (DART_USE_IDENTITY_JACOBIAN)
      Mat6 A = AdTMatrix(b.Tcj);
      for (int k = 0; k < 6; k++)
        for (int r = 0; r < 6; r++) b.S[k][r] = A(r, k);
    } else if (b.jtype == NBL_JOINT_SCREW) {
      // AdT(T_cj, [axis; axis pitch / 2 pi])   ScrewJoint.cpp:160-179
      b.S[0] = AdT(b.Tcj, mk6(b.axis, b.axis * (b.pitch / (2.0 * M_PI))));
    } else if (b.jtype == NBL_JOINT_BALL) {
      // getAdTMatrix(T_cj).leftCols<3>()      BallJoint.cpp:441-452 (DART_USE_IDENTITY_JACOBIAN)
      Mat6 A = AdTMatrix(b.Tcj);
      for (int k = 0; k < 3; k++)
        for (int r = 0; r < 6; r++) b.S[k][r] = A(r, k);
    }
    if (b.parent >= 0) m.bodies[b.parent].children.push_back(i);
  }
  auto cp = [&](const double* p, VecX& v, double dflt) {
    v.assign(m.n, dflt);
    if (p) for (int i = 0; i < m.n; i++) v[i] = p[i];
  };
  const double inf = INFINITY;
  cp(d->damping, m.damping, 0); cp(d->spring, m.spring, 0); cp(d->rest, m.rest, 0);
  cp(d->pos_lo, m.posLo, -inf); cp(d->pos_hi, m.posHi, inf);
  m.selfCollision.assign(m.nb, 0);
  if (d->body_self_collision) for (int i = 0; i < m.nb; i++) m.selfCollision[i] = d->body_self_collision[i];
  m.limitEnforced.assign(m.n, 0);
  if (d->dof_limit_enforced) for (int i = 0; i < m.n; i++) m.limitEnforced[i] = d->dof_limit_enforced[i] != 0;
  cp(d->vel_lo, m.velLo, -inf); cp(d->vel_hi, m.velHi, inf);
  cp(d->force_lo, m.forceLo, -inf); cp(d->force_hi, m.forceHi, inf);
  m.gravity = mk3(d->gravity[0], d->gravity[1], d->gravity[2]);
  m.dt = d->dt;
  m.actionMap.assign(d->action_map, d->action_map + d->n_action);
  for (int i = 0; i < d->n_boxes; i++) {
    BoxCollider bc;
    bc.body = d->box_body[i];
    bc.T = loadIso(d->box_T + 12 * i);
    bc.size = mk3(d->box_size[3 * i], d->box_size[3 * i + 1], d->box_size[3 * i + 2]);
    bc.mu = d->box_mu[i];
    bc.shape = d->box_shape ? d->box_shape[i] : NBL_SHAPE_BOX;
    bc.restitution = d->box_restitution ? d->box_restitution[i] : 0.0;
    if (d->box_node && d->box_node_parent) { bc.node = d->box_node[i]; bc.nodeParent = d->box_node_parent[i]; }
    m.boxes.push_back(bc);
  }
  m.maxContacts = d->max_contacts;
  m.clippingDepth = d->contact_clipping_depth;
  m.fallbackCfm = d->fallback_cfm;
  m.penetrationCorrection = d->penetration_correction != 0;
  m.skeleton.resize(m.nb);
  for (int i = 0; i < m.nb; i++) {
    if (d->body_skeleton) m.skeleton[i] = d->body_skeleton[i];
    else { int r = i; while (m.bodies[r].parent >= 0) r = m.bodies[r].parent; m.skeleton[i] = r; }   // default: one skeleton per tree
  }
  return m;
}

// Per-body kinematic cache (BodyNode::mWorldTransform, mVelocity, mPartialAcceleration)
struct Kin {
  Iso Trel, Tworld;
  Vec6 V, eta;
};

// Joint transform Q(q): RevoluteJoint.cpp:203-211, PrismaticJoint.cpp, FreeJoint.cpp:74-81,1027-1044
inline Iso jointQ(const Body& b, const s_t* q) {
  Iso Q = isoIdentity();
  if (b.jtype == NBL_JOINT_REVOLUTE) Q.R = expAngular(b.axis * q[b.dofOff]);
  else if (b.jtype == NBL_JOINT_PRISMATIC) Q.p = b.axis * q[b.dofOff];
  else if (b.jtype == NBL_JOINT_FREE) {
    Q.R = expMapRot(mk3(q[b.dofOff], q[b.dofOff + 1], q[b.dofOff + 2]));
    Q.p = mk3(q[b.dofOff + 3], q[b.dofOff + 4], q[b.dofOff + 5]);
  } else if (b.jtype == NBL_JOINT_BALL) {
    Q.R = expMapRot(mk3(q[b.dofOff], q[b.dofOff + 1], q[b.dofOff + 2]));   // BallJoint.cpp:91-95, 422-438
  } else if (b.jtype == NBL_JOINT_SCREW) {
    // math::expMap([axis; h axis] q), ScrewJoint.cpp:217-232: translation parallel to the rotation axis, so exp = (R(axis q), h axis q)
    Q.R = expAngular(b.axis * q[b.dofOff]);
    Q.p = b.axis * (b.pitch / (2.0 * M_PI) * q[b.dofOff]);
  }
  return Q;
}

inline Vec6 jointTwist(const Body& b, const s_t* v) {  // S * dq
  Vec6 s = zero6();
  for (int k = 0; k < b.ndof; k++) s = s + b.S[k] * v[b.dofOff + k];
  return s;
}

// Forward kinematics: T = T_pj Q T_cj^-1; V = AdInvT(T, V_parent) + S dq; eta = ad(V, S dq) + dS dq
// (detail/GenericJoint.hpp:1803-1824; dS = 0 for every joint type on the path)
inline void kinematics(const Model& m, const s_t* q, const s_t* v, std::vector<Kin>& kin) {
  kin.resize(m.nb);
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    Kin& k = kin[i];
    k.Trel = b.Tpj * jointQ(b, q) * b.TcjInv;
    Vec6 Sdq = jointTwist(b, v);
    if (b.parent >= 0) {
      k.Tworld = kin[b.parent].Tworld * k.Trel;
      k.V = AdInvT(k.Trel, kin[b.parent].V) + Sdq;
    } else {
      k.Tworld = k.Trel;
      k.V = Sdq;
    }
    k.eta = ad(k.V, Sdq);
  }
}

// Articulated-body pass cache
struct Art {
  Mat6 AI;       // BodyNode::mArtInertia
  s_t psi[36];   // Joint::mInvProjArtInertia (ndof x ndof, row-major with stride ndof)
  Vec6 AIS[6];   // AI * S columns
};

inline void invertSmall(const s_t* A, int k, s_t* out) {
  // math::inverse<ConfigSpaceT>: closed form up to 4, LDLT otherwise (ConfigurationSpace.hpp:48-65)
  if (k == 0) return;
  if (k == 1) { out[0] = 1.0 / A[0]; return; }
  MatX a(k, k), ai;
  for (int i = 0; i < k * k; i++) a.d[i] = A[i];
  spdInverse(a, ai);
  for (int i = 0; i < k * k; i++) out[i] = ai.d[i];
}

// BodyNode::updateArtInertia leaf->root (BodyNode.cpp:2046-2073; GenericJoint.hpp:2168-2185, 2276-2301)
inline void articulatedInertias(const Model& m, const std::vector<Kin>& kin, std::vector<Art>& art) {
  art.resize(m.nb);
  for (int i = m.nb - 1; i >= 0; i--) {
    const Body& b = m.bodies[i];
    Art& a = art[i];
    a.AI = b.G;
    for (int c : b.children) {
      const Body& cb = m.bodies[c];
      const Art& ca = art[c];
      Mat6 PI = ca.AI;
      // PI -= AIS * psi * AIS^T
      for (int r = 0; r < 6; r++)
        for (int cc = 0; cc < 6; cc++) {
          s_t s = 0;
          for (int k1 = 0; k1 < cb.ndof; k1++)
            for (int k2 = 0; k2 < cb.ndof; k2++) s += ca.AIS[k1][r] * ca.psi[k1 * cb.ndof + k2] * ca.AIS[k2][cc];
          PI(r, cc) -= s;
        }
      a.AI = a.AI + transformInertia(inverse(kin[c].Trel), PI);
    }
    s_t proj[36];
    for (int k = 0; k < b.ndof; k++) a.AIS[k] = a.AI * b.S[k];
    for (int k1 = 0; k1 < b.ndof; k1++)
      for (int k2 = 0; k2 < b.ndof; k2++) proj[k1 * b.ndof + k2] = dot(b.S[k1], a.AIS[k2]);
    invertSmall(proj, b.ndof, a.psi);
  }
}

// Skeleton::computeForwardDynamics (Skeleton.cpp:13296-13314).
// Returns joint accelerations; optionally the body accelerations and transmitted forces.
inline void forwardDynamics(const Model& m, const std::vector<Kin>& kin, const std::vector<Art>& art, const s_t* q,
                            const s_t* v, const s_t* tau, s_t* qdd, std::vector<Vec6>* bodyAcc = nullptr) {
  std::vector<Vec6> bias(m.nb), acc(m.nb);
  std::vector<s_t> total(m.n > 0 ? m.n : 1);
  // leaf -> root: BodyNode::updateBiasForce (BodyNode.cpp:2076-2114)
  for (int i = m.nb - 1; i >= 0; i--) {
    const Body& b = m.bodies[i];
    Vec6 Fg = b.G * AdInvRLinear(kin[i].Tworld, m.gravity);
    Vec6 B = -dad(kin[i].V, b.G * kin[i].V) - Fg;  // no external force on this path
    for (int c : b.children) {
      const Body& cb = m.bodies[c];
      // GenericJoint::addChildBiasForceToDynamic (GenericJoint.hpp:2395-2421)
      Vec6 Spsiu = zero6();
      for (int k1 = 0; k1 < cb.ndof; k1++) {
        s_t pu = 0;
        for (int k2 = 0; k2 < cb.ndof; k2++) pu += art[c].psi[k1 * cb.ndof + k2] * total[cb.dofOff + k2];
        Spsiu = Spsiu + cb.S[k1] * pu;
      }
      Vec6 beta = bias[c] + art[c].AI * (kin[c].eta + Spsiu);
      B = B + dAdInvT(kin[c].Trel, beta);
    }
    bias[i] = B;
    // GenericJoint::updateTotalForceDynamic (GenericJoint.hpp:2554-2571)
    Vec6 bodyForce = art[i].AI * kin[i].eta + B;
    for (int k = 0; k < b.ndof; k++) {
      int d = b.dofOff + k;
      s_t springForce = -m.spring[d] * (q[d] - m.rest[d] + v[d] * m.dt);
      s_t dampingForce = -m.damping[d] * v[d];
      total[d] = tau[d] + springForce + dampingForce - dot(b.S[k], bodyForce);
    }
  }
  // root -> leaf: updateAccelerationFD (BodyNode.cpp:2159-2185; GenericJoint.hpp:2656-2676)
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    Vec6 parentAcc = (b.parent >= 0) ? acc[b.parent] : zero6();
    Vec6 Xa = AdInvT(kin[i].Trel, parentAcc);
    Vec6 AIXa = art[i].AI * Xa;
    s_t rhs[6];
    for (int k = 0; k < b.ndof; k++) rhs[k] = total[b.dofOff + k] - dot(b.S[k], AIXa);
    Vec6 A = Xa + kin[i].eta;
    for (int k1 = 0; k1 < b.ndof; k1++) {
      s_t a = 0;
      for (int k2 = 0; k2 < b.ndof; k2++) a += art[i].psi[k1 * b.ndof + k2] * rhs[k2];
      qdd[b.dofOff + k1] = a;
      A = A + b.S[k1] * a;
    }
    acc[i] = A;
  }
  if (bodyAcc) *bodyAcc = acc;
}

// Impulse-based forward dynamics with an arbitrary set of body impulses
// (BodyNode::updateBiasImpulse BodyNode.cpp:2117-2138, updateVelocityChangeFD :2188-2215;
//  GenericJoint.hpp:2482-2498, 2607-2613, 2713-2725).  impulses[i] is the constraint impulse on body i
// expressed in its own frame (BodyNode::mConstraintImpulse).  Returns delta joint velocities.
inline void impulseDynamics(const Model& m, const std::vector<Kin>& kin, const std::vector<Art>& art,
                            const std::vector<Vec6>& impulses, s_t* delV, std::vector<Vec6>* bodyDelV = nullptr,
                            const s_t* jointImpulses = nullptr) {   // jointImpulses: Joint::mConstraintImpulses (joint-limit rows)
  std::vector<Vec6> bias(m.nb), dV(m.nb);
  std::vector<s_t> total(m.n > 0 ? m.n : 1);
  for (int i = m.nb - 1; i >= 0; i--) {
    const Body& b = m.bodies[i];
    Vec6 B = -impulses[i];
    for (int c : b.children) {
      const Body& cb = m.bodies[c];
      // addChildBiasImpulseToDynamic: beta = childBias + AI * S * psi * totalImpulse
      Vec6 Spsiu = zero6();
      for (int k1 = 0; k1 < cb.ndof; k1++) {
        s_t pu = 0;
        for (int k2 = 0; k2 < cb.ndof; k2++) pu += art[c].psi[k1 * cb.ndof + k2] * total[cb.dofOff + k2];
        Spsiu = Spsiu + cb.S[k1] * pu;
      }
      Vec6 beta = bias[c] + art[c].AI * Spsiu;
      B = B + dAdInvT(kin[c].Trel, beta);
    }
    bias[i] = B;
    // GenericJoint::updateTotalImpulse: mTotalImpulse = mConstraintImpulses - S^T biasImpulse
    for (int k = 0; k < b.ndof; k++) total[b.dofOff + k] = (jointImpulses ? jointImpulses[b.dofOff + k] : 0.0) - dot(b.S[k], B);
  }
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    Vec6 parentDV = (b.parent >= 0) ? dV[b.parent] : zero6();
    Vec6 X = AdInvT(kin[i].Trel, parentDV);
    Vec6 AIX = art[i].AI * X;
    s_t rhs[6];
    for (int k = 0; k < b.ndof; k++) rhs[k] = total[b.dofOff + k] - dot(b.S[k], AIX);
    Vec6 D = X;
    for (int k1 = 0; k1 < b.ndof; k1++) {
      s_t a = 0;
      for (int k2 = 0; k2 < b.ndof; k2++) a += art[i].psi[k1 * b.ndof + k2] * rhs[k2];
      delV[b.dofOff + k1] = a;
      D = D + b.S[k1] * a;
    }
    dV[i] = D;
  }
  if (bodyDelV) *bodyDelV = dV;
}

// Recursive Newton-Euler inverse dynamics  tau = M(q) a + C(q, v)  (gravity optional).
// Used for: C+g  (Skeleton::updateCoriolisAndGravityForces Skeleton.cpp:12914-12943 with
// BodyNode::updateCombinedVector/aggregateCombinedVector BodyNode.cpp:2455-2506; a = 0),
// and M columns (Skeleton::updateMassMatrix Skeleton.cpp:12372-12434; v = 0, no gravity, a = e_j).
// Gravity enters as a body force exactly as in the reference (mFgravity).
inline void inverseDynamics(const Model& m, const std::vector<Kin>& kinQ /* Trel/Tworld only */, const s_t* v,
                            const s_t* a, bool withVelocity, bool withGravity, s_t* tauOut,
                            std::vector<Vec6>* Vout = nullptr, std::vector<Vec6>* Aout = nullptr,
                            std::vector<Vec6>* Fout = nullptr) {
  std::vector<Vec6> V(m.nb), A(m.nb), F(m.nb);
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    Vec6 Sdq = withVelocity ? jointTwist(b, v) : zero6();
    Vec6 Sddq = jointTwist(b, a);
    Vec6 Vp = (b.parent >= 0) ? AdInvT(kinQ[i].Trel, V[b.parent]) : zero6();
    Vec6 Ap = (b.parent >= 0) ? AdInvT(kinQ[i].Trel, A[b.parent]) : zero6();
    V[i] = Vp + Sdq;
    A[i] = Ap + ad(V[i], Sdq) + Sddq;
  }
  for (int i = m.nb - 1; i >= 0; i--) {
    const Body& b = m.bodies[i];
    Vec6 f = b.G * A[i] - dad(V[i], b.G * V[i]);
    if (withGravity) f = f - b.G * AdInvRLinear(kinQ[i].Tworld, m.gravity);
    for (int c : b.children) f = f + dAdInvT(kinQ[c].Trel, F[c]);
    F[i] = f;
    for (int k = 0; k < b.ndof; k++) tauOut[b.dofOff + k] = dot(b.S[k], f);
  }
  if (Vout) *Vout = V;
  if (Aout) *Aout = A;
  if (Fout) *Fout = F;
}

inline MatX massMatrix(const Model& m, const std::vector<Kin>& kin) {
  MatX M(m.n, m.n);
  VecX a(m.n, 0.0), col(m.n, 0.0), zero(m.n, 0.0);
  for (int j = 0; j < m.n; j++) {
    a.assign(m.n, 0.0);
    a[j] = 1.0;
    inverseDynamics(m, kin, zero.data(), a.data(), false, false, col.data());
    for (int i = 0; i < m.n; i++) M(i, j) = col[i];
  }
  return M;
}

// Skeleton::getInvMassMatrix.  The reference builds it column by column with unit-force ABA sweeps
// (Skeleton.cpp:12621-12654) or, when weld joints are present, as M.llt().solve(I) (:12588-12608).
// Here: unit-impulse ABA sweeps on joint space, i.e. the same recursion with generalized impulses.
inline MatX invMassMatrix(const Model& m, const std::vector<Kin>& kin, const std::vector<Art>& art) {
  (void)art;
  MatX M = massMatrix(m, kin), Minv;
  spdInverse(M, Minv);
  return Minv;
}

// Position-space relative Jacobian column(s) H(q): delta T_rel = T_rel * hat(H dq).
// Revolute/Prismatic: H = S.  Free: AdTJacFixed(T_cj, blkdiag(expMapJac(q)^T, expMapRot(q)^T))
// (FreeJoint.cpp:790-823, getRelativeJacobianInPositionSpaceStatic)
inline void positionJacobian(const Body& b, const s_t* q, Vec6 H[6]) {
  if (b.jtype != NBL_JOINT_FREE && b.jtype != NBL_JOINT_BALL) {
    for (int k = 0; k < b.ndof; k++) H[k] = b.S[k];
    return;
  }
  Vec3 r = mk3(q[b.dofOff], q[b.dofOff + 1], q[b.dofOff + 2]);
  Mat3 Jt = transpose(expMapJac(r));
  if (b.jtype == NBL_JOINT_BALL) {   // BallJoint.cpp:282-289: AdTJacFixed(T_cj, [expMapJac(q)^T; 0])
    for (int k = 0; k < 3; k++) H[k] = AdT(b.Tcj, mk6(mk3(Jt(0, k), Jt(1, k), Jt(2, k)), mk3(0, 0, 0)));
    return;
  }
  Mat3 Rt = transpose(expMapRot(r));
  for (int k = 0; k < 3; k++) {
    Vec6 colA = mk6(mk3(Jt(0, k), Jt(1, k), Jt(2, k)), mk3(0, 0, 0));
    Vec6 colL = mk6(mk3(0, 0, 0), mk3(Rt(0, k), Rt(1, k), Rt(2, k)));
    H[k] = AdT(b.Tcj, colA);
    H[3 + k] = AdT(b.Tcj, colL);
  }
}

// Directional derivative of inverse dynamics  tau = ID(q, v, a)  along (dq, dv) with a held fixed.
// This is the column recursion the reference implements in
//   BodyNode::computeJacobianOfCForward/Backward (BodyNode.cpp:3440-3597, 3832-3962)  [dC/dq, dC/dv]
//   BodyNode::computeJacobianOfMForward/Backward (BodyNode.cpp:2972-3037, 3111-3205)  [d(M x)/dq]
// restated as one tangent (forward-mode) sweep per column.  Gravity is carried as the equivalent
// base acceleration -g (identical value, simpler derivative).
struct IDNominal {
  std::vector<Vec6> V, A, F;
};
inline void idNominal(const Model& m, const std::vector<Kin>& kin, const s_t* v, const s_t* a, bool withVelocity,
                      bool withGravity, IDNominal& nom) {
  nom.V.resize(m.nb); nom.A.resize(m.nb); nom.F.resize(m.nb);
  Vec6 a0 = withGravity ? mk6(mk3(0, 0, 0), -m.gravity) : zero6();
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    Vec6 Sdq = withVelocity ? jointTwist(b, v) : zero6();
    Vec6 Vp = (b.parent >= 0) ? AdInvT(kin[i].Trel, nom.V[b.parent]) : zero6();
    Vec6 Ap = AdInvT(kin[i].Trel, (b.parent >= 0) ? nom.A[b.parent] : a0);
    nom.V[i] = Vp + Sdq;
    nom.A[i] = Ap + ad(nom.V[i], Sdq) + jointTwist(b, a);
  }
  for (int i = m.nb - 1; i >= 0; i--) {
    const Body& b = m.bodies[i];
    Vec6 f = b.G * nom.A[i] - dad(nom.V[i], b.G * nom.V[i]);
    for (int c : b.children) f = f + dAdInvT(kin[c].Trel, nom.F[c]);
    nom.F[i] = f;
  }
}
inline void idTangent(const Model& m, const std::vector<Kin>& kin, const IDNominal& nom, const s_t* q, const s_t* v,
                      bool withVelocity, bool withGravity, const s_t* dq, const s_t* dv, s_t* dtau) {
  std::vector<Vec6> dV(m.nb), dA(m.nb), dF(m.nb), h(m.nb);
  Vec6 a0 = withGravity ? mk6(mk3(0, 0, 0), -m.gravity) : zero6();
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    Vec6 H[6];
    positionJacobian(b, q, H);
    Vec6 hi = zero6();
    for (int k = 0; k < b.ndof; k++) hi = hi + H[k] * dq[b.dofOff + k];
    h[i] = hi;
    Vec6 Sdq = withVelocity ? jointTwist(b, v) : zero6();
    Vec6 Sddv = jointTwist(b, dv);
    Vec6 XVp = (b.parent >= 0) ? AdInvT(kin[i].Trel, nom.V[b.parent]) : zero6();
    Vec6 XAp = AdInvT(kin[i].Trel, (b.parent >= 0) ? nom.A[b.parent] : a0);
    Vec6 XdVp = (b.parent >= 0) ? AdInvT(kin[i].Trel, dV[b.parent]) : zero6();
    Vec6 XdAp = (b.parent >= 0) ? AdInvT(kin[i].Trel, dA[b.parent]) : zero6();
    dV[i] = XdVp - ad(hi, XVp) + Sddv;
    dA[i] = XdAp - ad(hi, XAp) + ad(dV[i], Sdq) + ad(nom.V[i], Sddv);
  }
  for (int i = m.nb - 1; i >= 0; i--) {
    const Body& b = m.bodies[i];
    Vec6 GV = b.G * nom.V[i];
    Vec6 df = b.G * dA[i] - dad(dV[i], GV) - dad(nom.V[i], b.G * dV[i]);
    for (int c : b.children) {
      // d(X_c^T F_c) = X_c^T (dF_c - ad^T(h_c) F_c)
      df = df + dAdInvT(kin[c].Trel, dF[c] - dad(h[c], nom.F[c]));
    }
    dF[i] = df;
    for (int k = 0; k < b.ndof; k++) dtau[b.dofOff + k] = dot(b.S[k], df);
  }
}

// Skeleton::getJacobianOfC(wrt) (Skeleton.cpp:1780-1830), getVelCJacobian (:2264-2269)
inline MatX jacobianOfC(const Model& m, const std::vector<Kin>& kin, const s_t* q, const s_t* v, bool wrtVelocity) {
  VecX zero(m.n, 0.0), dir(m.n, 0.0), col(m.n, 0.0);
  IDNominal nom;
  idNominal(m, kin, v, zero.data(), true, true, nom);
  MatX J(m.n, m.n);
  for (int j = 0; j < m.n; j++) {
    dir.assign(m.n, 0.0);
    dir[j] = 1.0;
    if (wrtVelocity) idTangent(m, kin, nom, q, v, true, true, zero.data(), dir.data(), col.data());
    else idTangent(m, kin, nom, q, v, true, true, dir.data(), zero.data(), col.data());
    for (int i = 0; i < m.n; i++) J(i, j) = col[i];
  }
  return J;
}
// Skeleton::getJacobianOfM(x, POSITION) (Skeleton.cpp:1833-1882): d(M(q) x)/dq
inline MatX jacobianOfMx(const Model& m, const std::vector<Kin>& kin, const s_t* q, const s_t* x) {
  VecX zero(m.n, 0.0), dir(m.n, 0.0), col(m.n, 0.0);
  IDNominal nom;
  idNominal(m, kin, zero.data(), x, false, false, nom);
  MatX J(m.n, m.n);
  for (int j = 0; j < m.n; j++) {
    dir.assign(m.n, 0.0);
    dir[j] = 1.0;
    idTangent(m, kin, nom, q, zero.data(), false, false, dir.data(), zero.data(), col.data());
    for (int i = 0; i < m.n; i++) J(i, j) = col[i];
  }
  return J;
}

// Position integration per joint: R^n Euler (ConfigurationSpace.hpp:145-180) and
// FreeJoint::integratePositionsExplicit (FreeJoint.cpp:922-929, identity-Jacobian branch):
//   Qnext = convertToTransform(pos) * convertToTransform(vel*dt); pos' = [logMap(R); p]
inline void integratePositions(const Model& m, const s_t* q, const s_t* v, s_t dt, s_t* qn) {
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    if (b.jtype == NBL_JOINT_FREE) {
      int o = b.dofOff;
      Iso Q, D;
      Q.R = expMapRot(mk3(q[o], q[o + 1], q[o + 2]));
      Q.p = mk3(q[o + 3], q[o + 4], q[o + 5]);
      D.R = expMapRot(mk3(v[o] * dt, v[o + 1] * dt, v[o + 2] * dt));
      D.p = mk3(v[o + 3] * dt, v[o + 4] * dt, v[o + 5] * dt);
      Iso N = Q * D;
      Vec3 r = logMap(N.R);
      for (int k = 0; k < 3; k++) { qn[o + k] = r[k]; qn[o + 3 + k] = N.p[k]; }
    } else if (b.jtype == NBL_JOINT_BALL) {
      // BallJoint::integratePositionsExplicit (BallJoint.cpp:333-349, identity-Jacobian branch): Rnext = R(q) R(dq dt)
      int o = b.dofOff;
      Vec3 r = logMap(expMapRot(mk3(q[o], q[o + 1], q[o + 2])) * expMapRot(mk3(v[o] * dt, v[o + 1] * dt, v[o + 2] * dt)));
      for (int k = 0; k < 3; k++) qn[o + k] = r[k];
    } else {
      for (int k = 0; k < b.ndof; k++) qn[b.dofOff + k] = q[b.dofOff + k] + v[b.dofOff + k] * dt;
    }
  }
}

// World::getPosPosJacobian / getVelPosJacobian (World.cpp:2415-2446): block diagonal per joint; identity
// resp. dt*I for R^n joints (GenericJoint.hpp:1428-1444), central finite differences for the free
// joint (FreeJoint.cpp:950-1007: eps 1e-6 for pos, 1e-7 for vel) — restated literally, FD included.
// ---- TEST INSTRUMENT (Model::exactPosJacobians; NOT the reference's behaviour): the same two Jacobians by differentiating the very
// formulas integratePositions composes - expMapRot (Geometry.cpp:539, with its Taylor branch below 1e-3) and logMap (Geometry.cpp:720,
// with its small-angle branch) - in FORWARD mode, entry by entry, in EXTENDED precision (long double: 64-bit mantissa on x86).
// Two things go wrong next to the log-map singularity (next rotation angle within `gap` of pi), and the instrument avoids both:
//   * the reference's central differences (eps 1e-6) divide the rounding error of logMap, ~1e-16 / gap^2, by 1e-6: off by ~5e-9 / gap^2;
//   * ANY evaluation of the analytic derivative in doubles cancels two O(1 / gap) terms (alpha' dtheta vee(R - R^T) against
//     alpha vee(dR - dR^T)) into an O(1) result: good to ~eps / gap^3 (this formula in doubles: 3e-7 at gap = 2e-3 against an 80-bit
//     five-point stencil; the device's reverse mode of the same formulas: 1.7e-7) - hence the extended precision here (1e-10 there).
// (The closed form J_r^-1(r') E^T J_r(r) of the IDEAL exponential and logarithm is well conditioned but is not the derivative of THIS
// function: the Taylor branch of expMapRot is not exactly orthogonal, logMap turns that 1e-11 into 1e-9 of angle, which changes with q:
// 1.5e-6 apart at gap = 2e-3, measured in 80-bit arithmetic on both sides.)
// tests/test_oracle_exact_pos_jacobians.py pins it against that stencil; tests/test_gpu_contact.py::test_cfg4_box_stack_8192_worlds
// uses it to PROVE that the worlds of cfg4 the device "misses" next to the singularity are the finite differences' error: with the
// switch on they agree with the device, with it off the reported errors reappear (tests/parity.py::gradient_tolerance cites both).
template <class X>
struct xpT {
struct M3 { X m[9]; X& operator()(int r, int c) { return m[3 * r + c]; } const X& operator()(int r, int c) const { return m[3 * r + c]; } };
struct V3x { X v[3]; };
static inline M3 mul(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j); return r; }
static inline M3 add(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
static inline M3 scale(X s, const M3& a) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = s * a.m[i]; return r; }
static inline M3 eye() { M3 r; for (int i = 0; i < 9; i++) r.m[i] = (i % 4 == 0) ? X(1.0L) : X(0.0L); return r; }
static inline M3 skew(const V3x& a) { M3 r; r.m[0] = 0; r.m[1] = -a.v[2]; r.m[2] = a.v[1]; r.m[3] = a.v[2]; r.m[4] = 0; r.m[5] = -a.v[0]; r.m[6] = -a.v[1]; r.m[7] = a.v[0]; r.m[8] = 0; return r; }
static inline X nrm(const V3x& a) { return std::sqrt(a.v[0] * a.v[0] + a.v[1] * a.v[1] + a.v[2] * a.v[2]); }
static inline M3 expMapRot(const V3x& q) {                              // Geometry.cpp:539-553, same branches
  const X theta = nrm(q);
  const M3 Q = skew(q), Q2 = mul(Q, Q);
  if (theta < X(1.0e-3L)) return add(add(eye(), Q), scale(X(0.5L), Q2));
  return add(add(eye(), scale(std::sin(theta) / theta, Q)), scale((1 - std::cos(theta)) / (theta * theta), Q2));
}
static inline M3 dExpMapRot(const V3x& q, const V3x& dq) {              // d expMapRot(q)[dq]
  const X theta = nrm(q);
  const M3 Q = skew(q), D = skew(dq);
  const M3 QD = add(mul(Q, D), mul(D, Q));
  if (theta < X(1.0e-3L)) return add(D, scale(X(0.5L), QD));           // the branch expMapRot takes there: I + Q + Q^2 / 2
  const X sn = std::sin(theta), cs = std::cos(theta);
  const X a = sn / theta, b = (1 - cs) / (theta * theta);
  const X dth = (q.v[0] * dq.v[0] + q.v[1] * dq.v[1] + q.v[2] * dq.v[2]) / theta;
  const X da = (cs / theta - sn / (theta * theta)) * dth;
  const X db = (sn / (theta * theta) - 2 * (1 - cs) / (theta * theta * theta)) * dth;
  return add(add(scale(da, Q), scale(a, D)), add(scale(db, mul(Q, Q)), scale(b, QD)));
}
static inline V3x dLogMap(const M3& R, const M3& dR) {                  // d logMap(R)[dR] on the branches logMap takes away from pi (Geometry.cpp:720-745)
  const X DART_EPSILON = 1e-6L;
  X c = X(0.5L) * (R(0, 0) + R(1, 1) + R(2, 2) - X(1.0L));
  c = c > X(1.0L) ? X(1.0L) : (c < -X(1.0L) ? -X(1.0L) : c);
  const X theta = std::acos(c);
  const X dc = X(0.5L) * (dR(0, 0) + dR(1, 1) + dR(2, 2));
  const X w[3] = {R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1)};
  const X dw[3] = {dR(2, 1) - dR(1, 2), dR(0, 2) - dR(2, 0), dR(1, 0) - dR(0, 1)};
  X alpha, dalpha;
  if (theta > DART_EPSILON) {
    const X sn = std::sin(theta), cs = std::cos(theta);
    alpha = X(0.5L) * theta / sn;
    dalpha = X(0.5L) * (sn - theta * cs) / (sn * sn) * (-dc / sn);  // theta = acos(c)
  } else {
    alpha = X(0.5L) + (X(1.0L) / X(12.0L)) * theta * theta;
    dalpha = -(X(1.0L) / X(6.0L)) * dc;                                // theta^2 ~ 2 (1 - c)
  }
  V3x o;
  for (int k = 0; k < 3; k++) o.v[k] = dalpha * w[k] + alpha * dw[k];
  return o;
}
};
template <class X>
inline void posJacobiansExactT(const Model& m, const s_t* q, const s_t* v, s_t dt, MatX& posPos, MatX& velPos) {
  typedef xpT<X> xp;
  posPos = identityX(m.n);
  velPos = MatX(m.n, m.n);
  for (int i = 0; i < m.n; i++) velPos(i, i) = dt;
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    if (b.jtype != NBL_JOINT_FREE && b.jtype != NBL_JOINT_BALL) continue;
    const int o = b.dofOff;
    const bool freeJ = b.jtype == NBL_JOINT_FREE;
    const X dtx = dt;
    const typename xp::V3x r = {{q[o], q[o + 1], q[o + 2]}}, wdt = {{v[o] * dtx, v[o + 1] * dtx, v[o + 2] * dtx}};
    const typename xp::M3 R = xp::expMapRot(r), E = xp::expMapRot(wdt), Rn = xp::mul(R, E);
    const X u[3] = {freeJ ? v[o + 3] * dtx : X(0.0L), freeJ ? v[o + 4] * dtx : X(0.0L), freeJ ? v[o + 5] * dtx : X(0.0L)};
    for (int j = 0; j < 3; j++) {
      typename xp::V3x e = {{0, 0, 0}}, edt = {{0, 0, 0}};
      e.v[j] = X(1.0L); edt.v[j] = dtx;
      const typename xp::M3 dRq = xp::dExpMapRot(r, e);                   // d R / d r_j
      const typename xp::V3x drq = xp::dLogMap(Rn, xp::mul(dRq, E));
      const typename xp::M3 dEw = xp::dExpMapRot(wdt, edt);               // d E / d w_j
      const typename xp::V3x drw = xp::dLogMap(Rn, xp::mul(R, dEw));
      for (int k = 0; k < 3; k++) { posPos(o + k, o + j) = (s_t)drq.v[k]; velPos(o + k, o + j) = (s_t)drw.v[k]; }
      if (freeJ) {
        // p' = p + R(r) (v_lin dt)  (Qnext = Q * D, FreeJoint.cpp:922-929)
        for (int k = 0; k < 3; k++) {
          posPos(o + 3 + k, o + j) = (s_t)(dRq(k, 0) * u[0] + dRq(k, 1) * u[1] + dRq(k, 2) * u[2]);   // d p' / d r_j
          posPos(o + k, o + 3 + j) = 0.0;                        // d r' / d p_j
          posPos(o + 3 + k, o + 3 + j) = k == j ? 1.0 : 0.0;     // d p' / d p_j
          velPos(o + 3 + k, o + j) = 0.0;                        // d p' / d w_j
          velPos(o + k, o + 3 + j) = 0.0;                        // d r' / d v_lin_j
          velPos(o + 3 + k, o + 3 + j) = (s_t)(R(k, j) * dtx);   // d p' / d v_lin_j
        }
      }
    }
  }
}
inline void posJacobians(const Model& m, const s_t* q, const s_t* v, s_t dt, MatX& posPos, MatX& velPos) {
  if (m.exactPosJacobians == 1) { posJacobiansExactT<long double>(m, q, v, dt, posPos, velPos); return; }
  if (m.exactPosJacobians == 2) { posJacobiansExactT<double>(m, q, v, dt, posPos, velPos); return; }   // the same formulas in doubles: what ANY double-precision analytic derivative can reach (eps / gap^3)
  posPos = identityX(m.n);
  velPos = MatX(m.n, m.n);
  for (int i = 0; i < m.n; i++) velPos(i, i) = dt;
  VecX qa(q, q + m.n), va(v, v + m.n), plus(m.n), minus(m.n);
  for (int i = 0; i < m.nb; i++) {
    const Body& b = m.bodies[i];
    if (b.jtype != NBL_JOINT_FREE && b.jtype != NBL_JOINT_BALL) continue;   // BallJoint.cpp:351-408: the same finite differences
    int o = b.dofOff;
    const int nd = b.ndof;
    for (int j = 0; j < nd; j++) {
      s_t EPS = 1e-6;
      VecX pert = qa;
      pert[o + j] += EPS;
      integratePositions(m, pert.data(), v, dt, plus.data());
      pert = qa;
      pert[o + j] -= EPS;
      integratePositions(m, pert.data(), v, dt, minus.data());
      for (int r = 0; r < nd; r++) posPos(o + r, o + j) = (plus[o + r] - minus[o + r]) / (2 * EPS);
      EPS = 1e-7;
      pert = va;
      pert[o + j] += EPS;
      integratePositions(m, q, pert.data(), dt, plus.data());
      pert = va;
      pert[o + j] -= EPS;
      integratePositions(m, q, pert.data(), dt, minus.data());
      for (int r = 0; r < nd; r++) velPos(o + r, o + j) = (plus[o + r] - minus[o + r]) / (2 * EPS);
    }
  }
}

}  // namespace nbo
