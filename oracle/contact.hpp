// oracle/contact.hpp — TEST INFRASTRUCTURE ONLY.
//
// Contact stage of World::step (ConstraintSolver::solve, ConstraintSolver.cpp:376-414) and the
// contact terms of the BackpropSnapshot Jacobians.  STAGE 1 of the build: no collision shapes are
// processed yet, so every world takes the "no clamping constraints" branches of the reference
// (BackpropSnapshot.cpp:521-524, 686-689).
#pragma once
#include "dynamics.hpp"

namespace nbo {

struct Contact {
  Vec3 point, normal;
  s_t depth;
  int type, bodyA, bodyB, boxA, boxB;
};

struct ContactResult {
  std::vector<Contact> contacts;
  int m = 0;
  MatX A;
  VecX b, x, lo, hi;
  std::vector<int> findex, rowClass;
  int numClamping = 0;
};

inline void solveContacts(const Model& m, const std::vector<Kin>& kin, const std::vector<Art>& art, const s_t* q,
                          const s_t* vPre, VecX& lcpCache, ContactResult& out, s_t* vOut, uint32_t* status) {
  (void)m; (void)kin; (void)art; (void)q; (void)vPre; (void)lcpCache; (void)vOut;
  out = ContactResult();
  *status = 0;
}

inline void contactJacobians(const Model& m, const std::vector<Kin>& kin, const std::vector<Art>& art, const s_t* q,
                             const s_t* v, const s_t* tau, const ContactResult& cr, const MatX& Minv, const VecX& C,
                             const MatX& dCdq, const MatX& dCdv, MatX& forceVel, MatX& velVel, MatX& posVel) {
  (void)m; (void)kin; (void)art; (void)q; (void)v; (void)tau; (void)cr; (void)Minv; (void)C; (void)dCdq; (void)dCdv;
  (void)forceVel; (void)velVel; (void)posVel;
}

}  // namespace nbo
