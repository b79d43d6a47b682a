// oracle/contact.hpp — TEST INFRASTRUCTURE ONLY.
//
// Contact stage of World::step restated from the reference:
//   ConstraintSolver::updateConstraints            dart/constraint/ConstraintSolver.cpp:540-613
//   ContactConstraint ctor / getInformation        dart/constraint/ContactConstraint.cpp:66-230, 361-514, 734-795
//   BoxedLcpConstraintSolver::buildLcpInputs       dart/constraint/BoxedLcpConstraintSolver.cpp:190-349
//   BoxedLcpConstraintSolver::solveLcp (cascade)   :352-789
//   ConstrainedGroupGradientMatrices               dart/neural/ConstrainedGroupGradientMatrices.cpp:177-339, 482-872
//   DifferentiableContactConstraint::getConstraintForces  dart/neural/DifferentiableContactConstraint.cpp:231-270
// One constrained group per world (all contacts solved as one LCP): true for every config in
// BASELINE.json; multi-group worlds (ConstraintSolver.cpp:742-786) are outside the scope.
#pragma once
#include "collision.hpp"
#include "dynamics.hpp"
#include "lcp.hpp"

namespace nbo {

enum RowClass { RC_NOT_CLAMPING = 0, RC_CLAMPING = 1, RC_UPPER_BOUND = 2, RC_ILLEGAL = 3 };

struct ContactResult {
  std::vector<Contact> contacts;   // active contact constraints, in LCP order
  std::vector<int> rowContact, rowDir;  // LCP row -> (contact, 0 = normal / 1,2 = tangents); rowContact = -1: a joint-limit row
  std::vector<int> rowDof;              // joint-limit rows (JointLimitConstraint): the DOF, -1 on contact rows
  std::vector<int> contactOffset;  // first row of each contact
  std::vector<Vec6> JA, JB;        // per row: ContactConstraint::mSpatialNormalA/B columns (body frames)
  std::vector<Vec3> dirs;          // per row: world force direction
  int m = 0;
  MatX A;                          // m x m (possibly with the fallback CFM on the diagonal)
  VecX b, x, lo, hi;
  std::vector<int> findex;
  MatX massed;                     // n x m: massed impulse tests  M^-1 J^T        (CGGM::measureConstraintImpulse)
  MatX Aall;                       // n x m: constraint forces in joint space      (DCC::getConstraintForces)
  // classification (CGGM::constructMatrices)
  std::vector<int> rowClass, clampingIndex, upperBoundIndex;
  int numClamping = 0, numUpperBound = 0;
  MatX E;                          // numUpperBound x numClamping
  VecX fc;                         // clamping constraint impulses
  s_t cfm = 0;                     // (single-group worlds; the backward pass reads cfmRow)
  VecX cfmRow;                     // per LCP row: the constraint-force-mixing constant its constrained group ended with (CFM_CONSTANTS)
  std::vector<int> rowGroup;       // per LCP row: index of its constrained group
  int numGroups = 0;
  VecX restCoeff;                  // per LCP row: ContactConstraint::getCoefficientOfRestitution (e if the contact bounced, else 0; 0 on friction rows)
  bool ignoreFriction = false, standardized = false;
  uint32_t status = 0;
};

// ContactConstraint::getTangentBasisMatrixODE (ContactConstraint.cpp:734-795), first frictional direction = Z
inline void tangentBasis(const Vec3& n, Vec3& t1, Vec3& t2) {
  const s_t EPS2 = 1e-12;
  Vec3 tangent = cross(mk3(0, 0, 1), n);
  if (dot(tangent, tangent) < EPS2) {
    tangent = cross(mk3(1, 0, 0), n);
    if (dot(tangent, tangent) < EPS2) {
      tangent = cross(mk3(0, 1, 0), n);
      if (dot(tangent, tangent) < EPS2) tangent = cross(mk3(0, 0, 1), n);
    }
  }
  t1 = normalized(tangent);
  t2 = cross(n, t1);
}

inline bool dofIsParentOf(const Model& m, int dofBody, int body) {
  while (body >= 0) {
    if (body == dofBody) return true;
    body = m.bodies[body].parent;
  }
  return false;
}

// ---- CGGM::constructMatrices + opportunisticallyStandardizeResults ------------------------------
struct Cggm {
  const Model* model;
  const MatX* Minv;
  ContactResult* cr;
  // registered LCP results
  VecX X, Hi, Lo, B, AColNorms;
  std::vector<int> FIndex;
  MatX A;
  s_t cfmConst = 0;
  bool ignoreFriction = false;
  bool standardized = false;

  bool isSolutionValid(const VecX& x) const { return isLCPSolutionValid(A, x, B, Hi, Lo, FIndex, ignoreFriction); }

  void constructMatrices() {
    const s_t CLAMPING_THRESHOLD = 1e-6;
    const int m = cr->m;
    std::vector<int>& cls = cr->rowClass;
    cls.assign(m, RC_NOT_CLAMPING);
    cr->clampingIndex.assign(m, -1);
    cr->upperBoundIndex.assign(m, -1);
    int numClamping = 0, numUpperBound = 0;
    for (int j = 0; j < m; j++) {
      if (AColNorms[j] < 1e-9) { cls[j] = RC_NOT_CLAMPING; continue; }
      const s_t force = X[j];
      s_t upperBound = Hi[j], lowerBound = Lo[j];
      const int fp = FIndex[j];
      if (fp != -1) { upperBound *= X[fp]; lowerBound *= X[fp]; }
      if (std::fabs(force) < CLAMPING_THRESHOLD) {
        if (fp != -1) {
          s_t normalForce = X[fp];
          if (std::fabs(normalForce) < CLAMPING_THRESHOLD) cls[j] = RC_NOT_CLAMPING;
          else if (ignoreFriction) cls[j] = RC_NOT_CLAMPING;
          else { cls[j] = RC_CLAMPING; cr->clampingIndex[j] = numClamping++; }
        } else cls[j] = RC_NOT_CLAMPING;
        continue;
      }
      const s_t tieBreak = 1e-5;
      if ((X[j] > lowerBound + tieBreak && X[j] < upperBound - tieBreak) ||
          (lowerBound - X[j] > 1e-2 || X[j] - upperBound > 1e-2)) {
        cls[j] = RC_CLAMPING;
        cr->clampingIndex[j] = numClamping++;
      } else if (lowerBound - X[j] > 1e-2 || X[j] - upperBound > 1e-2) {
        cls[j] = RC_ILLEGAL;
      } else if (fp != -1 && std::fabs(X[fp]) > 1e-9 && AColNorms[fp] > 1e-9 && ((fp > j) || cls[fp] == RC_CLAMPING)) {
        cls[j] = RC_UPPER_BOUND;
        cr->upperBoundIndex[j] = numUpperBound++;
      } else cls[j] = RC_NOT_CLAMPING;
    }
    cr->numClamping = numClamping;
    cr->numUpperBound = numUpperBound;
    cr->E = MatX(numUpperBound, numClamping);
    cr->fc.assign(numClamping, 0.0);
    for (int j = 0; j < m; j++) {
      if (cls[j] == RC_CLAMPING) cr->fc[cr->clampingIndex[j]] = X[j];
      if (cls[j] == RC_UPPER_BOUND) {
        const int fp = FIndex[j];
        const s_t ub = X[fp] * Hi[j], lb = X[fp] * Lo[j];
        if (std::fabs(X[j] - ub) < std::fabs(X[j] - lb)) cr->E(cr->upperBoundIndex[j], cr->clampingIndex[fp]) = Hi[j];
        else cr->E(cr->upperBoundIndex[j], cr->clampingIndex[fp]) = Lo[j];
      }
    }
    cr->cfm = cfmConst;
    cr->ignoreFriction = ignoreFriction;
    standardize();
  }

  // Q = A_c^T Minv (A_c + A_ub E) + cfm I   (or the clamping block of A when there are no upper-bound rows)
  MatX buildQ() const {
    const int m = cr->m, nc = cr->numClamping, nu = cr->numUpperBound, n = model->n;
    MatX Q(nc, nc);
    if (nu == 0) {
      for (int r = 0; r < m; r++)
        if (cr->rowClass[r] == RC_CLAMPING)
          for (int c = 0; c < m; c++)
            if (cr->rowClass[c] == RC_CLAMPING) Q(cr->clampingIndex[r], cr->clampingIndex[c]) = A(r, c);
      return Q;
    }
    MatX Ac(n, nc), Aub(n, nu);
    for (int j = 0; j < m; j++) {
      if (cr->rowClass[j] == RC_CLAMPING) for (int i = 0; i < n; i++) Ac(i, cr->clampingIndex[j]) = cr->Aall(i, j);
      if (cr->rowClass[j] == RC_UPPER_BOUND) for (int i = 0; i < n; i++) Aub(i, cr->upperBoundIndex[j]) = cr->Aall(i, j);
    }
    MatX AcubE = matmul(Aub, cr->E);
    for (size_t i = 0; i < AcubE.d.size(); i++) AcubE.d[i] += Ac.d[i];
    Q = matmul(transposeX(Ac), matmul(*Minv, AcubE));
    for (int i = 0; i < nc; i++) Q(i, i) += cfmConst;
    return Q;
  }

  bool standardize() {
    const s_t CLAMPING_THRESHOLD = 1e-6;
    const int m = cr->m;
    standardized = true;
    if (m == 0) return true;
    if (cr->numClamping == 0) {
      VecX zero(m, 0.0);
      if (isSolutionValid(zero)) { X = zero; return true; }
      standardized = false;
      return false;
    }
    MatX Q = buildQ();
    VecX bc(cr->numClamping, 0.0);
    for (int j = 0; j < m; j++) if (cr->rowClass[j] == RC_CLAMPING) bc[cr->clampingIndex[j]] = B[j];
    VecX f_c = codSolve(Q, bc);
    VecX originalFc = cr->fc;
    bool anyNewlyNotClamping = false;
    VecX newX(m, 0.0);
    for (int i = 0; i < m; i++) {
      const int ci = cr->clampingIndex[i], ui = cr->upperBoundIndex[i];
      if (ci != -1) {
        newX[i] = f_c[ci];
        if (std::fabs(f_c[ci]) < CLAMPING_THRESHOLD && std::fabs(X[i]) > CLAMPING_THRESHOLD && FIndex[i] == -1) anyNewlyNotClamping = true;
      }
      if (ui != -1) {
        const int fp = FIndex[i];
        s_t originalMultiple = originalFc[cr->clampingIndex[fp]] / X[i];
        s_t cleanMultiple = (std::fabs(originalMultiple - Hi[i]) < std::fabs(originalMultiple - Lo[i])) ? Hi[i] : Lo[i];
        newX[i] = f_c[cr->clampingIndex[fp]] * cleanMultiple;
      }
    }
    if (isSolutionValid(newX)) {
      X = newX;
      cr->fc = f_c;
      if (anyNewlyNotClamping) constructMatrices();
      return true;
    }
    standardized = false;
    return false;
  }
};

// Stages 1-3 of BoxedLcpConstraintSolver::solveLcp (:461-687) on a boxed LCP given as plain arrays: reduce + Dantzig with early
// termination -> (on failure) CFM on the diagonal + reduce + PGS from the pre-solve x -> (on failure) friction dropped + PGS from
// zero, with the reference's validity checks between the stages.  xBackup: the pre-solve x (mXBackup).  Out: X, the CFM that
// ended up on the diagonal (0 when stage 1 succeeded), whether friction was dropped, NBL_ST_* bits.
inline void lcpCascade(const MatX& A, const VecX& b, const VecX& lo, const VecX& hi, const std::vector<int>& findex, const VecX& xBackup,
                       s_t fallbackCfm, VecX& X, s_t& cfm, bool& hadToIgnoreFriction, uint32_t& status) {
  const int mrows = (int)b.size();
  bool success = false;
  cfm = 0.0; hadToIgnoreFriction = false; status = 0;
  X = xBackup;
  // ---- stage 1: reduce + Dantzig with early termination (:461-522) ----
  {
    LcpProblem p;
    p.A = A; p.x = X; p.b = b; p.hi = hi; p.lo = lo; p.findex = findex;
    MatX mapOut = reduceLcp(p);
    int ok = dantzigSolve(p, true);
    if (ok == 1) {
      VecX xr = matvec(mapOut, p.x);
      X = xr;
      success = isLCPSolutionValid(A, X, b, hi, lo, findex, false);
      if (success) status |= NBL_ST_LCP_PIVOT;
    } else if (ok == -1) {
      // oracle/_ref not available: behave as a Dantzig failure (goes on to the PGS fallback) and say so
      status |= 0x80000000u;
    }  // on failure mX keeps its pre-solve value (:489 `if (success) mX = mapOut * mXReduced`)
  }
  bool nan = false;
  for (s_t v : X) if (std::isnan(v)) nan = true;
  if (nan) { success = false; X.assign(mrows, 0.0); status |= NBL_ST_NAN; }

  MatX ABackup = A;
  if (!success) {
    cfm = fallbackCfm;
    for (int i = 0; i < mrows; i++) ABackup(i, i) += cfm;
  }
  // ---- stage 2: CFM + PGS (:539-597) ----
  if (!success) {
    LcpProblem p;
    p.A = ABackup; p.x = xBackup; p.b = b; p.hi = hi; p.lo = lo; p.findex = findex;
    MatX mapOut = reduceLcp(p);
    success = pgsSolve(p);
    if (success) {
      X = matvec(mapOut, p.x);
      if (!isLCPSolutionValid(ABackup, X, b, hi, lo, findex, false)) success = false;
      else status |= NBL_ST_LCP_PGS;
    }
  }
  // ---- stage 3: drop friction, PGS again (:606-677) ----
  if (!success) {
    hadToIgnoreFriction = true;
    LcpProblem p;
    p.A = ABackup; p.x = xBackup; p.b = b; p.hi = hi; p.lo = lo; p.findex = findex;
    MatX mapOut = removeFrictionLcp(p);
    p.x.assign(p.x.size(), 0.0);
    success = pgsSolve(p);
    X = matvec(mapOut, p.x);
    status |= NBL_ST_LCP_NOFRIC;
    if (!success) status |= NBL_ST_LCP_FAILED;
  }
  nan = false;
  for (s_t v : X) if (std::isnan(v)) nan = true;
  if (nan) { X.assign(mrows, 0.0); status |= NBL_ST_NAN; }
}

inline void solveContacts(const Model& m, const std::vector<Kin>& kin, const std::vector<Art>& art, const s_t* q,
                          const s_t* vPre, VecX& lcpCache, ContactResult& out, s_t* vOut, uint32_t* status) {
  out = ContactResult();
  *status = 0;
  const int n = m.n;
  // ---- joint-limit constraints (ConstraintSolver.cpp:641-696, JointLimitConstraint::update :182-237): a DOF of a joint that
  //      enforces its limits, at or below its lower / at or above its upper limit, is one LCP row after the contact rows ----
  std::vector<int> limDof, limSide;                              // side: -1 lower limit active, +1 upper
  for (int d = 0; d < n; d++) {
    if (!m.limitEnforced[d]) continue;
    if (q[d] - m.posLo[d] <= 0.0) { limDof.push_back(d); limSide.push_back(-1); }
    else if (q[d] - m.posHi[d] >= 0.0) { limDof.push_back(d); limSide.push_back(+1); }
  }
  const int L = (int)limDof.size();
  if (m.boxes.empty() && L == 0) return;
  // ---- collision detection at q_t, filter by penetration depth (ConstraintSolver.cpp:563-613) ----
  std::vector<Contact> all;
  bool seenListFull = false;
  collideAll(m, kin, all, &seenListFull);
  const size_t seenCap = (size_t)deviceSeenPoints(m);
  for (size_t ci = 0; ci < all.size(); ci++) {
    const Contact& c = all[ci];
    if (dot(c.normal, c.normal) < 1e-12) continue;           // Contact::isZeroNormal
    if (c.depth < 0.0 || c.depth > m.clippingDepth) continue;
    if (c.bodyA < 0 && c.bodyB < 0) continue;                 // neither body reactive -> constraint inactive
    // (the device's duplicate filter remembers 16 or 32 distinct points per world - collision.hpp deviceSeenPoints -, kept or dropped by
    //  the depth filter: a contact kept after that flags the world, see below)
    if (ci >= seenCap) seenListFull = true;
    out.contacts.push_back(c);
  }
  const int C = (int)out.contacts.size();
  if (m.maxContacts > 0 && seenListFull) *status |= 0x80u;      // NBL_ST_CONTACT_OVERFLOW, see above
  if (C == 0 && L == 0) return;
  if (C > 0) *status |= NBL_ST_CONTACT;
  if (L > 0) *status |= NBL_ST_JOINT_LIMIT;
  // The reference has no contact cap; the device path keeps max_contacts and flags the world.  The oracle solves with all of
  // them (like the reference) and raises the same flag, so that a comparison knows which worlds the device truncated.
  // (a joint-limit row takes one contact slot of the device)
  if (m.maxContacts > 0 && C + L > m.maxContacts) *status |= 0x80u;   // NBL_ST_CONTACT_OVERFLOW


  // body velocities at the post-ABA, pre-contact velocity (ContactConstraint::getRelVelocity)
  std::vector<Kin> kinPre;
  kinematics(m, q, vPre, kinPre);

  // ---- ContactConstraint ctor: per-row body-frame Jacobians ----
  std::vector<s_t> mu(C);
  std::vector<int> dim(C);
  for (int c = 0; c < C; c++) {
    const Contact& ct = out.contacts[c];
    s_t muA = m.boxes[ct.boxA].mu, muB = m.boxes[ct.boxB].mu;
    mu[c] = muA < muB ? muA : muB;
    dim[c] = mu[c] > 1e-3 ? 3 : 1;                             // DART_FRICTION_COEFF_THRESHOLD
    out.contactOffset.push_back((int)out.rowContact.size());
    Vec3 t1, t2;
    tangentBasis(ct.normal, t1, t2);
    Vec3 d[3] = {ct.normal, t1, t2};
    for (int k = 0; k < dim[c]; k++) {
      out.rowContact.push_back(c);
      out.rowDir.push_back(k);
      out.dirs.push_back(d[k]);
      Vec6 ja = zero6(), jb = zero6();
      if (ct.bodyA >= 0) {
        const Iso& TA = kin[ct.bodyA].Tworld;
        Vec3 dirA = tmul(TA.R, d[k]), pA = apply(inverse(TA), ct.point);
        ja = mk6(cross(pA, dirA), dirA);
      }
      if (ct.bodyB >= 0) {
        const Iso& TB = kin[ct.bodyB].Tworld;
        Vec3 dirB = tmul(TB.R, -d[k]), pB = apply(inverse(TB), ct.point);
        jb = mk6(cross(pB, dirB), dirB);
      }
      out.JA.push_back(ja);
      out.JB.push_back(jb);
    }
  }
  out.rowDof.assign(out.rowContact.size(), -1);
  const int contactRows = (int)out.rowContact.size();
  for (int l = 0; l < L; l++) {
    out.rowContact.push_back(-1); out.rowDir.push_back(0); out.rowDof.push_back(limDof[l]);
    out.dirs.push_back(mk3(0, 0, 0)); out.JA.push_back(zero6()); out.JB.push_back(zero6());
  }
  std::vector<int> dofBodyOf(n, -1);
  for (int bi = 0; bi < m.nb; bi++) for (int k = 0; k < m.bodies[bi].ndof; k++) dofBodyOf[m.bodies[bi].dofOff + k] = bi;
  const int mrows = (int)out.rowContact.size();
  out.m = mrows;
  out.A = MatX(mrows, mrows);
  out.b.assign(mrows, 0.0); out.lo.assign(mrows, 0.0); out.hi.assign(mrows, 0.0);
  VecX bTerms(mrows, 0.0);
  out.restCoeff.assign(mrows, 0.0);
  out.findex.assign(mrows, -1);
  out.massed = MatX(n, mrows);
  out.Aall = MatX(n, mrows);

  // ---- getInformation: b, lo, hi, findex (restitution 0 and penetration correction off by default) ----
  for (int r = contactRows; r < mrows; r++) {
    // JointLimitConstraint::getInformation (:240-290): b = -qdot + bouncing velocity; the error allowance is 0 (DART_ERROR_ALLOWANCE), so
    // the bouncing velocity is -+0 * ERP / dt = 0; x starts from 0 (the constraint objects are rebuilt every step: life time 0)
    const int l = r - contactRows;
    out.b[r] = -vPre[limDof[l]];
    bTerms[r] = std::fabs(vPre[limDof[l]]);
    if (limSide[l] < 0) { out.lo[r] = 0.0; out.hi[r] = INFINITY; }
    else { out.lo[r] = -INFINITY; out.hi[r] = 0.0; }
    out.findex[r] = -1;
  }
  for (int r = 0; r < contactRows; r++) {
    const Contact& ct = out.contacts[out.rowContact[r]];
    s_t rel = 0;
    if (ct.bodyA >= 0) rel -= dot(out.JA[r], kinPre[ct.bodyA].V);
    if (ct.bodyB >= 0) rel -= dot(out.JB[r], kinPre[ct.bodyB].V);
    out.b[r] = rel;
    for (int k = 0; k < 6; k++) {                                 // sum of the magnitudes of the terms (Model::lcpNoiseBound)
      if (ct.bodyA >= 0) bTerms[r] += std::fabs(out.JA[r][k] * kinPre[ct.bodyA].V[k]);
      if (ct.bodyB >= 0) bTerms[r] += std::fabs(out.JB[r][k] * kinPre[ct.bodyB].V[k]);
    }
    if (out.rowDir[r] == 0) {
      // "Bouncing" of getInformation (ContactConstraint.cpp:393-441, ctor :95-110).  A: penetration correction, only when the
      // world enables it (ConstraintSolver.cpp:69-71: off by default; DART_ERROR_ALLOWANCE 0, DART_ERP 0.01, DART_MAX_ERV 1e-3, :45-47)
      s_t bouncingVelocity = ct.depth - 0.0;
      if (bouncingVelocity < 0.0) bouncingVelocity = 0.0;
      else {
        bouncingVelocity *= 0.01 * (1.0 / m.dt);
        if (bouncingVelocity > 1e-3) bouncingVelocity = 1e-3;
      }
      if (!m.penetrationCorrection) bouncingVelocity = 0;
      // B: restitution, e = e_A e_B; the contact bounces when e > 1e-3 and e * (approach speed) > 0.1
      const s_t e = m.boxes[ct.boxA].restitution * m.boxes[ct.boxB].restitution;
      if (e > 1e-3) {
        const s_t restitutionVel = rel * e;
        if (restitutionVel > 1e-1) {
          if (restitutionVel > bouncingVelocity) bouncingVelocity = restitutionVel > 1e+2 ? 1e+2 : restitutionVel;
          out.restCoeff[r] = e;                               // getCoefficientOfRestitution(): only when it really bounced
        }
      }
      out.b[r] += bouncingVelocity;
    }
    if (out.rowDir[r] == 0) { out.lo[r] = 0.0; out.hi[r] = INFINITY; out.findex[r] = -1; }
    else {
      s_t f = mu[out.rowContact[r]];
      out.lo[r] = -f; out.hi[r] = f;
      out.findex[r] = out.contactOffset[out.rowContact[r]];
    }
  }

  // ---- impulse tests: rows of A and massed impulse tests (BoxedLcpConstraintSolver.cpp:250-320): a unit impulse on every dimension
  //      of every constraint in turn; its own and the later constraints read their velocity change, earlier ones are mirrored ----
  std::vector<int> consFirst, consDim;                            // the constraints in LCP order: contacts, then joint-limit rows
  for (int c = 0; c < C; c++) { consFirst.push_back(out.contactOffset[c]); consDim.push_back(dim[c]); }
  for (int l = 0; l < L; l++) { consFirst.push_back(contactRows + l); consDim.push_back(1); }
  const int nCons = (int)consFirst.size();
  for (int c = 0; c < nCons; c++) {
    for (int k = 0; k < consDim[c]; k++) {
      const int row = consFirst[c] + k;
      std::vector<Vec6> imps(m.nb, zero6()), dV;
      VecX jimp(n, 0.0);
      if (out.rowContact[row] >= 0) {
        const Contact& ct = out.contacts[out.rowContact[row]];
        if (ct.bodyA >= 0) imps[ct.bodyA] = imps[ct.bodyA] + out.JA[row];
        if (ct.bodyB >= 0) imps[ct.bodyB] = imps[ct.bodyB] + out.JB[row];
      } else jimp[out.rowDof[row]] = 1.0;                        // JointLimitConstraint::applyUnitImpulse (:293-318)
      VecX delV(n, 0.0);
      impulseDynamics(m, kin, art, imps, delV.data(), &dV, jimp.data());
      for (int i = 0; i < n; i++) out.massed(i, row) = delV[i];
      for (int c2 = c; c2 < nCons; c2++)
        for (int k2 = 0; k2 < consDim[c2]; k2++) {
          const int col = consFirst[c2] + k2;
          s_t v = 0;
          if (out.rowContact[col] >= 0) {                          // ContactConstraint::getVelocityChange
            const Contact& ct2 = out.contacts[out.rowContact[col]];
            if (ct2.bodyA >= 0) v += dot(out.JA[col], dV[ct2.bodyA]);
            if (ct2.bodyB >= 0) v += dot(out.JB[col], dV[ct2.bodyB]);
          } else v = delV[out.rowDof[col]];                        // JointLimitConstraint::getVelocityChange (:321-349, withCfm false)
          out.A(row, col) = v;
        }
      for (int c2 = 0; c2 < c; c2++)
        for (int k2 = 0; k2 < consDim[c2]; k2++) {
          const int col = consFirst[c2] + k2;
          out.A(row, col) = out.A(col, row);
        }
    }
  }
  if (m.lcpNoiseUlps > 0) {                                       // test instrument, see Model::lcpNoiseUlps
    const uint64_t sample = m.lcpNoiseSample++;
    s_t amax = 0;
    for (int i = 0; i < mrows; i++) for (int j = 0; j < mrows; j++) amax = std::max(amax, std::fabs(out.A(i, j)));
    for (int i = 0; i < mrows; i++)
      for (int j = i; j < mrows; j++) {
        uint64_t h = m.lcpNoiseSeed * 0x9E3779B97F4A7C15ull + sample * 0xBF58476D1CE4E5B9ull + (uint64_t)(i * 64 + j) * 0x94D049BB133111EBull;
        h ^= h >> 31; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 29;
        const s_t e = (s_t)((int)(h % 3) - 1) * (s_t)m.lcpNoiseUlps * 2.220446049250313e-16;
        if (m.lcpNoiseAbsolute) {
          if (out.A(i, j) == 0 && out.A(j, i) == 0) continue;   // structural zeros (rows of other constrained groups, decoupled rows) stay
          out.A(i, j) += e * amax;
          if (j != i) out.A(j, i) += e * amax;
        } else {
          out.A(i, j) *= 1.0 + e;
          if (j != i) out.A(j, i) *= 1.0 + e;
        }
      }
    s_t bmax = 0;                                                 // ... and the same on the right-hand side b = -J v (+ bounce)
    for (int i = 0; i < mrows; i++) bmax = std::max(bmax, std::fabs(out.b[i]));
    for (int i = 0; i < mrows; i++) {
      uint64_t h = m.lcpNoiseSeed * 0x9E3779B97F4A7C15ull + sample * 0xBF58476D1CE4E5B9ull + (uint64_t)(4096 + i) * 0x94D049BB133111EBull;
      h ^= h >> 31; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 29;
      const s_t e = (s_t)((int)(h % 3) - 1) * (s_t)m.lcpNoiseUlps * 2.220446049250313e-16;
      if (m.lcpNoiseAbsolute) { if (out.b[i] != 0) out.b[i] += e * bmax; }
      else out.b[i] *= 1.0 + e;
    }
  }

  // ---- constraint forces in joint space: A_c columns (DCC.cpp:231-270, 2961-2988; Joint.cpp:1176-1181) ----
  // (a joint-limit constraint is not a contact constraint: DifferentiableContactConstraint gives it a zero world force, DCC.cpp:51-99,
  //  so its column of A_c is zero - in the standardisation's Q when friction rows sit on their bounds, and in the backward pass)
  for (int r = 0; r < contactRows; r++) {
    const Contact& ct = out.contacts[out.rowContact[r]];
    Vec6 F = mk6(cross(ct.point, out.dirs[r]), out.dirs[r]);
    for (int bi = 0; bi < m.nb; bi++) {
      const Body& bd = m.bodies[bi];
      if (bd.ndof == 0) continue;
      bool pa = ct.bodyA >= 0 && dofIsParentOf(m, bi, ct.bodyA), pb = ct.bodyB >= 0 && dofIsParentOf(m, bi, ct.bodyB);
      s_t mult = (pa && pb) ? 0.0 : (pa ? 1.0 : (pb ? -1.0 : 0.0));
      for (int k = 0; k < bd.ndof; k++) {
        if (mult == 0) { out.Aall(bd.dofOff + k, r) = 0; continue; }
        Vec6 tw = AdT(kin[bi].Tworld, bd.S[k]);
        out.Aall(bd.dofOff + k, r) = dot(tw, F) * mult;
      }
    }
  }

  // ---- constrained groups (ConstraintSolver::buildConstrainedGroups :724-780, ContactConstraint::uniteSkeletons :879-907) ----
  // Skeletons connected by a contact between two reactive bodies form one group; a contact with a world-fixed collider connects
  // nothing.  Groups are numbered by their first constraint, a group's rows keep the world's constraint order.
  {
    std::vector<int> parentOf;                                   // union-find over skeleton ids
    auto find = [&](int s_) { while (parentOf[s_] != s_) s_ = parentOf[s_]; return s_; };
    int maxSk = 0;
    for (int i = 0; i < m.nb; i++) maxSk = std::max(maxSk, m.skeleton[i]);
    parentOf.resize(maxSk + 1);
    for (int i = 0; i <= maxSk; i++) parentOf[i] = i;
    // BodyNode::isReactive (BodyNode.cpp:2394-2418): the body depends on at least one generalized coordinate
    auto reactive = [&](int body) { for (; body >= 0; body = m.bodies[body].parent) if (m.bodies[body].ndof > 0) return true; return false; };
    for (const Contact& ct : out.contacts) {
      if (!reactive(ct.bodyA) || !reactive(ct.bodyB)) continue;
      const int ra = find(m.skeleton[ct.bodyA]), rb = find(m.skeleton[ct.bodyB]);
      if (ra != rb) parentOf[rb] = ra;
    }
    std::vector<int> groupOfRoot(maxSk + 1, -1);
    out.rowGroup.assign(mrows, 0);
    out.numGroups = 0;
    for (int r = 0; r < mrows; r++) {
      int skelBody;
      if (out.rowContact[r] >= 0) {
        const Contact& ct = out.contacts[out.rowContact[r]];
        skelBody = reactive(ct.bodyA) ? ct.bodyA : ct.bodyB;
      } else skelBody = dofBodyOf[out.rowDof[r]];                  // JointLimitConstraint::getRootSkeleton: the joint's skeleton
      const int root = find(m.skeleton[skelBody]);
      if (groupOfRoot[root] < 0) groupOfRoot[root] = out.numGroups++;
      out.rowGroup[r] = groupOfRoot[root];
    }
  }

  // ---- per group: warm start / guess, stage 0, stages 1-3, registration (BoxedLcpConstraintSolver.cpp:190-789 runs once per
  //      constrained group, ConstraintSolver.cpp:800-811) ----
  MatX Minv = invMassMatrix(m, kin, art);
  if (m.lcpAlternateA) {                                          // test instrument, see Model::lcpAlternateA
    MatX Jt(n, mrows);                                            // column r: the generalized force of a unit impulse on row r
    for (int r = 0; r < mrows; r++)
      for (int d_ = 0; d_ < n; d_++) Jt(d_, r) = out.rowContact[r] >= 0 ? out.Aall(d_, r) : (d_ == out.rowDof[r] ? 1.0 : 0.0);
    MatX MJ = matmul(Minv, Jt);
    for (int i = 0; i < mrows; i++)
      for (int j = 0; j < mrows; j++) {
        if (out.A(i, j) == 0 && out.A(j, i) == 0) continue;       // structural zeros stay
        s_t s_ = 0;
        for (int d_ = 0; d_ < n; d_++) s_ += Jt(d_, i) * MJ(d_, j);
        out.A(i, j) = s_;
      }
  }
  if (m.lcpCacheSlots && (int)lcpCache.size() == 3 * nCons) {      // Model::lcpCacheSlots: three entries per constraint -> the reference's rows
    VecX compact(mrows, 0.0);
    for (int c = 0; c < nCons; c++) for (int k = 0; k < consDim[c]; k++) compact[consFirst[c] + k] = lcpCache[3 * c + k];
    lcpCache = compact;
  } else if (m.lcpCacheSlots) lcpCache.clear();
  if (m.lcpNoiseBound > 0) {                                       // test instrument, see Model::lcpNoiseBound
    MatX Jt(n, mrows);
    for (int r = 0; r < mrows; r++)
      for (int d_ = 0; d_ < n; d_++) Jt(d_, r) = out.rowContact[r] >= 0 ? out.Aall(d_, r) : (d_ == out.rowDof[r] ? 1.0 : 0.0);
    MatX MJ = matmul(Minv, Jt);
    const uint64_t sample = m.lcpNoiseSample++;
    auto draw = [&](uint64_t idx) {
      uint64_t h = m.lcpNoiseSeed * 0x9E3779B97F4A7C15ull + sample * 0xBF58476D1CE4E5B9ull + idx * 0x94D049BB133111EBull;
      h ^= h >> 31; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 29;
      return (s_t)((int)(h % 3) - 1) * (s_t)m.lcpNoiseBound * 2.220446049250313e-16;
    };
    for (int i = 0; i < mrows; i++)
      for (int j = i; j < mrows; j++) {
        if (out.A(i, j) == 0 && out.A(j, i) == 0) continue;       // structural zeros stay
        s_t terms = 0;
        for (int d_ = 0; d_ < n; d_++) terms += std::fabs(Jt(d_, i) * MJ(d_, j));
        const s_t e = draw((uint64_t)(i * 64 + j)) * terms;
        out.A(i, j) += e;
        if (j != i) out.A(j, i) += e;
      }
    for (int i = 0; i < mrows; i++) if (out.b[i] != 0) out.b[i] += draw((uint64_t)(4096 + i)) * bTerms[i];
  }
  const bool haveCache = (int)lcpCache.size() == mrows;
  VecX X(mrows, 0.0);
  out.cfmRow.assign(mrows, 0.0);
  out.rowClass.assign(mrows, RC_NOT_CLAMPING);
  VecX eRow(mrows, 0.0), fcRow(mrows, 0.0);                      // per row: its entry of E / of f_c (merged below)
  MatX aGradAll = out.A;
  bool allStage0 = true, allStandardized = true;
  for (int gi = 0; gi < out.numGroups; gi++) {
    std::vector<int> idx, local(mrows, -1);
    for (int r = 0; r < mrows; r++) if (out.rowGroup[r] == gi) { local[r] = (int)idx.size(); idx.push_back(r); }
    const int mg = (int)idx.size();
    ContactResult sub;
    sub.m = mg;
    sub.A = MatX(mg, mg); sub.b.assign(mg, 0.0); sub.lo.assign(mg, 0.0); sub.hi.assign(mg, 0.0); sub.findex.assign(mg, -1);
    sub.Aall = MatX(n, mg);
    for (int i = 0; i < mg; i++) {
      for (int j = 0; j < mg; j++) sub.A(i, j) = out.A(idx[i], idx[j]);
      sub.b[i] = out.b[idx[i]]; sub.lo[i] = out.lo[idx[i]]; sub.hi[i] = out.hi[idx[i]];
      sub.findex[i] = out.findex[idx[i]] < 0 ? -1 : local[out.findex[idx[i]]];
      for (int d_ = 0; d_ < n; d_++) sub.Aall(d_, i) = out.Aall(d_, idx[i]);
    }
    // warm start / guess (:202-208, 332-337)
    VecX Xg(mg, 0.0);
    if (haveCache) for (int i = 0; i < mg; i++) Xg[i] = lcpCache[idx[i]];
    else Xg = guessSolution(sub.A, sub.b, sub.findex);
    const VecX XBackup = Xg;
    Cggm g;
    g.model = &m; g.Minv = &Minv; g.cr = &sub;
    VecX aColNorms(mg, 0.0);
    for (int c = 0; c < mg; c++) { s_t s_ = 0; for (int r = 0; r < mg; r++) s_ += sub.A(r, c) * sub.A(r, c); aColNorms[c] = s_; }
    MatX aGrad = sub.A;
    // stage 0: classify the cached x and solve the active set in closed form (:434-457)
    s_t cfm = 0.0;
    g.X = Xg; g.Hi = sub.hi; g.Lo = sub.lo; g.FIndex = sub.findex; g.B = sub.b; g.AColNorms = aColNorms; g.A = aGrad;
    g.cfmConst = cfm; g.ignoreFriction = false;
    g.constructMatrices();
    bool success = g.standardized;
    const bool shortCircuit = success;
    bool hadToIgnoreFriction = false;
    if (success) Xg = g.X;
    else allStage0 = false;
    // stages 1-3 (:461-687)
    if (!success && (int)m.lcpForced.size() == mrows) {              // test instrument, see Model::lcpForced
      for (int i = 0; i < mg; i++) Xg[i] = m.lcpForced[idx[i]];
      MatX Av = sub.A;
      if (m.lcpForcedCfm) { cfm = m.fallbackCfm; for (int i = 0; i < mg; i++) { Av(i, i) += cfm; aGrad(i, i) += cfm; } }
      if (isLCPSolutionValid(Av, Xg, sub.b, sub.hi, sub.lo, sub.findex, false)) *status |= m.lcpForcedCfm ? NBL_ST_LCP_PGS : NBL_ST_LCP_PIVOT;
      else *status |= 0x40000000u;
    } else if (!success) {
      uint32_t st13 = 0;
      lcpCascade(sub.A, sub.b, sub.lo, sub.hi, sub.findex, XBackup, m.fallbackCfm, Xg, cfm, hadToIgnoreFriction, st13);
      *status |= st13;
      if (cfm != 0.0) for (int i = 0; i < mg; i++) aGrad(i, i) += cfm;
    }
    // re-register + classify + standardize with the fresh solution (:718-736)
    if (!shortCircuit) {
      g.X = Xg; g.A = aGrad; g.cfmConst = cfm; g.ignoreFriction = hadToIgnoreFriction;
      g.constructMatrices();
      if (g.standardized) Xg = g.X;
    }
    if (!g.standardized) allStandardized = false;
    // merge the group's rows into the world's vectors
    for (int i = 0; i < mg; i++) {
      const int r = idx[i];
      X[r] = Xg[i];
      out.cfmRow[r] = cfm;
      aGradAll(r, r) = aGrad(i, i);
      out.rowClass[r] = sub.rowClass[i];
      if (sub.rowClass[i] == RC_CLAMPING) fcRow[r] = sub.fc[sub.clampingIndex[i]];
      if (sub.rowClass[i] == RC_UPPER_BOUND) eRow[r] = sub.E(sub.upperBoundIndex[i], sub.clampingIndex[sub.findex[i]]);
    }
    if (out.numGroups == 1) { out.cfm = cfm; out.ignoreFriction = hadToIgnoreFriction; }
  }
  if (allStage0) *status |= NBL_ST_LCP_STAGE0;
  if (allStandardized) *status |= NBL_ST_STANDARDIZED;
  // the world-level classification vectors of the backward pass (BackpropSnapshot assembles the groups' matrices, :4215-4409;
  // clamping / upper-bound rows are numbered in row order here, which is a permutation of the reference's group-major order)
  out.clampingIndex.assign(mrows, -1); out.upperBoundIndex.assign(mrows, -1);
  out.numClamping = 0; out.numUpperBound = 0;
  for (int r = 0; r < mrows; r++) {
    if (out.rowClass[r] == RC_CLAMPING) out.clampingIndex[r] = out.numClamping++;
    if (out.rowClass[r] == RC_UPPER_BOUND) out.upperBoundIndex[r] = out.numUpperBound++;
  }
  out.E = MatX(out.numUpperBound, out.numClamping);
  out.fc.assign(out.numClamping, 0.0);
  for (int r = 0; r < mrows; r++) {
    if (out.rowClass[r] == RC_CLAMPING) out.fc[out.clampingIndex[r]] = fcRow[r];
    if (out.rowClass[r] == RC_UPPER_BOUND) out.E(out.upperBoundIndex[r], out.clampingIndex[out.findex[r]]) = eRow[r];
  }
  out.A = aGradAll;
  out.x = X;
  out.standardized = allStandardized;
  lcpCache = X;  // mX persists inside the solver (BoxedLcpConstraintSolver.cpp:176-187)
  if (m.lcpCacheSlots) {
    lcpCache.assign(3 * nCons, 0.0);
    for (int c = 0; c < nCons; c++) for (int k = 0; k < consDim[c]; k++) lcpCache[3 * c + k] = X[consFirst[c] + k];
  }

  // ---- applyImpulse + computeImpulseForwardDynamics (ContactConstraint.cpp:630-684, Skeleton.cpp:13571-13595) ----
  std::vector<Vec6> imps(m.nb, zero6());
  VecX jimp(n, 0.0);
  bool any = false;
  for (int r = 0; r < mrows; r++) {
    if (out.rowContact[r] >= 0) {
      const Contact& ct = out.contacts[out.rowContact[r]];
      if (ct.bodyA >= 0) imps[ct.bodyA] = imps[ct.bodyA] + out.JA[r] * X[r];
      if (ct.bodyB >= 0) imps[ct.bodyB] = imps[ct.bodyB] + out.JB[r] * X[r];
    } else jimp[out.rowDof[r]] += X[r];                            // JointLimitConstraint::applyImpulse (:364-381)
    any = true;
  }
  if (any) {
    VecX delV(n, 0.0);
    impulseDynamics(m, kin, art, imps, delV.data(), nullptr, jimp.data());
    for (int i = 0; i < n; i++) vOut[i] = vPre[i] + delV[i];
  }
  out.status = *status;
}

// ---------------------------------------------------------------------------------------------
// Backward: contact terms of the BackpropSnapshot Jacobians
// ---------------------------------------------------------------------------------------------
enum DofContactType { DCT_NONE = 0, DCT_VERTEX, DCT_FACE, DCT_EDGE_A, DCT_EDGE_B, DCT_SELF_COLLISION, DCT_UNSUPPORTED,
                      DCT_SPHERE_TO_BOX, DCT_BOX_TO_SPHERE, DCT_SPHERE_A, DCT_SPHERE_B,
                      DCT_SPHERE_TO_PIPE, DCT_PIPE_TO_SPHERE, DCT_PIPE_A, DCT_PIPE_B };

// DifferentiableContactConstraint::getDofContactType (DCC.cpp:116-228), box-box contact types only
inline int dofContactType(const Model& m, const Contact& ct, int dofBody) {
  bool pa = ct.bodyA >= 0 && dofIsParentOf(m, dofBody, ct.bodyA), pb = ct.bodyB >= 0 && dofIsParentOf(m, dofBody, ct.bodyB);
  if (pa && pb) return DCT_SELF_COLLISION;
  if (!pa && !pb) return DCT_NONE;
  if (pa) {
    if (ct.type == CT_FACE_VERTEX) return DCT_FACE;
    if (ct.type == CT_VERTEX_FACE) return DCT_VERTEX;
    if (ct.type == CT_EDGE_EDGE) return DCT_EDGE_A;
    if (ct.type == CT_SPHERE_BOX) return DCT_SPHERE_TO_BOX;
    if (ct.type == CT_BOX_SPHERE) return DCT_BOX_TO_SPHERE;
    if (ct.type == CT_SPHERE_SPHERE) return DCT_SPHERE_A;
    if (ct.type == CT_SPHERE_PIPE) return DCT_SPHERE_TO_PIPE;
    if (ct.type == CT_PIPE_SPHERE) return DCT_PIPE_TO_SPHERE;
    if (ct.type == CT_PIPE_PIPE) return DCT_PIPE_A;
    return DCT_UNSUPPORTED;
  }
  if (ct.type == CT_FACE_VERTEX) return DCT_VERTEX;
  if (ct.type == CT_VERTEX_FACE) return DCT_FACE;
  if (ct.type == CT_EDGE_EDGE) return DCT_EDGE_B;
  if (ct.type == CT_SPHERE_BOX) return DCT_BOX_TO_SPHERE;
  if (ct.type == CT_BOX_SPHERE) return DCT_SPHERE_TO_BOX;
  if (ct.type == CT_SPHERE_SPHERE) return DCT_SPHERE_B;
  if (ct.type == CT_SPHERE_PIPE) return DCT_PIPE_TO_SPHERE;
  if (ct.type == CT_PIPE_SPHERE) return DCT_SPHERE_TO_PIPE;
  if (ct.type == CT_PIPE_PIPE) return DCT_PIPE_B;
  return DCT_UNSUPPORTED;
}

// math::getContactPointGradient (Geometry.cpp:1129-1236); edge-edge contacts call it with radiusA = radiusB = 1
inline Vec3 contactPointGradient(const Vec3& aP, const Vec3& aPg, const Vec3& aD, const Vec3& aDg, const Vec3& bP,
                                 const Vec3& bPg, const Vec3& bD, const Vec3& bDg, s_t radiusA, s_t radiusB) {
  Vec3 p = bP - aP, d_p = bPg - aPg;
  s_t uaub = dot(aD, bD), d_uaub = dot(aDg, bD) + dot(aD, bDg);
  s_t q1 = dot(aD, p), d_q1 = dot(aDg, p) + dot(aD, d_p);
  s_t q2 = -dot(bD, p), d_q2 = -dot(bDg, p) - dot(bD, d_p);
  s_t d = 1 - uaub * uaub, d_d = -2 * d_uaub * uaub;
  auto over = [](const Vec3& x, s_t s) { return mk3(x[0] / s, x[1] / s, x[2] / s); };   // Eigen's vector / scalar divides every component
  if (d <= 0) return over(aPg * radiusB + bPg * radiusA, radiusA + radiusB);
  s_t e = 1.0 / d, d_e = -(1.0 / (d * d)) * d_d;
  s_t alpha = (q1 + uaub * q2) * e, d_alpha = (q1 + uaub * q2) * d_e + (d_q1 + d_uaub * q2 + uaub * d_q2) * e;
  s_t beta = (uaub * q1 + q2) * e, d_beta = (uaub * q1 + q2) * d_e + (d_uaub * q1 + uaub * d_q1 + d_q2) * e;
  return over((aPg + alpha * aDg + d_alpha * aD) * radiusB + (bPg + beta * bDg + d_beta * bD) * radiusA, radiusA + radiusB);
}
// math::closestPointOnLineGradient (Geometry.cpp:4427-4445)
inline Vec3 closestPointOnLineGradient(const Vec3& pointOnLine, const Vec3& pointOnLineGradient, const Vec3& lineDirection,
                                       const Vec3& lineDirectionGradient, const Vec3& goalPoint, const Vec3& goalPointGradient) {
  s_t offset = dot(lineDirection, pointOnLine);
  s_t dOffset = dot(lineDirectionGradient, pointOnLine) + dot(lineDirection, pointOnLineGradient);
  s_t goalOffset = dot(lineDirection, goalPoint);
  s_t dGoalOffset = dot(lineDirectionGradient, goalPoint) + dot(lineDirection, goalPointGradient);
  s_t relative = goalOffset - offset, dRelative = dGoalOffset - dOffset;
  return pointOnLineGradient + relative * lineDirectionGradient + dRelative * lineDirection;
}
inline Vec3 contactPointGradient(const Vec3& aP, const Vec3& aPg, const Vec3& aD, const Vec3& aDg, const Vec3& bP,
                                 const Vec3& bPg, const Vec3& bD, const Vec3& bDg) {
  Vec3 p = bP - aP, d_p = bPg - aPg;
  s_t uaub = dot(aD, bD), d_uaub = dot(aDg, bD) + dot(aD, bDg);
  s_t q1 = dot(aD, p), d_q1 = dot(aDg, p) + dot(aD, d_p);
  s_t q2 = -dot(bD, p), d_q2 = -dot(bDg, p) - dot(bD, d_p);
  s_t d = 1 - uaub * uaub, d_d = -2 * d_uaub * uaub;
  if (d <= 0) return 0.5 * (aPg + bPg);
  s_t e = 1.0 / d, d_e = -(1.0 / (d * d)) * d_d;
  s_t alpha = (q1 + uaub * q2) * e, d_alpha = (q1 + uaub * q2) * d_e + (d_q1 + d_uaub * q2 + uaub * d_q2) * e;
  s_t beta = (uaub * q1 + q2) * e, d_beta = (uaub * q1 + q2) * d_e + (d_uaub * q1 + uaub * d_q1 + d_q2) * e;
  return 0.5 * ((aPg + alpha * aDg + d_alpha * aD) + (bPg + beta * bDg + d_beta * bD));
}

// ContactConstraint::getTangentBasisMatrixODEGradient (ContactConstraint.cpp:800-876)
inline void tangentBasisGradient(const Vec3& n, const Vec3& g, Vec3& dt1, Vec3& dt2) {
  const s_t EPS2 = 1e-12;
  Vec3 crs = mk3(0, 0, 1);
  Vec3 tangent = cross(crs, n);
  if (dot(tangent, tangent) < EPS2) {
    crs = mk3(1, 0, 0); tangent = cross(crs, n);
    if (dot(tangent, tangent) < EPS2) {
      crs = mk3(0, 1, 0); tangent = cross(crs, n);
      if (dot(tangent, tangent) < EPS2) { crs = mk3(0, 0, 1); tangent = cross(crs, n); }
    }
  }
  s_t tn = norm(tangent);
  tangent = mk3(tangent[0] / tn, tangent[1] / tn, tangent[2] / tn);          // Eigen's normalize() and `/= tangentNorm` divide (no reciprocal):
  Vec3 gd = cross(crs, g);                                                    // pinned against the reference's own function,
  gd = mk3(gd[0] / tn, gd[1] / tn, gd[2] / tn);                               // tests/test_oracle_ref_geometry.py
  Vec3 gradOfTangent = (std::fabs(tn - 1.0) > 1e-6) ? gd - dot(gd, tangent) * tangent : gd;
  dt1 = gradOfTangent;
  dt2 = cross(g, tangent) + cross(n, gradOfTangent);
}

struct ContactGrad {
  const Model& m;
  const std::vector<Kin>& kin;
  const s_t* q;
  const ContactResult& cr;
  std::vector<Vec6> axis;     // per dof: Joint::getWorldAxisScrewForVelocity
  std::vector<Vec6> posTwist; // per dof: Joint::getWorldAxisScrewForPosition
  std::vector<int> dofBody;
  ContactGrad(const Model& m_, const std::vector<Kin>& k_, const s_t* q_, const ContactResult& c_) : m(m_), kin(k_), q(q_), cr(c_) {
    axis.resize(m.n); posTwist.resize(m.n); dofBody.resize(m.n);
    for (int bi = 0; bi < m.nb; bi++) {
      const Body& bd = m.bodies[bi];
      Vec6 H[6];
      positionJacobian(bd, q, H);
      for (int k = 0; k < bd.ndof; k++) {
        axis[bd.dofOff + k] = AdT(kin[bi].Tworld, bd.S[k]);
        posTwist[bd.dofOff + k] = AdT(kin[bi].Tworld, H[k]);
        dofBody[bd.dofOff + k] = bi;
      }
    }
  }
  // getContactPositionGradient (DCC.cpp:328-445)
  Vec3 positionGradient(const Contact& ct, int dof) const {
    int type = dofContactType(m, ct, dofBody[dof]);
    if (type == DCT_FACE || type == DCT_NONE || type == DCT_UNSUPPORTED) return mk3(0, 0, 0);
    const Vec6& tw = posTwist[dof];
    Vec3 w = head(tw), v = tail(tw);
    auto gradTheta = [&](const Vec3& pt) { return (norm(w) > 1e-6) ? cross(w, pt) + v : v; };  // math::gradientWrtTheta(.,.,0)
    if (type == DCT_VERTEX || type == DCT_SELF_COLLISION) return gradTheta(ct.point);
    auto unlock = [&](Vec3 g) {   // remove the motion along every locked face normal (DCC.cpp:351-369)
      for (int k = 0; k < 3; k++) if (ct.faceLocked[k]) g = g - dot(ct.faceNormal[k], g) * ct.faceNormal[k];
      return g;
    };
    if (type == DCT_SPHERE_A) return (ct.radiusB / (ct.radiusA + ct.radiusB)) * gradTheta(ct.centerA);   // DCC.cpp:342-346
    if (type == DCT_SPHERE_B) return (ct.radiusA / (ct.radiusA + ct.radiusB)) * gradTheta(ct.centerB);
    if (type == DCT_SPHERE_TO_BOX) return unlock(gradTheta(ct.sphereCenter));                              // :352-373
    if (type == DCT_BOX_TO_SPHERE) return gradTheta(ct.point) + unlock(-1.0 * gradTheta(ct.sphereCenter)); // :374-403
    if (type == DCT_SPHERE_TO_PIPE) {                                                                       // DCC.cpp:484-495
      s_t weight = ct.pipeRadius / (ct.sphereRadius + ct.pipeRadius);
      Vec3 raw = gradTheta(ct.sphereCenter);
      Vec3 par = dot(ct.pipeDir, raw) * ct.pipeDir;
      return par + weight * (raw - par);
    }
    if (type == DCT_PIPE_TO_SPHERE) {                                                                       // :496-509
      Vec3 raw = closestPointOnLineGradient(ct.pipeFixedPoint, gradTheta(ct.pipeFixedPoint), ct.pipeDir, cross(w, ct.pipeDir),
                                            ct.sphereCenter, mk3(0, 0, 0));
      return (ct.sphereRadius / (ct.sphereRadius + ct.pipeRadius)) * raw;
    }
    if (type == DCT_PIPE_A)                                                                                 // :510-528
      return contactPointGradient(ct.edgeAFixedPoint, gradTheta(ct.edgeAFixedPoint), ct.edgeADir, cross(w, ct.edgeADir),
                                  ct.edgeBFixedPoint, mk3(0, 0, 0), ct.edgeBDir, mk3(0, 0, 0), ct.radiusA, ct.radiusB);
    if (type == DCT_PIPE_B)                                                                                 // :529-547
      return contactPointGradient(ct.edgeAFixedPoint, mk3(0, 0, 0), ct.edgeADir, mk3(0, 0, 0), ct.edgeBFixedPoint,
                                  gradTheta(ct.edgeBFixedPoint), ct.edgeBDir, cross(w, ct.edgeBDir), ct.radiusA, ct.radiusB);
    if (type == DCT_EDGE_A)
      return contactPointGradient(ct.edgeAFixedPoint, gradTheta(ct.edgeAFixedPoint), ct.edgeADir, cross(w, ct.edgeADir),
                                  ct.edgeBFixedPoint, mk3(0, 0, 0), ct.edgeBDir, mk3(0, 0, 0));
    return contactPointGradient(ct.edgeAFixedPoint, mk3(0, 0, 0), ct.edgeADir, mk3(0, 0, 0), ct.edgeBFixedPoint,
                                gradTheta(ct.edgeBFixedPoint), ct.edgeBDir, cross(w, ct.edgeBDir));
  }
  // getContactNormalGradient (DCC.cpp:594-735)
  Vec3 normalGradient(const Contact& ct, int dof) const {
    int type = dofContactType(m, ct, dofBody[dof]);
    if (type == DCT_VERTEX || type == DCT_NONE || type == DCT_UNSUPPORTED) return mk3(0, 0, 0);
    Vec3 w = head(posTwist[dof]);
    if (type == DCT_FACE || type == DCT_SELF_COLLISION) return cross(w, ct.normal);
    if (type == DCT_SPHERE_A || type == DCT_SPHERE_B) {                                                   // DCC.cpp:626-645
      Vec3 v = tail(posTwist[dof]);
      auto gradTheta = [&](const Vec3& pt) { return (norm(w) > 1e-6) ? cross(w, pt) + v : v; };
      s_t nrm = norm(ct.centerA - ct.centerB);
      Vec3 pg = (1.0 / nrm) * gradTheta(type == DCT_SPHERE_A ? ct.centerA : ct.centerB);
      pg = pg - dot(ct.normal, pg) * ct.normal;
      return type == DCT_SPHERE_A ? pg : -1.0 * pg;
    }
    if (type == DCT_SPHERE_TO_BOX || type == DCT_BOX_TO_SPHERE) {                                         // DCC.cpp:646-709
      Vec3 v = tail(posTwist[dof]);
      auto gradTheta = [&](const Vec3& pt) { return (norm(w) > 1e-6) ? cross(w, pt) + v : v; };
      s_t nrm = norm(ct.sphereCenter - ct.point);
      Vec3 cpg = positionGradient(ct, dof);
      Vec3 spg = type == DCT_SPHERE_TO_BOX ? gradTheta(ct.sphereCenter) : mk3(0, 0, 0);
      if (nrm > 1e-5) { cpg = (1.0 / nrm) * cpg; spg = (1.0 / nrm) * spg; }
      // BOX_SPHERE: normal = contact point - sphere centre; SPHERE_BOX: the opposite
      Vec3 total = ct.type == CT_BOX_SPHERE ? cpg - spg : spg - cpg;
      return total - dot(total, ct.normal) * ct.normal;
    }
    if (type == DCT_SPHERE_TO_PIPE || type == DCT_PIPE_TO_SPHERE) {                                        // DCC.cpp:819-861
      Vec3 v = tail(posTwist[dof]);
      auto gradTheta = [&](const Vec3& pt) { return (norm(w) > 1e-6) ? cross(w, pt) + v : v; };
      s_t nrm = norm(ct.pipeClosestPoint - ct.sphereCenter);
      Vec3 ng;
      if (type == DCT_SPHERE_TO_PIPE) {
        ng = gradTheta(ct.sphereCenter);
        ng = ng - dot(ng, ct.pipeDir) * ct.pipeDir;
      } else
        ng = closestPointOnLineGradient(ct.pipeFixedPoint, gradTheta(ct.pipeFixedPoint), ct.pipeDir, cross(w, ct.pipeDir),
                                        ct.sphereCenter, mk3(0, 0, 0));
      ng = (1.0 / nrm) * ng;
      ng = ng - dot(ct.normal, ng) * ct.normal;
      // the normal points from object 2 to object 1: the moving side is object 1 exactly when the contact type names it first
      const bool first = type == DCT_SPHERE_TO_PIPE ? ct.type == CT_SPHERE_PIPE : ct.type == CT_PIPE_SPHERE;
      return first ? ng : -1.0 * ng;
    }
    if (type == DCT_PIPE_A || type == DCT_PIPE_B) {                                                        // DCC.cpp:862-938
      Vec3 v = tail(posTwist[dof]);
      auto gradTheta = [&](const Vec3& pt) { return (norm(w) > 1e-6) ? cross(w, pt) + v : v; };
      const bool a = type == DCT_PIPE_A;
      const Vec3 z = mk3(0, 0, 0);
      const Vec3 aPg = a ? gradTheta(ct.edgeAFixedPoint) : z, aDg = a ? cross(w, ct.edgeADir) : z;
      const Vec3 bPg = a ? z : gradTheta(ct.edgeBFixedPoint), bDg = a ? z : cross(w, ct.edgeBDir);
      Vec3 cA = contactPointGradient(ct.edgeAFixedPoint, aPg, ct.edgeADir, aDg, ct.edgeBFixedPoint, bPg, ct.edgeBDir, bDg, 0.0, 1.0);
      Vec3 cB = contactPointGradient(ct.edgeAFixedPoint, aPg, ct.edgeADir, aDg, ct.edgeBFixedPoint, bPg, ct.edgeBDir, bDg, 1.0, 0.0);
      Vec3 ng = (1.0 / norm(ct.edgeAClosestPoint - ct.edgeBClosestPoint)) * (cA - cB);
      return ng - dot(ct.normal, ng) * ct.normal;
    }
    s_t sign = dot(cross(ct.edgeBDir, ct.edgeADir), ct.normal) < 0 ? -1.0 : 1.0;
    if (type == DCT_EDGE_A) return sign * cross(ct.edgeBDir, cross(w, ct.edgeADir));
    return sign * cross(cross(w, ct.edgeBDir), ct.edgeADir);
  }
  // getContactForceGradient (DCC.cpp:1092-1111)
  Vec3 forceGradient(const Contact& ct, int dir, int dof) const {
    int type = dofContactType(m, ct, dofBody[dof]);
    if (type == DCT_VERTEX || type == DCT_NONE) return mk3(0, 0, 0);
    Vec3 ng = normalGradient(ct, dof);
    if (dir == 0 || dot(ng, ng) <= 1e-12) return ng;
    Vec3 d1, d2;
    tangentBasisGradient(ct.normal, ng, d1, d2);
    return dir == 1 ? d1 : d2;
  }
  // getContactWorldForceGradient (DCC.cpp:1115-1128)
  Vec6 worldForceGradient(int row, int dof) const {
    const Contact& ct = cr.contacts[cr.rowContact[row]];
    Vec3 fg = forceGradient(ct, cr.rowDir[row], dof), pg = positionGradient(ct, dof);
    return mk6(cross(ct.point, fg) + cross(pg, cr.dirs[row]), fg);
  }
  // getScrewAxisForForceGradient (DCC.cpp:1226-1316). The FreeJoint intra-joint branch
  // (FreeJoint.cpp:1195-1237) is the same derivative d/dq_l [Ad(T_world(child)) S_i] = ad(s_l^pos, s_i)
  // written out explicitly, so one expression covers both.
  Vec6 screwAxisGradient(int screwDof, int rotateDof) const {
    int bs = dofBody[screwDof], br = dofBody[rotateDof];
    if (bs == br) {
      if (m.bodies[bs].ndof == 1) return zero6();
    } else if (!dofIsParentOf(m, br, bs)) return zero6();
    return ad(posTwist[rotateDof], axis[screwDof]);
  }
  s_t multiple(int row, int dof) const {
    const Contact& ct = cr.contacts[cr.rowContact[row]];
    bool pa = ct.bodyA >= 0 && dofIsParentOf(m, dofBody[dof], ct.bodyA), pb = ct.bodyB >= 0 && dofIsParentOf(m, dofBody[dof], ct.bodyB);
    return (pa && pb) ? 0.0 : (pa ? 1.0 : (pb ? -1.0 : 0.0));
  }
  // getConstraintForcesJacobian (DCC.cpp:1505-1649, the "slow, but known to be correct version")
  MatX constraintForcesJacobian(int row) const {
    const int n = m.n;
    MatX J(n, n);
    const Contact& ct = cr.contacts[cr.rowContact[row]];
    Vec6 F = mk6(cross(ct.point, cr.dirs[row]), cr.dirs[row]);
    std::vector<Vec6> fg(n);
    for (int wrt = 0; wrt < n; wrt++) fg[wrt] = worldForceGradient(row, wrt);
    for (int r = 0; r < n; r++) {
      s_t mult = multiple(row, r);
      if (mult == 0.0) continue;
      for (int wrt = 0; wrt < n; wrt++) J(r, wrt) = mult * (dot(screwAxisGradient(r, wrt), F) + dot(axis[r], fg[wrt]));
    }
    return J;
  }
};

inline MatX pinvCod(const MatX& Q) {
  const int n = Q.r;
  MatX P(Q.c, n);
  for (int k = 0; k < n; k++) {
    VecX e(n, 0.0);
    e[k] = 1.0;
    VecX x = codSolve(Q, e);
    for (int i = 0; i < Q.c; i++) P(i, k) = x[i];
  }
  return P;
}
inline MatX codSolveMat(const MatX& Q, const MatX& B) {
  MatX X(Q.c, B.c);
  for (int k = 0; k < B.c; k++) {
    VecX b(B.r);
    for (int i = 0; i < B.r; i++) b[i] = B(i, k);
    VecX x = codSolve(Q, b);
    for (int i = 0; i < Q.c; i++) X(i, k) = x[i];
  }
  return X;
}
inline MatX addX(const MatX& a, const MatX& b, s_t sb = 1.0) { MatX o = a; for (size_t i = 0; i < o.d.size(); i++) o.d[i] += sb * b.d[i]; return o; }
inline MatX scaleX(const MatX& a, s_t s) { MatX o = a; for (auto& v : o.d) v *= s; return o; }


// BackpropSnapshot::getBounceApproximationJacobian (:1131-1226), restated literally: with A_b = the A_c columns of the clamping
// rows that bounced (restitution coefficient > 0), W[(j n + k), i] = a_i(j) a_i(k), center = vec(I):
//   q = center - (W^T).completeOrthogonalDecomposition().solve(restitutionDiagonals + W^T center),  X.col(i) = q[i n .. i n + n).
// posPos and velPos are multiplied by X from the right (:1304-1305, :1372-1373).  Identity when nothing bounced.
inline MatX bounceApproximationJacobian(const Model& m, const ContactResult& cr) {
  const int n = m.n;
  std::vector<int> rows;
  for (int j = 0; j < cr.m; j++) if (cr.rowClass.size() && cr.rowClass[j] == RC_CLAMPING && cr.restCoeff[j] > 0) rows.push_back(j);
  const int nb = (int)rows.size();
  if (nb == 0) return identityX(n);
  MatX Wt(nb, n * n);
  VecX rhs(nb, 0.0), center(n * n, 0.0);
  for (int i = 0; i < n; i++) center[i * n + i] = 1.0;
  for (int i = 0; i < nb; i++) {
    for (int j = 0; j < n; j++)
      for (int k = 0; k < n; k++) Wt(i, j * n + k) = cr.Aall(j, rows[i]) * cr.Aall(k, rows[i]);
    s_t wc = 0;
    for (int j = 0; j < n * n; j++) wc += Wt(i, j) * center[j];
    rhs[i] = cr.restCoeff[rows[i]] + wc;
  }
  VecX y = codSolve(Wt, rhs);          // minimum-norm solution of the under-determined system, like Eigen's COD
  MatX X(n, n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) X(j, i) = center[i * n + j] - y[i * n + j];
  return X;
}
// getControlForceVelJacobian / getVelVelJacobian / getPosVelJacobian with clamping constraints
// (BackpropSnapshot.cpp:482-574, 643-759, 762-821, 980-1066, 2723-2774, 2889-3039, 3088-3146, 3657-3747)
inline void contactJacobians(const Model& m, const std::vector<Kin>& kin, const std::vector<Art>& art, const s_t* q,
                             const s_t* v, const s_t* tau, const VecX& vPre, const ContactResult& cr, const MatX& Minv,
                             const VecX& C, const MatX& dCdq, const MatX& dCdv, MatX& forceVel, MatX& velVel, MatX& posVel) {
  (void)art;
  const int n = m.n, nc = cr.numClamping, nu = cr.numUpperBound, mrows = cr.m;
  const s_t dt = m.dt;
  ContactGrad cg(m, kin, q, cr);
  MatX Ac(n, nc), Aub(n, nu);
  std::vector<int> clampRows(nc), ubRows(nu);
  for (int j = 0; j < mrows; j++) {
    if (cr.rowClass[j] == RC_CLAMPING) { clampRows[cr.clampingIndex[j]] = j; for (int i = 0; i < n; i++) Ac(i, cr.clampingIndex[j]) = cr.Aall(i, j); }
    if (cr.rowClass[j] == RC_UPPER_BOUND) { ubRows[cr.upperBoundIndex[j]] = j; for (int i = 0; i < n; i++) Aub(i, cr.upperBoundIndex[j]) = cr.Aall(i, j); }
  }
  const MatX& E = cr.E;
  MatX AcubE = nu > 0 ? addX(Ac, matmul(Aub, E)) : Ac;
  const VecX& f_c = cr.fc;
  MatX D(n, n), K(n, n);
  for (int i = 0; i < n; i++) { D(i, i) = m.damping[i]; K(i, i) = m.spring[i]; }
  MatX I = identityX(n);
  MatX AcT = transposeX(Ac);

  // per-row constraint-force Jacobians (cached like mWorldConstraintJacCache)
  std::vector<MatX> rowJac(mrows);
  for (int j = 0; j < mrows; j++)
    if (cr.rowClass[j] == RC_CLAMPING || cr.rowClass[j] == RC_UPPER_BOUND)
      rowJac[j] = cr.rowContact[j] >= 0 ? cg.constraintForcesJacobian(j) : MatX(n, n);   // a joint-limit row: zero world force, zero Jacobian
  auto jacClamping = [&](const VecX& f0) { MatX r(n, n); for (int i = 0; i < nc; i++) r = addX(r, rowJac[clampRows[i]], f0[i]); return r; };
  auto jacClampingT = [&](const VecX& v0) { MatX r(nc, n); for (int i = 0; i < nc; i++) { VecX row = matTvec(rowJac[clampRows[i]], v0); for (int k = 0; k < n; k++) r(i, k) = row[k]; } return r; };
  auto jacUpper = [&](const VecX& Ef0) { MatX r(n, n); for (int i = 0; i < nu; i++) r = addX(r, rowJac[ubRows[i]], Ef0[i]); return r; };
  auto jacUpperT = [&](const VecX& v0) { MatX r(nu, n); for (int i = 0; i < nu; i++) { VecX row = matTvec(rowJac[ubRows[i]], v0); for (int k = 0; k < n; k++) r(i, k) = row[k]; } return r; };
  // getJacobianOfMinv(f, POSITION) = -Minv * d(M (Minv f))/dq   (Skeleton.cpp:2025-2082)
  auto jacMinv = [&](const VecX& f) { VecX w = matvec(Minv, f); return scaleX(matmul(Minv, jacobianOfMx(m, kin, q, w.data())), -1.0); };

  // Q and its factorisation
  MatX Q = matmul(AcT, matmul(Minv, AcubE));
  for (int i = 0; i < nc; i++) Q(i, i) += cr.cfmRow[clampRows[i]];   // getConstraintForceMixingDiagonal: every group's own constant
  VecX bvec(nc);
  for (int i = 0; i < nc; i++) bvec[i] = cr.b[clampRows[i]];

  // ---- dB for the three wrts (getJacobianOfLCPOffsetClampingSubset :3088-3146) ----
  // bounce diagonals: 1 + restitution coefficient of the clamping row (CGGM.cpp:770, getBounceDiagonals)
  VecX bounce(nc, 1.0);
  for (int i = 0; i < nc; i++) bounce[i] = 1.0 + cr.restCoeff[clampRows[i]];
  auto scaleRows = [&](MatX M_) { for (int i = 0; i < M_.r; i++) for (int k = 0; k < M_.c; k++) M_(i, k) *= bounce[i]; return M_; };
  MatX dvPre_dv = addX(I, scaleX(matmul(Minv, addX(addX(dCdv, D), K, dt)), dt), -1.0);
  MatX dB_vel = scaleRows(scaleX(matmul(AcT, dvPre_dv), -1.0));
  MatX dB_force = scaleRows(scaleX(matmul(AcT, Minv), -dt));
  VecX f(n);
  for (int i = 0; i < n; i++) f[i] = tau[i] - C[i] - m.damping[i] * v[i] - m.spring[i] * (q[i] - m.rest[i] + dt * v[i]);
  MatX dMinv_f = jacMinv(f);
  MatX dAcT_vf = jacClampingT(vPre);
  MatX inner = addX(addX(dMinv_f, matmul(Minv, dCdq), -1.0), matmul(Minv, K), -1.0);
  MatX dB_pos = scaleRows(scaleX(addX(dAcT_vf, scaleX(matmul(AcT, inner), dt)), -1.0));

  // ---- dF_c (getJacobianOfConstraintForce) ----
  MatX dFc_vel = codSolveMat(Q, dB_vel), dFc_force = codSolveMat(Q, dB_force);
  // dQ_b for POSITION (getJacobianOfLCPConstraintMatrixClampingSubset)
  MatX Qinv = pinvCod(Q);
  if (m.pinvNoiseUlps > 0) {                                      // test instrument, see Model::pinvNoiseUlps
    const uint64_t sample = m.pinvNoiseSample++;
    for (int i = 0; i < nc; i++)
      for (int j = 0; j < nc; j++) {
        uint64_t h = m.lcpNoiseSeed * 0x9E3779B97F4A7C15ull + sample * 0xBF58476D1CE4E5B9ull + (uint64_t)(8192 + i * 512 + j) * 0x94D049BB133111EBull;
        h ^= h >> 31; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        Qinv(i, j) *= 1.0 + (s_t)((int)(h % 3) - 1) * (s_t)m.pinvNoiseUlps * 2.220446049250313e-16;
      }
  }
  auto dQ = [&](const VecX& rhs) {
    VecX Ar = matvec(AcubE, rhs);
    MatX t1 = jacClampingT(matvec(Minv, Ar));
    MatX in2 = jacClamping(rhs);
    if (nu > 0) in2 = addX(in2, jacUpper(matvec(E, rhs)));
    MatX t2 = matmul(AcT, addX(jacMinv(Ar), matmul(Minv, in2)));
    return addX(t1, t2);
  };
  auto dQT = [&](const VecX& rhs) {
    if (nu == 0) return dQ(rhs);
    VecX Ar = matvec(Ac, rhs);
    VecX MAr = matvec(Minv, Ar);
    MatX jm = jacMinv(Ar);
    MatX jc = jacClamping(rhs);
    MatX first = addX(jacClampingT(MAr), matmul(AcT, addX(jm, matmul(Minv, jc))));
    MatX second = matmul(transposeX(E), addX(jacUpperT(MAr), matmul(transposeX(Aub), addX(jm, matmul(Minv, jc)))));
    return addX(first, second);
  };
  VecX Qinv_b = codSolve(Q, bvec);
  MatX imprecision = addX(identityX(nc), matmul(Q, Qinv), -1.0);
  s_t impNorm2 = 0;
  for (s_t x : imprecision.d) impNorm2 += x * x;
  MatX dQ_b = scaleX(codSolveMat(Q, dQ(Qinv_b)), -1.0);
  bool imprecise = !(impNorm2 < 1e-18);
  if (imprecise) {
    VecX ib = matvec(imprecision, bvec);
    dQ_b = addX(dQ_b, codSolveMat(Q, matmul(transposeX(Qinv), dQT(ib))));
    MatX IQQ = addX(identityX(nc), matmul(Qinv, Q), -1.0);
    VecX t = matvec(transposeX(Qinv), Qinv_b);
    dQ_b = addX(dQ_b, matmul(IQQ, dQT(t)));
  }
  MatX dFc_pos = addX(dQ_b, codSolveMat(Q, dB_pos));

  // ---- getVelJacobianWrt (:980-1066) ----
  forceVel = matmul(Minv, addX(matmul(AcubE, dFc_force), scaleX(I, dt)));
  MatX velJacVel = addX(I, matmul(Minv, addX(matmul(AcubE, dFc_vel), scaleX(dCdv, dt), -1.0)));
  velVel = addX(addX(velJacVel, scaleX(matmul(Minv, D), dt), -1.0), scaleX(matmul(Minv, K), dt * dt), -1.0);
  VecX r(n);
  VecX Af = matvec(AcubE, f_c);
  for (int i = 0; i < n; i++) r[i] = dt * f[i] + Af[i];
  MatX dM = jacMinv(r);
  MatX dA_c = jacClamping(f_c);
  MatX dA_ubE(n, n);
  if (nu > 0) dA_ubE = jacUpper(matvec(E, f_c));
  MatX innerP = addX(addX(addX(matmul(AcubE, dFc_pos), dA_c), dA_ubE), scaleX(dCdq, dt), -1.0);
  posVel = addX(addX(dM, matmul(Minv, innerP)), scaleX(matmul(Minv, K), dt), -1.0);
}

}  // namespace nbo
