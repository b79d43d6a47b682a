// contact_backward.hip — matrix-free adjoint of the contact stage (one world per lane).
//
// The reference differentiates  v' = v_pre + M^-1 Abar f_c,  f_c = Q^+ b,  Q = A_c^T M^-1 Abar + cfm I,
// b = -A_c^T v_pre  by forming dense Jacobians (BackpropSnapshot::getVelJacobianWrt :980-1066,
// getJacobianOfConstraintForce :2723-2774, getJacobianOfLCPConstraintMatrixClampingSubset :2889-3039 with
// the full pseudo-inverse derivative, getJacobianOfLCPOffsetClampingSubset :3088-3146,
// getJacobianOfClampingConstraints{,Transpose} :3657-3747, DCC::getConstraintForcesJacobian
// DifferentiableContactConstraint.cpp:1505-1649).  Here the same vector-Jacobian product is evaluated
// without any n x n matrix.  With g = dL/dv':
//   lambda1 = M^-1 g                     fbar = Abar^T lambda1           mu = (Q^+)^T fbar
//   g_vpre  = g - A_c mu                 (feeds the unconstrained backward sweep)
//   dL = sum_k alpha_k^T dQ beta_k + mu^T db + lambda1^T d(M^-1 Abar) f_c   with the three (alpha, beta) pairs of
//        d(Q^+) = -Q^+ dQ Q^+ + Q^+ Q^+T dQ^T (I - Q Q^+) + (I - Q^+ Q) dQ^T Q^+T Q^+ :
//        (-mu, f_c), ((I - Q Q^+) b, Q^+ mu), (Q^+T f_c, fbar - Q^T mu)
//   every q-dependence then reduces to
//     (a) sum_rows (dA_row/dq)^T z_row  with z_row a combination of {lambda1, v_pre, p_k = M^-1 Abar beta_k, s_k = M^-1 A_c alpha_k}
//         (M^-1 of a contact combination is a combination of the saved massed impulse tests), evaluated by
//         walking the ancestor chain of the two contact bodies:  A_row[i] = sigma_i s_i . F  =>
//         d/dq_l = ad(s_l^pos, s_i) . F  [l above i]  +  s_i . dF/dq_l   (vertex / face contact model, DCC.cpp:328-445, 594-735, 1092-1128)
//     (b) -d(adj^T M(q) acc)/dq for four (adj, acc) pairs: one reverse-mode Newton-Euler sweep each (v = 0, no gravity).
#include "lcp_dev.hpp"

namespace nbl {

DEV double qEntry(const LcpView& V, const Classes& K, double cfm, int r, int s) {  // Q[cidx r][cidx s], r and s clamping rows
  double q = V.A(r, s);
  if (K.nu > 0 && (s % 3) == 0)
    for (int u = s + 1; u < s + 3 && u < V.m; u++)
      if (K.cls[u] == RC_UPPER_BOUND) q += K.E[u] * V.A(r, u);
  if (r == s) q += cfm;
  return q;
}

// Recompute the forward tree state, decide whether the contact adjoint is active for this world (any clamping row),
// and, if so, lambda1 = M^-1 g (two tree sweeps) for the dense kernel that follows.
__global__ __launch_bounds__(64) void k_bwd_recompute(DevModel mdl, const DevBody* __restrict__ bodies,
                                                      const DevDof* __restrict__ dofs, int64_t B,
                                                      const double* __restrict__ saved, SavedLayout lay,
                                                      const double* __restrict__ gnext, double* __restrict__ ws,
                                                      double* __restrict__ lws) {
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  Ctx c = makeCtx(mdl, bodies, dofs, ws, B, b, const_cast<double*>(saved), &lay);
  const int n = mdl.n;
  const double* tau = saved + (int64_t)2 * n * B;
  if (lay.treeRows > 0) { for (int i = 0; i < c.nb; i++) { zeroN(c, i, WS_BIMP, 6); zeroN(c, i, WS_FACC, 18); } }   // forward state comes from the record
  else abaSweeps<true>(c, saved, saved + (int64_t)n * B, [&](int d) -> double { return tau[(int64_t)d * B + b]; }, [](int, double) {});
  const double* gvn = gnext + (int64_t)n * B;
  LaneMem L; L.base = lws; L.B = B; L.b = b;
  const int m = 3 * (int)saved[(int64_t)lay.nc * B + b];
  bool active = false;
  for (int r = 0; r < m; r++) if (saved[(int64_t)(lay.cls + r) * B + b] == 1.0) active = true;
  L.at(LB_FLAG) = active ? 1.0 : 0.0;
  if (!active) for (int d = 0; d < n; d++) { L.at(LB_GVP + d) = gvn[(int64_t)d * B + b]; L.at(LB_QX + d) = 0; }
  if (!__any(active)) return;
  // lambda1 = M^-1 g (all lanes of the wave take part in the sweeps)
  minvSweeps(c, [&](int d) -> double { return gvn[(int64_t)d * B + b]; });
  for (int i = 0; i < c.nb; i++) {
    const DevBody& bd = bodies[i];
    for (int k = 0; k < bd.ndof; k++) L.at(LB_LAM1 + bd.dofOff + k) = wsAt(c, i, WS_UIMP + k);
  }
}

// ---- kernel A: dense (c x c) part of the adjoint ----
__global__ __launch_bounds__(LCP_LANES) void k_bwd_contact_a(DevModel mdl, const DevBody* __restrict__ bodies,
                                                      const DevDof* __restrict__ dofs, const DevContactModel* __restrict__ cm,
                                                      int64_t B, double* __restrict__ saved, SavedLayout lay,
                                                      const double* __restrict__ gnext, double* __restrict__ ws,
                                                      double* __restrict__ lws) {
  extern __shared__ __attribute__((aligned(16))) double ldsq[];   // Q factor + Cholesky factor, LCP_LANES worlds (see k_contact_solve)
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  LaneMem QL; QL.base = ldsq; QL.B = (int)blockDim.x; QL.b = threadIdx.x;
  Ctx c = makeCtx(mdl, bodies, dofs, ws, B, b, saved, &lay);
  const int n = mdl.n;
  const double* gvn = gnext + (int64_t)n * B;
  LaneMem L; L.base = lws; L.B = B; L.b = b;
  LaneMem SV; SV.base = saved; SV.B = B; SV.b = b;
  const LaneMem DN = denseMem(saved, lay, B, b);
  const int nC = (int)SV.at(lay.nc);
  const int m = 3 * nC;
  // classes as stored by the forward pass
  Classes K;
  K.nc = 0; K.nu = 0;
  for (int r = 0; r < m; r++) {
    const double cv = SV.at(lay.cls + r);
    K.cidx[r] = -1; K.uidx[r] = -1; K.E[r] = 0; K.cls[r] = RC_NOT_CLAMPING;
    if (cv == 1.0) { K.cls[r] = RC_CLAMPING; K.cidx[r] = K.nc++; }
    else if (cv == 2.0 || cv == -2.0) { K.cls[r] = RC_UPPER_BOUND; K.uidx[r] = K.nu++; }
  }
  if (L.at(LB_FLAG) == 0.0) return;   // set by k_bwd_recompute together with lambda1 (LB_LAM1)
  LcpView V;
  V.mem = DN; V.offA = lay.A; V.m = m;
  for (int ci = 0; ci < nC; ci++) {
    const int r0 = lay.contacts + ci * CR_SIZE;
    const double muA = cm->boxes[(int)SV.at(r0 + CR_BOXA)].mu, muB = cm->boxes[(int)SV.at(r0 + CR_BOXB)].mu;
    V.mu[ci] = muA < muB ? muA : muB;
  }
  for (int r = 0; r < m; r++) if (K.cls[r] == RC_UPPER_BOUND) K.E[r] = SV.at(lay.cls + r) > 0 ? V.hi(r) : V.lo(r);
  const double cfm = SV.at(lay.cfm);
  const int nc = K.nc;
  int rowOf[MAXR];
  for (int r = 0; r < m; r++) if (K.cls[r] == RC_CLAMPING) rowOf[K.cidx[r]] = r;

  double fc[MAXR], bc[MAXR], fbar[MAXR], mu[MAXR], tmp[MAXR];
  // fbar = Abar^T lambda1
  for (int i = 0; i < nc; i++) {
    const int r = rowOf[i];
    fc[i] = SV.at(lay.x + r);
    double s = 0;
    for (int d = 0; d < n; d++) {
      double a = DN.at(lay.aall + d * MAX_ROWS + r);
      if (K.nu > 0 && (r % 3) == 0)
        for (int u = r + 1; u < r + 3 && u < m; u++)
          if (K.cls[u] == RC_UPPER_BOUND) a += K.E[u] * DN.at(lay.aall + d * MAX_ROWS + u);
      s += a * L.at(LB_LAM1 + d);
    }
    fbar[i] = s;
  }
  {
    double Bv[MAXR];
    for (int r = 0; r < m; r++) Bv[r] = SV.at(lay.b + r);
    buildQ(V, K, cfm, QL, 0, Bv, bc);   // Q into LDS, bc = clamping entries of b
  }
  CodFactor F;
  F.ld = MAXR; F.offQR = 0; F.offChol = MAXR * MAXR; F.c = nc;
  codFactor(QL, F);
  for (int i = 0; i < nc; i++) tmp[i] = fbar[i];
  codSolveT(QL, F, tmp, mu);                                   // mu = (Q^+)^T fbar
  double al[3][MAXR], be[3][MAXR], fls[MAXR];
  // Q^+ b: equals the applied impulses f_c when the results were standardised; when the solver's raw x was kept
  // (PGS / frictionless fallback without a valid standardisation) the reference still differentiates Q^+ b here
  // (Qfactored.solve(b), BackpropSnapshot.cpp:2934) while A_c f_c terms use the applied impulses (:1003, 1057-1058)
  for (int i = 0; i < nc; i++) tmp[i] = bc[i];
  codSolve(QL, F, tmp, fls);
  // pair 1: (-mu, Q^+ b)
  for (int i = 0; i < nc; i++) { al[0][i] = -mu[i]; be[0][i] = fls[i]; }
  // pair 2: ((I - Q Q^+) b, Q^+ mu)
  for (int i = 0; i < nc; i++) {
    double s = 0;
    for (int j = 0; j < nc; j++) s += qEntry(V, K, cfm, rowOf[i], rowOf[j]) * fls[j];
    al[1][i] = bc[i] - s;
  }
  for (int i = 0; i < nc; i++) tmp[i] = mu[i];
  codSolve(QL, F, tmp, be[1]);
  // pair 3: (Q^+T Q^+ b, fbar - Q^T mu)
  for (int i = 0; i < nc; i++) tmp[i] = fls[i];
  codSolveT(QL, F, tmp, al[2]);
  for (int i = 0; i < nc; i++) {
    double s = 0;
    for (int j = 0; j < nc; j++) s += qEntry(V, K, cfm, rowOf[j], rowOf[i]) * mu[j];
    be[2][i] = fbar[i] - s;
  }
  // s_k = M^-1 A_c alpha_k, p_k = M^-1 Abar beta_k from the saved massed impulse tests; g_vpre = g - A_c mu
  for (int d = 0; d < n; d++) {
    double sk[3] = {0, 0, 0}, pk[3] = {0, 0, 0}, acmu = 0;
    for (int i = 0; i < nc; i++) {
      const int r = rowOf[i];
      const double ms = DN.at(lay.massed + d * MAX_ROWS + r);
      double mb = ms;
      if (K.nu > 0 && (r % 3) == 0)
        for (int u = r + 1; u < r + 3 && u < m; u++)
          if (K.cls[u] == RC_UPPER_BOUND) mb += K.E[u] * DN.at(lay.massed + d * MAX_ROWS + u);
      for (int k = 0; k < 3; k++) { sk[k] += al[k][i] * ms; pk[k] += be[k][i] * mb; }
      acmu += mu[i] * DN.at(lay.aall + d * MAX_ROWS + r);
    }
    for (int k = 0; k < 3; k++) { L.at(LB_S + k * MAX_DOF_CONTACT + d) = sk[k]; L.at(LB_P + k * MAX_DOF_CONTACT + d) = pk[k]; }
    L.at(LB_GVP + d) = gvn[(int64_t)d * B + b] - acmu;
  }
  // coefficients of z_row on the bases [lambda1, v_pre, p1, p2, p3, s1, s2, s3]
  for (int r = 0; r < MAX_ROWS; r++) {
    double cf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < m) {
      if (K.cls[r] == RC_CLAMPING) {
        const int i = K.cidx[r];
        cf[0] = fc[i]; cf[1] = -mu[i];
        for (int k = 0; k < 3; k++) { cf[2 + k] = al[k][i]; cf[5 + k] = be[k][i]; }
      } else if (K.cls[r] == RC_UPPER_BOUND) {
        const int i = K.cidx[r - (r % 3)];
        cf[0] = K.E[r] * fc[i];
        for (int k = 0; k < 3; k++) cf[5 + k] = K.E[r] * be[k][i];
      }
    }
    for (int k = 0; k < 8; k++) L.at(LB_COEF + r * 8 + k) = cf[k];
  }
}

// ---- contact-geometry pieces shared by the one-world-per-lane and the one-world-per-wavefront kernels ----
struct ContactRec { V3 p, nrm, eAP, eAD, eBP, eBD; int type, bA, bB; };
DEV ContactRec loadContactRec(const LaneMem& SV, const SavedLayout& lay, const DevContactModel* __restrict__ cm, int ci) {
  const int r0 = lay.contacts + ci * CR_SIZE;
  ContactRec R;
  R.p = mk3(SV.at(r0 + CR_POINT), SV.at(r0 + CR_POINT + 1), SV.at(r0 + CR_POINT + 2));
  R.nrm = mk3(SV.at(r0 + CR_NORMAL), SV.at(r0 + CR_NORMAL + 1), SV.at(r0 + CR_NORMAL + 2));
  R.type = (int)SV.at(r0 + CR_TYPE);
  R.bA = cm->boxes[(int)SV.at(r0 + CR_BOXA)].body; R.bB = cm->boxes[(int)SV.at(r0 + CR_BOXB)].body;
  R.eAP = mk3(SV.at(r0 + CR_EA_FIXED), SV.at(r0 + CR_EA_FIXED + 1), SV.at(r0 + CR_EA_FIXED + 2));
  R.eAD = mk3(SV.at(r0 + CR_EA_DIR), SV.at(r0 + CR_EA_DIR + 1), SV.at(r0 + CR_EA_DIR + 2));
  R.eBP = mk3(SV.at(r0 + CR_EB_FIXED), SV.at(r0 + CR_EB_FIXED + 1), SV.at(r0 + CR_EB_FIXED + 2));
  R.eBD = mk3(SV.at(r0 + CR_EB_DIR), SV.at(r0 + CR_EB_DIR + 1), SV.at(r0 + CR_EB_DIR + 2));
  return R;
}
// tangent basis and the pieces of its derivative (ContactConstraint.cpp:734-876)
struct TangentFrame { V3 crs, t1, t2; double tn; bool project; };
DEV TangentFrame tangentFrameOf(V3 nrm) {
  TangentFrame F;
  V3 crs = mk3(0, 0, 1), tng = cross(crs, nrm);
  if (dot(tng, tng) < 1e-12) { crs = mk3(1, 0, 0); tng = cross(crs, nrm);
    if (dot(tng, tng) < 1e-12) { crs = mk3(0, 1, 0); tng = cross(crs, nrm);
      if (dot(tng, tng) < 1e-12) { crs = mk3(0, 0, 1); tng = cross(crs, nrm); } } }
  F.crs = crs; F.tn = norm3(tng);
  F.t1 = (1.0 / F.tn) * tng; F.t2 = cross(nrm, F.t1);
  F.project = fabs(F.tn - 1.0) > 1e-6;
  return F;
}
// Adjoint terms of one contact row (k = 0 normal, 1 / 2 tangents) given Z_all = world twist of body A minus that of
// body B under the joint rates z_row: what a position twist of a DOF on the vertex side / face side / edge A / edge B
// contributes through dF/dq.
struct RowTerms { V6 vertexTerm, faceTerm, edgeTermA, edgeTermB; };
DEV RowTerms contactRowTerms(const ContactRec& R, const TangentFrame& TF, int k, V3 d, V6 Zall) {
  const V3 p = R.p, nrm = R.nrm, t1 = TF.t1, crs = TF.crs;
  const double tn = TF.tn;
  const bool project = TF.project;
  RowTerms out;
  // vertex-type term: Z_all . dF/dq_l = s_l^pos . [p x cv; cv],  cv = d x Z_all.w
  V3 cv = cross(d, Zall.w);
  out.vertexTerm = mk6(cross(p, cv), cv);
  // face-type term: Z_all . dF/dq_l = w_l . a   (dn = w x n, tangents through the basis derivative)
  V3 cc = Zall.v + cross(Zall.w, p);
  V3 aFace;
  auto t1Adjoint = [&](V3 x) -> V3 {   // a with x . dt1(w) = a . w
    V3 xp = project ? x - dot(x, t1) * t1 : x;
    return cross(nrm, (1.0 / tn) * cross(xp, crs));
  };
  if (k == 0) aFace = cross(nrm, cc);
  else if (k == 1) aFace = t1Adjoint(cc);
  else aFace = cross(nrm, cross(t1, cc)) + t1Adjoint(cross(cc, nrm));
  out.faceTerm = mk6(aFace, mk3(0, 0, 0));
  // edge-edge contacts (DCC.cpp:397-424, 700-735; math::getContactPointGradient Geometry.cpp:1129-1236):
  // the contact point is the midpoint of the closest points of the two edge lines, the normal follows
  // +-eB x eA.  Both are linear in the position twist [w; u] of the moving DOF; their adjoints:
  out.edgeTermA = zero6(); out.edgeTermB = zero6();
  V3 hN = mk3(0, 0, 0);   // cc . d(dir) = hN . dn for a normal that moves by dn (direction k follows through the tangent basis)
  if (R.type >= CT_EDGE_EDGE) {
    if (k == 0) hN = cc;
    else if (k == 1) { V3 xp = project ? cc - dot(cc, t1) * t1 : cc; hN = (1.0 / tn) * cross(xp, crs); }
    else { V3 x2 = cross(cc, nrm); V3 xp = project ? x2 - dot(x2, t1) * t1 : x2; hN = cross(t1, cc) + (1.0 / tn) * cross(xp, crs); }
  }
  // Sphere contacts (DCC.cpp:116-228 types, :328-403 point, :626-709 normal).  Both the contact point and the normal are linear
  // in the position twist s = [w; u] of the moving DOF, dp = Mp s and dn = Mn s, so its share is Mp^T cv + Mn^T hN.  With
  // g_x(s) = w x x + u, whose adjoint is x -> [x_pt x x; x]:
  //   sphere side of a sphere-box contact:  dp = P g_c,                dn = sigma (1 - n n^T)(P - 1) g_c / |c - p|
  //   box side:                             dp = g_p - P g_c,          dn = sigma (1 - n n^T) dp / |c - p|
  //   (c the sphere centre, P removes the locked face normals, sigma = +1 for BOX_SPHERE: n = p - c, -1 for SPHERE_BOX)
  //   sphere A of a sphere-sphere contact:  dp = rB/(rA+rB) g_cA,      dn = +(1 - n n^T) g_cA / |cA - cB|   (B: mirrored, -)
  if (R.type == CT_SPHERE_BOX || R.type == CT_BOX_SPHERE) {
    const V3 c = R.eAP, n0 = R.eAD, n1 = R.eBP, n2 = R.eBD;
    auto P = [&](V3 x) -> V3 { return x - dot(n0, x) * n0 - dot(n1, x) * n1 - dot(n2, x) * n2; };
    auto adj = [&](V3 pt, V3 x) -> V6 { return mk6(cross(pt, x), x); };
    const double len = norm3(c - p), invLen = len > 1e-5 ? 1.0 / len : 1.0;
    const double sigma = R.type == CT_BOX_SPHERE ? 1.0 : -1.0;
    const V3 y = (sigma * invLen) * (hN - dot(nrm, hN) * nrm);
    const V6 sphereSide = adj(c, P(cv)) + adj(c, P(y) - y);
    const V6 boxSide = adj(p, cv) - adj(c, P(cv)) + adj(p, y) - adj(c, P(y));
    const bool aIsSphere = R.type == CT_SPHERE_BOX;
    out.edgeTermA = aIsSphere ? sphereSide : boxSide;
    out.edgeTermB = aIsSphere ? boxSide : sphereSide;
  }
  if (R.type == CT_SPHERE_SPHERE) {
    const V3 cA = R.eAP, cB = R.eBP;
    const double rA = R.eAD.x, rB = R.eAD.y, invL = 1.0 / norm3(cA - cB);
    const V3 y = invL * (hN - dot(nrm, hN) * nrm);
    const V3 xa = (rB / (rA + rB)) * cv + y, xb = (rA / (rA + rB)) * cv - y;
    out.edgeTermA = mk6(cross(cA, xa), xa);
    out.edgeTermB = mk6(cross(cB, xb), xb);
  }
  if (R.type == CT_EDGE_EDGE) {
    const V3 eAP = R.eAP, eAD = R.eAD, eBP = R.eBP, eBD = R.eBD;
    const double sgnN = dot(cross(eBD, eAD), nrm) < 0 ? -1.0 : 1.0;
    V3 pv = eBP - eAP;
    const double uaub = dot(eAD, eBD), q1 = dot(eAD, pv), q2 = -dot(eBD, pv), dd = 1 - uaub * uaub;
    V3 gPa, gDa, gPb, gDb;
    if (dd <= 0) { gPa = 0.5 * cv; gDa = mk3(0, 0, 0); gPb = 0.5 * cv; gDb = mk3(0, 0, 0); }
    else {
      const double e = 1.0 / dd, N1 = q1 + uaub * q2, N2 = uaub * q1 + q2, alpha = N1 * e, beta = N2 * e;
      const double ca = dot(cv, eAD), cb = dot(cv, eBD), k2 = 2 * uaub * e * e;
      gPa = 0.5 * (cv + ca * (e * uaub * eBD - e * eAD) + cb * (e * eBD - e * uaub * eAD));
      gDa = 0.5 * (alpha * cv + ca * ((k2 * N1 + e * q2) * eBD + e * pv) + cb * ((k2 * N2 + e * q1) * eBD + (e * uaub) * pv));
      gPb = 0.5 * (cv + ca * (e * eAD - e * uaub * eBD) + cb * (e * uaub * eAD - e * eBD));
      gDb = 0.5 * (beta * cv + ca * ((k2 * N1 + e * q2) * eAD - (e * uaub) * pv) + cb * ((k2 * N2 + e * q1) * eAD - e * pv));
    }
    out.edgeTermA = mk6(cross(eAP, gPa) + cross(eAD, gDa) + sgnN * cross(eAD, cross(hN, eBD)), gPa);
    out.edgeTermB = mk6(cross(eBP, gPb) + cross(eBD, gDb) + sgnN * cross(eBD, cross(eAD, hN)), gPb);
  }
  return out;
}

// ---- kernel B: tree part of the adjoint ----
DEV V6 ldField(const Ctx& c, int body, int base, int f) { return ldV6(c, body, base + 6 * f); }

__global__ __launch_bounds__(64) void k_bwd_contact_b(DevModel mdl, const DevBody* __restrict__ bodies,
                                                      const DevDof* __restrict__ dofs, const DevContactModel* __restrict__ cm,
                                                      int64_t B, double* __restrict__ saved, SavedLayout lay,
                                                      double* __restrict__ ws, double* __restrict__ lws,
                                                      uint32_t* __restrict__ gradStatus) {
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  Ctx c = makeCtx(mdl, bodies, dofs, ws, B, b, saved, &lay);
  const int n = mdl.n;
  LaneMem L; L.base = lws; L.B = B; L.b = b;
  LaneMem SV; SV.base = saved; SV.B = B; SV.b = b;
  const bool active = L.at(LB_FLAG) != 0.0;
  if (!__any(active)) return;
  const double* q = saved;
  // ---- pass 1 (root->leaf): twist fields of the nine joint-rate vectors ----
  for (int i = 0; i < c.nb; i++) {
    const DevBody& bd = bodies[i];
    T12 T = ldT(c, i), TW = ldTAt(c, i, WS_TW);
    for (int f = 0; f < NFIELD; f++) {
      auto rate = [&](int d) -> double {
        if (f == 0) return L.at(LB_LAM1 + d);
        if (f == 1) return SV.at(lay.vpre + d);
        if (f <= 4) return L.at(LB_P + (f - 2) * MAX_DOF_CONTACT + d);
        if (f <= 7) return L.at(LB_S + (f - 5) * MAX_DOF_CONTACT + d);
        return SV.at(lay.w + d);
      };
      V6 tw;
      if (bd.jtype == JT_FREE) {
        const int o = bd.dofOff;
        tw = AdT(cT(bd.Tcj), mk6(mk3(rate(o), rate(o + 1), rate(o + 2)), mk3(rate(o + 3), rate(o + 4), rate(o + 5))));
      } else tw = rate(bd.dofOff) * cV6(bd.S);
      if (bd.parent >= 0) tw = tw + AdInvT(T, ldField(c, bd.parent, WS_FB, f));
      stV6(c, i, WS_FB + 6 * f, tw);
      if (f < 8) stV6(c, i, WS_FW + 6 * f, AdT(TW, tw));
    }
    zeroN(c, i, WS_PAIRF, 48);
    zeroN(c, i, WS_XI, 6);
  }
  // ---- pass 2 (leaf->root): -d(adj^T M acc)/dq for (lambda1, w), (s_k, p_k) ----
  const int ADJ[4] = {0, 5, 6, 7}, ACC[4] = {8, 2, 3, 4};
  for (int d = 0; d < n; d++) L.at(LB_QX + d) = 0.0;
  for (int i = c.nb - 1; i >= 0; i--) {
    const DevBody& bd = bodies[i];
    T12 T = ldT(c, i);
    S6 G = cS6(bd.G);
    V6 xi = zero6();
    for (int k = 0; k < 4; k++) {
      V6 adj = ldField(c, i, WS_FB, ADJ[k]), acc = ldField(c, i, WS_FB, ACC[k]);
      V6 Fk = mul(G, acc) + ldV6(c, i, WS_PAIRF + 6 * k);
      V6 Ak = mul(G, adj) + ldV6(c, i, WS_PAIRA + 6 * k);
      if (bd.parent >= 0) {
        xi = xi + dad(AdInvT(T, ldField(c, bd.parent, WS_FB, ADJ[k])), Fk) + dad(AdInvT(T, ldField(c, bd.parent, WS_FB, ACC[k])), Ak);
        addV6(c, bd.parent, WS_PAIRF + 6 * k, dAdInvT(T, Fk));
        addV6(c, bd.parent, WS_PAIRA + 6 * k, dAdInvT(T, Ak));
      }
    }
    double qb[6];
    applyHt(bd, q, B, b, xi, qb);
    if (active) for (int k = 0; k < bd.ndof; k++) L.at(LB_QX + bd.dofOff + k) -= qb[k];
  }
  // ---- pass 3: contact geometry, per row walk up from body A and body B ----
  uint32_t gst = 0;
  if (active) {
    const int nC = (int)SV.at(lay.nc);
    for (int ci = 0; ci < nC; ci++) {
      const ContactRec CR = loadContactRec(SV, lay, cm, ci);
      const V3 p = CR.p, nrm = CR.nrm;
      const int type = CR.type, bA = CR.bA, bB = CR.bB;
      if (bA >= 0 && bB >= 0 && (cm->ancestors[bA] & cm->ancestors[bB])) gst |= 0x2u;  // self-collision chains unsupported
      const TangentFrame TF = tangentFrameOf(nrm);
      V3 dirs[3] = {nrm, TF.t1, TF.t2};
      for (int k = 0; k < 3; k++) {
        const int row = 3 * ci + k;
        double cf[8];
        bool any = false;
        for (int e = 0; e < 8; e++) { cf[e] = L.at(LB_COEF + row * 8 + e); any = any || cf[e] != 0.0; }
        if (!any) continue;
        V3 d = dirs[k];
        V6 Fw = mk6(cross(p, d), d);
        auto twistOf = [&](int body) -> V6 {   // world twist of `body` under the joint rates z_row
          V6 z = zero6();
          if (body < 0) return z;
          for (int e = 0; e < 8; e++) if (cf[e] != 0.0) z = z + cf[e] * ldField(c, body, WS_FW, e);
          return z;
        };
        V6 TA = twistOf(bA), TB = twistOf(bB);
        const RowTerms RT = contactRowTerms(CR, TF, k, d, TA - TB);
        const V6 vertexTerm = RT.vertexTerm, faceTerm = RT.faceTerm, edgeTermA = RT.edgeTermA, edgeTermB = RT.edgeTermB;
        const bool aIsVertex = (type == CT_VERTEX_FACE);
        for (int side = 0; side < 2; side++) {
          const int start = side == 0 ? bA : bB;
          const V6 Tend = side == 0 ? TA : TB;
          const double sgn = side == 0 ? 1.0 : -1.0;
          const bool vertexSide = (side == 0) == aIsVertex;
          for (int l = start; l >= 0; l = bodies[l].parent) {
            const int par = bodies[l].parent;
            V6 Zl = sgn * (Tend - twistOf(par));
            V6 add = -dad(Zl, Fw);
            if (type == CT_VERTEX_FACE || type == CT_FACE_VERTEX) add = add + (vertexSide ? vertexTerm : faceTerm);
            else if (type >= CT_EDGE_EDGE) add = add + (side == 0 ? edgeTermA : edgeTermB);   // edge-edge and the sphere types
            addV6(c, l, WS_XI, add);
          }
        }
      }
    }
  }
  for (int i = 0; i < c.nb; i++) {
    const DevBody& bd = bodies[i];
    V6 xiW = ldV6(c, i, WS_XI);
    double qb[6];
    applyHt(bd, q, B, b, dAdT(ldTAt(c, i, WS_TW), xiW), qb);
    if (active) for (int k = 0; k < bd.ndof; k++) L.at(LB_QX + bd.dofOff + k) += qb[k];
  }
  if (gradStatus && gst) atomicOr(gradStatus, gst);
}

// ---- kernel C: unconstrained backward sweep driven by g_vpre, plus the contact position cotangent ----
__global__ __launch_bounds__(64) void k_bwd_final(DevModel mdl, const DevBody* __restrict__ bodies,
                                                  const DevDof* __restrict__ dofs, int64_t B,
                                                  const double* __restrict__ saved, SavedLayout lay,
                                                  const double* __restrict__ gnext,
                                                  double* __restrict__ gstate, double* __restrict__ gaction,
                                                  double* __restrict__ ws, const double* __restrict__ lws, int treeInWs) {
  (void)treeInWs;   // kept slots come from the workspace (k_tree_to_lanes) when lay carries no tree block
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  Ctx c = makeCtx(mdl, bodies, dofs, ws, B, b, const_cast<double*>(saved), &lay);
  const int n = mdl.n;
  const double* q = saved;
  const double* v = saved + (int64_t)n * B;
  const double* tau = saved + (int64_t)2 * n * B;
  auto gvp = [&](int d) -> double { return lws[(int64_t)(LB_GVP + d) * B + b]; };
  auto qx = [&](int d) -> double { return lws[(int64_t)(LB_QX + d) * B + b]; };
  minvSweeps(c, [&](int d) -> double { return c.dt * gvp(d); });
  reverseSweep(c, q, v, tau, gnext, gvp, qx, gstate, gstate + (int64_t)n * B, gaction);
}

}  // namespace nbl
