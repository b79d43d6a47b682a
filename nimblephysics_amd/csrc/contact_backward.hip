// contact_backward.hip — matrix-free adjoint of the contact stage: the derivation, the contact-geometry pieces shared with the
// wavefront-per-world kernels (coop_kernels.hip), and the one-world-per-lane tree sweeps of the memory-lean mode (NBL_SAVE_TREE=0).
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

namespace NBL_NS {

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

// ---- contact-geometry pieces of k_bwd_contact_b_coop ----
struct ContactRec { V3 p, nrm, eAP, eAD, eBP, eBD; int type, bA, bB; double depth, radA, radB; };   // radA / radB: the colliders' radii (spheres, capsules)
template <bool CAPS = false>
DEV ContactRec loadContactRec(const LaneMem& SV, const SavedLayout& lay, const DevContactModel* __restrict__ cm, int ci) {
  const int r0 = lay.contacts + ci * CR_SIZE;
  ContactRec R;
  R.p = mk3(SV.at(r0 + CR_POINT), SV.at(r0 + CR_POINT + 1), SV.at(r0 + CR_POINT + 2));
  R.nrm = mk3(SV.at(r0 + CR_NORMAL), SV.at(r0 + CR_NORMAL + 1), SV.at(r0 + CR_NORMAL + 2));
  R.type = (int)SV.at(r0 + CR_TYPE);
  const int codeA = (int)SV.at(r0 + CR_BOXA), codeB = (int)SV.at(r0 + CR_BOXB);   // (a joint-limit row never reaches the backward pass: its coefficients are zero)
  const DevBox& boxA = cm->boxes[codeA < MAX_BOXES ? codeA : 0];
  const DevBox& boxB = cm->boxes[codeB < MAX_BOXES ? codeB : 0];
  R.bA = boxA.body; R.bB = boxB.body;
  R.depth = 0; R.radA = 0; R.radB = 0;
  if (CAPS) { R.depth = SV.at(r0 + CR_DEPTH); R.radA = boxA.half[0]; R.radB = boxB.half[0]; }
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
struct RowTerms { V6 vertexTerm, faceTerm, edgeTermA, edgeTermB; V3 commonAngular; };   // commonAngular: see CT_EDGE_EDGE below (self-collision)
// CAPS: the model has capsule colliders (their contact types are compiled into that instantiation only: registers)
template <bool CAPS = false>
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
  out.edgeTermA = zero6(); out.edgeTermB = zero6(); out.commonAngular = mk3(0, 0, 0);
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
  // Closest points of two lines (math::getContactPointGradient, Geometry.cpp:1129-1236): the point wA * closestA + wB * closestB
  // moves linearly with the position twist of the DOF carrying line A / line B; adjoints of that map applied to x, accumulated.
  V3 gPa = mk3(0, 0, 0), gDa = gPa, gPb = gPa, gDb = gPa;
  auto lineAdjoint = [&](V3 x, double wA, double wB) {
    const V3 eAP = R.eAP, eAD = R.eAD, eBP = R.eBP, eBD = R.eBD;
    const V3 pv = eBP - eAP;
    const double uaub = dot(eAD, eBD), q1 = dot(eAD, pv), q2 = -dot(eBD, pv), dd = 1 - uaub * uaub;
    if (dd <= 0) { gPa = gPa + wA * x; gPb = gPb + wB * x; return; }
    const double e = 1.0 / dd, N1 = q1 + uaub * q2, N2 = uaub * q1 + q2, alpha = N1 * e, beta = N2 * e;
    const double ca = wA * dot(x, eAD), cb = wB * dot(x, eBD), k2 = 2 * uaub * e * e;
    gPa = gPa + (wA * x + ca * (e * uaub * eBD - e * eAD) + cb * (e * eBD - e * uaub * eAD));
    gDa = gDa + ((wA * alpha) * x + ca * ((k2 * N1 + e * q2) * eBD + e * pv) + cb * ((k2 * N2 + e * q1) * eBD + (e * uaub) * pv));
    gPb = gPb + (wB * x + ca * (e * eAD - e * uaub * eBD) + cb * (e * uaub * eAD - e * eBD));
    gDb = gDb + ((wB * beta) * x + ca * ((k2 * N1 + e * q2) * eAD - (e * uaub) * pv) + cb * ((k2 * N2 + e * q1) * eAD - e * pv));
  };
  if (R.type == CT_EDGE_EDGE) {
    const V3 eAD = R.eAD, eBD = R.eBD;
    const double sgnN = dot(cross(eBD, eAD), nrm) < 0 ? -1.0 : 1.0;
    lineAdjoint(cv, 0.5, 0.5);
    out.edgeTermA = mk6(cross(R.eAP, gPa) + cross(eAD, gDa) + sgnN * cross(eAD, cross(hN, eBD)), gPa);
    out.edgeTermB = mk6(cross(R.eBP, gPb) + cross(eBD, gDb) + sgnN * cross(eBD, cross(eAD, hN)), gPb);
    // Self-collision: a DOF above BOTH bodies moves the contact rigidly in the reference (DofContactType::SELF_COLLISION: dn = w x n,
    // DCC.cpp:594-610).  The two edge models add up to dn = sgn (eB x (w x eA) + (w x eB) x eA) = w x (sgn eB x eA) (Jacobi identity) - the
    // edge normal model differentiates the UNNORMALISED cross product of the annotated edges, which is the contact normal only up to its
    // length for a separating-axis edge contact and not at all for a clipped face contact annotated as an edge contact - so such a DOF is
    // owed dn = w x (n - sgn eB x eA): hN . dn = w . ((n - sgn eB x eA) x hN), added at the lowest common ancestor of the two bodies
    // (k_bwd_contact_b_coop).
    out.commonAngular = cross(nrm - sgnN * cross(eBD, eAD), hN);
    // ... and the point: the edge model moves the midpoint of the closest points of the two edge LINES (math::getContactPoint), the
    // reference's SELF_COLLISION the contact point itself (gradientWrtTheta(point), DCC.cpp:336-340) - the same point for a pure
    // edge-edge contact of the separating-axis test, not for the clipped face contacts that are annotated as edge contacts
    // (DARTCollide.cpp:1280-1381): the DOF is owed dp = w x (p - p_model), cv . dp = w . ((p - p_model) x cv)
    {
      const V3 pv = R.eBP - R.eAP;
      const double uaub = dot(eAD, eBD), q1 = dot(eAD, pv), q2 = -dot(eBD, pv), dd = 1 - uaub * uaub;
      V3 pModel = 0.5 * (R.eAP + R.eBP);
      if (dd > 0) {
        const double e = 1.0 / dd, alpha = (q1 + uaub * q2) * e, beta = (uaub * q1 + q2) * e;
        pModel = 0.5 * ((R.eAP + alpha * eAD) + (R.eBP + beta * eBD));
      }
      out.commonAngular = out.commonAngular + cross(p - pModel, cv);
    }
  }
  // Capsule contacts (DCC.cpp:484-547 point, :819-938 normal).  PIPE_PIPE: the contact point divides the closest points of the two
  // axis lines by the radii, the normal is their difference normalised: the same line map with the weights (1, -1).
  if (CAPS && R.type == CT_PIPE_PIPE) {
    const double rsum = R.radA + R.radB, rA = R.radA / rsum, rB = R.radB / rsum, dist = rsum - R.depth;
    const V3 y = (1.0 / dist) * (hN - dot(nrm, hN) * nrm);
    lineAdjoint(cv, rB / (rA + rB), rA / (rA + rB));
    lineAdjoint(y, 1.0, -1.0);
    out.edgeTermA = mk6(cross(R.eAP, gPa) + cross(R.eAD, gDa), gPa);
    out.edgeTermB = mk6(cross(R.eBP, gPb) + cross(R.eBD, gDb), gPb);
  }
  // SPHERE_PIPE / PIPE_SPHERE (c sphere centre, e pipe direction, f the pipe's fixed point, rel = e . (c - f)):
  //   sphere side: dp = [w + (1 - w) e e^T] g_c,  w = r_pipe / (r_sphere + r_pipe);   dn = sigma (1 - n n^T)(1 - e e^T) g_c / dist
  //   pipe side:   dp = r_sphere / (r_sphere + r_pipe) raw,  raw = g_f + rel (w x e) + ((w x e).(c - f) - e.g_f) e
  //                (math::closestPointOnLineGradient);                                 dn = -sigma (1 - n n^T) raw / dist
  //   sigma = +1 when the sphere is the first object (the normal points from the second object to the first)
  if (CAPS && (R.type == CT_SPHERE_PIPE || R.type == CT_PIPE_SPHERE)) {
    const V3 c = R.eAP, e = R.eAD, f = R.eBP;
    const double dist = R.eBD.x, rs = R.eBD.y, rp = R.eBD.z;
    const bool sphereIsA = R.type == CT_SPHERE_PIPE;
    const double sigma = sphereIsA ? 1.0 : -1.0, wgt = rp / (rs + rp);
    const V3 y = (sigma / dist) * (hN - dot(nrm, hN) * nrm);
    const V3 xs = wgt * cv + ((1 - wgt) * dot(e, cv)) * e + (y - dot(e, y) * e);
    const V6 sphereSide = mk6(cross(c, xs), xs);
    const V3 xp = (rs / (rs + rp)) * cv - y;
    const double rel = dot(e, c - f), xe = dot(xp, e);
    const V6 pipeSide = mk6(cross(f, xp) + rel * cross(e, xp) + xe * cross(e, c), xp - xe * e);
    out.edgeTermA = sphereIsA ? sphereSide : pipeSide;
    out.edgeTermB = sphereIsA ? pipeSide : sphereSide;
  }
  return out;
}

DEV V6 ldField(const Ctx& c, int body, int base, int f) { return ldV6(c, body, base + 6 * f); }

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

}  // namespace NBL_NS
