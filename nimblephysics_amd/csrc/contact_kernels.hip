// contact_kernels.hip — contact stage of the batched step (forward), one world per lane.
//
//   k_contact_detect   collision detection at q_t + depth filter      ConstraintSolver.cpp:563-613, DARTCollide.cpp:764-1450
//   k_contact_rows     per-row body wrenches, b, unit-impulse tests -> A and the massed impulse tests M^-1 J^T
//                                                                     ContactConstraint.cpp:66-230, 361-514, 517-607; BoxedLcpConstraintSolver.cpp:190-349
//   k_contact_solve    stage 0 of the LCP cascade: classify the warm start / guess, least-squares
//                      standardisation on the active set, validity check; v' = v_pre + M^-1 J^T x
//                                                                     BoxedLcpConstraintSolver.cpp:434-457, CGGM.cpp:218-339, 482-872, LCPUtils.cpp:12-140
// Lanes whose warm start is not a valid LCP solution need the pivoting / PGS stages
// (BoxedLcpConstraintSolver.cpp:461-677); they are flagged NBL_ST_LCP_FAILED for now and get zero
// impulses, exactly what the reference does when every stage fails (:679-687).
#include "collision_dev.hpp"
#include "dantzig_dev.hpp"
#include "lcp_dev.hpp"

namespace nbl {

DEV double& svAt(double* saved, int row, int64_t B, int64_t b) { return saved[(int64_t)row * B + b]; }
// the world-major dense block of world b (SavedLayout)
DEV double* denseOf(double* saved, const SavedLayout& lay, int64_t B, int64_t b) { return saved + (int64_t)lay.total * B + b * (int64_t)lay.dense; }
DEV LaneMem denseMem(double* saved, const SavedLayout& lay, int64_t B, int64_t b) { LaneMem m; m.base = denseOf(saved, lay, B, b); m.B = 1; m.b = 0; return m; }

// ContactConstraint::getTangentBasisMatrixODE (ContactConstraint.cpp:734-795)
DEV void tangentBasis(V3 n, V3& t1, V3& t2) {
  const double EPS2 = 1e-12;
  V3 t = cross(mk3(0, 0, 1), n);
  if (dot(t, t) < EPS2) {
    t = cross(mk3(1, 0, 0), n);
    if (dot(t, t) < EPS2) {
      t = cross(mk3(0, 1, 0), n);
      if (dot(t, t) < EPS2) t = cross(mk3(0, 0, 1), n);
    }
  }
  t1 = unit3(t);
  t2 = cross(n, t1);
}

__global__ __launch_bounds__(64) void k_contact_detect(DevModel mdl, const DevBody* __restrict__ bodies,
                                                       const DevContactModel* __restrict__ cm, int64_t B,
                                                       double* __restrict__ saved, SavedLayout lay,
                                                       uint32_t* __restrict__ status, double* __restrict__ ws, int doTwists,
                                                       uint32_t* __restrict__ failCount, int ppw) {
  // the counter of the unresolved-worlds list of this slice starts at zero for the solve kernel that follows on the stream
  // (a separate hipMemsetAsync node cost ~6 us of every forward step)
  if (failCount && blockIdx.x == 0 && threadIdx.x == 0) *failCount = 0u;
  NBL_PHASE(56);
  // ppw lanes per world (1, 2 or 4): the narrow phases of ppw collider pairs of a world run side by side, each lane parks its
  // candidate contacts in LDS, and the world's first lane then accepts them in pair order - exactly the order and the filters
  // of the one-lane loop, at about 1 / ppw of its dependent chain (two foot-ground pairs: 74k -> ~40k cycles).
  extern __shared__ __attribute__((aligned(16))) double stage[];   // [thread][8 candidates][CR_SIZE] + counts (ppw > 1 only)
  const int pl = (int)threadIdx.x % ppw, wl = (int)blockDim.x / ppw;
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * wl + (int)threadIdx.x / ppw;
  const bool valid = b < mdl.b1;
  if (!valid && ppw == 1) return;
  const int64_t bs = valid ? b : mdl.b1 - 1;       // lanes of a padding world repeat the last world's narrow phase, store nothing
  Ctx c = makeCtx(mdl, bodies, nullptr, ws, B, bs, saved, &lay);
  __shared__ double keptP[MAX_CONTACTS * 3 * 64];   // accepted contact points of the workgroup's worlds, [contact][xyz][lane]
  __shared__ double clipBuf[48 * 64];               // clip polygons of boxBox, [entry][lane]
  LaneBuf clip; clip.base = clipBuf + threadIdx.x;
  int nC = 0;
  bool overflow = false, edge = false;
  // accept one candidate (lane pl == 0 of the world, or the only lane): postProcess + depth filter + append to the record
  auto acceptRec = [&](const double* ct, int stride, int pi) {   // ct: CR layout, element e at ct[e * stride]
    const V3 pt = mk3(ct[(CR_POINT + 0) * stride], ct[(CR_POINT + 1) * stride], ct[(CR_POINT + 2) * stride]);
    const V3 nr = mk3(ct[(CR_NORMAL + 0) * stride], ct[(CR_NORMAL + 1) * stride], ct[(CR_NORMAL + 2) * stride]);
    const double depth = ct[CR_DEPTH * stride];
    // skip points within 3e-12 of an accepted contact (DARTCollisionDetector.cpp:360-400); the accepted points are kept in
    // LDS (reading them back from the record would be a global round trip per comparison)
    bool close = false;
    for (int e = 0; e < nC; e++) {
      const V3 d = pt - mk3(keptP[(e * 3 + 0) * 64 + threadIdx.x], keptP[(e * 3 + 1) * 64 + threadIdx.x], keptP[(e * 3 + 2) * 64 + threadIdx.x]);
      if (norm3(d) < 3.0e-12) { close = true; break; }
    }
    if (close) return;
    if (dot(nr, nr) < 1e-12) return;
    if (depth < 0.0 || depth > cm->clippingDepth) return;
    if (nC >= cm->maxContacts) { overflow = true; return; }
    const int r0 = lay.contacts + nC * CR_SIZE;
    keptP[(nC * 3 + 0) * 64 + threadIdx.x] = pt.x; keptP[(nC * 3 + 1) * 64 + threadIdx.x] = pt.y; keptP[(nC * 3 + 2) * 64 + threadIdx.x] = pt.z;
#pragma unroll
    for (int e = 0; e < CR_SIZE; e++) {
      double v = ct[e * stride];
      if (e == CR_BOXA) v = (double)cm->pairA[pi];
      if (e == CR_BOXB) v = (double)cm->pairB[pi];
      svAt(saved, r0 + e, B, b) = v;
    }
    if ((int)ct[CR_TYPE * stride] == CT_EDGE_EDGE) edge = true;
    nC++;
  };
  auto toRec = [&](const DevContact& ct, double* out) {   // DevContact -> CR layout (collider indices are filled in by acceptRec)
    out[CR_POINT] = ct.point.x; out[CR_POINT + 1] = ct.point.y; out[CR_POINT + 2] = ct.point.z;
    out[CR_NORMAL] = ct.normal.x; out[CR_NORMAL + 1] = ct.normal.y; out[CR_NORMAL + 2] = ct.normal.z;
    out[CR_DEPTH] = ct.depth; out[CR_TYPE] = (double)ct.type; out[CR_BOXA] = 0.0; out[CR_BOXB] = 0.0;
    out[CR_EA_FIXED] = ct.edgeAFixed.x; out[CR_EA_FIXED + 1] = ct.edgeAFixed.y; out[CR_EA_FIXED + 2] = ct.edgeAFixed.z;
    out[CR_EA_DIR] = ct.edgeADir.x; out[CR_EA_DIR + 1] = ct.edgeADir.y; out[CR_EA_DIR + 2] = ct.edgeADir.z;
    out[CR_EB_FIXED] = ct.edgeBFixed.x; out[CR_EB_FIXED + 1] = ct.edgeBFixed.y; out[CR_EB_FIXED + 2] = ct.edgeBFixed.z;
    out[CR_EB_DIR] = ct.edgeBDir.x; out[CR_EB_DIR + 1] = ct.edgeBDir.y; out[CR_EB_DIR + 2] = ct.edgeBDir.z;
  };
  // narrow phase of collider pair pi, every contact handed to emit(ct)
  auto runPair = [&](int pi, auto emit) {
    const DevBox& ba = cm->boxes[cm->pairA[pi]];
    const DevBox& bb = cm->boxes[cm->pairB[pi]];
    T12 Ta = cT(ba.T), Tb = cT(bb.T);
    if (ba.body >= 0) Ta = mulT(ldTAt(c, ba.body, WS_TW), Ta);
    if (bb.body >= 0) Tb = mulT(ldTAt(c, bb.body, WS_TW), Tb);
    // dispatch on the two shape types (collide(), DARTCollide.cpp:5030-5260)
    const V3 ha = mk3(ba.half[0], ba.half[1], ba.half[2]), hb = mk3(bb.half[0], bb.half[1], bb.half[2]);
    const bool sa = ba.shape == SHAPE_SPHERE, sb = bb.shape == SHAPE_SPHERE;
    if (sa && sb) sphereSphere(ba.half[0], Ta, bb.half[0], Tb, cm->clippingDepth, emit);
    else if (sa) sphereBoxPair(true, ba.half[0], Ta, hb, Tb, cm->clippingDepth, emit);
    else if (sb) sphereBoxPair(false, bb.half[0], Tb, ha, Ta, cm->clippingDepth, emit);
    else boxBox(Ta, ha, Tb, hb, cm->clippingDepth, clip, emit);
  };
  const int nPairs = cm->nPairs;
  if (ppw == 1) {
    for (int pi = 0; pi < nPairs; pi++)
      runPair(pi, [&](const DevContact& ct) { double rec[CR_SIZE]; toRec(ct, rec); acceptRec(rec, 1, pi); });
  } else {
    // the world's first lane accepts the contacts of ITS pair directly (it is the first pair of the group, so the order holds);
    // only the other lanes park theirs: (ppw - 1) staging slots per world
    const int wI = (int)threadIdx.x / ppw;
    double* mine = stage + (size_t)(wI * (ppw - 1) + (pl > 0 ? pl - 1 : 0)) * (8 * CR_SIZE);
    int* counts = reinterpret_cast<int*>(stage + (size_t)wl * (ppw - 1) * (8 * CR_SIZE));
    for (int p0 = 0; p0 < nPairs; p0 += ppw) {       // ppw pairs of every world at a time
      int cnt = 0;
      if (pl == 0) {
        if (valid) runPair(p0, [&](const DevContact& ct) { double rec[CR_SIZE]; toRec(ct, rec); acceptRec(rec, 1, p0); });
      } else {
        if (p0 + pl < nPairs) runPair(p0 + pl, [&](const DevContact& ct) { if (cnt < 8) { toRec(ct, mine + cnt * CR_SIZE); cnt++; } });
        counts[wI * (ppw - 1) + pl - 1] = cnt;
      }
      __syncthreads();
      if (pl == 0 && valid) {
        for (int q = 1; q < ppw && p0 + q < nPairs; q++) {
          const double* src = stage + (size_t)(wI * (ppw - 1) + q - 1) * (8 * CR_SIZE);
          const int cq = counts[wI * (ppw - 1) + q - 1];
          for (int k = 0; k < cq; k++) acceptRec(src + k * CR_SIZE, 1, p0 + q);
        }
      }
      __syncthreads();
    }
    if (pl != 0 || !valid) return;
  }
  NBL_PHASE(59);
  // NOTE: the duplicate filter above only sees contacts that were kept; the reference compares against every
  // contact of the total result including ones later dropped by the depth filter.  Those can only coincide
  // with a kept point if they are the same point, which the depth filter treats identically.
  svAt(saved, lay.nc, B, b) = (double)nC;
  uint32_t st = 0;
  if (nC > 0) st |= 0x1u;
  if (overflow) st |= 0x80u;
  (void)edge;
  if (status) status[b] = st;
  NBL_PHASE(60);
  if (!doTwists || !__any(nC > 0)) return;   // k_step_forward_coop already left the twists
  // body twists at the pre-contact velocity (BodyNode::getSpatialVelocity after integrateVelocities) -> WS_VTW (the dead bias
  // accumulator slot; WS_A keeps the accelerations for the backward pass), for the relative velocities b = -J^T V of the contact-row kernel
  const double* vpre = saved + (int64_t)lay.vpre * B;
  for (int i = 0; i < c.nb; i++) {
    const DevBody& bd = bodies[i];
    V6 V = jointTwist(bd, vpre, B, b);
    if (bd.parent >= 0) V = V + AdInvT(ldT(c, i), ldV6(c, bd.parent, WS_VTW));
    stV6(c, i, WS_VTW, V);
  }
}

// ---------------------------------------------------------------------------------------------
// rows: wrenches, b, impulse tests
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_contact_rows(DevModel mdl, const DevBody* __restrict__ bodies,
                                                     const DevContactModel* __restrict__ cm, int64_t B,
                                                     double* __restrict__ saved, SavedLayout lay, double* __restrict__ ws,
                                                     double* __restrict__ lws) {
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  Ctx c = makeCtx(mdl, bodies, nullptr, ws, B, b, saved, &lay);
  LaneMem L;
  L.base = lws; L.B = B; L.b = b;
  const int n = mdl.n;
  const int nC = (int)svAt(saved, lay.nc, B, b);
  if (!__any(nC > 0)) return;
  const double* vpre = saved + (int64_t)lay.vpre * B;
  double* dn = denseOf(saved, lay, B, b);

  // body twists at the pre-contact velocity: WS_VTW, left by k_contact_detect / k_step_forward_coop
  (void)vpre;
  // per-row body-frame wrenches (mSpatialNormalA/B) and b = -J^T V
  int bodyA[MAX_CONTACTS], bodyB[MAX_CONTACTS];
  for (int ci = 0; ci < MAX_CONTACTS; ci++) {
    bodyA[ci] = -1; bodyB[ci] = -1;
    if (ci >= nC) continue;
    const int r0 = lay.contacts + ci * CR_SIZE;
    V3 p = mk3(svAt(saved, r0 + CR_POINT, B, b), svAt(saved, r0 + CR_POINT + 1, B, b), svAt(saved, r0 + CR_POINT + 2, B, b));
    V3 nrm = mk3(svAt(saved, r0 + CR_NORMAL, B, b), svAt(saved, r0 + CR_NORMAL + 1, B, b), svAt(saved, r0 + CR_NORMAL + 2, B, b));
    const int boxA = (int)svAt(saved, r0 + CR_BOXA, B, b), boxB = (int)svAt(saved, r0 + CR_BOXB, B, b);
    const int bA = cm->boxes[boxA].body, bB = cm->boxes[boxB].body;
    bodyA[ci] = bA; bodyB[ci] = bB;
    V3 t1, t2;
    tangentBasis(nrm, t1, t2);
    V3 d[3] = {nrm, t1, t2};
    for (int k = 0; k < 3; k++) {
      const int row = 3 * ci + k;
      V6 F = mk6(cross(p, d[k]), d[k]);  // world wrench of a unit impulse along d at p
      double rel = 0;
      V6 ja = zero6(), jb = zero6();
      if (bA >= 0) { ja = dAdT(ldTAt(c, bA, WS_TW), F); rel -= dot(ja, ldV6(c, bA, WS_VTW)); }
      if (bB >= 0) { jb = dAdT(ldTAt(c, bB, WS_TW), -F); rel -= dot(jb, ldV6(c, bB, WS_VTW)); }
      double a6[6];
      toArr(ja, a6);
      for (int e = 0; e < 6; e++) L.at(LW_JA + row * 6 + e) = a6[e];
      toArr(jb, a6);
      for (int e = 0; e < 6; e++) L.at(LW_JB + row * 6 + e) = a6[e];
      svAt(saved, lay.b + row, B, b) = rel;   // getRelVelocity; restitution 0, penetration correction off
      // constraint forces in joint space (DCC::getConstraintForces): A_c[i] = sigma_i s_i . F
      const uint64_t mA = bA >= 0 ? cm->ancestors[bA] : 0ull, mB = bB >= 0 ? cm->ancestors[bB] : 0ull;
      for (int i = 0; i < c.nb; i++) {
        const DevBody& bd = bodies[i];
        const bool pa = (mA >> i) & 1ull, pb = (mB >> i) & 1ull;
        const double mult = (pa && pb) ? 0.0 : (pa ? 1.0 : (pb ? -1.0 : 0.0));
        if (bd.jtype != JT_FREE) {
          double val = 0;
          if (mult != 0.0) val = mult * dot(cV6(bd.S), dAdT(ldTAt(c, i, WS_TW), F));
          dn[lay.aall + bd.dofOff * MAX_ROWS + row] = val;
        } else {
          double v6[6] = {0, 0, 0, 0, 0, 0};
          if (mult != 0.0) toArr(dAdT(cT(bd.Tcj), dAdT(ldTAt(c, i, WS_TW), F)), v6);
          for (int e = 0; e < 6; e++) dn[lay.aall + (bd.dofOff + e) * MAX_ROWS + row] = mult * v6[e];
        }
      }
    }
  }
  // ---- unit-impulse tests, three rows (one contact) per pair of sweeps ----
  for (int ci = 0; ci < MAX_CONTACTS; ci++) {
    if (!__any(ci < nC)) break;
    const bool active = ci < nC;
    const int bA = active ? bodyA[ci] : -1, bB = active ? bodyB[ci] : -1;
    const int ACC[3] = {WS_BIMP, WS_FACC, WS_ABAR};
    for (int i = 0; i < c.nb; i++) { zeroN(c, i, WS_BIMP, 6); zeroN(c, i, WS_FACC, 12); }
    // leaf -> root: BodyNode::updateBiasImpulse (BodyNode.cpp:2117-2138)
    for (int i = c.nb - 1; i >= 0; i--) {
      const DevBody& bd = bodies[i];
      T12 T = ldT(c, i);
      V6 Bi[3];
      for (int k = 0; k < 3; k++) {
        Bi[k] = ldV6(c, i, ACC[k]);
        const int row = 3 * ci + k;
        if (i == bA) { double a6[6]; for (int e = 0; e < 6; e++) a6[e] = L.at(LW_JA + row * 6 + e); Bi[k] = Bi[k] - fromArr(a6); }
        if (i == bB) { double a6[6]; for (int e = 0; e < 6; e++) a6[e] = L.at(LW_JB + row * 6 + e); Bi[k] = Bi[k] - fromArr(a6); }
      }
      if (bd.jtype != JT_FREE) {
        V6 S = cV6(bd.S), AIS = ldV6(c, i, WS_AIS);
        double psi = wsAt(c, i, WS_PSI);
        for (int k = 0; k < 3; k++) {
          double uimp = -dot(S, Bi[k]);                    // GenericJoint.hpp:2607-2613 (no joint constraint impulse)
          wsAt(c, i, WS_UIMP + k) = uimp;
          if (bd.parent >= 0) addV6(c, bd.parent, ACC[k], dAdInvT(T, Bi[k] + (psi * uimp) * AIS));  // :2482-2498
        }
      } else {
        const int US[3] = {WS_UIMP, WS_W, WS_VBAR};
        for (int k = 0; k < 3; k++) {
          double pj[6];
          toArr(dAdT(cT(bd.Tcj), Bi[k]), pj);
          for (int e = 0; e < 6; e++) wsAt(c, i, US[k] + e) = -pj[e];
        }
      }
    }
    // root -> leaf: BodyNode::updateVelocityChangeFD (BodyNode.cpp:2188-2215)
    for (int i = 0; i < c.nb; i++) {
      const DevBody& bd = bodies[i];
      T12 T = ldT(c, i);
      if (bd.jtype != JT_FREE) {
        V6 S = cV6(bd.S), AIS = ldV6(c, i, WS_AIS);
        double psi = wsAt(c, i, WS_PSI);
        for (int k = 0; k < 3; k++) {
          V6 X = bd.parent >= 0 ? AdInvT(T, ldV6(c, bd.parent, ACC[k])) : zero6();
          double dq = psi * (wsAt(c, i, WS_UIMP + k) - dot(AIS, X));   // GenericJoint.hpp:2713-2725
          stV6(c, i, ACC[k], X + dq * S);
          if (active) dn[lay.massed + bd.dofOff * MAX_ROWS + 3 * ci + k] = dq;
        }
      } else {
        const int US[3] = {WS_UIMP, WS_W, WS_VBAR};
        S6 AI = ldS6(c, i, WS_AI);
        LDL6 f;
        for (int e = 0; e < 15; e++) f.l[e] = wsAt(c, i, WS_PSI + e);
        for (int e = 0; e < 6; e++) f.d[e] = wsAt(c, i, WS_PSI + 15 + e);
        for (int k = 0; k < 3; k++) {
          V6 X = bd.parent >= 0 ? AdInvT(T, ldV6(c, bd.parent, ACC[k])) : zero6();
          double r[6], pj[6];
          toArr(dAdT(cT(bd.Tcj), mul(AI, X)), pj);
          for (int e = 0; e < 6; e++) r[e] = wsAt(c, i, US[k] + e) - pj[e];
          ldl6Solve(f, r);
          stV6(c, i, ACC[k], X + AdT(cT(bd.Tcj), fromArr(r)));
          if (active) for (int e = 0; e < 6; e++) dn[lay.massed + (bd.dofOff + e) * MAX_ROWS + 3 * ci + k] = r[e];
        }
      }
    }
    // rows 3ci..3ci+2 of A: relative-velocity response of every row of the contacts c2 >= ci; earlier
    // ones mirrored (BoxedLcpConstraintSolver.cpp:250-320)
    if (active) {
      for (int c2 = ci; c2 < nC; c2++) {
        const int b2A = bodyA[c2], b2B = bodyB[c2];
        V6 dVA[3], dVB[3];
        for (int k = 0; k < 3; k++) {
          dVA[k] = b2A >= 0 ? ldV6(c, b2A, ACC[k]) : zero6();
          dVB[k] = b2B >= 0 ? ldV6(c, b2B, ACC[k]) : zero6();
        }
        for (int k2 = 0; k2 < 3; k2++) {
          const int col = 3 * c2 + k2;
          double ja[6], jb[6];
          for (int e = 0; e < 6; e++) { ja[e] = L.at(LW_JA + col * 6 + e); jb[e] = L.at(LW_JB + col * 6 + e); }
          V6 JA = fromArr(ja), JB = fromArr(jb);
          for (int k = 0; k < 3; k++) {
            double val = 0;
            if (b2A >= 0) val += dot(JA, dVA[k]);
            if (b2B >= 0) val += dot(JB, dVB[k]);
            dn[lay.A + (3 * ci + k) * MAX_ROWS + col] = val;
          }
        }
      }
      for (int c2 = 0; c2 < ci; c2++)
        for (int k2 = 0; k2 < 3; k2++)
          for (int k = 0; k < 3; k++)
            dn[lay.A + (3 * ci + k) * MAX_ROWS + 3 * c2 + k2] = dn[lay.A + (3 * c2 + k2) * MAX_ROWS + 3 * ci + k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// stage 0 solve + apply
// ---------------------------------------------------------------------------------------------
DEV void contactOutputs(const LaneMem& SV, const LaneMem& DN, const SavedLayout& lay, int n, int m, const double* X, const Classes& K, double cfm,
                        double* __restrict__ cacheOut, double* __restrict__ nv, int64_t B, int64_t b) {
  SV.at(lay.pflag) = 0.0;   // no pseudo-inverse saved by the one-world-per-lane path
  for (int r = 0; r < MAX_ROWS; r++) {
    SV.at(lay.x + r) = r < m ? X[r] : 0.0;
    SV.at(lay.cls + r) = r < m ? (K.cls[r] == RC_UPPER_BOUND ? (K.E[r] > 0 ? 2.0 : -2.0) : (double)K.cls[r]) : 0.0;
  }
  SV.at(lay.cfm) = cfm;
  if (cacheOut) {
    for (int r = 0; r < MAX_ROWS; r++) cacheOut[(int64_t)r * B + b] = r < m ? X[r] : 0.0;
    cacheOut[(int64_t)MAX_ROWS * B + b] = (double)m;
  }
  // v' = v_pre + M^-1 J^T x   (applyImpulse + computeImpulseForwardDynamics)
  for (int d = 0; d < n; d++) {
    double w = 0;
    for (int r = 0; r < m; r++) w += DN.at(lay.massed + d * MAX_ROWS + r) * X[r];
    SV.at(lay.w + d) = w;
    nv[(int64_t)d * B + b] = SV.at(lay.vpre + d) + w;
  }
}

DEV void loadLcpView(LcpView& V, const LaneMem& SV, const LaneMem& DN, const SavedLayout& lay, const DevContactModel* cm, int nC) {
  V.mem = DN; V.offA = lay.A; V.m = 3 * nC;
  for (int ci = 0; ci < nC; ci++) {
    const int r0 = lay.contacts + ci * CR_SIZE;
    const double muA = cm->boxes[(int)SV.at(r0 + CR_BOXA)].mu, muB = cm->boxes[(int)SV.at(r0 + CR_BOXB)].mu;
    V.mu[ci] = muA < muB ? muA : muB;
  }
}

// The dense per-world matrices (Q / its QR factor and the Cholesky factor of R1 R1^T, 2 x 24 x 24 doubles) are
// staged in LDS: LCP_LANES worlds per workgroup, element e of world l at lds[e * LCP_LANES + l] (conflict-free),
// 16 x 9216 B = 144 KiB of the CU's 160 KiB.  The factorisation is a chain of dependent accesses, so LDS
// latency instead of L2 latency is what matters; the 256 workgroups of a B = 4096 launch cover every CU.
// Worlds whose warm start / guess does not standardise to a valid LCP solution are appended to `failList`
// and finished by k_contact_cascade (compacted slow path).
__global__ __launch_bounds__(LCP_LANES) void k_contact_solve(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                      double* __restrict__ saved, SavedLayout lay,
                                                      const double* __restrict__ cacheIn, double* __restrict__ cacheOut,
                                                      double* __restrict__ next, uint32_t* __restrict__ status,
                                                      double* __restrict__ lws, int32_t* __restrict__ failList,
                                                      uint32_t* __restrict__ failCount) {
  extern __shared__ __attribute__((aligned(16))) double ldsq[];
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  const int n = mdl.n;
  const int nC = (int)svAt(saved, lay.nc, B, b);
  const int m = 3 * nC;
  LaneMem L;
  L.base = ldsq; L.B = (int)blockDim.x; L.b = threadIdx.x;
  LaneMem SV;
  SV.base = saved; SV.B = B; SV.b = b;
  const LaneMem DN = denseMem(saved, lay, B, b);
  double* nv = next + (int64_t)n * B;
  uint32_t st = status ? status[b] : 0u;
  // cache layout: MAX_ROWS values + the row count they belong to
  if (m == 0) {
    if (cacheOut) { for (int r = 0; r < MAX_ROWS; r++) cacheOut[(int64_t)r * B + b] = 0; cacheOut[(int64_t)MAX_ROWS * B + b] = 0; }
    for (int r = 0; r < MAX_ROWS; r++) { SV.at(lay.x + r) = 0; SV.at(lay.cls + r) = 0; }
    SV.at(lay.cfm) = 0; SV.at(lay.pflag) = 0;
    for (int d = 0; d < n; d++) SV.at(lay.w + d) = 0;
    return;
  }
  LcpView V;
  loadLcpView(V, SV, DN, lay, cm, nC);
  double Bv[MAXR], X[MAXR], colNorm[MAXR];
  for (int r = 0; r < m; r++) Bv[r] = SV.at(lay.b + r);
  for (int cc = 0; cc < m; cc++) { double s = 0; for (int r = 0; r < m; r++) { double a = V.A(r, cc); s += a * a; } colNorm[cc] = s; }

  // ---- warm start, or LCPUtils::guessSolution when the cache belongs to another row count; standardise ----
  const bool haveCache = cacheIn && ((int)cacheIn[(int64_t)MAX_ROWS * B + b] == m);
  if (haveCache) { for (int r = 0; r < m; r++) X[r] = cacheIn[(int64_t)r * B + b]; }
  double X0[MAXR];
  Classes K;
  const bool ok = laneStage0(V, L, haveCache, X, X0, Bv, colNorm, K);
  // the pre-solve x (mXBackup) is what the PGS fallback starts from (BoxedLcpConstraintSolver.cpp:541-547)
  for (int r = 0; r < MAX_ROWS; r++) lws[(int64_t)(LW_JA + r) * B + b] = r < m ? X0[r] : 0.0;
  if (ok) {
    st |= 0x2u | 0x100u;
    contactOutputs(SV, DN, lay, n, m, X, K, 0.0, cacheOut, nv, B, b);
  } else {
    const uint32_t slot = atomicAdd(failCount, 1u);
    failList[slot] = (int32_t)b;
  }
  if (status) status[b] = st;
}

// ---------------------------------------------------------------------------------------------
// stages 1-3 of the cascade for the worlds stage 0 could not resolve (compacted list)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LCP_LANES) void k_contact_cascade(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                        double* __restrict__ saved, SavedLayout lay,
                                                        double* __restrict__ cacheOut, double* __restrict__ next,
                                                        uint32_t* __restrict__ status, double* __restrict__ lws,
                                                        const int32_t* __restrict__ failList,
                                                        const uint32_t* __restrict__ failCount) {
  extern __shared__ __attribute__((aligned(16))) double ldsq[];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= *failCount) return;
  const int64_t b = failList[t];
  const int n = mdl.n;
  LaneMem L;
  L.base = ldsq; L.B = (int)blockDim.x; L.b = threadIdx.x;
  LaneMem SV;
  SV.base = saved; SV.B = B; SV.b = b;
  const LaneMem DN = denseMem(saved, lay, B, b);
  const int nC = (int)SV.at(lay.nc);
  const int m = 3 * nC;
  double* nv = next + (int64_t)n * B;
  uint32_t st = status ? status[b] : 0u;
  LcpView V;
  loadLcpView(V, SV, DN, lay, cm, nC);
  double Bv[MAXR], X[MAXR], X0[MAXR], colNorm[MAXR];
  for (int r = 0; r < m; r++) { Bv[r] = SV.at(lay.b + r); X0[r] = lws[(int64_t)(LW_JA + r) * B + b]; X[r] = X0[r]; }
  for (int cc = 0; cc < m; cc++) { double s = 0; for (int r = 0; r < m; r++) { double a = V.A(r, cc); s += a * a; } colNorm[cc] = s; }
  const int OFFA = 0, OFFL = MAXR * MAXR;
  auto loadProblem = [&](RedLcp& P, double cfmDiag, const double* x0) {
    P.n = m; P.nOrig = m;
    for (int i = 0; i < m; i++) {
      P.x[i] = x0[i]; P.b[i] = Bv[i]; P.lo[i] = V.lo(i); P.hi[i] = V.hi(i); P.findex[i] = V.findex(i); P.mapTo[i] = i;
      for (int j = 0; j < m; j++) L.at(OFFA + i * MAXR + j) = V.A(i, j) + (i == j ? cfmDiag : 0.0);
    }
  };
  auto hasNan = [&](const double* x) { bool bad = false; for (int r = 0; r < m; r++) if (x[r] != x[r]) bad = true; return bad; };
  bool success = false, ignoreFriction = false;
  double cfm = 0.0;
  RedLcp P;
  // ---- stage 1: reduce + Dantzig with early termination (:461-522) ----
  loadProblem(P, 0.0, X0);
  lcpReduce(L, OFFA, P);
  if (dantzigSolve(L, OFFA, OFFL, P)) {
    for (int o = 0; o < m; o++) X[o] = P.x[P.mapTo[o]];
    success = lcpValid(V, X, Bv, false, 0.0);
    if (success) st |= 0x4u;
  }
  if (hasNan(X)) { success = false; for (int r = 0; r < m; r++) X[r] = 0; st |= 0x40u; }
  if (!success) {
    cfm = cm->fallbackCfm;
    // ---- stage 2: CFM + PGS from the pre-solve x (:539-597) ----
    loadProblem(P, cfm, X0);
    lcpReduce(L, OFFA, P);
    if (pgsSolve(L, OFFA, P)) {
      for (int o = 0; o < m; o++) X[o] = P.x[P.mapTo[o]];
      success = lcpValid(V, X, Bv, false, cfm);
      if (success) st |= 0x8u;
    }
  }
  if (!success) {
    // ---- stage 3: drop friction, PGS from zero (:606-677) ----
    ignoreFriction = true;
    loadProblem(P, cfm, X0);
    lcpRemoveFriction(L, OFFA, P);
    for (int i = 0; i < P.n; i++) P.x[i] = 0.0;
    const bool ok3 = pgsSolve(L, OFFA, P);
    for (int o = 0; o < m; o++) X[o] = P.mapTo[o] >= 0 ? P.x[P.mapTo[o]] : 0.0;
    st |= 0x10u;
    if (!ok3) st |= 0x20u;
  }
  if (hasNan(X)) { for (int r = 0; r < m; r++) X[r] = 0; st |= 0x40u; }
  // ---- register the fresh solution, classify, standardise (:718-736) ----
  CodFactor F;
  F.ld = MAXR; F.offQR = 0; F.offChol = MAXR * MAXR;
  Classes K;
  if (standardizeLoop(V, L, F, X, Bv, colNorm, cfm, ignoreFriction, 0u, K)) st |= 0x100u;
  contactOutputs(SV, DN, lay, n, m, X, K, cfm, cacheOut, nv, B, b);
  if (status) status[b] = st;
}

}  // namespace nbl
