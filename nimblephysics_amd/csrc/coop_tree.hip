// coop_tree.hip — the tree sweeps with ONE WORLD PER WAVEFRONT, lane = body.
//
// The one-world-per-lane tree kernels (kernels.hip) are a ~2e4-instruction dependent chain per wave whose duration does
// not depend on the batch size up to B ~ 16k: at B = 4096 they keep < 10 % of the chip busy.  Here a wavefront owns one
// world, lane i owns body i, the per-body state of the sweeps lives in LDS (lds[slot * nbp + body], conflict-free), the
// bodies of one tree level run together (level-synchronous) and children add to their parent one sibling rank at a time.
// The sweep code itself is the single-source code of kernels.hip (abaSweeps / minvSweeps / reverseSweep instantiated for
// CoopCtx).  The forward tree state is exchanged with the backward pass through the world-major tree block of the saved
// record ([slot][nbp] per world = the LDS image, copied with coalesced 8-byte-per-lane streams).
#include "coop_wave_dev.hpp"

namespace NBL_NS {

constexpr int TREE_WPB_MAX = 8;   // worlds (wavefronts) per workgroup: they share one LDS copy of the model constants

// LDS of a workgroup: [DevBody x nb][DevDof x n][wpb x (coopRows<P> x nbp + nFree x coopFreeExtra<P> doubles)].  With lane = body
// the model constants are indexed per lane, so the scalar-load path of the one-world-per-lane kernels is gone; a
// per-workgroup LDS copy keeps them at LDS latency instead of 15 divergent global loads per field.
template <int P> __host__ __device__ constexpr int coopWorldDoubles(int nbp, int nFree) { return coopRows<P>() * nbp + nFree * coopFreeExtra<P>(); }
// returns false for a padding world
template <int P>
DEV bool coopTreeSetup(CoopCtxT<P>& c, const DevModel& mdl, const DevBody* __restrict__ bodies, const DevDof* __restrict__ dofs,
                       double* lds, int64_t B, uint32_t bid = blockIdx.x, uint32_t nblk = gridDim.x) {
  DevBody* lb = reinterpret_cast<DevBody*>(lds);
  DevDof* ld = reinterpret_cast<DevDof*>(lb + mdl.nb);
  double* st = reinterpret_cast<double*>(ld + mdl.n);
  {
    double* dstB = reinterpret_cast<double*>(lb);
    const double* srcB = reinterpret_cast<const double*>(bodies);
    const int cntB = mdl.nb * (int)(sizeof(DevBody) / sizeof(double));
    const int cntD = mdl.n * (int)(sizeof(DevDof) / sizeof(double));
    // DevBody[] and DevDof[] are contiguous in LDS: one copy loop, 4 loads in flight per thread (their latency is paid once)
    const double* srcD = reinterpret_cast<const double*>(dofs);
    const int cnt = cntB + cntD, step = (int)blockDim.x;
    for (int i0 = (int)threadIdx.x; i0 < cnt; i0 += 4 * step) {
      double tmp[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int idx = i0 + u * step;
        if (idx < cnt) tmp[u] = idx < cntB ? srcB[idx] : srcD[idx - cntB];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int idx = i0 + u * step;
        if (idx < cnt) dstB[idx] = tmp[u];
      }
    }
  }
  __syncthreads();   // the only workgroup-wide barrier: from here on every wavefront runs on its own
  // A wavefront carries wpw = mdl.pad worlds (host: 64 / nbp, e.g. 4 for a 16-body model): lanes [s * nbp, (s + 1) * nbp) are
  // the bodies of its s-th world, each world with its own LDS image.  The sweeps only ever talk to LDS through (c.lds, body)
  // and to HBM through (c.b, dof), so they are unaware of the packing; it divides the wave-instructions per world by wpw
  // (with one world per wave 15 of 64 lanes did the arithmetic of the metric model).
  const int tl = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
  const int wpw = mdl.pad > 0 ? mdl.pad : 1;
  int sub = tl / mdl.nbp;
  const bool spare = sub >= wpw;            // lanes beyond the last packed world (64 not a multiple of nbp): idle
  if (spare) sub = wpw - 1;
  const int64_t first = mdl.b0 + (coopWorld(bid, nblk) * wpb + wv) * wpw;
  // worlds past the end of the slice repeat its last world (same inputs, same values stored to the same addresses) instead of
  // branching around every store
  const int64_t b = first + sub < mdl.b1 ? first + sub : mdl.b1 - 1;
  const int perWorld = coopWorldDoubles<P>(mdl.nbp, mdl.nFree);
  c.bodies = lb; c.dofs = ld; c.lds = st + (size_t)(wv * wpw + sub) * perWorld; c.ldsFree = c.lds + coopRows<P>() * mdl.nbp;
  c.nbp = mdl.nbp; c.B = B; c.b = b;
  c.nb = mdl.nb; c.n = mdl.n; c.dt = mdl.dt;
  c.g = mk3(mdl.gravity[0], mdl.gravity[1], mdl.gravity[2]);
  c.lane = spare ? 63 + mdl.nb : tl - sub * mdl.nbp;   // body index; >= nb: no body
  const bool on = c.lane < mdl.nb;
  c.level = on ? lb[c.lane].level : -1;
  c.rank = on ? lb[c.lane].rank : -1;
  c.maxLevel = mdl.maxLevel; c.maxRank = mdl.maxRank;
  return first < mdl.b1;
}
DEV double* treeBlock(double* saved, const SavedLayout& lay, int64_t B, int64_t b) {
  return saved + ((int64_t)lay.total + lay.dense) * B + b * (int64_t)lay.treeRows;
}
// kept slots: LDS image (compact rows + free-joint blocks) <-> the compact tree block of the record (treeCompactRow, model_dev.hpp).
// Lanes are laid over (row, body) so that a wavefront moves 64 / nbp rows per trip without integer divisions in the loop.
template <int P, bool STORE>
DEV void coopCopyTree(const CoopCtxT<P>& c, double* blk) {
  // every packed world moves its own image: lane = body, U rows in flight per lane so that the global-load (LDS-read) latency
  // is paid once per U rows.  Kept rows of the profile: FWD rows 0..70, BWD rows 0..30 (coopRowSlot).
  constexpr int KEPT = P == PROF_FWD ? 71 : 31, U = 8;
  const int body = c.lane;
  if (body < c.nbp) {
    // the articulated inertia (21 of the forward profile's 71 kept rows) has ONE reader per tree: the 6 x 6 solve of a free-joint root
    // (kernels.hip, k_contact_rows_coop's free-joint block) - the other bodies' rows stay in LDS (30 % of the block's write traffic)
    const bool keepsAI = !(STORE && P == PROF_FWD) || (body < c.nb && c.bodies[body].jtype == JT_FREE);
    const int fiMine = body < c.nb ? c.bodies[body].freeIdx : -1;
#pragma unroll        // (fully unrolled: the row -> slot -> block-row maps fold to constants)
    for (int r0 = 0; r0 < KEPT; r0 += U) {
      double tmp[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = r0 + u;
        if (r < KEPT) tmp[u] = STORE ? c.lds[r * c.nbp + body] : blk[treeCompactRow(coopRowSlot<P>(r)) * c.nbp + body];   // (the rows the backward profile keeps are rows of the block)
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = r0 + u;
        if (r < KEPT) {
          if (STORE) {
            const int cr = treeCompactRow(coopRowSlot<P>(r));
            if (cr >= 0) blk[cr * c.nbp + body] = tmp[u];
            else if (keepsAI && P == PROF_FWD) blk[TREE_ROWS * c.nbp + fiMine * TREE_FREE + (-1 - cr)] = tmp[u];   // AI of a free-joint body
          } else c.lds[r * c.nbp + body] = tmp[u];
        }
      }
    }
  }
  for (int fb = 0; fb < c.nb; fb++) {
    const int fi = c.bodies[fb].freeIdx;
    if (fi < 0 || body >= c.nbp) continue;
    for (int e = body; e < coopFreeExtra<P>(); e += c.nbp) {
      double* at = blk + TREE_ROWS * c.nbp + fi * TREE_FREE + (-1 - treeCompactRow(coopFreeSlot<P>(e)));   // (every slot of the free-joint LDS block lives in the free-joint part)
      if (STORE) *at = c.ldsFree[fi * coopFreeExtra<P>() + e];
      else c.ldsFree[fi * coopFreeExtra<P>() + e] = *at;
    }
  }
}
template <int P>
DEV void coopStoreTree(const CoopCtxT<P>& c, double* saved, const SavedLayout& lay) {
  waveFence();
  coopCopyTree<P, true>(c, treeBlock(saved, lay, c.B, c.b));
}
template <int P>
DEV void coopLoadTree(const CoopCtxT<P>& c, const double* saved, const SavedLayout& lay) {
  coopCopyTree<P, false>(c, treeBlock(const_cast<double*>(saved), lay, c.B, c.b));
  waveFence();
}

// ---- Featherstone ABA in the WORLD frame (lane = body) -----------------------------------------------------------------
// In one common frame the articulated inertias and bias forces of the children ADD into the parent without the congruence /
// coadjoint transforms of the body-frame recursion (GenericJoint.hpp:2168-2185, 2395-2421), and twists / accelerations pass
// from parent to child unchanged.  The heavy per-body work (joint transform, world inertia Ad*^-1 G Ad^-1, own bias force) is
// done once per body by all lanes together; what stays inside the level-synchronous loops is ~250 instructions per level
// instead of ~950.  Scalars (psi, u, qdd) are frame invariant, so the results equal the body-frame sweeps of kernels.hip to
// round-off; the kept slots are written in the body-frame convention every consumer expects (V, A, AI*S converted once at the
// end; AI only for the free-joint root, the only body whose AI is read later).  A free joint is a tree root and is handled in
// its body frame exactly like abaSweeps does.
//   scratch slots reused for world-frame data of the lane's body: WS_AI = AI^W (accumulated), WS_FACC = B^W (accumulated),
//   WS_W = twist V^W, WS_VBAR = acceleration A^W  (parent <-> child exchange; everything else lives in registers)
template <class TauFn, class EmitFn>
DEV void abaSweepsWorld(const CoopCtxT<PROF_FWD>& c, const double* __restrict__ q, const double* __restrict__ v, TauFn tauAt, EmitFn emit) {
  const int64_t B = c.B, b = c.b;
  const int i = c.lane;
  const bool on = i < c.nb;
  const DevBody& bd = c.bodies[on ? i : 0];
  const bool isFree = bd.jtype == JT_FREE;
  NBL_PHASE(1);
  // ---- joint transform (all bodies together) ----
  T12 T;
  if (on) {
    T12 Q;
    if (bd.jtype == JT_REVOLUTE) {
      const double qi = q[bd.dofOff * B + b];
      Q.R = expAngular(mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi));  // RevoluteJoint.cpp:203-211
      Q.p = mk3(0, 0, 0);
    } else if (bd.jtype == JT_PRISMATIC) {
      const double qi = q[bd.dofOff * B + b];
      Q.R = eye3();
      Q.p = mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi);
    } else if (bd.jtype == JT_SCREW) {                   // expMap([axis; h axis] q) = (R(axis q), h axis q), ScrewJoint.cpp:217-232
      const double qi = q[bd.dofOff * B + b], hq = bd.screwRate * qi;
      Q.R = expAngular(mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi));
      Q.p = mk3(bd.axis[0] * hq, bd.axis[1] * hq, bd.axis[2] * hq);
    } else if (bd.jtype == JT_FREEC) {
      // a free joint below the root: six coincident axes at zero displacement, the first body carries [exp(q_r), q_p] (FreeJoint.cpp:74-81)
      const int o = bd.dofOff;
      Q.R = bd.ballComp == 0 ? expMapRot(mk3(q[(o + 0) * B + b], q[(o + 1) * B + b], q[(o + 2) * B + b])) : eye3();
      Q.p = bd.ballComp == 0 ? mk3(q[(o + 3) * B + b], q[(o + 4) * B + b], q[(o + 5) * B + b]) : mk3(0, 0, 0);
    } else if (bd.jtype == JT_BALL) {
      // the x body of the triple carries the joint rotation exp(q) (BallJoint.cpp:91-95, 422-438); the y and z bodies sit on it at zero angle
      Q.R = bd.ballComp == 0 ? expMapRot(mk3(q[(bd.dofOff + 0) * B + b], q[(bd.dofOff + 1) * B + b], q[(bd.dofOff + 2) * B + b])) : eye3();
      Q.p = mk3(0, 0, 0);
    } else {
      Q.R = expMapRot(mk3(q[(bd.dofOff + 0) * B + b], q[(bd.dofOff + 1) * B + b], q[(bd.dofOff + 2) * B + b]));  // FreeJoint.cpp:74-81
      Q.p = mk3(q[(bd.dofOff + 3) * B + b], q[(bd.dofOff + 4) * B + b], q[(bd.dofOff + 5) * B + b]);
    }
    T = mulT(mulT(cT(bd.Tpj), Q), cT(bd.TcjInv));
    stT(c, i, T);
  }
  const V6 SdqB = on ? jointTwist(bd, v, B, b) : zero6();   // S dq in the body frame
  NBL_PHASE(2);
  // ---- sweep 1 (root -> leaf): world transforms and twists ----
  // The "world frame" of every spatial quantity below has its origin at the ROOT of the lane's tree, not at the world's: moments about
  // a far origin would cost digits with the SQUARE of the distance (inertias m p^2 against the body's own, which the joint projections
  // must recover by cancellation; 1 km from the origin: 1e-7).  A pure translation of the frame, one per tree (trees do not exchange
  // anything here): sums over bodies, twists passed down and wrenches passed up stay transform-free.  BodyNode::mWorldTransform itself
  // is kept as it is (WS_TW).
  waveFence();
  const V3 worldOrigin = ldT(c, bd.root).p;
  T12 TW = T;
  V6 Vw = zero6(), SdqW = zero6();
  forBodiesDown(c, [&](int) {
    if (bd.parent >= 0) TW = mulT(ldTAt(c, bd.parent, WS_TW), T);
    stTAt(c, i, WS_TW, TW);                                  // BodyNode::mWorldTransform
    TW.p = TW.p - worldOrigin;                               // from here on: the lane's body in the shifted frame
    SdqW = AdT(TW, SdqB);
    Vw = bd.parent >= 0 ? ldV6(c, bd.parent, WS_W) + SdqW : SdqW;
    stV6(c, i, WS_W, Vw);
  });
  NBL_PHASE(3);
  // ---- own inertia and bias in the world frame (all bodies together) ----
  const V6 Sw = (on && !isFree) ? AdT(TW, cV6(bd.S)) : zero6();
  const V6 etaW = ad(Vw, SdqW);                               // GenericJoint.hpp:1803-1824 (dS = 0); zero for the root
  if (on) {
    const S6 Gw = congruenceToParent(TW, cS6(bd.G));
    stS6(c, i, WS_AI, Gw);
    stV6(c, i, WS_FACC, -dad(Vw, mul(Gw, Vw)));               // BodyNode.cpp:2076-2114; gravity rides on the base acceleration
  }
  waveFence();
  NBL_PHASE(4);
  // ---- sweep 2 (leaf -> root): articulated inertias, bias forces, joint-space total force ----
  V6 AISw = zero6();
  double psi = 0.0, u = 0.0;
  // joint-space applied force of the lane's 1-DOF joint, fetched before the level loop (a global load inside it would be
  // waited for once per level).  GenericJoint.hpp:2554-2571: spring uses q - q0 + dt*v, damping explicit
  double u0 = 0.0;
  if (on && !isFree) {
    const int d = bd.dofOff;
    const DevDof& df = c.dofs[d];
    const double qd = q[d * B + b], vd = v[d * B + b];
    u0 = tauAt(d) - df.spring * (qd - df.rest + vd * c.dt) - df.damping * vd;
  }
  forBodiesUp(c, [&](int) {
    S6 AI = ldS6(c, i, WS_AI);
    const V6 Bf = ldV6(c, i, WS_FACC);
    const V6 AIeta = mul(AI, etaW);
    if (!isFree) {
      AISw = mul(AI, Sw);
      psi = 1.0 / dot(Sw, AISw);                              // GenericJoint.hpp:2276-2301
      u = u0 - dot(Sw, AIeta + Bf);
      wsAt(c, i, WS_PSI) = psi;
      wsAt(c, i, WS_U) = u;
      if (bd.parent >= 0) {
        const V6 beta = Bf + AIeta + (psi * u) * AISw;        // GenericJoint.hpp:2395-2421
        rank1Sub(AI, AISw, psi);                              // PI = AI - AIS psi AIS^T  (GenericJoint.hpp:2168-2185)
        parentTurn(c, [&]() { addS6(c, bd.parent, WS_AI, AI); addV6(c, bd.parent, WS_FACC, beta); });
      }
    } else {
      // free joint as a tree root, in its body frame (abaSweeps): projected inertia S^T AI S with S = Ad(T_cj)
      const T12 TWi = invT(TW);
      const S6 AIb = congruenceToParent(TWi, AI);
      stS6(c, i, WS_AI, AIb);
      const LDL6 f = ldl6(congruenceToParent(cT(bd.TcjInv), AIb));
#pragma unroll
      for (int k = 0; k < 15; k++) wsAt(c, i, WS_PSI + k) = f.l[k];
#pragma unroll
      for (int k = 0; k < 6; k++) wsAt(c, i, WS_PSI + 15 + k) = f.d[k];
      const V6 proj = dAdT(cT(bd.Tcj), dAdT(TW, AIeta + Bf));   // S^T (AI eta + B), body frame
      double pj[6];
      toArr(proj, pj);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int d = bd.dofOff + k;
        const DevDof& df = c.dofs[d];
        const double qd = q[d * B + b], vd = v[d * B + b];
        wsAt(c, i, WS_U + k) = tauAt(d) - df.spring * (qd - df.rest + vd * c.dt) - df.damping * vd - pj[k];
      }
    }
  });
  NBL_PHASE(5);
  // ---- sweep 3 (root -> leaf): accelerations ----
  const V6 a0 = mk6(mk3(0, 0, 0), -c.g);
  V6 Aw = zero6();
  double qdd = 0.0;
  forBodiesDown(c, [&](int) {
    const V6 Ap = bd.parent >= 0 ? ldV6(c, bd.parent, WS_VBAR) : a0;
    if (!isFree) {
      qdd = psi * (u - dot(AISw, Ap));                        // GenericJoint.hpp:2656-2676
      Aw = Ap + etaW + qdd * Sw;
    } else {
      const V6 XA = AdInvT(TW, Ap);
      const S6 AIb = ldS6(c, i, WS_AI);
      LDL6 f;
#pragma unroll
      for (int k = 0; k < 15; k++) f.l[k] = wsAt(c, i, WS_PSI + k);
#pragma unroll
      for (int k = 0; k < 6; k++) f.d[k] = wsAt(c, i, WS_PSI + 15 + k);
      const V6 proj = dAdT(cT(bd.Tcj), mul(AIb, XA));
      double r[6], pj[6];
      toArr(proj, pj);
#pragma unroll
      for (int k = 0; k < 6; k++) r[k] = wsAt(c, i, WS_U + k) - pj[k];
      ldl6Solve(f, r);
#pragma unroll
      for (int k = 0; k < 6; k++) emit(bd.dofOff + k, r[k]);
      Aw = AdT(TW, XA + AdInvT(TW, etaW) + AdT(cT(bd.Tcj), fromArr(r)));
    }
    stV6(c, i, WS_VBAR, Aw);
  });
  if (on && !isFree) {                                        // all 1-DOF joints together: the load of v is waited for once
    double acc = qdd;
    if (bd.jtype == JT_BALL) {
      // the chain's axes turn with its own rates, the ball joint's do not (constant S in the child frame): the child accelerates
      // equally in both when qdd_ball = qdd_chain + (wy wz, -wx wz, wx wy), the Lie brackets of the chain's own axis velocities
      const int d0 = bd.dofOff - bd.ballComp;
      const double wx = v[(int64_t)d0 * B + b], wy = v[(int64_t)(d0 + 1) * B + b], wz = v[(int64_t)(d0 + 2) * B + b];
      acc += bd.ballComp == 0 ? wy * wz : (bd.ballComp == 1 ? -wx * wz : wx * wy);
    } else if (bd.jtype == JT_FREEC) {
      // the same for the six axes of a free joint: the angular brackets, and w x u for the translations (all rotations come first)
      const int d0 = bd.dofOff - bd.ballComp, cmp = bd.ballComp;
      const V3 wv = mk3(v[(int64_t)d0 * B + b], v[(int64_t)(d0 + 1) * B + b], v[(int64_t)(d0 + 2) * B + b]);
      const V3 uv = mk3(v[(int64_t)(d0 + 3) * B + b], v[(int64_t)(d0 + 4) * B + b], v[(int64_t)(d0 + 5) * B + b]);
      acc += cmp < 3 ? pick3(mk3(wv.y * wv.z, -wv.x * wv.z, wv.x * wv.y), cmp) : pick3(cross(wv, uv), cmp - 3);
    }
    emit(bd.dofOff, acc);
  }
  NBL_PHASE(6);
  // ---- kept slots in the body-frame convention of the consumers ----
  if (on) {
    stV6(c, i, WS_V, AdInvT(TW, Vw));
    stV6(c, i, WS_A, AdInvT(TW, Aw));
    if (!isFree) stV6(c, i, WS_AIS, dAdT(TW, AISw));
  }
  waveFence();
}

// the ABA of the forward step: world frame with lane = body, body frame with lane = world
template <class TauFn, class EmitFn>
DEV void stepAba(const CoopCtxT<PROF_FWD>& c, const double* __restrict__ q, const double* __restrict__ v, TauFn tauAt, EmitFn emit) {
  abaSweepsWorld(c, q, v, tauAt, emit);
}

// World::step without contact + (contact models) the body twists at the pre-contact velocity
DEV void stepForwardCoopBody(const DevModel& mdl, const DevBody* __restrict__ bodies, const DevDof* __restrict__ dofs, int64_t B,
                             const double* __restrict__ state, const double* __restrict__ action, double* __restrict__ next,
                             double* __restrict__ saved, uint32_t* __restrict__ status, const SavedLayout& lay, int withTwists,
                             double* ldsTree, uint32_t bid, uint32_t nblk) {
  CoopCtxT<PROF_FWD> c;
  NBL_PHASE(0);
  if (!coopTreeSetup(c, mdl, bodies, dofs, ldsTree, B, bid, nblk)) return;
  const int64_t b = c.b;
  bodies = c.bodies; dofs = c.dofs;   // the LDS copies
  stepForwardCore(c, state, action, next, saved, lay);
  NBL_PHASE(8);
  if (withTwists) {
    // BodyNode::getSpatialVelocity after integrateVelocities -> WS_VTW, for b = -J^T V of the contact rows.  Each lane
    // reads back the new velocities of its own body's DOFs, which it stored itself (program order of one lane).
    const double* nv = next + (int64_t)mdl.n * B;
    const int i = c.lane;
    const bool on = i < c.nb;
    T12 TW = on ? ldTAt(c, i, WS_TW) : T12();
    if (on) TW.p = TW.p - ldTAt(c, bodies[i].root, WS_TW).p;           // frame origin at the root of the tree (see abaSweepsWorld)
    V6 Vw = on ? AdT(TW, jointTwist(bodies[i], nv, B, b)) : zero6();   // own joint twist, world frame
    forBodiesDown(c, [&](int) {                                          // world twists are prefix sums down the tree
      if (bodies[i].parent >= 0) Vw = Vw + ldV6(c, bodies[i].parent, WS_W);
      stV6(c, i, WS_W, Vw);
    });
    if (on) stV6(c, i, WS_VTW, AdInvT(TW, Vw));
    waveFence();
  }
  NBL_PHASE(9);
  if (saved && lay.treeRows > 0) coopStoreTree(c, saved, lay);
  if (status) {
    // NBL_ST_NAN for worlds whose unconstrained step is not finite (poisoned inputs, a singular model): every lane looks at what it wrote
    // for its own body's DOFs, the lanes of a world vote
    bool bad = false;
    if (c.lane < c.nb) {
      const DevBody& bd = bodies[c.lane];
      for (int k = 0; k < bd.ndof; k++) {
        const int64_t d = bd.dofOff + k;
        bad = bad || !__builtin_isfinite(next[d * B + b]) || !__builtin_isfinite(next[((int64_t)mdl.n + d) * B + b]);
      }
    }
    const uint64_t votes = (uint64_t)__ballot(bad ? 1 : 0);
    const int grp = (int)(threadIdx.x & 63u) / c.nbp;
    const uint64_t gmask = c.nbp >= 64 ? ~0ull : ((1ull << c.nbp) - 1ull) << (grp * c.nbp);
    if (c.lane == 0) status[b] = (votes & gmask) != 0ull ? 0x40u : 0u;
  }
  NBL_PHASE(10);
}
__global__ __launch_bounds__(64 * TREE_WPB_MAX) NBL_WAVES(NBL_W_FWD) void k_step_forward_coop(DevModel mdl, const DevBody* __restrict__ bodies,
                                                          const DevDof* __restrict__ dofs, int64_t B,
                                                          const double* __restrict__ state, const double* __restrict__ action,
                                                          double* __restrict__ next, double* __restrict__ saved,
                                                          uint32_t* __restrict__ status, SavedLayout lay, int withTwists) {
  extern __shared__ __attribute__((aligned(16))) double ldsTree[];
  stepForwardCoopBody(mdl, bodies, dofs, B, state, action, next, saved, status, lay, withTwists, ldsTree, blockIdx.x, gridDim.x);
}

// The forward tree kernel AND the narrow phase in one launch (models with colliders, the default): the first `nDetect` workgroups run the
// narrow phase of wl worlds each (contactDetectBody with its own forward kinematics of the collider bodies - contacts are detected at
// q_t, they need nothing the tree sweeps compute), the others the tree sweeps.  The two are independent, the narrow phase is a
// ~40 us chain of one lane per collider pair on 1/8 of the chip, and as a launch of its own it stood between the tree kernel and the
// contact-row kernel of every step; streams cannot provide the overlap (a fifth stream in flight halves the throughput).
__global__ __launch_bounds__(64 * TREE_WPB_MAX) NBL_WAVES(NBL_W_FWD) void k_forward_detect_coop(DevModel mdl, const DevBody* __restrict__ bodies,
                                                          const DevDof* __restrict__ dofs, const DevContactModel* __restrict__ cm, int64_t B,
                                                          const double* __restrict__ state, const double* __restrict__ action,
                                                          double* __restrict__ next, double* __restrict__ saved,
                                                          uint32_t* __restrict__ status, SavedLayout lay, double* __restrict__ ws,
                                                          uint32_t* __restrict__ failCount, int ppw, int wl, int nDetect) {
  extern __shared__ __attribute__((aligned(16))) double ldsTree[];
  if ((int)blockIdx.x < nDetect) {
    NBL_PHASE_FIRST(18);
    const int ls = wl * ppw;                              // narrow-phase lanes of the workgroup
#if NBL_GENERAL
    const int seenPts = 2 * cm->maxContacts < SEEN_POINTS ? 2 * cm->maxContacts : SEEN_POINTS;   // (contactDetectBody: by the MODEL's slots)
#else
    constexpr int seenPts = SEEN_POINTS;
#endif
    double* keptP = ldsTree;                              // seenPts * 3 * ls
    double* clipBuf = keptP + seenPts * 3 * ls;           // 48 * ls
    double* stage = clipBuf + 48 * ls;
    // the body constants of the forward kinematics from LDS (the two feet of a world sit on different lanes: indexed per lane, the
    // constants would be ~190 dependent global loads per lane)
    const size_t stageDoubles = ppw > 1 ? (size_t)wl * (ppw - 1) * (8 * CR_SIZE) + ((size_t)wl * (ppw - 1) + 1) / 2 : 0;
    DevBody* lb = reinterpret_cast<DevBody*>(stage + ((stageDoubles + 1) & ~(size_t)1));
    {
      double* dst = reinterpret_cast<double*>(lb);
      const double* src = reinterpret_cast<const double*>(bodies);
      const int cnt = mdl.nb * (int)(sizeof(DevBody) / sizeof(double));
      for (int i = (int)threadIdx.x; i < cnt; i += (int)blockDim.x) dst[i] = src[i];
    }
    // ... and the collider model (pairs -> colliders -> bodies -> ancestor chains are dependent loads, per lane: through global memory they
    // were 30 k of the 70 k cycles of a lane's narrow phase)
    DevContactModel* lcm = reinterpret_cast<DevContactModel*>(lb + mdl.nb);
    {
      static_assert(sizeof(DevContactModel) % sizeof(double) == 0 && sizeof(DevBody) % sizeof(double) == 0, "copied as doubles");
      double* dst = reinterpret_cast<double*>(lcm);
      const double* src = reinterpret_cast<const double*>(cm);
      for (int i = (int)threadIdx.x; i < (int)(sizeof(DevContactModel) / sizeof(double)); i += (int)blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    double* fkT = reinterpret_cast<double*>(lcm + 1);      // [wl][bodies on the collider chains][12] joint transforms (contactDetectBody)
    contactDetectBody(mdl, lb, lcm, B, saved, lay, status, ws, 0, failCount, ppw, state, (int)blockIdx.x, wl, keptP, clipBuf, stage, fkT, ls);
    return;
  }
  stepForwardCoopBody(mdl, bodies, dofs, B, state, action, next, saved, status, lay, 1, ldsTree, blockIdx.x - (uint32_t)nDetect,
                      gridDim.x - (uint32_t)nDetect);
}

// ---- backward sweeps in the WORLD frame (lane = body) ---------------------------------------------------------------------
// The lane's body in world coordinates, built once from the body-frame kept slots of the tree block.
struct WorldBody {
  T12 TW;
  V6 Sw, AISw, Vw, Aw;   // joint axis, AI*S, twist, acceleration (1-DOF joints; Sw / AISw unused for the free root)
  double psi;
  bool on, isFree;
};
// scratch slots of the backward sweeps (LDS, lane = body): WS_W = W^W (twist of lambda), WS_BIMP = impulse accumulator,
// WS_FACC / WS_ABAR / WS_VBAR = adjoint accumulators, WS_UIMP = V^W and WS_BACC = A^W for the children's reads
DEV WorldBody loadWorldBody(const CoopCtxT<PROF_BWD>& c) {
  WorldBody wb;
  const int i = c.lane;
  wb.on = i < c.nb;
  const DevBody& bd = c.bodies[wb.on ? i : 0];
  wb.isFree = bd.jtype == JT_FREE;
  wb.Sw = wb.AISw = wb.Vw = wb.Aw = zero6();
  wb.psi = 0.0;
  if (wb.on) {
    wb.TW = ldTAt(c, i, WS_TW);
    wb.TW.p = wb.TW.p - ldTAt(c, bd.root, WS_TW).p;          // frame origin at the root of the tree (see abaSweepsWorld)
    wb.Vw = AdT(wb.TW, ldV6(c, i, WS_V));
    wb.Aw = AdT(wb.TW, ldV6(c, i, WS_A));
    if (!wb.isFree) { wb.Sw = AdT(wb.TW, cV6(bd.S)); wb.AISw = dAdInvT(wb.TW, ldV6(c, i, WS_AIS)); wb.psi = wsAt(c, i, WS_PSI); }
  }
  return wb;
}

// lambda = M^-1 rhs (minvSweeps of kernels.hip) in the world frame: impulses add up the tree and twists pass down it
// without transforms.  lam[k]: the lane's DOFs; W^W is left in WS_W for the children and returned.
template <class RhsFn>
DEV V6 minvSweepsWorld(const CoopCtxT<PROF_BWD>& c, const WorldBody& wb, RhsFn rhsAt, double (&lam)[6]) {
  const int i = c.lane;
  const DevBody& bd = c.bodies[wb.on ? i : 0];
  forBodies(c, [&](int) { zeroN(c, i, WS_BIMP, 6); });
  double uimp[6] = {0, 0, 0, 0, 0, 0};
  double rhs[6] = {0, 0, 0, 0, 0, 0};                   // the lane's right-hand sides, fetched before the level loop
  if (wb.on) {
    if (!wb.isFree) rhs[0] = rhsAt(bd.dofOff);
    else {
#pragma unroll
      for (int k = 0; k < 6; k++) rhs[k] = rhsAt(bd.dofOff + k);
    }
  }
  forBodiesUp(c, [&](int) {
    const V6 Bi = ldV6(c, i, WS_BIMP);
    if (!wb.isFree) {
      uimp[0] = rhs[0] - dot(wb.Sw, Bi);
      if (bd.parent >= 0) {
        const V6 up = Bi + (wb.psi * uimp[0]) * wb.AISw;
        parentTurn(c, [&]() { addV6(c, bd.parent, WS_BIMP, up); });
      }
    } else {
      double pj[6];
      toArr(dAdT(cT(bd.Tcj), dAdT(wb.TW, Bi)), pj);
#pragma unroll
      for (int k = 0; k < 6; k++) uimp[k] = rhs[k] - pj[k];
    }
  });
  V6 Ww = zero6();
  forBodiesDown(c, [&](int) {
    const V6 Wp = bd.parent >= 0 ? ldV6(c, bd.parent, WS_W) : zero6();
    if (!wb.isFree) {
      lam[0] = wb.psi * (uimp[0] - dot(wb.AISw, Wp));
      Ww = Wp + lam[0] * wb.Sw;
    } else {
      const S6 AI = ldS6(c, i, WS_AI);
      LDL6 f;
#pragma unroll
      for (int k = 0; k < 15; k++) f.l[k] = wsAt(c, i, WS_PSI + k);
#pragma unroll
      for (int k = 0; k < 6; k++) f.d[k] = wsAt(c, i, WS_PSI + 15 + k);
      double r[6], pj[6];
      toArr(dAdT(cT(bd.Tcj), mul(AI, AdInvT(wb.TW, Wp))), pj);
#pragma unroll
      for (int k = 0; k < 6; k++) r[k] = uimp[k] - pj[k];
      ldl6Solve(f, r);
#pragma unroll
      for (int k = 0; k < 6; k++) lam[k] = r[k];
      Ww = Wp + AdT(wb.TW, AdT(cT(bd.Tcj), fromArr(r)));
    }
    stV6(c, i, WS_W, Ww);
  });
  return Ww;
}

// Reverse-mode Newton-Euler sweep at (q, v, qdd) with joint adjoint lambda (reverseSweep of kernels.hip) in the world frame.
// Everything that does not involve the children's accumulations (three G-products, the ad / dad terms, the projections, the
// per-DOF epilogue including the free joint's exp/log VJP) is done by all bodies together; the level loop only adds the
// accumulators, forms F / Abar / Vbar and hands them to the parent.
template <class GvFn, class GvPreFn, class QxFn>
DEV void reverseSweepWorld(const CoopCtxT<PROF_BWD>& c, const WorldBody& wb, V6 Ww, const double (&lam)[6], const double* __restrict__ q,
                           const double* __restrict__ v, const double* __restrict__ tau, const double* __restrict__ gqn, GvFn gvAt,
                           GvPreFn gvPreAt, QxFn qExtraAt, double* __restrict__ gq, double* __restrict__ gv, double* __restrict__ gaction) {
  const int64_t B = c.B, b = c.b;
  const int i = c.lane;
  const DevBody& bd = c.bodies[wb.on ? i : 0];
  const DevDof* dofs = c.dofs;
  V6 Floc = zero6(), AbarLoc = zero6(), VbarLoc = zero6(), SdqW = zero6();
  if (wb.on) {
    const S6 Gw = congruenceToParent(wb.TW, cS6(bd.G));
    const V6 GV = mul(Gw, wb.Vw);
    Floc = mul(Gw, wb.Aw) - dad(wb.Vw, GV);                  // transmitted force at (q, v, qdd), own part
    AbarLoc = mul(Gw, Ww);
    VbarLoc = dad(Ww, GV) - mul(Gw, ad(wb.Vw, Ww));
    SdqW = AdT(wb.TW, jointTwist(bd, v, B, b));
    zeroN(c, i, WS_FACC, 18);
    stV6(c, i, WS_UIMP, wb.Vw);
    stV6(c, i, WS_BACC, wb.Aw);
  }
  waveFence();
  NBL_PHASE(25);
  V6 F = zero6(), Abar = zero6(), Vbar = zero6();
  forBodiesUp(c, [&](int) {
    F = Floc + ldV6(c, i, WS_FACC);
    Abar = AbarLoc + ldV6(c, i, WS_ABAR);
    Vbar = VbarLoc - dad(SdqW, Abar) + ldV6(c, i, WS_VBAR);
    if (bd.parent >= 0) parentTurn(c, [&]() { addV6(c, bd.parent, WS_FACC, F); addV6(c, bd.parent, WS_ABAR, Abar); addV6(c, bd.parent, WS_VBAR, Vbar); });
  });
  NBL_PHASE(26);
  const V6 a0 = mk6(mk3(0, 0, 0), -c.g);
  V6 xi = zero6(), tmp = zero6();
  if (wb.on) {
    const V6 Vp = bd.parent >= 0 ? ldV6(c, bd.parent, WS_UIMP) : zero6();
    const V6 Ap = bd.parent >= 0 ? ldV6(c, bd.parent, WS_BACC) : a0;
    const V6 Wp = bd.parent >= 0 ? ldV6(c, bd.parent, WS_W) : zero6();
    xi = dAdT(wb.TW, dad(Wp, F) + dad(Ap, Abar) + dad(Vp, Vbar));   // adjoint of the joint transform, body frame
    tmp = dAdT(wb.TW, dad(wb.Vw, Abar) + Vbar);                     // body frame
  }
  // a ball joint's positions act through the transform of the x body of its triple only: the y and z bodies take its adjoint
  // (handed over in the accumulator slots, which the sweep above has consumed)
  const bool isBall = wb.on && bd.jtype == JT_BALL, isFreeC = wb.on && bd.jtype == JT_FREEC;
  if ((isBall || isFreeC) && bd.ballComp == 0) stV6(c, i, WS_FACC, xi);
  waveFence();
  if ((isBall || isFreeC) && bd.ballComp > 0) xi = ldV6(c, i - bd.ballComp, WS_FACC);
  if (!wb.on) return;
  double qb[6], vb[6], pp[6], vp[6];
  applyHt(bd, q, B, b, xi, qb);
  const int o = bd.dofOff;
  double ballExtra = 0.0;                       // velocity cotangent through the closed-form acceleration term of a ball joint
  if (isBall) {
    vb[0] = dot(cV6(bd.S), tmp);
    const int d0 = o - bd.ballComp, cmp = bd.ballComp;
    const V3 r = mk3(q[(int64_t)(d0 + 0) * B + b], q[(int64_t)(d0 + 1) * B + b], q[(int64_t)(d0 + 2) * B + b]);
    const V3 w = mk3(v[(int64_t)(d0 + 0) * B + b], v[(int64_t)(d0 + 1) * B + b], v[(int64_t)(d0 + 2) * B + b]);
    const V3 grn = mk3(gqn[(int64_t)(d0 + 0) * B + b], gqn[(int64_t)(d0 + 1) * B + b], gqn[(int64_t)(d0 + 2) * B + b]);
    V3 posr, velw;
    so3IntegrationVjp(r, w, c.dt, grn, posr, velw);
    pp[0] = cmp == 0 ? posr.x : (cmp == 1 ? posr.y : posr.z);
    vp[0] = cmp == 0 ? velw.x : (cmp == 1 ? velw.y : velw.z);
    // v' = v + dt (qdd_chain + delta(w)),  delta = (wy wz, -wx wz, wx wy)
    const double g0 = gvPreAt(d0), g1 = gvPreAt(d0 + 1), g2 = gvPreAt(d0 + 2);
    ballExtra = c.dt * (cmp == 0 ? (-w.z * g1 + w.y * g2) : (cmp == 1 ? (w.z * g0 + w.x * g2) : (w.y * g0 - w.x * g1)));
  } else if (isFreeC) {
    vb[0] = dot(cV6(bd.S), tmp);
    const int d0 = o - bd.ballComp, cmp = bd.ballComp;
    auto at3 = [&](const double* x, int k0) { return mk3(x[(int64_t)(d0 + k0) * B + b], x[(int64_t)(d0 + k0 + 1) * B + b], x[(int64_t)(d0 + k0 + 2) * B + b]); };
    const V3 wv = at3(v, 0), uv = at3(v, 3);
    double posT[6], velT[6];
    se3IntegrationVjp(at3(q, 0), wv, uv, c.dt, at3(gqn, 0), at3(gqn, 3), posT, velT);
#pragma unroll
    for (int k = 0; k < 6; k++) if (k == cmp) { pp[0] = posT[k]; vp[0] = velT[k]; }
    // v' = v + dt (qdd_chain + delta),  delta = [(wy wz, -wx wz, wx wy); w x u]
    const V3 ga = mk3(gvPreAt(d0), gvPreAt(d0 + 1), gvPreAt(d0 + 2)), gl = mk3(gvPreAt(d0 + 3), gvPreAt(d0 + 4), gvPreAt(d0 + 5));
    const V3 dW = mk3(-wv.z * ga.y + wv.y * ga.z, wv.z * ga.x + wv.x * ga.z, wv.y * ga.x - wv.x * ga.y) + cross(uv, gl);
    const V3 dU = cross(gl, wv);
    ballExtra = c.dt * (cmp < 3 ? pick3(dW, cmp) : pick3(dU, cmp - 3));
  } else if (!wb.isFree) {
    vb[0] = dot(cV6(bd.S), tmp);
    pp[0] = gqn[(int64_t)o * B + b];            // posPos = 1, velPos = dt  (GenericJoint.hpp:1428-1444)
    vp[0] = c.dt * pp[0];
  } else {
    toArr(dAdT(cT(bd.Tcj), tmp), vb);
    V3 r = mk3(q[(o + 0) * B + b], q[(o + 1) * B + b], q[(o + 2) * B + b]);
    V3 w = mk3(v[(o + 0) * B + b], v[(o + 1) * B + b], v[(o + 2) * B + b]);
    V3 vl = mk3(v[(o + 3) * B + b], v[(o + 4) * B + b], v[(o + 5) * B + b]);
    M3 R = expMapRot(r);
    // VJP of q' = [logMap(R E); p + R vl dt]  (exact reverse-mode of FreeJoint.cpp:922-929, see reverseSweep)
    V3 grn = mk3(gqn[(o + 0) * B + b], gqn[(o + 1) * B + b], gqn[(o + 2) * B + b]);
    V3 gpn = mk3(gqn[(o + 3) * B + b], gqn[(o + 4) * B + b], gqn[(o + 5) * B + b]);
    M3 E = expMapRot(c.dt * w);
    M3 Rn = mul(R, E);
    M3 Rnb = logMap_vjp(Rn, grn);
    M3 Rb = mulABt(Rnb, E);                 // dL/dR from R' = R E
    M3 Eb = mulAtB(R, Rnb);
    V3 vdt = c.dt * vl;
    Rb.m[0] += gpn.x * vdt.x; Rb.m[1] += gpn.x * vdt.y; Rb.m[2] += gpn.x * vdt.z;   // p' = p + R vdt
    Rb.m[3] += gpn.y * vdt.x; Rb.m[4] += gpn.y * vdt.y; Rb.m[5] += gpn.y * vdt.z;
    Rb.m[6] += gpn.z * vdt.x; Rb.m[7] += gpn.z * vdt.y; Rb.m[8] += gpn.z * vdt.z;
    V3 posr = expMapRot_vjp(r, Rb);
    V3 velw = c.dt * expMapRot_vjp(c.dt * w, Eb);
    V3 vell = c.dt * tmul(R, gpn);
    pp[0] = posr.x; pp[1] = posr.y; pp[2] = posr.z; pp[3] = gpn.x; pp[4] = gpn.y; pp[5] = gpn.z;   // posPos^T gq'
    vp[0] = velw.x; vp[1] = velw.y; vp[2] = velw.z; vp[3] = vell.x; vp[4] = vell.y; vp[5] = vell.z;   // velPos^T gq'
  }
  for (int k = 0; k < bd.ndof; k++) {
    const int d = o + k;
    const DevDof& df = dofs[d];
    const double lm = lam[k];
    double gt = lm;
    double gvo = gvAt(d) + vp[k] + ballExtra - (vb[k] + df.damping * lm + c.dt * df.spring * lm);
    double gqo = pp[k] - (qb[k] + df.spring * lm) + qExtraAt(d);
    // clipLossGradientsToBounds (BackpropSnapshot.cpp:425-479)
    const double qd = q[(int64_t)d * B + b], vd = v[(int64_t)d * B + b], td = tau[(int64_t)d * B + b];
    if ((qd == df.posLo && gqo > 0) || (qd == df.posHi && gqo < 0)) gqo = 0;
    if ((vd == df.velLo && gvo > 0) || (vd == df.velHi && gvo < 0)) gvo = 0;
    if ((td == df.forceLo && gt > 0) || (td == df.forceHi && gt < 0)) gt = 0;
    gq[(int64_t)d * B + b] = gqo;
    gv[(int64_t)d * B + b] = gvo;
    if (df.actionIndex >= 0) gaction[(int64_t)df.actionIndex * B + b] = gt;
  }
}

// contact adjoint activity flag and lambda1 = M^-1 g (k_bwd_recompute)
__global__ __launch_bounds__(64 * TREE_WPB_MAX) NBL_WAVES(NBL_W_RECOMP) void k_bwd_recompute_coop(DevModel mdl, const DevBody* __restrict__ bodies,
                                                           const DevDof* __restrict__ dofs, int64_t B,
                                                           const double* __restrict__ saved, SavedLayout lay,
                                                           const double* __restrict__ gnext, double* __restrict__ lws) {
  extern __shared__ __attribute__((aligned(16))) double ldsTree[];
  CoopCtxT<PROF_BWD> c;
  NBL_PHASE(11);
  if (!coopTreeSetup(c, mdl, bodies, dofs, ldsTree, B)) return;
  NBL_PHASE(12);
  const int64_t b = c.b;
  bodies = c.bodies; dofs = c.dofs;   // the LDS copies
  const int n = mdl.n;
  const double* gvn = gnext + (int64_t)n * B;
  // any clamping row in this lane's world?  The lanes of a world share the rows out, vote, and read their group's bits
  const int m = 3 * (int)saved[(int64_t)lay.nc * B + b];
  bool mine = false;
  if (c.lane < c.nbp) for (int r = c.lane; r < MAX_ROWS; r += c.nbp) mine = mine || (r < m && saved[(int64_t)(lay.cls + r) * B + b] == 1.0);
  const uint64_t votes = (uint64_t)__ballot(mine ? 1 : 0);
  const int grp = (int)(threadIdx.x & 63u) / c.nbp;
  const uint64_t gmask = c.nbp >= 64 ? ~0ull : ((1ull << c.nbp) - 1ull) << (grp * c.nbp);
  const bool active = c.lane < c.nbp && (votes & gmask) != 0ull;
  if (c.lane == 0) lws[(int64_t)LB_FLAG * B + b] = active ? 1.0 : 0.0;
  if (!active) {
    forDofs(c, [&](int d) { lws[(int64_t)(LB_GVP + d) * B + b] = gvn[(int64_t)d * B + b]; lws[(int64_t)(LB_QX + d) * B + b] = 0.0; });
    return;
  }
  NBL_PHASE(13);
  coopLoadTree(c, saved, lay);
  NBL_PHASE(14);
  const WorldBody wb = loadWorldBody(c);
  NBL_PHASE(15);
  double lam[6];
  minvSweepsWorld(c, wb, [&](int d) -> double { return gvn[(int64_t)d * B + b]; }, lam);
  NBL_PHASE(16);
  if (wb.on) {
    const DevBody& bd = bodies[c.lane];
    for (int k = 0; k < bd.ndof; k++) lws[(int64_t)(LB_LAM1 + bd.dofOff + k) * B + b] = lam[k];
  }
  NBL_PHASE(17);
}

// unconstrained backward sweep driven by g_vpre, plus the contact position cotangent (k_bwd_final); with lws == nullptr
// the whole backward pass of a model without colliders (k_step_backward)
__global__ __launch_bounds__(64 * TREE_WPB_MAX) NBL_WAVES(NBL_W_BFINAL) void k_bwd_final_coop(DevModel mdl, const DevBody* __restrict__ bodies,
                                                       const DevDof* __restrict__ dofs, int64_t B,
                                                       const double* __restrict__ saved, SavedLayout lay,
                                                       const double* __restrict__ gnext, double* __restrict__ gstate,
                                                       double* __restrict__ gaction, const double* __restrict__ lws,
                                                       double* __restrict__ lamOut) {
  extern __shared__ __attribute__((aligned(16))) double ldsTree[];
  CoopCtxT<PROF_BWD> c;
  NBL_PHASE(20);
  if (!coopTreeSetup(c, mdl, bodies, dofs, ldsTree, B)) return;
  NBL_PHASE(21);
  const int64_t b = c.b;
  const int n = mdl.n;
  const double* q = saved;
  const double* v = saved + (int64_t)n * B;
  const double* tau = saved + (int64_t)2 * n * B;
  const double* gvn = gnext + (int64_t)n * B;
  coopLoadTree(c, saved, lay);
  NBL_PHASE(22);
  const WorldBody wb = loadWorldBody(c);
  NBL_PHASE(23);
  auto gvp = [&](int d) -> double { return lws ? lws[(int64_t)(LB_GVP + d) * B + b] : gvn[(int64_t)d * B + b]; };
  auto qx = [&](int d) -> double { return lws ? lws[(int64_t)(LB_QX + d) * B + b] : 0.0; };
  double lam[6];
  const V6 Ww = minvSweepsWorld(c, wb, [&](int d) -> double { return c.dt * gvp(d); }, lam);
  NBL_PHASE(24);
  // with bouncing contacts the reference multiplies velPos by its bounce approximation: the correction to velPos^T gq' comes from
  // k_bwd_bounce as an extra velocity cotangent (the one to posPos^T gq' is already inside LB_QX)
  auto gvFinal = [&](int d) -> double { return gvp(d) + ((lws && mdl.hasBounce) ? lws[(int64_t)(LB_VX + d) * B + b] : 0.0); };
  reverseSweepWorld(c, wb, Ww, lam, q, v, tau, gnext, gvFinal, gvp, qx, gstate, gstate + (int64_t)n * B, gaction);
  NBL_PHASE(27);
  if (lamOut && wb.on) {   // lambda = dL/dtau on every DOF, where the one-world-per-lane kernels leave it (k_bwd_inertia reads it)
    const int nd = c.bodies[c.lane].ndof;
    for (int k = 0; k < nd; k++) lamOut[((int64_t)c.lane * WS_PER_BODY + WS_UIMP + k) * B + b] = lam[k];
  }
}

// Compact world-major tree blocks (model_dev.hpp)  ->  the lane-interleaved kept slots of the workspace ws[(body * 288 + slot) * B + b],
// for the one-world-per-lane k_bwd_final / k_step_backward (the heavy reverse sweep is VALU-issue bound with lane = body:
// 3 of 64 lanes busy; one world per lane stays the better shape for it).  LDS-tiled transpose, both sides coalesced.
__global__ __launch_bounds__(256) void k_tree_to_lanes(const double* __restrict__ saved, SavedLayout lay, int nb, int64_t B,
                                                       int64_t b0, int64_t b1, double* __restrict__ ws, const DevBody* __restrict__ bodies) {
  __shared__ double tile[32][33];
  const double* blk = saved + ((int64_t)lay.total + lay.dense) * B;
  const int64_t cols = lay.treeRows;                     // entries per world
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = b0 + (int64_t)blockIdx.y * 32;   // c: entry, r: world
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int64_t r = r0 + k, cc = c0 + tx;
    if (r < b1 && cc < cols) tile[k][tx] = blk[r * cols + cc];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int64_t cc = c0 + k, r = r0 + tx;
    if (r < b1 && cc < cols) {
      int slot, body;
      if (cc < (int64_t)TREE_ROWS * lay.treeNbp) { slot = treeRowSlot((int)(cc / lay.treeNbp)); body = (int)(cc % lay.treeNbp); }
      else {                                                  // the free-joint part: entry of the fi-th free-joint body
        const int e = (int)(cc - (int64_t)TREE_ROWS * lay.treeNbp), fi = e / TREE_FREE;
        slot = treeFreeSlot(e % TREE_FREE); body = nb;
        for (int i = 0; i < nb; i++) if (bodies[i].freeIdx == fi) body = i;
      }
      if (body < nb) ws[((int64_t)body * WS_PER_BODY + slot) * B + r] = tile[tx][k];
    }
  }
}

}  // namespace NBL_NS
