// kernels.hip — hand-written HIP kernels (gfx950) for the batched differentiable timestep.
//
// Execution model: ONE WORLD PER LANE.  A wavefront holds 64 consecutive worlds; every per-DOF
// load/store of a wavefront is one coalesced 512-byte line of the [dof][B] arrays.  The body loop
// is the (short, sequential) outer loop inside the kernel; its trip count and every branch on the
// joint type are wave-uniform (all worlds share the model), so the tree sweeps run without
// divergence and the model constants are fetched with scalar loads.
//
// Forward  (k_step_forward):  World::step, dart/simulation/World.cpp:221-333
//   sweep 1 root->leaf  kinematics                     detail/GenericJoint.hpp:1803-1824
//   sweep 2 leaf->root  articulated inertia + bias     BodyNode.cpp:2046-2114, GenericJoint.hpp:2168-2185, 2276-2301, 2395-2421, 2554-2571
//   sweep 3 root->leaf  accelerations                  BodyNode.cpp:2159-2185, GenericJoint.hpp:2656-2676
//   v' = v + dt*qdd ; q' = integrate(q, v_t, dt)       GenericJoint.hpp:1410-1426, FreeJoint.cpp:922-929
//
// Backward (k_step_backward): the vector-Jacobian product that BackpropSnapshot::backprop
// (dart/neural/BackpropSnapshot.cpp:121-194) obtains from five dense n x n Jacobians is computed
// here MATRIX-FREE in O(n) per world:
//   lambda = M^-1 (dt * gv')                           two sweeps reusing the articulated inertias
//                                                      (same recursion as Skeleton::updateInvMassMatrix, Skeleton.cpp:12573-12660)
//   (dID/dq)^T lambda, (dID/dv)^T lambda               one reverse-mode sweep of the Newton-Euler
//                                                      recursion at (q, v, qdd)  — replaces the O(n^2)
//                                                      column recursions of BodyNode.cpp:2972-3205, 3440-3962
//   g_tau = lambda
//   g_v   = gv' + velPos^T gq' - (dID/dv)^T lambda - (D + dt K) lambda
//   g_q   = posPos^T gq' - (dID/dq)^T lambda - K lambda
// which is algebraically identical to Appendix A.6 of SURVEY.md (forceVel = dt M^-1, velVel, posVel)
// because d(qdd) = M^-1 (d tau_eff - dID|_{qdd fixed}).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "model_dev.hpp"
#include "spatial_dev.hpp"

namespace NBL_NS {

struct Ctx {
  const DevBody* __restrict__ bodies;
  const DevDof* __restrict__ dofs;
  double* __restrict__ ws;
  double* __restrict__ tree;   // slots < WS_KEEP: element (body, slot) at tree[body * tBody + slot * tSlot] (the saved record's
  int64_t tBody, tSlot;        // lane-interleaved tree block, or the workspace itself), or - tCompact - in the compact world-major
  int tCompact;                // tree block of the record (treeCompactRow, model_dev.hpp; tSlot = nbp)
  int64_t B, b;
  int nb, n;
  double dt;
  V3 g;
};

DEV double& wsAt(const Ctx& c, int body, int slot) {
  if (slot >= WS_KEEP) return c.ws[((int64_t)body * WS_PER_BODY + slot) * c.B + c.b];
  if (c.tCompact) {
    const int r = treeCompactRow(slot);
    if (r >= 0) return c.tree[r * c.tSlot + body];
    // AI / PSI[1..20] / U[1..5]: the record holds them for free-joint bodies only; for any other body they are scratch (workspace)
    const int fi = c.bodies[body].freeIdx;
    return fi >= 0 ? c.tree[TREE_ROWS * c.tSlot + fi * TREE_FREE + (-1 - r)] : c.ws[((int64_t)body * WS_PER_BODY + slot) * c.B + c.b];
  }
  return c.tree[body * c.tBody + slot * c.tSlot];
}

// One world per WAVEFRONT, lane = body: the whole per-body state of the sweeps lives in LDS, lds[slot * nbp + body]
// (conflict-free), bodies of one tree level are processed together (level-synchronous sweeps) and children add to their
// parent one sibling rank at a time.  The sweep code below is single-source for both execution models: it is written
// against the small vocabulary  wsAt / forBodiesDown / forBodiesUp / forBodies / forDofs / parentAdd*.
// LDS image of a world for the lane = body kernels: rows [row][nbp] for the slots a kernel family touches (compact map,
// coopSlotMap) plus, per free-joint body, a small block for the slots only a free joint owns (the 6 x 6 LDL^T of its
// projected inertia, its 6 joint forces, and - in the backward kernels - its articulated inertia).  Two profiles:
constexpr int PROF_FWD = 0, PROF_BWD = 1;
template <int P> __host__ __device__ constexpr int coopRows() { return P == PROF_FWD ? 89 : 73; }
template <int P> __host__ __device__ constexpr int coopFreeExtra() { return P == PROF_FWD ? 25 : 41; }
constexpr int COOP_UNMAPPED = -1000000;
// row (>= 0), or -(1 + offset) in the free-joint block, or COOP_UNMAPPED when the profile does not hold the slot
template <int P> DEV int coopSlotMap(int s) {
  if (P == PROF_FWD) {
    if (s <= WS_PSI) return s;                                  // T V AI AIS PSI[0]            rows 0..45
    if (s < WS_BACC) return -(1 + (s - WS_PSI - 1));            // PSI[1..20]                   free 0..19
    if (s < WS_U) return 46 + (s - WS_BACC);                    // BACC / VTW                   rows 46..51
    if (s == WS_U) return 52;                                   // U[0]
    if (s < WS_A) return -(1 + 20 + (s - WS_U - 1));            // U[1..5]                      free 20..24
    if (s < WS_TW) return 53 + (s - WS_A);                      // A                            rows 53..58
    if (s < WS_KEEP) return 59 + (s - WS_TW);                   // TW                           rows 59..70
    if (s >= WS_W && s < WS_W + 6) return 71 + (s - WS_W);      // scratch of the world-frame ABA
    if (s >= WS_FACC && s < WS_FACC + 6) return 77 + (s - WS_FACC);
    if (s >= WS_VBAR && s < WS_VBAR + 6) return 83 + (s - WS_VBAR);
    return COOP_UNMAPPED;
  } else {
    if (s >= WS_TW && s < WS_KEEP) return s - WS_TW;            // TW                           rows 0..11
    if (s >= WS_V && s < WS_V + 6) return 12 + (s - WS_V);
    if (s >= WS_A && s < WS_A + 6) return 18 + (s - WS_A);
    if (s >= WS_AIS && s < WS_AIS + 6) return 24 + (s - WS_AIS);
    if (s == WS_PSI) return 30;
    if (s > WS_PSI && s < WS_BACC) return -(1 + (s - WS_PSI - 1));   // PSI[1..20]              free 0..19
    if (s >= WS_AI && s < WS_AI + 21) return -(1 + 20 + (s - WS_AI)); // AI (free-joint root)   free 20..40
    if (s >= WS_BACC && s < WS_BACC + 6) return 31 + (s - WS_BACC);
    if (s >= WS_BIMP && s < WS_VBAR + 6) return 37 + (s - WS_BIMP);  // BIMP UIMP W FACC ABAR VBAR   rows 37..72
    return COOP_UNMAPPED;
  }
}

// inverse maps for the tree-block copies: kept slot held by row r (or -1 for a scratch row) / by entry e of the free-joint block
template <int P> DEV int coopRowSlot(int r) {
  if (P == PROF_FWD) {
    if (r <= 45) return r;
    if (r < 52) return WS_BACC + (r - 46);
    if (r == 52) return WS_U;
    if (r < 59) return WS_A + (r - 53);
    if (r < 71) return WS_TW + (r - 59);
    return -1;
  } else {
    if (r < 12) return WS_TW + r;
    if (r < 18) return WS_V + (r - 12);
    if (r < 24) return WS_A + (r - 18);
    if (r < 30) return WS_AIS + (r - 24);
    if (r == 30) return WS_PSI;
    return -1;   // BACC (A^W exchange) and the accumulators are scratch in the backward kernels
  }
}
template <int P> DEV int coopFreeSlot(int e) {
  if (P == PROF_FWD) return e < 20 ? WS_PSI + 1 + e : WS_U + 1 + (e - 20);
  return e < 20 ? WS_PSI + 1 + e : WS_AI + (e - 20);
}

// One world per WAVEFRONT, lane = body: the bodies of one tree level are processed together (level-synchronous sweeps) and
// children add to their parent one sibling rank at a time.  The sweep code is written against the small vocabulary
// wsAt / forBodiesDown / forBodiesUp / forBodies / forDofs / parentTurn shared with the one-world-per-lane Ctx.
template <int P>
struct CoopCtxT {
  const DevBody* __restrict__ bodies;
  const DevDof* __restrict__ dofs;
  double* lds;       // rows [coopRows<P>()][nbp]
  double* ldsFree;   // [nFree][coopFreeExtra<P>()]
  int nbp;
  int64_t B, b;
  int nb, n;
  double dt;
  V3 g;
  int lane, level, rank, maxLevel, maxRank;
};
template <int P>
DEV double& wsAt(const CoopCtxT<P>& c, int body, int slot) {
  const int r = coopSlotMap<P>(slot);
  return r >= 0 ? c.lds[r * c.nbp + body] : c.ldsFree[c.bodies[body].freeIdx * coopFreeExtra<P>() + (-1 - r)];
}
DEV void waveFence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

template <class F> DEV void forBodiesDown(const Ctx& c, F f) { for (int i = 0; i < c.nb; i++) f(i); }        // root -> leaf
template <class F> DEV void forBodiesUp(const Ctx& c, F f) { for (int i = c.nb - 1; i >= 0; i--) f(i); }    // leaf -> root
template <class F> DEV void forBodies(const Ctx& c, F f) { for (int i = 0; i < c.nb; i++) f(i); }            // independent
template <class F> DEV void forDofs(const Ctx& c, F f) { for (int d = 0; d < c.n; d++) f(d); }
template <int P, class F> DEV void forBodiesDown(const CoopCtxT<P>& c, F f) {
  for (int l = 0; l <= c.maxLevel; l++) { if (c.level == l) f(c.lane); waveFence(); }
}
template <int P, class F> DEV void forBodiesUp(const CoopCtxT<P>& c, F f) {
  for (int l = c.maxLevel; l >= 0; l--) { if (c.level == l) f(c.lane); waveFence(); }
}
template <int P, class F> DEV void forBodies(const CoopCtxT<P>& c, F f) { if (c.lane < c.nb) f(c.lane); waveFence(); }
template <int P, class F> DEV void forDofs(const CoopCtxT<P>& c, F f) { if (c.lane < c.nb) for (int d = c.lane; d < c.n; d += c.nb) f(d); }
// Children add into their parent's accumulators.  One world per lane: plain read-modify-write.  Lane = body: the siblings of
// one level add TOGETHER with LDS atomics (ds_add_f64, no return value: nothing to wait for); lanes that hit the same
// address are serialised by the LDS unit in a fixed order, so the sums are reproducible.  (The earlier scheme - siblings
// taking turns by rank with read / add / write - cost maxRank + 1 exposed LDS round trips per level.)
template <class F> DEV void parentTurn(const Ctx&, F f) { f(); }
template <int P, class F> DEV void parentTurn(const CoopCtxT<P>&, F f) { f(); }
DEV void accAdd(double& dst, double x, const Ctx&) { dst += x; }
template <int P> DEV void accAdd(double& dst, double x, const CoopCtxT<P>&) {
  __hip_atomic_fetch_add(&dst, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <class C> DEV V6 ldV6(const C& c, int body, int slot) {
  double a[6];
#pragma unroll
  for (int k = 0; k < 6; k++) a[k] = wsAt(c, body, slot + k);
  return fromArr(a);
}
template <class C> DEV void stV6(const C& c, int body, int slot, V6 x) {
  double a[6];
  toArr(x, a);
#pragma unroll
  for (int k = 0; k < 6; k++) wsAt(c, body, slot + k) = a[k];
}
template <class C> DEV void addV6(const C& c, int body, int slot, V6 x) {
  double a[6];
  toArr(x, a);
#pragma unroll
  for (int k = 0; k < 6; k++) accAdd(wsAt(c, body, slot + k), a[k], c);
}
template <class C> DEV void zeroN(const C& c, int body, int slot, int cnt) {
  for (int k = 0; k < cnt; k++) wsAt(c, body, slot + k) = 0.0;
}
template <class C> DEV T12 ldT(const C& c, int body) {
  T12 T;
#pragma unroll
  for (int k = 0; k < 9; k++) T.R.m[k] = wsAt(c, body, WS_T + k);
  T.p = mk3(wsAt(c, body, WS_T + 9), wsAt(c, body, WS_T + 10), wsAt(c, body, WS_T + 11));
  return T;
}
template <class C> DEV void stT(const C& c, int body, const T12& T) {
#pragma unroll
  for (int k = 0; k < 9; k++) wsAt(c, body, WS_T + k) = T.R.m[k];
  wsAt(c, body, WS_T + 9) = T.p.x; wsAt(c, body, WS_T + 10) = T.p.y; wsAt(c, body, WS_T + 11) = T.p.z;
}
template <class C> DEV T12 ldTAt(const C& c, int body, int slot) {
  T12 T;
#pragma unroll
  for (int k = 0; k < 9; k++) T.R.m[k] = wsAt(c, body, slot + k);
  T.p = mk3(wsAt(c, body, slot + 9), wsAt(c, body, slot + 10), wsAt(c, body, slot + 11));
  return T;
}
template <class C> DEV void stTAt(const C& c, int body, int slot, const T12& T) {
#pragma unroll
  for (int k = 0; k < 9; k++) wsAt(c, body, slot + k) = T.R.m[k];
  wsAt(c, body, slot + 9) = T.p.x; wsAt(c, body, slot + 10) = T.p.y; wsAt(c, body, slot + 11) = T.p.z;
}
template <class C> DEV S6 ldS6(const C& c, int body, int slot) {
  S6 A;
#pragma unroll
  for (int k = 0; k < 21; k++) A.a[k] = wsAt(c, body, slot + k);
  return A;
}
template <class C> DEV void stS6(const C& c, int body, int slot, const S6& A) {
#pragma unroll
  for (int k = 0; k < 21; k++) wsAt(c, body, slot + k) = A.a[k];
}
template <class C> DEV void addS6(const C& c, int body, int slot, const S6& A) {
#pragma unroll
  for (int k = 0; k < 21; k++) accAdd(wsAt(c, body, slot + k), A.a[k], c);
}

DEV T12 cT(const double* t) {  // wave-uniform constant -> scalar loads
  T12 T;
#pragma unroll
  for (int k = 0; k < 9; k++) T.R.m[k] = t[k];
  T.p = mk3(t[9], t[10], t[11]);
  return T;
}
DEV S6 cS6(const double* g) {
  S6 A;
#pragma unroll
  for (int k = 0; k < 21; k++) A.a[k] = g[k];
  return A;
}
DEV V6 cV6(const double* s) { return mk6(mk3(s[0], s[1], s[2]), mk3(s[3], s[4], s[5])); }

// joint twist S*dq in the child frame
DEV V6 jointTwist(const DevBody& bd, const double* __restrict__ v, int64_t B, int64_t b) {
  if (bd.jtype == JT_FREE) {
    V6 x = mk6(mk3(v[(bd.dofOff + 0) * B + b], v[(bd.dofOff + 1) * B + b], v[(bd.dofOff + 2) * B + b]),
               mk3(v[(bd.dofOff + 3) * B + b], v[(bd.dofOff + 4) * B + b], v[(bd.dofOff + 5) * B + b]));
    return AdT(cT(bd.Tcj), x);  // S = Ad(T_cj), FreeJoint.cpp:1049-1056
  }
  return v[bd.dofOff * B + b] * cV6(bd.S);
}

// T_parent->child of one body at the positions q ([n][B]): T_pj Q(q) T_cj^-1 with Q of the joint type (RevoluteJoint.cpp:203-211,
// PrismaticJoint, ScrewJoint.cpp:217-232, FreeJoint.cpp:74-81, BallJoint.cpp:91-95; ball joints and free joints below the root are
// chains of coincident single-axis bodies whose first one carries the exponential map).  The same expressions as the tree kernels'
// first sweep; used by the narrow phase when it runs next to the forward tree kernel instead of after it.
DEV T12 jointRelTransform(const DevBody& bd, const double* __restrict__ q, int64_t B, int64_t b) {
  T12 Q;
  if (bd.jtype == JT_REVOLUTE) {
    const double qi = q[bd.dofOff * B + b];
    Q.R = expAngular(mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi));
    Q.p = mk3(0, 0, 0);
  } else if (bd.jtype == JT_PRISMATIC) {
    const double qi = q[bd.dofOff * B + b];
    Q.R = eye3();
    Q.p = mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi);
  } else if (bd.jtype == JT_SCREW) {
    const double qi = q[bd.dofOff * B + b], hq = bd.screwRate * qi;
    Q.R = expAngular(mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi));
    Q.p = mk3(bd.axis[0] * hq, bd.axis[1] * hq, bd.axis[2] * hq);
  } else if (bd.jtype == JT_FREEC) {
    const int o = bd.dofOff;
    Q.R = bd.ballComp == 0 ? expMapRot(mk3(q[(o + 0) * B + b], q[(o + 1) * B + b], q[(o + 2) * B + b])) : eye3();
    Q.p = bd.ballComp == 0 ? mk3(q[(o + 3) * B + b], q[(o + 4) * B + b], q[(o + 5) * B + b]) : mk3(0, 0, 0);
  } else if (bd.jtype == JT_BALL) {
    Q.R = bd.ballComp == 0 ? expMapRot(mk3(q[(bd.dofOff + 0) * B + b], q[(bd.dofOff + 1) * B + b], q[(bd.dofOff + 2) * B + b])) : eye3();
    Q.p = mk3(0, 0, 0);
  } else {
    Q.R = expMapRot(mk3(q[(bd.dofOff + 0) * B + b], q[(bd.dofOff + 1) * B + b], q[(bd.dofOff + 2) * B + b]));
    Q.p = mk3(q[(bd.dofOff + 3) * B + b], q[(bd.dofOff + 4) * B + b], q[(bd.dofOff + 5) * B + b]);
  }
  return mulT(mulT(cT(bd.Tpj), Q), cT(bd.TcjInv));
}

// ---------------------------------------------------------------------------------------------
// The three ABA sweeps.  q, v: [n][B];  tau fetched through tauAt(d).  Leaves T, V, AI, AIS, psi,
// A in the workspace; joint accelerations are handed to `emit(d, qdd)`.
// ---------------------------------------------------------------------------------------------
template <bool BACKWARD, class C, class TauFn, class EmitFn>
DEV void abaSweeps(const C& c, const double* __restrict__ q, const double* __restrict__ v, TauFn tauAt, EmitFn emit) {
  const int64_t B = c.B, b = c.b;
  // ---- sweep 1: kinematics ----
  forBodiesDown(c, [&](int i) {
    const DevBody& bd = c.bodies[i];
    T12 Q;
    if (bd.jtype == JT_REVOLUTE) {
      double qi = q[bd.dofOff * B + b];
      Q.R = expAngular(mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi));  // RevoluteJoint.cpp:203-211
      Q.p = mk3(0, 0, 0);
    } else if (bd.jtype == JT_PRISMATIC) {
      double qi = q[bd.dofOff * B + b];
      Q.R = eye3();
      Q.p = mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi);
    } else if (bd.jtype == JT_SCREW) {                   // expMap([axis; h axis] q) = (R(axis q), h axis q), ScrewJoint.cpp:217-232
      const double qi = q[bd.dofOff * B + b], hq = bd.screwRate * qi;
      Q.R = expAngular(mk3(bd.axis[0] * qi, bd.axis[1] * qi, bd.axis[2] * qi));
      Q.p = mk3(bd.axis[0] * hq, bd.axis[1] * hq, bd.axis[2] * hq);
    } else {
      Q.R = expMapRot(mk3(q[(bd.dofOff + 0) * B + b], q[(bd.dofOff + 1) * B + b], q[(bd.dofOff + 2) * B + b]));  // FreeJoint.cpp:74-81
      Q.p = mk3(q[(bd.dofOff + 3) * B + b], q[(bd.dofOff + 4) * B + b], q[(bd.dofOff + 5) * B + b]);
    }
    T12 T = mulT(mulT(cT(bd.Tpj), Q), cT(bd.TcjInv));
    V6 V = jointTwist(bd, v, B, b);
    if (bd.parent >= 0) V = V + AdInvT(T, ldV6(c, bd.parent, WS_V));
    stT(c, i, T);
    stTAt(c, i, WS_TW, bd.parent >= 0 ? mulT(ldTAt(c, bd.parent, WS_TW), T) : T);  // BodyNode::mWorldTransform
    stV6(c, i, WS_V, V);
    zeroN(c, i, WS_AI, 21);
    zeroN(c, i, WS_BACC, 6);
    if (BACKWARD) zeroN(c, i, WS_BIMP, 6), zeroN(c, i, WS_FACC, 18);  // FACC, ABAR, VBAR are contiguous
  });
  // ---- sweep 2: articulated inertias, bias forces, joint-space total force ----
  forBodiesUp(c, [&](int i) {
    const DevBody& bd = c.bodies[i];
    T12 T = ldT(c, i);
    V6 V = ldV6(c, i, WS_V);
    S6 G = cS6(bd.G);
    S6 AI = ldS6(c, i, WS_AI);
    addTo(AI, G);
    V6 Sdq = jointTwist(bd, v, B, b);
    V6 eta = ad(V, Sdq);                                  // GenericJoint.hpp:1803-1824 (dS = 0)
    V6 Bf = ldV6(c, i, WS_BACC) - dad(V, mul(G, V));      // BodyNode.cpp:2076-2114; gravity rides on the base acceleration
    V6 AIeta = mul(AI, eta);
    stS6(c, i, WS_AI, AI);
    if (bd.jtype != JT_FREE) {
      const int d = bd.dofOff;
      const DevDof& df = c.dofs[d];
      V6 S = cV6(bd.S);
      V6 AIS = mul(AI, S);
      double psi = 1.0 / dot(S, AIS);                     // GenericJoint.hpp:2276-2301
      double qd = q[d * B + b], vd = v[d * B + b];
      // GenericJoint.hpp:2554-2571: spring uses q - q0 + dt*v, damping explicit
      double u = tauAt(d) - df.spring * (qd - df.rest + vd * c.dt) - df.damping * vd - dot(S, AIeta + Bf);
      stV6(c, i, WS_AIS, AIS);
      wsAt(c, i, WS_PSI) = psi;
      wsAt(c, i, WS_U) = u;
      if (bd.parent >= 0) {
        V6 beta = Bf + AIeta + (psi * u) * AIS;           // GenericJoint.hpp:2395-2421
        rank1Sub(AI, AIS, psi);                           // PI = AI - AIS psi AIS^T  (GenericJoint.hpp:2168-2185)
        const S6 toParentAI = congruenceToParent(T, AI);
        const V6 toParentB = dAdInvT(T, beta);
        parentTurn(c, [&]() { addS6(c, bd.parent, WS_AI, toParentAI); addV6(c, bd.parent, WS_BACC, toParentB); });
      }
    } else {
      // free joint as a tree root: projected inertia S^T AI S with S = Ad(T_cj); LDL^T stands in
      // for math::inverse<SE3Space> (ConfigurationSpace.hpp:48-65)
      LDL6 f = ldl6(congruenceToParent(cT(bd.TcjInv), AI));
#pragma unroll
      for (int k = 0; k < 15; k++) wsAt(c, i, WS_PSI + k) = f.l[k];
#pragma unroll
      for (int k = 0; k < 6; k++) wsAt(c, i, WS_PSI + 15 + k) = f.d[k];
      V6 proj = dAdT(cT(bd.Tcj), AIeta + Bf);             // S^T (AI eta + B)
      double pj[6];
      toArr(proj, pj);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int d = bd.dofOff + k;
        const DevDof& df = c.dofs[d];
        double qd = q[d * B + b], vd = v[d * B + b];
        wsAt(c, i, WS_U + k) = tauAt(d) - df.spring * (qd - df.rest + vd * c.dt) - df.damping * vd - pj[k];
      }
    }
  });
  // ---- sweep 3: accelerations ----
  const V6 a0 = mk6(mk3(0, 0, 0), -c.g);
  forBodiesDown(c, [&](int i) {
    const DevBody& bd = c.bodies[i];
    T12 T = ldT(c, i);
    V6 V = ldV6(c, i, WS_V);
    V6 Sdq = jointTwist(bd, v, B, b);
    V6 eta = ad(V, Sdq);
    V6 XA = AdInvT(T, bd.parent >= 0 ? ldV6(c, bd.parent, WS_A) : a0);
    V6 A;
    if (bd.jtype != JT_FREE) {
      V6 AIS = ldV6(c, i, WS_AIS);
      double qdd = wsAt(c, i, WS_PSI) * (wsAt(c, i, WS_U) - dot(AIS, XA));   // GenericJoint.hpp:2656-2676
      A = XA + eta + qdd * cV6(bd.S);
      emit(bd.dofOff, qdd);
    } else {
      S6 AI = ldS6(c, i, WS_AI);
      LDL6 f;
#pragma unroll
      for (int k = 0; k < 15; k++) f.l[k] = wsAt(c, i, WS_PSI + k);
#pragma unroll
      for (int k = 0; k < 6; k++) f.d[k] = wsAt(c, i, WS_PSI + 15 + k);
      V6 proj = dAdT(cT(bd.Tcj), mul(AI, XA));
      double r[6], pj[6];
      toArr(proj, pj);
#pragma unroll
      for (int k = 0; k < 6; k++) r[k] = wsAt(c, i, WS_U + k) - pj[k];
      ldl6Solve(f, r);
#pragma unroll
      for (int k = 0; k < 6; k++) emit(bd.dofOff + k, r[k]);
      A = XA + eta + AdT(cT(bd.Tcj), fromArr(r));
    }
    stV6(c, i, WS_A, A);
  });
}

// Where the kept slots live: the tree block of the saved record when it has one (lane-interleaved rows, or world-major
// [slot][nbp] blocks when lay.treeNbp > 0), else the workspace.
DEV Ctx makeCtx(const DevModel& mdl, const DevBody* bodies, const DevDof* dofs, double* ws, int64_t B, int64_t b,
                double* saved = nullptr, const SavedLayout* lay = nullptr) {
  Ctx c;
  c.bodies = bodies; c.dofs = dofs; c.ws = ws; c.B = B; c.b = b;
  if (saved && lay && lay->treeRows > 0) {
    double* blk = saved + ((int64_t)lay->total + lay->dense) * B;
    if (lay->treeNbp > 0) { c.tree = blk + b * (int64_t)lay->treeRows; c.tBody = 1; c.tSlot = lay->treeNbp; c.tCompact = 1; }
    else { c.tree = blk + b; c.tBody = (int64_t)WS_KEEP * B; c.tSlot = B; c.tCompact = 0; }
  } else { c.tree = ws + b; c.tBody = (int64_t)WS_PER_BODY * B; c.tSlot = B; c.tCompact = 0; }
  c.nb = mdl.nb; c.n = mdl.n; c.dt = mdl.dt;
  c.g = mk3(mdl.gravity[0], mdl.gravity[1], mdl.gravity[2]);
  return c;
}

// ---------------------------------------------------------------------------------------------
// Forward kernel
// ---------------------------------------------------------------------------------------------
template <class TauFn, class EmitFn>
DEV void stepAba(const Ctx& c, const double* __restrict__ q, const double* __restrict__ v, TauFn tauAt, EmitFn emit) {
  abaSweeps<false>(c, q, v, tauAt, emit);
}

// World::step without contact for the world(s) of context c: ABA, v' = v + dt qdd, q' = integrate(q, v_t, dt), and the
// rows of the saved record BackpropSnapshot captures (q_t, v_t, tau_t, mLastPreConstraintVelocity).  Single source for
// the one-world-per-lane and the one-world-per-wavefront kernels.
template <class C>
DEV void stepForwardCore(const C& c, const double* __restrict__ state, const double* __restrict__ action,
                         double* __restrict__ next, double* __restrict__ saved, const SavedLayout& lay) {
  const DevBody* bodies = c.bodies;
  const DevDof* dofs = c.dofs;
  const int64_t B = c.B, b = c.b;
  const int n = c.n;
  const double* q = state;
  const double* v = state + (int64_t)n * B;
  auto tauAt = [&](int d) -> double {
    int ai = dofs[d].actionIndex;                       // World::setAction: unmapped control forces are 0 (World.cpp:2061-2086)
    return ai >= 0 ? action[(int64_t)ai * B + b] : 0.0;
  };
  double* nq = next;
  double* nv = next + (int64_t)n * B;
  const int vpreRow = saved ? lay.vpre : -1;
  auto emit = [&](int d, double qdd) {                  // GenericJoint.hpp:1410-1414
    const double x = v[(int64_t)d * B + b] + c.dt * qdd;
    nv[(int64_t)d * B + b] = x;
    if (vpreRow >= 0) saved[(int64_t)(vpreRow + d) * B + b] = x;   // mLastPreConstraintVelocity (World.cpp:236-239)
  };
  stepAba(c, q, v, tauAt, emit);
  NBL_PHASE(7);

  // positions integrate with the PRE-step velocity (World.cpp:307-333, mParallelVelocityAndPositionUpdates)
  forBodies(c, [&](int i) {
    const DevBody& bd = bodies[i];
    const int o = bd.dofOff;
    if (bd.jtype == JT_FREE) {
      V3 r = mk3(q[(o + 0) * B + b], q[(o + 1) * B + b], q[(o + 2) * B + b]);
      V3 p = mk3(q[(o + 3) * B + b], q[(o + 4) * B + b], q[(o + 5) * B + b]);
      V3 w = mk3(v[(o + 0) * B + b], v[(o + 1) * B + b], v[(o + 2) * B + b]);
      V3 vl = mk3(v[(o + 3) * B + b], v[(o + 4) * B + b], v[(o + 5) * B + b]);
      M3 R = expMapRot(r);
      M3 E = expMapRot(c.dt * w);                        // FreeJoint.cpp:922-929: Q * convertToTransform(vel*dt)
      V3 rn = logMap(mul(R, E));
      V3 pn = p + mul(R, c.dt * vl);
      nq[(o + 0) * B + b] = rn.x; nq[(o + 1) * B + b] = rn.y; nq[(o + 2) * B + b] = rn.z;
      nq[(o + 3) * B + b] = pn.x; nq[(o + 4) * B + b] = pn.y; nq[(o + 5) * B + b] = pn.z;
    } else if (bd.jtype == JT_FREEC) {
      // FreeJoint::integratePositionsExplicit (FreeJoint.cpp:922-929) of a free joint below the root: each of its six bodies writes its component
      const int d0 = o - bd.ballComp, cmp = bd.ballComp;
      auto at3 = [&](const double* x, int k0) { return mk3(x[(int64_t)(d0 + k0) * B + b], x[(int64_t)(d0 + k0 + 1) * B + b], x[(int64_t)(d0 + k0 + 2) * B + b]); };
      const M3 R = expMapRot(at3(q, 0));
      const V3 rn = logMap(mul(R, expMapRot(c.dt * at3(v, 0)))), pn = at3(q, 3) + mul(R, c.dt * at3(v, 3));
      nq[(int64_t)o * B + b] = cmp < 3 ? pick3(rn, cmp) : pick3(pn, cmp - 3);
    } else if (bd.jtype == JT_BALL) {
      // BallJoint::integratePositionsExplicit (BallJoint.cpp:333-349): R' = R(q) R(w dt); each of the triple's bodies writes its component
      const int d0 = o - bd.ballComp;
      const V3 r = mk3(q[(int64_t)(d0 + 0) * B + b], q[(int64_t)(d0 + 1) * B + b], q[(int64_t)(d0 + 2) * B + b]);
      const V3 w = mk3(v[(int64_t)(d0 + 0) * B + b], v[(int64_t)(d0 + 1) * B + b], v[(int64_t)(d0 + 2) * B + b]);
      const V3 rn = logMap(mul(expMapRot(r), expMapRot(c.dt * w)));
      nq[(int64_t)o * B + b] = bd.ballComp == 0 ? rn.x : (bd.ballComp == 1 ? rn.y : rn.z);
    } else {
      nq[(int64_t)o * B + b] = q[(int64_t)o * B + b] + c.dt * v[(int64_t)o * B + b];
    }
  });
  if (saved) {  // what BackpropSnapshot captures: q_t, v_t, tau_t (BackpropSnapshot.cpp:33-118)
    forDofs(c, [&](int d) {
      saved[(int64_t)d * B + b] = q[(int64_t)d * B + b];
      saved[(int64_t)(n + d) * B + b] = v[(int64_t)d * B + b];
      saved[(int64_t)(2 * n + d) * B + b] = tauAt(d);
    });
  }
}

__global__ __launch_bounds__(64) void k_step_forward(DevModel mdl, const DevBody* __restrict__ bodies,
                                                     const DevDof* __restrict__ dofs, int64_t B,
                                                     const double* __restrict__ state, const double* __restrict__ action,
                                                     double* __restrict__ next, double* __restrict__ saved,
                                                     uint32_t* __restrict__ status, double* __restrict__ ws, SavedLayout lay) {
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  Ctx c = makeCtx(mdl, bodies, dofs, ws, B, b, saved, &lay);
  stepForwardCore(c, state, action, next, saved, lay);
  if (status) {                                         // NBL_ST_NAN for a non-finite unconstrained step (like k_step_forward_coop)
    bool bad = false;
    for (int d = 0; d < 2 * mdl.n; d++) bad = bad || !__builtin_isfinite(next[(int64_t)d * B + b]);
    status[b] = bad ? 0x40u : 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward building blocks
// ---------------------------------------------------------------------------------------------
// lambda = M^-1 rhs using the articulated inertias left in the workspace by abaSweeps: leaf->root
// impulse sweep then root->leaf sweep (same recursion as Skeleton::updateInvMassMatrix,
// Skeleton.cpp:12573-12660).  Leaves lambda per DOF in WS_UIMP (+k) of its body and the body twists
// W_i = X W_parent + S lambda_i in WS_W.
template <class C, class RhsFn>
DEV void minvSweeps(const C& c, RhsFn rhsAt) {
  const DevBody* bodies = c.bodies;
  forBodies(c, [&](int i) { zeroN(c, i, WS_BIMP, 6); });
  forBodiesUp(c, [&](int i) {
    const DevBody& bd = bodies[i];
    V6 Bi = ldV6(c, i, WS_BIMP);
    if (bd.jtype != JT_FREE) {
      const int d = bd.dofOff;
      double uimp = rhsAt(d) - dot(cV6(bd.S), Bi);
      wsAt(c, i, WS_UIMP) = uimp;
      if (bd.parent >= 0) {
        V6 beta = Bi + (wsAt(c, i, WS_PSI) * uimp) * ldV6(c, i, WS_AIS);
        const V6 up = dAdInvT(ldT(c, i), beta);
        parentTurn(c, [&]() { addV6(c, bd.parent, WS_BIMP, up); });
      }
    } else {
      double pj[6];
      toArr(dAdT(cT(bd.Tcj), Bi), pj);
#pragma unroll
      for (int k = 0; k < 6; k++) wsAt(c, i, WS_UIMP + k) = rhsAt(bd.dofOff + k) - pj[k];
    }
  });
  forBodiesDown(c, [&](int i) {
    const DevBody& bd = bodies[i];
    T12 T = ldT(c, i);
    V6 XW = bd.parent >= 0 ? AdInvT(T, ldV6(c, bd.parent, WS_W)) : zero6();
    V6 W;
    if (bd.jtype != JT_FREE) {
      double lam = wsAt(c, i, WS_PSI) * (wsAt(c, i, WS_UIMP) - dot(ldV6(c, i, WS_AIS), XW));
      wsAt(c, i, WS_UIMP) = lam;
      W = XW + lam * cV6(bd.S);
    } else {
      S6 AI = ldS6(c, i, WS_AI);
      LDL6 f;
#pragma unroll
      for (int k = 0; k < 15; k++) f.l[k] = wsAt(c, i, WS_PSI + k);
#pragma unroll
      for (int k = 0; k < 6; k++) f.d[k] = wsAt(c, i, WS_PSI + 15 + k);
      double r[6], pj[6];
      toArr(dAdT(cT(bd.Tcj), mul(AI, XW)), pj);
#pragma unroll
      for (int k = 0; k < 6; k++) r[k] = wsAt(c, i, WS_UIMP + k) - pj[k];
      ldl6Solve(f, r);
#pragma unroll
      for (int k = 0; k < 6; k++) wsAt(c, i, WS_UIMP + k) = r[k];
      W = XW + AdT(cT(bd.Tcj), fromArr(r));
    }
    stV6(c, i, WS_W, W);
  });
}

// Position-space Jacobian transpose of joint i applied to a body-frame adjoint xi:  H_i^T xi
// (H = S for 1-DOF joints; free joint: Ad(T_cj) blkdiag(expMapJac(r)^T, R^T), FreeJoint.cpp:790-823)
DEV void applyHt(const DevBody& bd, const double* __restrict__ q, int64_t B, int64_t b, V6 xi, double* out) {
  if (bd.jtype == JT_BALL) {
    // ball joint (BallJoint.cpp:282-289): H = [expMapJac(q)^T; 0] in the frame of the x body of the triple; xi = THAT body's adjoint
    const int d0 = bd.dofOff - bd.ballComp;
    const V3 y = mul(expMapJac(mk3(q[(int64_t)(d0 + 0) * B + b], q[(int64_t)(d0 + 1) * B + b], q[(int64_t)(d0 + 2) * B + b])), xi.w);
    out[0] = bd.ballComp == 0 ? y.x : (bd.ballComp == 1 ? y.y : y.z);
    return;
  }
  if (bd.jtype == JT_FREEC) {
    // free joint below the root (FreeJoint.cpp:790-823): H = blkdiag(expMapJac(r)^T, R^T) in the frame of the first of its six bodies
    const int d0 = bd.dofOff - bd.ballComp, cmp = bd.ballComp;
    const V3 r = mk3(q[(int64_t)(d0 + 0) * B + b], q[(int64_t)(d0 + 1) * B + b], q[(int64_t)(d0 + 2) * B + b]);
    out[0] = cmp < 3 ? pick3(mul(expMapJac(r), xi.w), cmp) : pick3(mul(expMapRot(r), xi.v), cmp - 3);
    return;
  }
  if (bd.jtype != JT_FREE) { out[0] = dot(cV6(bd.S), xi); return; }
  const int o = bd.dofOff;
  V6 y = dAdT(cT(bd.Tcj), xi);
  V3 r = mk3(q[(o + 0) * B + b], q[(o + 1) * B + b], q[(o + 2) * B + b]);
  V3 qbr = mul(expMapJac(r), y.w), qbp = mul(expMapRot(r), y.v);
  out[0] = qbr.x; out[1] = qbr.y; out[2] = qbr.z; out[3] = qbp.x; out[4] = qbp.y; out[5] = qbp.z;
}

// Reverse-mode Newton-Euler sweep at (q, v, qdd) with joint adjoint lambda (WS_UIMP/WS_W from
// minvSweeps) + the per-DOF epilogue.  gvAt(d): cotangent of the pre-contact velocity;
// qExtraAt(d): additional position cotangent from the contact stage (0 without contact).
template <class C, class GvFn, class QxFn>
DEV void reverseSweep(const C& c, const double* __restrict__ q, const double* __restrict__ v,
                      const double* __restrict__ tau, const double* __restrict__ gqn, GvFn gvAt, QxFn qExtraAt,
                      double* __restrict__ gq, double* __restrict__ gv, double* __restrict__ gaction) {
  const DevBody* bodies = c.bodies;
  const DevDof* dofs = c.dofs;
  const int64_t B = c.B, b = c.b;
  forBodies(c, [&](int i) { zeroN(c, i, WS_FACC, 18); });
  const V6 a0 = mk6(mk3(0, 0, 0), -c.g);
  forBodiesUp(c, [&](int i) {
    const DevBody& bd = bodies[i];
    T12 T = ldT(c, i);
    V6 V = ldV6(c, i, WS_V), A = ldV6(c, i, WS_A), W = ldV6(c, i, WS_W);
    S6 G = cS6(bd.G);
    V6 GV = mul(G, V);
    V6 F = mul(G, A) - dad(V, GV) + ldV6(c, i, WS_FACC);          // transmitted force at (q, v, qdd)
    V6 Abar = mul(G, W) + ldV6(c, i, WS_ABAR);
    V6 Sdq = jointTwist(bd, v, B, b);
    V6 Vbar = dad(W, GV) - mul(G, ad(V, W)) - dad(Sdq, Abar) + ldV6(c, i, WS_VBAR);
    V6 tmp = dad(V, Abar) + Vbar;
    V6 XVp = zero6(), XAp = AdInvT(T, a0), XWp = zero6();
    if (bd.parent >= 0) {
      XVp = AdInvT(T, ldV6(c, bd.parent, WS_V));
      XAp = AdInvT(T, ldV6(c, bd.parent, WS_A));
      XWp = AdInvT(T, ldV6(c, bd.parent, WS_W));
      const V6 upF = dAdInvT(T, F), upA = dAdInvT(T, Abar), upV = dAdInvT(T, Vbar);
      parentTurn(c, [&]() { addV6(c, bd.parent, WS_FACC, upF); addV6(c, bd.parent, WS_ABAR, upA); addV6(c, bd.parent, WS_VBAR, upV); });
    }
    V6 xi = dad(XWp, F) + dad(XAp, Abar) + dad(XVp, Vbar);          // adjoint of the joint transform, body frame
    double qb[6], vb[6], pp[6], vp[6];
    applyHt(bd, q, B, b, xi, qb);
    const int o = bd.dofOff;
    if (bd.jtype != JT_FREE) {
      vb[0] = dot(cV6(bd.S), tmp);
      pp[0] = gqn[(int64_t)o * B + b];            // posPos = 1, velPos = dt  (GenericJoint.hpp:1428-1444)
      vp[0] = c.dt * pp[0];
    } else {
      toArr(dAdT(cT(bd.Tcj), tmp), vb);
      V3 r = mk3(q[(o + 0) * B + b], q[(o + 1) * B + b], q[(o + 2) * B + b]);
      V3 w = mk3(v[(o + 0) * B + b], v[(o + 1) * B + b], v[(o + 2) * B + b]);
      V3 vl = mk3(v[(o + 3) * B + b], v[(o + 4) * B + b], v[(o + 5) * B + b]);
      M3 R = expMapRot(r);
      // VJP of q' = [logMap(R E); p + R vl dt]  (exact reverse-mode of FreeJoint.cpp:922-929; the
      // reference differentiates the same expression by central differences, :950-1007)
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
      double lam = wsAt(c, i, WS_UIMP + k);
      double gt = lam;
      double gvo = gvAt(d) + vp[k] - (vb[k] + df.damping * lam + c.dt * df.spring * lam);
      double gqo = pp[k] - (qb[k] + df.spring * lam) + qExtraAt(d);
      // clipLossGradientsToBounds (BackpropSnapshot.cpp:425-479)
      double qd = q[(int64_t)d * B + b], vd = v[(int64_t)d * B + b], td = tau[(int64_t)d * B + b];
      if ((qd == df.posLo && gqo > 0) || (qd == df.posHi && gqo < 0)) gqo = 0;
      if ((vd == df.velLo && gvo > 0) || (vd == df.velHi && gvo < 0)) gvo = 0;
      if ((td == df.forceLo && gt > 0) || (td == df.forceHi && gt < 0)) gt = 0;
      gq[(int64_t)d * B + b] = gqo;
      gv[(int64_t)d * B + b] = gvo;
      if (df.actionIndex >= 0) gaction[(int64_t)df.actionIndex * B + b] = gt;
    }
  });
}

// ---------------------------------------------------------------------------------------------
// Backward kernel (no clamping contact constraints in the whole batch model)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_step_backward(DevModel mdl, const DevBody* __restrict__ bodies,
                                                      const DevDof* __restrict__ dofs, int64_t B,
                                                      const double* __restrict__ saved, SavedLayout lay,
                                                      const double* __restrict__ gnext,
                                                      double* __restrict__ gstate, double* __restrict__ gaction,
                                                      double* __restrict__ ws, int treeInWs) {
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= mdl.b1) return;
  Ctx c = makeCtx(mdl, bodies, dofs, ws, B, b, const_cast<double*>(saved), &lay);
  const int n = mdl.n;
  const double* q = saved;
  const double* v = saved + (int64_t)n * B;
  const double* tau = saved + (int64_t)2 * n * B;
  const double* gqn = gnext;
  const double* gvn = gnext + (int64_t)n * B;
  auto tauAt = [&](int d) -> double { return tau[(int64_t)d * B + b]; };
  auto emit = [&](int, double) {};
  if (lay.treeRows > 0 || treeInWs) { for (int i = 0; i < c.nb; i++) { zeroN(c, i, WS_BIMP, 6); zeroN(c, i, WS_FACC, 18); } }   // forward state comes from the record
  else abaSweeps<true>(c, q, v, tauAt, emit);
  auto gvAt = [&](int d) -> double { return gvn[(int64_t)d * B + b]; };
  minvSweeps(c, [&](int d) -> double { return c.dt * gvn[(int64_t)d * B + b]; });
  reverseSweep(c, q, v, tau, gqn, gvAt, [](int) -> double { return 0.0; }, gstate, gstate + (int64_t)n * B, gaction);
}

// [B][d] <-> [d][B] transposes through LDS (the Python surface stacks the reference's 1-D state
// vectors world-major; the kernels want DOF-major so a wavefront's loads coalesce).
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ src, double* __restrict__ dst, int64_t rows,
                                                   int64_t cols) {
  // src is [rows][cols] row-major; dst is [cols][rows]
  __shared__ double tile[32][33];
  int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    int64_t r = r0 + k, cc = c0 + tx;
    if (r < rows && cc < cols) tile[k][tx] = src[r * cols + cc];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int64_t cc = c0 + k, r = r0 + tx;
    if (r < rows && cc < cols) dst[cc * rows + r] = tile[tx][k];
  }
}

}  // namespace NBL_NS
