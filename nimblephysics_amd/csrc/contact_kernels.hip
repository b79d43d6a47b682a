// contact_kernels.hip — the narrow phase of the batched step (forward), a few lanes per world.
//
//   k_contact_detect   collision detection at q_t + depth filter      ConstraintSolver.cpp:563-613, DARTCollide.cpp:764-1450
// The rest of the contact stage (rows, LCP stage 0, the fallback cascade) is one world per wavefront: coop_kernels.hip.
#include "collision_dev.hpp"
#include "lcp_dev.hpp"

namespace NBL_NS {

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

// The narrow phase of `wl` worlds x `ppw` lanes (threads tid < wl * ppw of the workgroup `bid`; any further threads of the workgroup
// only take part in its barriers).  keptP: seenPts * 3 * ls doubles, clipBuf: 48 * ls doubles (ls >= wl * ppw: the lane stride), stage: the staging area of the
// ppw > 1 scheme - all LDS.  qFk != nullptr: the world transforms of the collider bodies are computed HERE from the positions (the
// kernel runs next to the forward tree kernel, not after it) and the status word is left alone: the contact count goes to the record
// with + 0.5 when contacts were dropped, k_contact_solve_coop raises NBL_ST_CONTACT / NBL_ST_CONTACT_OVERFLOW from it.
DEV void contactDetectBody(const DevModel& mdl, const DevBody* __restrict__ bodies, const DevContactModel* __restrict__ cm, int64_t B,
                           double* __restrict__ saved, const SavedLayout& lay, uint32_t* __restrict__ status, double* __restrict__ ws,
                           int doTwists, uint32_t* __restrict__ failCount, int ppw, const double* __restrict__ qFk, int bid, int wl,
                           double* keptP, double* clipBuf, double* stage, double* fkT = nullptr, int ls = 64) {
  const int tid = (int)threadIdx.x;
  NBL_PHASE_FIRST(19);
  // points the duplicate filter remembers per world: twice the contact slots - of the BUILD in the 24- / 48-row builds, of the MODEL in the
  // general builds (whose 128 / 256 per world made the narrow phase's LDS 110 kB per 16 worlds whatever the model asked for)
#if NBL_GENERAL
  const int seenPts = 2 * cm->maxContacts < SEEN_POINTS ? 2 * cm->maxContacts : SEEN_POINTS;
#else
  constexpr int seenPts = SEEN_POINTS;
#endif
  // ---- qFk with fkT (the narrow phase next to the forward tree kernel): the joint transforms T_parent->child of every body on an ancestor
  //      chain of a collider, for the wl worlds of the workgroup, by ALL its threads - (world, body) items side by side instead of one lane
  //      per collider pair walking its 7-joint chain alone (an exponential map with its sine and cosine per joint: 55 k of the 140 k cycles
  //      of these workgroups).  The chains are then products of LDS-resident transforms, in the same order: identical arithmetic. ----
  uint64_t fkNeed = 0ull;
  if (qFk && fkT) {
    for (int i = 0; i < cm->nBoxes; i++) { const int bdy = cm->boxes[i].body; if (bdy >= 0) fkNeed |= cm->ancestors[bdy]; }
    const int nNeed = __builtin_popcountll(fkNeed);
    for (int item = tid; item < wl * nNeed; item += (int)blockDim.x) {
      const int wI = item / nNeed, k = item - wI * nNeed;
      uint64_t mk = fkNeed;
      for (int j = 0; j < k; j++) mk &= mk - 1;
      const int body = __builtin_ctzll(mk);
      const int64_t bw = mdl.b0 + (int64_t)bid * wl + wI;
      const T12 T = jointRelTransform(bodies[body], qFk, B, bw < mdl.b1 ? bw : mdl.b1 - 1);
      double* dst = fkT + (size_t)item * 12;
#pragma unroll
      for (int e = 0; e < 9; e++) dst[e] = T.R.m[e];
      dst[9] = T.p.x; dst[10] = T.p.y; dst[11] = T.p.z;
    }
    __syncthreads();
  }
  // the counter of the unresolved-worlds list of this slice starts at zero for the solve kernel that follows on the stream
  // (a separate hipMemsetAsync node cost ~6 us of every forward step)
  if (failCount && bid == 0 && tid == 0) *failCount = 0u;
  NBL_PHASE_FIRST(56);
  // ppw lanes per world (1, 2 or 4): the narrow phases of ppw collider pairs of a world run side by side, each lane parks its
  // candidate contacts in LDS, and the world's first lane then accepts them in pair order - exactly the order and the filters
  // of the one-lane loop, at about 1 / ppw of its dependent chain (two foot-ground pairs: 74k -> ~40k cycles).
  const bool extra = tid >= wl * ppw;                 // threads of a wider workgroup: barriers only
  const int pl = tid % ppw;
  const int64_t b = mdl.b0 + (int64_t)bid * wl + tid / ppw;
  const bool valid = !extra && b < mdl.b1;
  if (!valid && ppw == 1) return;
  const int64_t bs = valid ? b : mdl.b1 - 1;       // lanes of a padding world repeat the last world's narrow phase, store nothing
  Ctx c = makeCtx(mdl, bodies, nullptr, ws, B, bs, saved, &lay);
  NBL_PHASE_FIRST(29);
  const int ltid = extra ? 0 : tid;
  LaneBuf clip; clip.base = clipBuf + ltid; clip.ls = ls;
  int nC = 0, nDropped = 0;
  bool overflow = false, edge = false;
  // accept one candidate (lane pl == 0 of the world, or the only lane): postProcess + depth filter + append to the record
  auto acceptRec = [&](const double* ct, int stride, int pi) {   // ct: CR layout, element e at ct[e * stride]
    const V3 pt = mk3(ct[(CR_POINT + 0) * stride], ct[(CR_POINT + 1) * stride], ct[(CR_POINT + 2) * stride]);
    const V3 nr = mk3(ct[(CR_NORMAL + 0) * stride], ct[(CR_NORMAL + 1) * stride], ct[(CR_NORMAL + 2) * stride]);
    const double depth = ct[CR_DEPTH * stride];
    // skip points within 3e-12 of a contact already in the collision result (DARTCollisionDetector.cpp:360-400: postProcess compares with
    // EVERY contact the detector has added so far - the constraint solver's depth / zero-normal filters come later, ConstraintSolver.cpp:
    // 598-601 - so a point the depth filter drops still shadows a later contact at the same place: a sphere on the corner of a box whose
    // corner is deep in the ground).  Seen points live in LDS: the kept contacts in slots 0 .. nC-1, dropped ones from the top down.
    bool close = false;
    for (int e = 0; e < nC; e++) {
      const V3 d = pt - mk3(keptP[(e * 3 + 0) * ls + ltid], keptP[(e * 3 + 1) * ls + ltid], keptP[(e * 3 + 2) * ls + ltid]);
      if (norm3(d) < 3.0e-12) { close = true; break; }
    }
    for (int e = seenPts - nDropped; e < seenPts && !close; e++) {
      const V3 d = pt - mk3(keptP[(e * 3 + 0) * ls + ltid], keptP[(e * 3 + 1) * ls + ltid], keptP[(e * 3 + 2) * ls + ltid]);
      if (norm3(d) < 3.0e-12) close = true;
    }
    if (close) return;
    // A full list cannot remember another point.  A contact that passes the depth filter after that may be the duplicate of a point that
    // was not remembered: the world is flagged like one with too many contacts (NBL_ST_CONTACT_OVERFLOW).
    const bool full = nC + nDropped >= seenPts;
    if (dot(nr, nr) < 1e-12 || depth < 0.0 || depth > cm->clippingDepth) {
      if (!full) {
        nDropped++;
        const int e = seenPts - nDropped;
        keptP[(e * 3 + 0) * ls + ltid] = pt.x; keptP[(e * 3 + 1) * ls + ltid] = pt.y; keptP[(e * 3 + 2) * ls + ltid] = pt.z;
      }
      return;
    }
    if (nC >= cm->maxContacts) { overflow = true; return; }
    if (full) { overflow = true; if (nDropped > 0) nDropped--; }   // (the kept point takes the slot of the last remembered dropped one)
    const int r0 = lay.contacts + nC * CR_SIZE;
    keptP[(nC * 3 + 0) * ls + ltid] = pt.x; keptP[(nC * 3 + 1) * ls + ltid] = pt.y; keptP[(nC * 3 + 2) * ls + ltid] = pt.z;
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
  // BodyNode::mWorldTransform of a collider body: from the tree block the forward kernel left, or - qFk - the product of the joint
  // transforms down its ancestor chain (ancestors are numbered before their descendants)
  auto worldT = [&](int body) -> T12 {
    if (!qFk) return ldTAt(c, body, WS_TW);
    if (fkT) {
      const int nNeed = __builtin_popcountll(fkNeed);
      const double* mine = fkT + (size_t)(ltid / ppw) * nNeed * 12;
      auto rel = [&](int i) -> T12 {
        const double* t = mine + 12 * __builtin_popcountll(fkNeed & ((1ull << i) - 1ull));
        T12 T;
#pragma unroll
        for (int e = 0; e < 9; e++) T.R.m[e] = t[e];
        T.p = mk3(t[9], t[10], t[11]);
        return T;
      };
      uint64_t chain = cm->ancestors[body];
      NBL_PHASE_FIRST(30);
      T12 TW = rel(__builtin_ctzll(chain));
      chain &= chain - 1;
      NBL_PHASE_FIRST(31);
      while (chain) {
        TW = mulT(TW, rel(__builtin_ctzll(chain)));
        chain &= chain - 1;
      }
      return TW;
    }
    uint64_t chain = cm->ancestors[body];
    T12 TW = jointRelTransform(bodies[__builtin_ctzll(chain)], qFk, B, bs);
    chain &= chain - 1;
    while (chain) {
      TW = mulT(TW, jointRelTransform(bodies[__builtin_ctzll(chain)], qFk, B, bs));
      chain &= chain - 1;
    }
    return TW;
  };
  // narrow phase of collider pair pi, every contact handed to emit(ct)
  auto runPair = [&](int pi, auto emit) {
    const DevBox& ba = cm->boxes[cm->pairA[pi]];
    const DevBox& bb = cm->boxes[cm->pairB[pi]];
    T12 Ta = cT(ba.T), Tb = cT(bb.T);
    if (ba.body >= 0) Ta = mulT(worldT(ba.body), Ta);
    if (bb.body >= 0) Tb = mulT(worldT(bb.body), Tb);
    NBL_PHASE_FIRST(28);
    // dispatch on the two shape types (collide(), DARTCollide.cpp:5030-5260)
    const V3 ha = mk3(ba.half[0], ba.half[1], ba.half[2]), hb = mk3(bb.half[0], bb.half[1], bb.half[2]);
    const bool sa = ba.shape == SHAPE_SPHERE, sb = bb.shape == SHAPE_SPHERE;
    const bool ca = ba.shape == SHAPE_CAPSULE, cb = bb.shape == SHAPE_CAPSULE;   // half = (radius, height / 2, -)
    if (ca && cb) capsuleCapsule(2 * ba.half[1], ba.half[0], Ta, 2 * bb.half[1], bb.half[0], Tb, cm->clippingDepth, emit);
    else if (sa && cb) sphereCapsulePair(true, ba.half[0], Ta, 2 * bb.half[1], bb.half[0], Tb, cm->clippingDepth, emit);
    else if (ca && sb) sphereCapsulePair(false, bb.half[0], Tb, 2 * ba.half[1], ba.half[0], Ta, cm->clippingDepth, emit);
    else if (ca || cb) {}   // capsule-box: refused at model creation
    else if (sa && sb) sphereSphere(ba.half[0], Ta, bb.half[0], Tb, cm->clippingDepth, emit);
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
    const int wI = ltid / ppw;
    double* mine = stage + (size_t)(wI * (ppw - 1) + (pl > 0 ? pl - 1 : 0)) * (8 * CR_SIZE);
    int* counts = reinterpret_cast<int*>(stage + (size_t)wl * (ppw - 1) * (8 * CR_SIZE));
    for (int p0 = 0; p0 < nPairs; p0 += ppw) {       // ppw pairs of every world at a time
      // ONE call for both kinds of lane - two calls with different continuations are two copies of the narrow phase, and the wavefront
      // runs the copies of a divergent branch one after the other (measured: 71 k cycles for the two pairs of a world, 35 k each)
      int cnt = 0;
      if (!extra && (pl == 0 ? valid : p0 + pl < nPairs))
        runPair(p0 + pl, [&](const DevContact& ct) {
          if (pl == 0) { double rec[CR_SIZE]; toRec(ct, rec); acceptRec(rec, 1, p0); }
          else if (cnt < 8) { toRec(ct, mine + cnt * CR_SIZE); cnt++; }
        });
      if (!extra && pl != 0) counts[wI * (ppw - 1) + pl - 1] = cnt;
      NBL_PHASE_FIRST(57);
      __syncthreads();
      NBL_PHASE_FIRST(58);
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
  NBL_PHASE_FIRST(59);
  // ---- joint-limit constraint rows (JointLimitConstraint::update, JointLimitConstraint.cpp:182-237): every limit-enforcing DOF at or
  //      below its lower / at or above its upper limit is appended as a pseudo-contact after the contacts (ConstraintSolver.cpp:641-696
  //      pushes the joint-limit constraints after the contact constraints) ----
  int nLim = 0;
  for (int k = 0; k < cm->nLimitDofs; k++) {
    const int d = cm->limitDof[k];
    const double qd = qFk ? qFk[(int64_t)d * B + b] : svAt(saved, lay.q + d, B, b);
    double sigma = 0.0;
    if (qd - cm->limitLo[k] <= 0.0) sigma = 1.0;
    else if (qd - cm->limitHi[k] >= 0.0) sigma = -1.0;
    if (sigma == 0.0) continue;
    if (nC >= cm->maxContacts) { overflow = true; continue; }
    const int r0 = lay.contacts + nC * CR_SIZE;
    const int body = cm->limitBody[k];
    for (int e = 0; e < CR_SIZE; e++) svAt(saved, r0 + e, B, b) = 0.0;
    svAt(saved, r0 + CR_NORMAL + 1, B, b) = 1.0;                  // (a unit vector: the tangent-basis code runs on every record)
    svAt(saved, r0 + CR_TYPE, B, b) = (double)CT_LIMIT;
    svAt(saved, r0 + CR_BOXA, B, b) = (double)(CR_BODY_CODE + 1 + body);
    svAt(saved, r0 + CR_BOXB, B, b) = (double)(CR_BODY_CODE + 1 + bodies[body].parent);
    svAt(saved, r0 + CR_EA_FIXED, B, b) = (double)d;
    svAt(saved, r0 + CR_EA_FIXED + 1, B, b) = sigma;
    nC++; nLim++;
  }
  svAt(saved, lay.nc, B, b) = (double)nC + ((qFk && overflow) ? 0.5 : 0.0);
  uint32_t st = 0;
  if (nC - nLim > 0) st |= 0x1u;
  if (nLim > 0) st |= 0x400u;
  if (overflow) st |= 0x80u;
  (void)edge;
  if (status && !qFk) status[b] |= st;  // the forward tree kernel initialised the word (0, or NBL_ST_NAN for a non-finite unconstrained step)
  NBL_PHASE_FIRST(60);
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

// lanes of a stand-alone narrow-phase workgroup (the lane stride of its static LDS buffers): the remembered points of the general
// instantiation (128 per world) do not fit 64 lanes' worth of LDS
constexpr int DETECT_LS = SEEN_POINTS > 32 ? 16 : 64;
__global__ __launch_bounds__(64) void k_contact_detect(DevModel mdl, const DevBody* __restrict__ bodies,
                                                       const DevContactModel* __restrict__ cm, int64_t B,
                                                       double* __restrict__ saved, SavedLayout lay,
                                                       uint32_t* __restrict__ status, double* __restrict__ ws, int doTwists,
                                                       uint32_t* __restrict__ failCount, int ppw) {
  extern __shared__ __attribute__((aligned(16))) double stage[];   // [thread][8 candidates][CR_SIZE] + counts (ppw > 1 only)
  __shared__ double keptP[SEEN_POINTS * 3 * DETECT_LS];   // accepted contact points of the workgroup's worlds, [contact][xyz][lane]
  __shared__ double clipBuf[48 * DETECT_LS];               // clip polygons of boxBox, [entry][lane]
  contactDetectBody(mdl, bodies, cm, B, saved, lay, status, ws, doTwists, failCount, ppw, nullptr, (int)blockIdx.x, (int)blockDim.x / ppw,
                    keptP, clipBuf, stage, nullptr, DETECT_LS);
}

}  // namespace NBL_NS
