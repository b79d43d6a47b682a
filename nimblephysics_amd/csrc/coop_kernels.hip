// coop_kernels.hip — one world per WAVEFRONT for the dense part of the contact stage (see coop_dev.hpp).
//   k_contact_rows_coop      contact Jacobians, b, the unit-impulse tests and A
//   k_contact_solve_coop     stage 0 of the LCP cascade + v' = v_pre + M^-1 J^T x; worlds it cannot resolve go to the
//                            compacted list of k_contact_cascade_stages / k_contact_cascade_final (stages 1-3)
//   k_bwd_contact_a_coop     the dense (c x c) part of the contact adjoint;  k_bwd_contact_b_coop  its tree part
//   k_bwd_bounce             the reference's bounce approximation (restitution)
#include "coop_dev.hpp"
#include "coop_dantzig_dev.hpp"
#include "coop_wave_dev.hpp"

namespace NBL_NS {

// Scratch rows of an unresolved world between k_contact_solve_coop, k_contact_cascade_stages and k_contact_cascade_final
// (rows LW_JA .. of the per-world contact scratch):
constexpr int LW_X0 = LW_JA;                            // MAX_ROWS: the pre-solve x (mXBackup) of every row
constexpr int LW_OKX = LW_JA + MAX_ROWS;                // MAX_ROWS: x of the rows whose constrained group stage 0 resolved ...
constexpr int LW_OKCLS = LW_JA + 2 * MAX_ROWS;          // MAX_ROWS: ... and their row class (+-2 = upper bound with E = +-mu)
constexpr int LW_STAGE_X = LW_JB;                       // 3 x MAX_ROWS candidate solutions (a group's rows only hold its own)
constexpr int LW_STAGE_FLAGS = LW_JB + 3 * MAX_ROWS;    // 3 x MAX_CONTACTS flag words [stage][group] (as doubles)
constexpr int LW_FAILMASK = LW_STAGE_FLAGS + 3 * MAX_CONTACTS;   // bit g: constrained group g was not resolved by stage 0
constexpr int LW_STAGE_CYCLES = LW_FAILMASK + 1;        // NBL_CASCADE_TIMING: cycles of the stage waves and of the final kernel
static_assert(LW_STAGE_CYCLES + 9 <= LW_TOTAL, "stage results must fit the contact scratch rows");

template <bool LIM = false>
DEV void coopLoadRow(CoopRow& R, int ln, int m, const double* __restrict__ saved, const double* __restrict__ dn,
                     const SavedLayout& lay, const DevContactModel* __restrict__ cm, int64_t B, int64_t b) {
  R.m = m;
  R.fric = (ln % 3) != 0;
  R.fp = ln < MAXR ? ln - (ln % 3) : 0;
  R.mu = 0.0; R.Bv = 0.0;
  const bool on = ln < m;
  if (on) {
    const int r0 = lay.contacts + (ln / 3) * CR_SIZE;
    const int cA = (int)saved[(int64_t)(r0 + CR_BOXA) * B + b], cB = (int)saved[(int64_t)(r0 + CR_BOXB) * B + b];
    const double muA = LIM ? crMuOf(cm, cA) : cm->boxes[cA].mu, muB = LIM ? crMuOf(cm, cB) : cm->boxes[cB].mu;
    if (LIM && cA >= CR_BODY_CODE) { R.lim = (ln % 3) == 0; R.neg = R.lim && saved[(int64_t)(r0 + CR_EA_FIXED + 1) * B + b] < 0.0; }
    R.mu = muA < muB ? muA : muB;
    if (!(R.mu > 1e-3)) R.mu = 0.0;   // frictionless contact: its tangent rows are empty (k_contact_rows_coop) and pinned to 0
    R.Bv = saved[(int64_t)(lay.b + ln) * B + b];
  }
  R.on = on;
  if (LIM) R.limMask = (RowMask)__ballot(R.lim ? 1 : 0);
  R.Acol = dn + lay.A + (on ? ln : 0);
  double cn = 0.0, cn1 = 0.0;
#pragma unroll 1
  for (int ib = 0; ib < MAXR; ib += 8) {
#pragma unroll
    for (int iq = 0; iq < 8; iq += 2) { const double x = R.a(ib + iq), y = R.a(ib + iq + 1); cn = fma(x, x, cn); cn1 = fma(y, y, cn1); }
  }
  R.colNorm = cn + cn1;
}

// Constrained groups of a world's contacts (ConstraintSolver::buildConstrainedGroups :724-780, ContactConstraint::uniteSkeletons
// :879-907): skeletons connected by a contact between two reactive bodies are one group, contacts with world-fixed colliders connect
// nothing; groups are numbered by their first contact.  Returns the number of groups, gid = the group of this lane's row.  The
// reference runs its whole solver cascade once per group; almost every world has one.
template <class W>
DEV int coopGroups(const W& w, const DevContactModel* __restrict__ cm, const double* __restrict__ saved, const SavedLayout& lay,
                   int64_t B, int64_t b, int m, int& gid) {
  const int ln = w.lane();
  const int nC = m / 3;
  int u = 0, v = 0;                       // lane c < nC: the skeleton(s) contact c acts on
  if (ln < nC) {
    const int r0 = lay.contacts + ln * CR_SIZE;
    const int bxA = (int)saved[(int64_t)(r0 + CR_BOXA) * B + b], bxB = (int)saved[(int64_t)(r0 + CR_BOXB) * B + b];
    const int bA = crBodyOf(cm, bxA), bB = crBodyOf(cm, bxB);   // (a joint-limit row: child and parent body of its joint, one skeleton)
    const int sA = bA >= 0 ? cm->skelOf[bA] : -1, sB = bB >= 0 ? cm->skelOf[bB] : -1;
    u = sA >= 0 ? sA : sB;
    v = (sA >= 0 && sB >= 0) ? sB : u;
    if (u < 0) { u = 0; v = 0; }
  }
  // the usual case first: every contact acts on one and the same skeleton (a robot standing on the ground)
  const int u0 = w.bcastI(u, 0);
  if (w.ballot(ln < nC && (u != u0 || v != u0)) == 0ull) { gid = 0; return 1; }
  int lab = ln;                           // lane s: label of skeleton s; uniting relabels a whole component, so one pass over the contacts is enough
  for (int c = 0; c < nC; c++) {
    const int uc = w.bcastI(u, c), vc = w.bcastI(v, c);
    const int lu = w.bcastI(lab, uc), lv = w.bcastI(lab, vc);
    const int mn = lu < lv ? lu : lv;
    if (lab == lu || lab == lv) lab = mn;
  }
  const int myLab = w.shflI(lab, u);      // lane c: the component of contact c
  uint32_t eq = 0;
  for (int c = 0; c < nC; c++) { const int lc = w.bcastI(myLab, c); if (myLab == lc) eq |= 1u << c; }
  const int firstOf = (ln < nC && eq != 0u) ? __builtin_ctz(eq) : 0;
  const uint32_t firstMask = (uint32_t)w.ballot(ln < nC && firstOf == ln);
  const int g = __builtin_popcount(firstMask & ((1u << firstOf) - 1u));
  gid = w.shflI(g, ln < MAXR ? ln / 3 : 0);
  return __builtin_popcount(firstMask);
}

// x, classes, cfm, warm start, v' = v_pre + M^-1 J^T x and (when valid) the pseudo-inverse of the final Q -> saved record
template <class W>
DEV uint32_t coopContactOutputs(const W& w, CoopLds& S, const CoopRow& R, int n, int m, double X, const CoopClasses& K, double cfm, bool pinvValidIn,
                            double* __restrict__ saved, const SavedLayout& lay, double* __restrict__ dn,
                            double* __restrict__ cacheOut, double* __restrict__ nv, int64_t B, int64_t b) {
  const int ln = w.lane();
  // Joint-limit rows: the reference's backward pass gives them a zero constraint-force column (model_dev.hpp), which takes them out of
  // every Jacobian: the record classifies them "not clamping", and a Q^+ that couples them to the contacts is not the backward pass's.
  // The warm-start cache carries the reference's sign of a negated (upper-limit) row.
  bool pinvValid = pinvValidIn;
  if ((K.clampMask & R.limMask) != 0) {   // (uniform; never taken by the instantiation without joint-limit rows)
    CoopClasses Kb = K;
    if (R.lim) Kb.cls = RC_NOT_CLAMPING;
    Kb.clampMask &= ~R.limMask;
    Kb.nc = rmPop(Kb.clampMask);
    if (Kb.nc > 0) {
      double a[MAXR];
      coopBuildQ(w, S, R, Kb, cfm, a);
      coopPinvOfQ(w, a, S, Kb);
    } else if (ln < MAXR) {
#pragma unroll 1
      for (int i = 0; i < MAXR; i++) S.P[i * CLD + ln] = 0.0;
    }
    w.sync();
    pinvValid = true;
  }
  if (ln < MAX_ROWS) {
    svAt(saved, lay.x + ln, B, b) = X;
    // (3: a joint-limit row that was clamping - "not clamping" to every reader, k_bwd_contact_a_coop looks at it for the precise-inverse switch)
    svAt(saved, lay.cls + ln, B, b) = R.lim ? (K.cls == RC_CLAMPING ? 3.0 : 0.0) : (K.cls == RC_UPPER_BOUND ? (K.E > 0 ? 2.0 : -2.0) : (double)K.cls);
    if (cacheOut) cacheOut[(int64_t)ln * B + b] = R.neg ? -X : X;
  }
  if (ln == MAX_ROWS && cacheOut) cacheOut[(int64_t)MAX_ROWS * B + b] = (double)m;
  if (ln < MAX_ROWS) svAt(saved, lay.cfm + ln, B, b) = cfm;     // this row's constant (its group's, CFM_CONSTANTS)
  if (ln == 0) svAt(saved, lay.pflag, B, b) = pinvValid ? 1.0 : 0.0;
  // v' = v_pre + M^-1 J^T x  (lane = DOF), and the velocity change the BACKWARD pass works with: the reference's Jacobians take
  // A_c f_c + A_ub E f_c (BackpropSnapshot.cpp:980-1066), the impulses of the clamping rows and, for a friction row on its bound, E times
  // its normal's - the same as x on a standardised world; on a world that fell through every solver stage the raw PGS iterate also
  // holds impulses below the classification thresholds, which those Jacobians do not know.  Joint-limit rows: never (no force column).
  double vNext = 0.0;
  {
    const double Xn = w.shfl(X, R.fp);
    const double xb = R.lim ? 0.0 : (K.cls == RC_CLAMPING ? X : (K.cls == RC_UPPER_BOUND ? K.E * Xn : 0.0));
    if (ln < MAXR) { S.vec[2][ln] = X; S.vec[1][ln] = xb; }
  }
  w.sync();
  if (ln < n) {
    double wd = 0.0, wd1 = 0.0, wb = 0.0, wb1 = 0.0;
#pragma unroll 1
    for (int rb = 0; rb < MAXR; rb += 8) {
#pragma unroll
      for (int rq = 0; rq < 8; rq += 2) {
        const int r = rb + rq;
        const double m0 = r < m ? dn[lay.massed + ln * MAX_ROWS + r] : 0.0, m1 = r + 1 < m ? dn[lay.massed + ln * MAX_ROWS + r + 1] : 0.0;   // columns >= m were never written
        wd = fma(m0, S.vec[2][r], wd); wd1 = fma(m1, S.vec[2][r + 1], wd1);
        wb = fma(m0, S.vec[1][r], wb); wb1 = fma(m1, S.vec[1][r + 1], wb1);
      }
    }
    wd += wd1;
    svAt(saved, lay.w + ln, B, b) = wb + wb1;
    vNext = svAt(saved, lay.vpre + ln, B, b) + wd;
    nv[(int64_t)ln * B + b] = vNext;
  }
  if (pinvValid && ln < MAXR) {
#pragma unroll 1
    for (int ib = 0; ib < MAXR; ib += 8) {
#pragma unroll
      for (int iq = 0; iq < 8; iq++) dn[lay.pinv + (ib + iq) * MAX_ROWS + ln] = S.P[(ib + iq) * CLD + ln];
    }
  }
  return w.ballot(!__builtin_isfinite(vNext)) != 0ull ? 0x40u : 0u;   // NBL_ST_NAN: a non-finite next velocity (poisoned inputs end up here)
}

// MULTI: the model has colliders on more than one skeleton, so a world can hold several constrained groups (decided when the model
// is created; the single-skeleton instantiation carries none of the group code).
template <bool MULTI>
__global__ __launch_bounds__(64) NBL_WAVES(NBL_W_SOLVE) void k_contact_solve_coop(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                           double* __restrict__ saved, SavedLayout lay,
                                                           const double* __restrict__ cacheIn, double* __restrict__ cacheOut,
                                                           double* __restrict__ next, uint32_t* __restrict__ status,
                                                           double* __restrict__ lws, int32_t* __restrict__ failList,
                                                           uint32_t* __restrict__ failCount) {
  __shared__ CoopLds S;
  const DevWave w;
  const int ln = w.lane();
  NBL_PHASE(40);
  const int64_t b = mdl.b0 + coopWorld(blockIdx.x, gridDim.x);
  if (b >= mdl.b1) return;
#ifdef NBL_CASCADE_TIMING
  const long long tSolve0 = clock64();
#define NBL_SOLVE_CYCLES() do { if (ln == 0) lws[(int64_t)(LW_STAGE_CYCLES + 4) * B + b] = (double)(clock64() - tSolve0); } while (0)
#else
#define NBL_SOLVE_CYCLES() do { } while (0)
#endif
  const int n = mdl.n;
  const double ncD = svAt(saved, lay.nc, B, b);
  const int nC = (int)ncD;
  const int m = 3 * nC;
  // the narrow phase that runs NEXT TO the forward tree kernel (k_forward_detect_coop) leaves the status word to this kernel: contacts
  // present, contacts dropped (count + 0.5).  Idempotent after the stand-alone narrow phase, which sets the bits itself.
  if (ln == 0 && status) status[b] |= ((!MULTI && nC > 0) ? 0x1u : 0u) | (ncD - (double)nC > 0.25 ? 0x80u : 0u);
  double* nv = next + (int64_t)n * B;
  double* dn = denseOf(saved, lay, B, b);
  if (m == 0) {
    if (cacheOut && ln <= MAX_ROWS) cacheOut[(int64_t)ln * B + b] = 0.0;
    if (ln < MAX_ROWS) { svAt(saved, lay.x + ln, B, b) = 0.0; svAt(saved, lay.cls + ln, B, b) = 0.0; }
    if (ln < MAX_ROWS) svAt(saved, lay.cfm + ln, B, b) = 0.0;
    if (ln == 0) svAt(saved, lay.pflag, B, b) = 0.0;
    if (ln < n) svAt(saved, lay.w + ln, B, b) = 0.0;
    NBL_SOLVE_CYCLES();
    return;
  }
  CoopRow R;
  coopLoadRow<MULTI>(R, ln, m, saved, dn, lay, cm, B, b);
  NBL_PHASE(41);
  if (MULTI && ln == 0 && status) {   // (joint-limit rows are pseudo-contacts of the record: NBL_ST_CONTACT counts the real ones)
    const int nLim = rmPop(R.limMask);
    status[b] |= (nC - nLim > 0 ? 0x1u : 0u) | (nLim > 0 ? 0x400u : 0u);
  }
  const bool haveCache = cacheIn && ((int)cacheIn[(int64_t)MAX_ROWS * B + b] == m);
  const double Xcache = (haveCache && ln < m) ? (R.neg ? -1.0 : 1.0) * cacheIn[(int64_t)ln * B + b] : 0.0;
  NBL_PHASE(42);
  int gid = 0;
  const int nGroups = MULTI ? coopGroups(w, cm, saved, lay, B, b, m, gid) : 1;
  if (!MULTI || nGroups == 1) {
    CoopStage0 out;
    coopStage0(w, S, R, haveCache, Xcache, out);
#ifdef NBL_CASCADE_TIMING
    if (ln == 0) {
      lws[(int64_t)(LW_STAGE_CYCLES + 5) * B + b] = (double)(out.tGuess - tSolve0);
      lws[(int64_t)(LW_STAGE_CYCLES + 6) * B + b] = (double)(clock64() - out.tGuess);
      lws[(int64_t)(LW_STAGE_CYCLES + 7) * B + b] = (double)out.nu;
      lws[(int64_t)(LW_STAGE_CYCLES + 8) * B + b] = (double)out.fast;
    }
#endif
    if (out.ok) {
      const uint32_t nanBit = coopContactOutputs(w, S, R, n, m, out.X, out.K, 0.0, out.pinvValid, saved, lay, dn, cacheOut, nv, B, b);
      if (ln == 0 && status) status[b] |= 0x2u | 0x100u | nanBit;
      NBL_PHASE(47);
    } else {
      // the pre-solve x (mXBackup) is what the PGS fallback starts from (BoxedLcpConstraintSolver.cpp:541-547)
      if (ln < MAX_ROWS) lws[(int64_t)(LW_X0 + ln) * B + b] = out.X0;
      if (ln == 0) { lws[(int64_t)LW_FAILMASK * B + b] = 1.0; const uint32_t slot = atomicAdd(failCount, 1u); failList[slot] = (int32_t)b; }
    }
    NBL_SOLVE_CYCLES();
    return;
  }
  // Several constrained groups: stage 0 group by group (rows of the other groups switched off).  Groups that resolve keep their
  // result whatever happens to the others; the world goes to the cascade kernels with the mask of the groups that did not.
  double X = 0.0, X0 = 0.0, E = 0.0;
  int cls = RC_NOT_CLAMPING;
  uint32_t failMask = 0u;
  for (int g = 0; g < nGroups; g++) {
    CoopRow Rg = R;
    Rg.on = R.on && gid == g;
    CoopStage0 out;
    coopStage0(w, S, Rg, haveCache, Xcache, out);
    if (Rg.on) { X = out.X; X0 = out.X0; cls = out.K.cls; E = out.K.E; }
    if (!out.ok) failMask |= 1u << g;
  }
  if (failMask == 0u) {
    CoopClasses K;
    K.cls = cls; K.E = E;
    K.clampMask = (RowMask)w.ballot(cls == RC_CLAMPING); K.ubMask = (RowMask)w.ballot(cls == RC_UPPER_BOUND);
    K.nc = rmPop(K.clampMask); K.nu = rmPop(K.ubMask);
    bool pinvValid = false;
    if (K.nc > 0) {                                       // Q^+ of the whole clamping set (block diagonal over the groups) for the record
      double a[MAXR];
      coopBuildQ(w, S, R, K, 0.0, a);
      coopPinvOfQ(w, a, S, K);
      pinvValid = true;
    }
    const uint32_t nanBit = coopContactOutputs(w, S, R, n, m, X, K, 0.0, pinvValid, saved, lay, dn, cacheOut, nv, B, b);
    if (ln == 0 && status) status[b] |= 0x2u | 0x100u | nanBit;
  } else {
    // rows of resolved groups: their result (x, class) waits in the scratch rows for k_contact_cascade_final
    if (ln < MAX_ROWS) {
      lws[(int64_t)(LW_X0 + ln) * B + b] = X0;
      lws[(int64_t)(LW_OKX + ln) * B + b] = X;
      lws[(int64_t)(LW_OKCLS + ln) * B + b] = cls == RC_UPPER_BOUND ? (E > 0 ? 2.0 : -2.0) : (double)cls;
    }
    if (ln == 0) { lws[(int64_t)LW_FAILMASK * B + b] = (double)failMask; const uint32_t slot = atomicAdd(failCount, 1u); failList[slot] = (int32_t)b; }
  }
}

// Stages 1-3 of the cascade for the worlds stage 0 could not resolve (compacted list): one WORKGROUP OF TWO WAVEFRONTS per
// failed world.  Wavefront 0 runs stage 1 (reduce + Dantzig); wavefront 1 runs stage 2 (CFM + reduce + PGS) and then, unless
// stage 2 already produced a valid solution (stage 3 is only ever consulted when it did not), stage 3 (friction dropped + PGS).
// The stages only share their inputs, see coop_dantzig_dev.hpp, so the Dantzig solve - the longest of the three - overlaps with
// both PGS stages; each wavefront leaves its candidate solutions and flags in the world's scratch rows and
// k_contact_cascade_final picks one in the reference's order of preference and standardises it.  The wavefronts of a group never
// wait for each other (DevWaveInGroup::sync is a wave-level fence, not a workgroup barrier).

template <bool MULTI>
__global__ __launch_bounds__(128) NBL_WAVES(NBL_W_STAGES) void k_contact_cascade_stages(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                               double* __restrict__ saved, SavedLayout lay,
                                                               double* __restrict__ lws, const int32_t* __restrict__ failList,
                                                               const uint32_t* __restrict__ failCount) {
  __shared__ CascadeLds C1;
  __shared__ PgsLds C2;
  if (blockIdx.x >= *failCount) return;
#ifdef NBL_CASCADE_TIMING
  const long long t0 = clock64();
#endif
  const DevWaveInGroup w;
  const int ln = w.lane();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t b = failList[blockIdx.x];
  const int m = 3 * (int)svAt(saved, lay.nc, B, b);
  double* dn = denseOf(saved, lay, B, b);
  CoopRow R;
  coopLoadRow<MULTI>(R, ln, m, saved, dn, lay, cm, B, b);
  const double X0 = ln < m ? lws[(int64_t)(LW_X0 + ln) * B + b] : 0.0;
  // the constrained groups stage 0 left unresolved, one after the other (almost always: the world's only group)
  const uint32_t failMask = MULTI ? (uint32_t)lws[(int64_t)LW_FAILMASK * B + b] : 1u;
  int gid = 0;
  if (MULTI) coopGroups(w, cm, saved, lay, B, b, m, gid);
  const bool rowOn = R.on;
#pragma unroll 1
  for (int g = 0; g < (MULTI ? MAX_CONTACTS : 1); g++) {
    if (!((failMask >> g) & 1u)) continue;
    R.on = rowOn && gid == g;
    CoopRow& Rg = R;
    auto publish = [&](int stage, const CoopStageResult& r) {
      if (Rg.on) lws[(int64_t)(LW_STAGE_X + stage * MAX_ROWS + ln) * B + b] = r.X;
      if (ln == 0) lws[(int64_t)(LW_STAGE_FLAGS + stage * MAX_CONTACTS + g) * B + b] = (double)r.flags;
    };
    CoopStageResult r;
    if (wave == 0) {
      coopCascadeStage1(w, C1, Rg, X0, r);
      publish(0, r);
#ifdef NBL_CASCADE_TIMING
      if (ln == 0) lws[(int64_t)(LW_STAGE_CYCLES + 0) * B + b] = (double)(clock64() - t0);
#endif
    } else {
      coopCascadeStage2(w, C2, Rg, X0, cm->fallbackCfm, r);
      publish(1, r);
#ifdef NBL_CASCADE_TIMING
      const long long t1 = clock64();
      if (ln == 0) lws[(int64_t)(LW_STAGE_CYCLES + 1) * B + b] = (double)(t1 - t0);
#endif
      const bool stage2Valid = (r.flags & (CS_SOLVED | CS_VALID)) == (CS_SOLVED | CS_VALID);
      r.X = 0.0; r.flags = 0;
      if (!stage2Valid) coopCascadeStage3(w, C2, Rg, X0, cm->fallbackCfm, r);
      publish(2, r);
#ifdef NBL_CASCADE_TIMING
      if (ln == 0) lws[(int64_t)(LW_STAGE_CYCLES + 2) * B + b] = (double)(clock64() - t1);
#endif
    }
    w.sync();
  }
}

// Select + standardise (per unresolved constrained group) + outputs for one unresolved world (one wavefront).  The candidates of the
// three stages come from `stageX` (3 x MAX_ROWS doubles, stride `sx` between entries) and `stageFlags` (3 x MAX_CONTACTS flag words as
// doubles, stride `sf`): the world's scratch rows in global memory (k_contact_cascade_final) or the workgroup's LDS (fused kernel).
template <bool MULTI, class W>
DEV void coopCascadeFinish(const W& w, CoopLds& S, const DevModel& mdl, const DevContactModel* __restrict__ cm, int64_t B, int64_t b,
                           double* __restrict__ saved, const SavedLayout& lay, double* __restrict__ cacheOut, double* __restrict__ next,
                           uint32_t* __restrict__ status, const double* __restrict__ lws, const double* stageX, int64_t sx,
                           const double* stageFlags, int64_t sf) {
  const int ln = w.lane();
  const int n = mdl.n;
  const int m = 3 * (int)svAt(saved, lay.nc, B, b);
  double* nv = next + (int64_t)n * B;
  double* dn = denseOf(saved, lay, B, b);
  const int row = ln < MAX_ROWS ? ln : 0;
  const double X0 = ln < m ? lws[(int64_t)(LW_X0 + ln) * B + b] : 0.0;
  CoopRow R;
  coopLoadRow<MULTI>(R, ln, m, saved, dn, lay, cm, B, b);
  const uint32_t failMask = MULTI ? (uint32_t)lws[(int64_t)LW_FAILMASK * B + b] : 1u;
  int gid = 0;
  const int nGroups = MULTI ? coopGroups(w, cm, saved, lay, B, b, m, gid) : 1;
  // rows of the groups stage 0 resolved come with their result; the others are filled in below, group by group
  double X = 0.0, cfmRow = 0.0, E = 0.0;
  int cls = RC_NOT_CLAMPING;
  if (nGroups > 1 && R.on && !((failMask >> gid) & 1u)) {   // (R.on: still every row of the world here)
    X = lws[(int64_t)(LW_OKX + row) * B + b];
    const double cv = lws[(int64_t)(LW_OKCLS + row) * B + b];
    cls = cv == 1.0 ? RC_CLAMPING : ((cv == 2.0 || cv == -2.0) ? RC_UPPER_BOUND : RC_NOT_CLAMPING);
    E = cls == RC_UPPER_BOUND ? (cv > 0 ? R.mu : -R.mu) : 0.0;
  }
  uint32_t st = 0x100u;          // standardised unless a group says otherwise
  bool pinvValid = false;
  const bool rowOn = R.on;
#pragma unroll 1
  for (int g = 0; g < (MULTI ? MAX_CONTACTS : 1); g++) {
    if (!((failMask >> g) & 1u)) continue;
    R.on = rowOn && gid == g;
    CoopRow& Rg = R;
    CoopStageResult r1, r2, r3;
    r1.X = Rg.on ? stageX[(int64_t)row * sx] : 0.0;
    r2.X = Rg.on ? stageX[(int64_t)(MAX_ROWS + row) * sx] : 0.0;
    r3.X = Rg.on ? stageX[(int64_t)(2 * MAX_ROWS + row) * sx] : 0.0;
    r1.flags = (int)stageFlags[(int64_t)(0 * MAX_CONTACTS + g) * sf];
    r2.flags = (int)stageFlags[(int64_t)(1 * MAX_CONTACTS + g) * sf];
    r3.flags = (int)stageFlags[(int64_t)(2 * MAX_CONTACTS + g) * sf];
    CoopCascadeOut out;
    coopCascadeSelect(w, S, Rg, Rg.on ? X0 : 0.0, cm->fallbackCfm, r1, r2, r3, out);
    if (Rg.on) { X = out.X; cfmRow = out.cfm; cls = out.K.cls; E = out.K.E; }
    st = (st & ~0x100u) | (out.st & ~0x100u) | (st & out.st & 0x100u);
    pinvValid = nGroups == 1 && out.pinvValid;
  }
  R.on = rowOn;
  CoopClasses K;
  K.cls = cls; K.E = E;
  K.clampMask = (RowMask)w.ballot(cls == RC_CLAMPING); K.ubMask = (RowMask)w.ballot(cls == RC_UPPER_BOUND);
  K.nc = rmPop(K.clampMask); K.nu = rmPop(K.ubMask);
  // The record always carries Q^+ of the final classification when there is a clamping row (of the whole world: block diagonal
  // over its groups, each block with its group's CFM), so that the backward pass never has to factorise.  Stages that end without
  // one (PGS results accepted as they are) and worlds with several groups pay for it here.
  if (!pinvValid && K.nc > 0) {
    double a[MAXR];
    coopBuildQ(w, S, R, K, cfmRow, a);
    coopPinvOfQ(w, a, S, K);
    pinvValid = true;
  }
  const uint32_t nanBit = coopContactOutputs(w, S, R, n, m, X, K, cfmRow, pinvValid, saved, lay, dn, cacheOut, nv, B, b);
  if (ln == 0 && status) status[b] |= st | nanBit;
}

template <bool MULTI>
__global__ __launch_bounds__(64) NBL_WAVES(NBL_W_CFINAL) void k_contact_cascade_final(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                              double* __restrict__ saved, SavedLayout lay,
                                                              double* __restrict__ cacheOut, double* __restrict__ next,
                                                              uint32_t* __restrict__ status, double* __restrict__ lws,
                                                              const int32_t* __restrict__ failList,
                                                              const uint32_t* __restrict__ failCount) {
  __shared__ CoopLds S;
  if (blockIdx.x >= *failCount) return;
#ifdef NBL_CASCADE_TIMING
  const long long t0 = clock64();
#endif
  const DevWave w;
  const int64_t b = failList[blockIdx.x];
  coopCascadeFinish<MULTI>(w, S, mdl, cm, B, b, saved, lay, cacheOut, next, status, lws, lws + (int64_t)LW_STAGE_X * B + b, B,
                           lws + (int64_t)LW_STAGE_FLAGS * B + b, B);
#ifdef NBL_CASCADE_TIMING
  if (w.lane() == 0) lws[(int64_t)(LW_STAGE_CYCLES + 3) * B + b] = (double)(clock64() - t0);
#endif
}

// Stages 1-3 AND the final part in ONE launch (opt-in, NBL_FUSED_CASCADE=1: measured slower with four slices in flight, nimble_amd.hip): the two stage wavefronts of a world leave their candidates in the
// workgroup's LDS, and whichever of the two arrives second goes on with select + standardise + outputs (coopCascadeFinish) on the LDS
// the stages no longer need.  A world's final part therefore starts when ITS stages are done instead of when the slowest world of the
// launch is - the launch boundary between k_contact_cascade_stages and k_contact_cascade_final made every world wait for the longest
// Dantzig run of the slice (tail: 2-3 x the mean) - and one launch and one pass over the scratch rows go away.
struct FusedCascadeLds {
  union {
    struct { CascadeLds C1; PgsLds C2; } stages;
    CoopLds S;
  } u;
  double stageX[3 * MAX_ROWS];
  double stageFlags[3 * MAX_CONTACTS];
  int arrive;
  int pad;
};

template <bool MULTI>
__global__ __launch_bounds__(128) NBL_WAVES(2) void k_contact_cascade_fused(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                              double* __restrict__ saved, SavedLayout lay,
                                                              double* __restrict__ cacheOut, double* __restrict__ next,
                                                              uint32_t* __restrict__ status, double* __restrict__ lws,
                                                              const int32_t* __restrict__ failList,
                                                              const uint32_t* __restrict__ failCount) {
  __shared__ FusedCascadeLds F;
  if (blockIdx.x >= *failCount) return;
  const DevWaveInGroup w;
  const int ln = w.lane();
  const int wave = (int)(threadIdx.x >> 6);
  if (threadIdx.x == 0) F.arrive = 0;
  if (threadIdx.x < 3 * MAX_ROWS) F.stageX[threadIdx.x] = 0.0;
  __syncthreads();   // the only workgroup-wide barrier: from here on the two wavefronts never wait for each other
  const int64_t b = failList[blockIdx.x];
  {
    const int m = 3 * (int)svAt(saved, lay.nc, B, b);
    double* dn = denseOf(saved, lay, B, b);
    CoopRow R;
    coopLoadRow<MULTI>(R, ln, m, saved, dn, lay, cm, B, b);
    const double X0 = ln < m ? lws[(int64_t)(LW_X0 + ln) * B + b] : 0.0;
    const uint32_t failMask = MULTI ? (uint32_t)lws[(int64_t)LW_FAILMASK * B + b] : 1u;
    int gid = 0;
    if (MULTI) coopGroups(w, cm, saved, lay, B, b, m, gid);
    const bool rowOn = R.on;
#pragma unroll 1
    for (int g = 0; g < (MULTI ? MAX_CONTACTS : 1); g++) {
      if (!((failMask >> g) & 1u)) continue;
      R.on = rowOn && gid == g;
      CoopRow& Rg = R;
      auto publish = [&](int stage, const CoopStageResult& r) {
        if (Rg.on) F.stageX[stage * MAX_ROWS + ln] = r.X;
        if (ln == 0) F.stageFlags[stage * MAX_CONTACTS + g] = (double)r.flags;
      };
      CoopStageResult r;
      if (wave == 0) {
        coopCascadeStage1(w, F.u.stages.C1, Rg, X0, r);
        publish(0, r);
      } else {
        coopCascadeStage2(w, F.u.stages.C2, Rg, X0, cm->fallbackCfm, r);
        publish(1, r);
        const bool stage2Valid = (r.flags & (CS_SOLVED | CS_VALID)) == (CS_SOLVED | CS_VALID);
        r.X = 0.0; r.flags = 0;
        if (!stage2Valid) coopCascadeStage3(w, F.u.stages.C2, Rg, X0, cm->fallbackCfm, r);
        publish(2, r);
      }
      w.sync();
    }
  }
  // arrival: the candidates above must be in LDS before this wavefront's increment is, and the wavefront that reads 1 (the second one)
  // must see both sets: a release / acquire increment with workgroup fences on both sides (the compiler may otherwise move the plain
  // stores past a relaxed atomic)
  int prev = 0;
  __threadfence_block();
  if (ln == 0) prev = __hip_atomic_fetch_add(&F.arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  __threadfence_block();
  prev = __builtin_amdgcn_readfirstlane(prev);
  if (prev == 0) return;
  w.sync();
  coopCascadeFinish<MULTI>(w, F.u.S, mdl, cm, B, b, saved, lay, cacheOut, next, status, lws, F.stageX, 1, F.stageFlags, 1);
}

// Self-test of the device Dantzig driver (nbl_selftest_lcp_dantzig): one wavefront per problem of a batch of n-row boxed LCPs
// with explicit bounds, exactly the code k_contact_cascade_stages runs in its stage 1.  Problems are dense [count][n * n] / [count][n].
__global__ __launch_bounds__(64) void k_selftest_dantzig(int count, int n, const double* __restrict__ A, const double* __restrict__ b,
                                                        const double* __restrict__ lo, const double* __restrict__ hi,
                                                        const int32_t* __restrict__ findex, double* __restrict__ x, int32_t* __restrict__ rc) {
  __shared__ CascadeLds C;
  const DevWave w;
  const int ln = w.lane();
  const int64_t pb = blockIdx.x;
  if (pb >= count) return;
  for (int i = ln; i < MAXR * CLD; i += 64) { C.A[i] = 0.0; C.L[i] = 0.0; }
  w.sync();
  if (ln < n) for (int j = 0; j < n; j++) C.A[ln * CLD + j] = A[(pb * n + ln) * n + j];
  w.sync();
  CoopLcpRow row;
  const bool on = ln < n;
  row.x = 0.0; row.b = on ? b[pb * n + ln] : 0.0; row.lo = on ? lo[pb * n + ln] : 0.0; row.hi = on ? hi[pb * n + ln] : 0.0;
  row.findex = on ? findex[pb * n + ln] : -1;
  const int r = coopDantzig(w, C, n, row);
  if (on) x[pb * n + ln] = row.x;
  if (ln == 0) rc[pb] = r;
}

// Self-test of the device pseudo-inverses (nbl_selftest_pinv): one wavefront per 24 x 24 matrix (row-major, masked rows / columns zero),
// route 0 = coopPinv (Householder QR + complete orthogonal decomposition), 1 = coopPinvSym (two Cholesky factorisations, symmetric
// positive semi-definite input), exactly the code the contact kernels run.
__global__ __launch_bounds__(64) void k_selftest_pinv(int count, const double* __restrict__ Q, const int32_t* __restrict__ cTrue, int route,
                                                     double* __restrict__ P, int32_t* __restrict__ rank) {
  __shared__ CoopLds S;
  const DevWave w;
  const int ln = w.lane();
  const int64_t pb = blockIdx.x;
  if (pb >= count) return;
  double a[MAXR];
  const double* q = Q + pb * MAXR * MAXR;
#pragma unroll
  for (int i = 0; i < MAXR; i++) a[i] = ln < MAXR ? q[i * MAXR + ln] : 0.0;
  const int r = route == 1 ? coopPinvSym(w, a, S, cTrue[pb]) : coopPinv(w, a, S, cTrue[pb]);
  if (ln < MAXR) {
#pragma unroll
    for (int i = 0; i < MAXR; i++) P[pb * MAXR * MAXR + i * MAXR + ln] = S.P[i * CLD + ln];
  }
  if (ln == 0) rank[pb] = r;
}

// Dense part of the contact adjoint, one world per wavefront (the header of contact_backward.hip derives the
// quantities), with lane = LCP row for the c-vectors and lane = DOF for the
// n-vectors.  Row-indexed vectors are zero outside the clamping set, which replaces the index compaction:
//   (Q x)_r   = (A xE)_r + cfm_r x_r      xE = x on clamping rows, E_u x_normal(u) on upper-bound rows  ("spread")
//   (Q^T y)_s = t_s + sum_{u in ub(s)} E_u t_u + cfm_s y_s,  t = A y                                     ("fold")
// Q^+ is read back from the saved record when the forward pass left it there (pflag), else recomputed.
__global__ __launch_bounds__(64) NBL_WAVES(NBL_W_BWDA) void k_bwd_contact_a_coop(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                           double* __restrict__ saved, SavedLayout lay,
                                                           const double* __restrict__ gnext, double* __restrict__ lws) {
  __shared__ CoopLds S;
  const DevWave w;
  const int ln = w.lane();
  const int64_t b = mdl.b0 + coopWorld(blockIdx.x, gridDim.x);
  if (b >= mdl.b1) return;
  const int n = mdl.n;
  const double* gvn = gnext + (int64_t)n * B;
  double* dn = denseOf(saved, lay, B, b);
  // ---- every global load that does not depend on another, in one batch (one memory round trip instead of ~8):
  //      rows beyond m / DOFs beyond n read valid memory with unused (possibly stale) values, masked below ----
  const int row = ln < MAXR ? ln : 0, dof = ln < n ? ln : 0;
  const double ncD = svAt(saved, lay.nc, B, b);
  const double pflagD = svAt(saved, lay.pflag, B, b);
  const double cfmRaw = svAt(saved, lay.cfm + row, B, b);   // this row's constraint-force-mixing constant (its constrained group's)
  const double cvRaw = svAt(saved, lay.cls + row, B, b);
  const double xRaw0 = svAt(saved, lay.x + row, B, b);
  const double bvRaw = svAt(saved, lay.b + row, B, b);
  const int r0c = lay.contacts + (row / 3) * CR_SIZE;
  const int bxA = (int)svAt(saved, r0c + CR_BOXA, B, b), bxB = (int)svAt(saved, r0c + CR_BOXB, B, b);
  const double muTab = cm->boxes[ln < MAX_BOXES ? ln : 0].mu;       // collider -> mu, looked up with ds_bpermute
  const double gMine = gvn[(int64_t)dof * B + b];   // cotangent of v' for this lane's DOF
  const double restRaw = svAt(saved, lay.rest + row / 3, B, b);   // restitution coefficient of this row's contact if it bounced
  double Acol[MAXR];
#pragma unroll
  for (int i = 0; i < MAXR; i++) Acol[i] = dn[lay.A + i * MAX_ROWS + row];
  {
    // Q^+ of the record goes straight to LDS (its registers are free again as soon as the values have arrived)
    double Pcol[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; i++) Pcol[i] = dn[lay.pinv + i * MAX_ROWS + row];
    if (ln < MAXR) {
#pragma unroll
      for (int i = 0; i < MAXR; i++) S.P[i * CLD + ln] = Pcol[i];
    }
  }
  const int m = 3 * (int)ncD;
  const bool rowOn = ln < m;
  // contact adjoint active <=> some row is clamping (the same test k_bwd_recompute_coop makes for LB_FLAG; computed here so
  // that this kernel does not wait for it: the two run concurrently on two streams)
  if (w.ballot(rowOn && cvRaw == 1.0) == 0ull) return;
#pragma unroll
  for (int i = 0; i < MAXR; i++) Acol[i] = (rowOn && i < m) ? Acol[i] : 0.0;   // columns / rows >= m were never written
  CoopRow R;
  R.m = m; R.fric = (ln % 3) != 0; R.fp = ln < MAXR ? ln - (ln % 3) : 0; R.on = rowOn;
  {
    const double muA = w.shfl(muTab, bxA), muB = w.shfl(muTab, bxB);
    R.mu = rowOn ? (muA < muB ? muA : muB) : 0.0;
  }
  R.Bv = rowOn ? bvRaw : 0.0;
  R.Acol = dn + lay.A + (rowOn ? ln : 0);
  R.colNorm = 0.0;   // only the forward solver's validity test uses it
  // classes as stored by the forward pass
  CoopClasses K;
  const double cv = rowOn ? cvRaw : 0.0;
  const double cfm = rowOn ? cfmRaw : 0.0;
  K.cls = cv == 1.0 ? RC_CLAMPING : ((cv == 2.0 || cv == -2.0) ? RC_UPPER_BOUND : RC_NOT_CLAMPING);
  K.E = K.cls == RC_UPPER_BOUND ? (cv > 0 ? R.mu : -R.mu) : 0.0;
  K.clampMask = (RowMask)w.ballot(K.cls == RC_CLAMPING);
  K.ubMask = (RowMask)w.ballot(K.cls == RC_UPPER_BOUND);
  K.nc = rmPop(K.clampMask);
  K.nu = rmPop(K.ubMask);
  const bool clamp = K.cls == RC_CLAMPING, isUb = K.cls == RC_UPPER_BOUND;
  const double xRaw = rowOn ? xRaw0 : 0.0;   // the impulses that were applied
  auto fold = [&](double t) -> double {   // normal rows collect E_u t_u of their contact's upper-bound rows
    if (K.nu == 0) return 0.0;
    const double et = isUb ? K.E * t : 0.0;
    const double f1 = w.shfl(et, ln + 1), f2 = w.shfl(et, ln + 2);
    return (!R.fric && ln + 2 < MAXR) ? f1 + f2 : 0.0;
  };
  auto spread = [&](double x) -> double {   // upper-bound rows ride on their normal row: E_u x_normal
    if (K.nu == 0) return clamp ? x : 0.0;
    const double xn = w.shfl(x, R.fp);
    return clamp ? x : (isUb ? K.E * xn : 0.0);
  };
  // g -> LDS (n <= MAX_DOF_CONTACT <= 64 entries, in the R buffer which is free until the factorisation)
  double* bc = S.R;
  if (ln < n) bc[ln] = gMine;
  w.sync();
  // fbar = Abar^T lambda1 with lambda1 = M^-1 g:  A_c^T M^-1 g = (M^-1 A_c)^T g, i.e. the saved impulse tests applied to g -
  // no M^-1 solve needed here (k_bwd_recompute_coop computes lambda1 for the tree part meanwhile)
  double t = 0.0;
  {
    // the column in chunks of eight loads in flight (all 40 at once cost 80 registers on top of the A / Q^+ columns: the kernel sat at
    // 256 VGPRs + 192 AGPRs, ONE wavefront per SIMD, and waited for an empty SIMD whenever another slice's kernels held the chip)
    double ta = 0.0, tb = 0.0;
#pragma unroll 1
    for (int d0 = 0; d0 < MAX_DOF_CONTACT; d0 += 8) {
      if (d0 >= n) break;
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = d0 + q < n ? dn[lay.massed + (d0 + q) * MAX_ROWS + row] : 0.0;
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        ta = fma(v[q], d0 + q < n ? bc[d0 + q] : 0.0, ta);
        tb = fma(v[q + 1], d0 + q + 1 < n ? bc[d0 + q + 1] : 0.0, tb);
      }
    }
    t = rowOn ? ta + tb : 0.0;
  }
  const double tf = fold(t);   // cross-lane: every lane takes part
  const double fbar = clamp ? t + tf : 0.0;
  w.sync();
  // Q^+
  if (pflagD != 0.0) w.sync();   // S.P was filled from the record at the top
  else {
    // cannot happen: the forward kernels always leave Q^+ of the final classification in the record; make it loud instead of silently wrong
    if (ln < MAXR) {
#pragma unroll
      for (int i = 0; i < MAXR; i++) S.P[i * CLD + ln] = __builtin_nan("");
    }
    w.sync();
  }
  // A (masked to the rows in use) moves to the G buffer, free once Q^+ exists: 48 registers less for the rest of the kernel
  if (ln < MAXR) {
#pragma unroll
    for (int i = 0; i < MAXR; i++) S.G[i * CLD + ln] = Acol[i];
  }
  w.sync();
  // A x for this lane's row (A is symmetric: row = column), x one entry per lane
  auto ax = [&](double xLane, int slot) -> double {
    if (ln < MAXR) S.vec[slot][ln] = rowOn ? xLane : 0.0;
    w.sync();
    double v0 = 0.0, v1 = 0.0;
#pragma unroll 1
    for (int jb = 0; jb < MAXR; jb += 8) {
#pragma unroll
      for (int jq = 0; jq < 8; jq += 2) { const int jx = jb + jq; v0 = fma(S.G[jx * CLD + row], S.vec[slot][jx], v0); v1 = fma(S.G[(jx + 1) * CLD + row], S.vec[slot][jx + 1], v1); }
    }
    return v0 + v1;
  };
  // ---- was Q inverted precisely?  The reference switches between two formulas for the derivative of Q^+ b
  // (BackpropSnapshot.cpp:2964-2984): when ||I - Q Q^+||_F^2 < 1e-18 it uses  -Q^+ dQ Q^+ b  alone, otherwise the full derivative
  // of the pseudo-inverse, whose two extra terms carry (I - Q Q^+) b and (I - Q^+ Q) next to Q^+T Q^+.  For an exactly rank
  // deficient Q (four coplanar corners) the extra terms are the correct ones; for a full-rank but ill-conditioned Q (the CFM
  // fallback: cond ~ 1e6) they are round-off amplified by |Q^+|^2 - the reference drops them there and so must we (measured on
  // cfg4 with the 0.1 kg cubes of box_stacking.skel: 2.7e-4 relative gradient error with them, 1e-9 without).
  // Q Q^+ on the clamping block: (Q X)_r = (A spread(X))_r + cfm X_r for the columns X of Q^+.
  bool precise;
  {
    double* XE = S.R;                        // spread(Q^+) row by row (the buffer is free between g and the coefficient vectors)
    if (ln < MAXR) {
      S.vec[3][ln] = cfm;                    // cfm_r by row index for the tiles below
      const int src = clamp ? row : (isUb ? R.fp : row);
      const double sc = clamp ? 1.0 : (isUb ? K.E : 0.0);
#pragma unroll
      for (int j = 0; j < MAXR; j++) XE[row * CLD + j] = sc * S.P[src * CLD + j];
    }
    w.sync();
    // Q Q^+ = A spread(Q^+) + cfm Q^+ is the one true GEMM of the contact adjoint (24 x 24 x 24): on the matrix cores.
    // v_mfma_f64_16x16x4_f64: A operand lane l = A[l & 15][l >> 4], B operand lane l = B[l >> 4][l & 15], D register g of lane l =
    // D[(l >> 4) + 4 g][l & 15].  Output padded to 32 x 32 (2 x 2 tiles), K = 24 in 6 steps: 24 MFMAs instead of 576 FMAs per lane
    // (48 rows: 3 x 3 tiles, 12 steps).
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int li = ln & 15, lk = ln >> 4;
    double acc = 0.0;
#pragma unroll
    for (int tr = 0; tr < MFMA_TILES; tr++) {
#pragma unroll
      for (int tj = 0; tj < MFMA_TILES; tj++) {
        v4d d = {0.0, 0.0, 0.0, 0.0};
        const int r = 16 * tr + li, j = 16 * tj + li;
#pragma unroll
        for (int ks = 0; ks < MAXR / 4; ks++) {
          const int k = 4 * ks + lk;
          const double av = S.G[k * CLD + (r < MAXR ? r : 0)], bv = XE[k * CLD + (j < MAXR ? j : 0)];   // A is symmetric: A[r][k] = G[k][r]
          d = __builtin_amdgcn_mfma_f64_16x16x4f64(r < MAXR ? av : 0.0, j < MAXR ? bv : 0.0, d, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int rr = 16 * tr + lk + 4 * g, jj = 16 * tj + li;
          const bool inBlock = rr < MAXR && jj < MAXR && ((K.clampMask >> rr) & 1u) && ((K.clampMask >> jj) & 1u);
          const double y = d[g] + S.vec[3][rr < MAXR ? rr : 0] * S.P[(rr < MAXR ? rr : 0) * CLD + (jj < MAXR ? jj : 0)];
          const double dlt = inBlock ? ((rr == jj) ? 1.0 : 0.0) - y : 0.0;
          acc = fma(dlt, dlt, acc);
        }
      }
    }
    w.sync();
    S.vec[0][ln] = acc;                      // 64 partial sums in the 96 doubles of S.vec
    w.sync();
    double imp2 = 0.0;
#pragma unroll 1
    for (int kb = 0; kb < 64; kb += 8) {       // (eight loads in flight, not 64)
#pragma unroll
      for (int k = 0; k < 8; k++) imp2 += S.vec[0][kb + k];
    }
    precise = imp2 < 1e-18;
    // A clamping joint-limit row is a zero row and column of the reference's Q = A_c^T M^-1 A_c (no constraint-force column), with its
    // group's CFM constant on the diagonal: without one Q is singular, ||I - Q Q^+||^2 >= 1 and the reference takes the full derivative
    if (w.ballot(rowOn && cvRaw == 3.0 && cfmRaw == 0.0) != 0ull) precise = false;
    w.sync();
  }
  // (scheduling fences between the six 24 x 24 products: each reads 48 LDS values; left alone the scheduler issues the loads of all of them
  // up front - 256 VGPRs + 192 AGPRs, one wavefront per SIMD)
  const double bcl = clamp ? R.Bv : 0.0;
  const double mu = coopPinvApply<DevWave, true>(w, S, fbar, 0);     // (Q^+)^T fbar
  coopSchedFence();
  const double fls = coopPinvApply<DevWave, false>(w, S, bcl, 1);    // Q^+ b, the reference's least-squares f_c (not the applied x: they differ when the cascade's answer is not standardised)
  coopSchedFence();
  double al[3], be[3];
  al[0] = -mu; be[0] = fls;
  {
    const double axv = ax(spread(fls), 2);
    al[1] = clamp ? bcl - (axv + cfm * fls) : 0.0;
  }
  coopSchedFence();
  be[1] = coopPinvApply<DevWave, false>(w, S, mu, 3);
  coopSchedFence();
  al[2] = coopPinvApply<DevWave, true>(w, S, fls, 0);
  coopSchedFence();
  {
    const double t2 = ax(mu, 1);
    const double t2f = fold(t2);
    be[2] = clamp ? fbar - (t2 + t2f + cfm * mu) : 0.0;
  }
  coopSchedFence();
  if (precise) { al[1] = 0.0; be[1] = 0.0; al[2] = 0.0; be[2] = 0.0; }   // -Q^+ dQ Q^+ b alone
  // bounce diagonals (CGGM.cpp:770, BackpropSnapshot.cpp:3099-3146): b = -beta A_c^T v_pre with beta = 1 + e on the normal rows that
  // bounced, so the adjoint of b reaches v_pre (and the contact geometry through A_c^T v_pre) scaled by beta; the pairs of dQ are not
  const double muB = (rowOn && (ln % 3) == 0) ? mu * (1.0 + restRaw) : mu;
  double beE[3];
  for (int k = 0; k < 3; k++) beE[k] = spread(be[k]);
  const double fcE = spread(xRaw);
  // coefficient vectors for the DOF lanes: [al0 al1 al2 beE0 beE1 beE2 mu] x 24 rows, in the (now free) R buffer
  w.sync();
  if (ln < MAXR) {
    for (int k = 0; k < 3; k++) { bc[k * MAXR + ln] = clamp ? al[k] : 0.0; bc[(3 + k) * MAXR + ln] = beE[k]; }
    bc[6 * MAXR + ln] = clamp ? muB : 0.0;
  }
  w.sync();
  if (ln < n) {
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int r0 = 0; r0 < MAXR; r0 += 6) {          // six rows (12 loads) in flight
      double ms[6], aa[6];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int r = r0 + q;
        ms[q] = r < m ? dn[lay.massed + ln * MAX_ROWS + r] : 0.0; aa[q] = r < m ? dn[lay.aall + ln * MAX_ROWS + r] : 0.0;   // columns >= m were never written
      }
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int r = r0 + q;
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] = fma(bc[k * MAXR + r], ms[q], acc[k]);
        acc[6] = fma(bc[6 * MAXR + r], aa[q], acc[6]);
      }
    }
    for (int k = 0; k < 3; k++) {
      lws[(int64_t)(LB_S + k * MAX_DOF_CONTACT + ln) * B + b] = acc[k];
      lws[(int64_t)(LB_P + k * MAX_DOF_CONTACT + ln) * B + b] = acc[3 + k];
    }
    lws[(int64_t)(LB_GVP + ln) * B + b] = gvn[(int64_t)ln * B + b] - acc[6];
  }
  // coefficients of z_row on the bases [lambda1, v_pre, p1, p2, p3, s1, s2, s3]
  if (ln < MAX_ROWS) {
    double cf[8];
    cf[0] = fcE; cf[1] = clamp ? -muB : 0.0;
    for (int k = 0; k < 3; k++) { cf[2 + k] = clamp ? al[k] : 0.0; cf[5 + k] = beE[k]; }
    for (int k = 0; k < 8; k++) lws[(int64_t)(LB_COEF + ln * 8 + k) * B + b] = cf[k];
  }
}

// The reference's bounce approximation of the position Jacobians (BackpropSnapshot::getBounceApproximationJacobian :1131-1226):
// posPos and velPos are multiplied from the right by X, the matrix closest to the identity with a_i^T X a_i = -e_i for the
// bouncing constraints (clamping normal rows whose contact bounced; a_i = their column of A_c, e_i their restitution coefficient).
// The reference sets this up as a least-squares problem with an (n^2 x nb) matrix W, W[:, i] = vec(a_i a_i^T), and solves it with a
// complete orthogonal decomposition; its minimum-norm solution has the closed form
//      X = I - sum_i c_i a_i a_i^T,   c = G^+ (e + |a|^2),   G_ij = (a_i . a_j)^2   (G = W^T W, |a_i|^2 = W[:, i] . vec(I)),
// so that X y = y - A_b (c o (A_b^T y)): two small products and one pseudo-inverse of at most 8 x 8 (coopPinv on masked lanes).
// One world per wavefront: computes y_q = posPos^T gq' and y_v = velPos^T gq' of the position integration (lane = DOF; the free
// joints' exp / log VJP like in the reverse sweep), and ADDS  X y_q - y_q  to the extra position cotangent LB_QX, writes
// X y_v - y_v to the extra velocity cotangent LB_VX; k_bwd_final_coop folds both in before clipLossGradientsToBounds.
__global__ __launch_bounds__(64) void k_bwd_bounce(DevModel mdl, const DevBody* __restrict__ bodies, int64_t B,
                                                   const double* __restrict__ savedC, SavedLayout lay,
                                                   const double* __restrict__ gnext, double* __restrict__ lws) {
  double* saved = const_cast<double*>(savedC);   // read only here (svAt / denseOf take the writable type)
  __shared__ CoopLds S;
  __shared__ double yq[MAX_DOF_CONTACT], yv[MAX_DOF_CONTACT];
  const DevWave w;
  const int ln = w.lane();
  const int64_t b = mdl.b0 + coopWorld(blockIdx.x, gridDim.x);
  if (b >= mdl.b1) return;
  const int n = mdl.n;
  const int row = ln < MAXR ? ln : 0;
  const int m = 3 * (int)svAt(saved, lay.nc, B, b);
  const double cv = svAt(saved, lay.cls + row, B, b);
  const double eRow = svAt(saved, lay.rest + row / 3, B, b);
  const bool bouncing = ln < m && (ln % 3) == 0 && cv == 1.0 && eRow > 0.0;
  const RowMask bmask = (RowMask)w.ballot(bouncing);
  if (bmask == 0) {                                    // nothing bounced in this world: X = I
    if (ln < n) lws[(int64_t)(LB_VX + ln) * B + b] = 0.0;
    return;
  }
  const double* dn = denseOf(saved, lay, B, b);
  const double* q = saved;
  const double* v = saved + (int64_t)n * B;
  // ---- y_q = posPos^T gq', y_v = velPos^T gq' (GenericJoint.hpp:1428-1444: identity and dt; FreeJoint.cpp:922-929 by reverse mode) ----
  if (ln < mdl.nb) {
    const DevBody& bd = bodies[ln];
    const int o = bd.dofOff;
    if (bd.jtype == JT_FREEC) {                         // a free joint below the root: the 6 x 6 blocks of the SE(3) integration
      const int d0 = o - bd.ballComp, cmp = bd.ballComp;
      auto at3 = [&](const double* x, int k0) { return mk3(x[(int64_t)(d0 + k0) * B + b], x[(int64_t)(d0 + k0 + 1) * B + b], x[(int64_t)(d0 + k0 + 2) * B + b]); };
      double posT[6], velT[6];
      se3IntegrationVjp(at3(q, 0), at3(v, 0), at3(v, 3), mdl.dt, at3(gnext, 0), at3(gnext, 3), posT, velT);
#pragma unroll
      for (int k = 0; k < 6; k++) if (k == cmp) { yq[o] = posT[k]; yv[o] = velT[k]; }
    } else if (bd.jtype == JT_BALL) {                   // BallJoint.cpp:351-408: the 3 x 3 blocks of the SO(3) integration
      const int d0 = o - bd.ballComp, cmp = bd.ballComp;
      V3 posr, velw;
      so3IntegrationVjp(mk3(q[(int64_t)(d0 + 0) * B + b], q[(int64_t)(d0 + 1) * B + b], q[(int64_t)(d0 + 2) * B + b]),
                        mk3(v[(int64_t)(d0 + 0) * B + b], v[(int64_t)(d0 + 1) * B + b], v[(int64_t)(d0 + 2) * B + b]), mdl.dt,
                        mk3(gnext[(int64_t)(d0 + 0) * B + b], gnext[(int64_t)(d0 + 1) * B + b], gnext[(int64_t)(d0 + 2) * B + b]), posr, velw);
      yq[o] = cmp == 0 ? posr.x : (cmp == 1 ? posr.y : posr.z);
      yv[o] = cmp == 0 ? velw.x : (cmp == 1 ? velw.y : velw.z);
    } else if (bd.jtype != JT_FREE) {
      const double g = gnext[(int64_t)o * B + b];
      yq[o] = g; yv[o] = mdl.dt * g;
    } else {
      const V3 r = mk3(q[(int64_t)(o + 0) * B + b], q[(int64_t)(o + 1) * B + b], q[(int64_t)(o + 2) * B + b]);
      const V3 wv = mk3(v[(int64_t)(o + 0) * B + b], v[(int64_t)(o + 1) * B + b], v[(int64_t)(o + 2) * B + b]);
      const V3 vl = mk3(v[(int64_t)(o + 3) * B + b], v[(int64_t)(o + 4) * B + b], v[(int64_t)(o + 5) * B + b]);
      const V3 grn = mk3(gnext[(int64_t)(o + 0) * B + b], gnext[(int64_t)(o + 1) * B + b], gnext[(int64_t)(o + 2) * B + b]);
      const V3 gpn = mk3(gnext[(int64_t)(o + 3) * B + b], gnext[(int64_t)(o + 4) * B + b], gnext[(int64_t)(o + 5) * B + b]);
      const M3 R = expMapRot(r), E = expMapRot(mdl.dt * wv);
      const M3 Rn = mul(R, E);
      const M3 Rnb = logMap_vjp(Rn, grn);
      M3 Rb = mulABt(Rnb, E);
      const M3 Eb = mulAtB(R, Rnb);
      const V3 vdt = mdl.dt * vl;
      Rb.m[0] += gpn.x * vdt.x; Rb.m[1] += gpn.x * vdt.y; Rb.m[2] += gpn.x * vdt.z;
      Rb.m[3] += gpn.y * vdt.x; Rb.m[4] += gpn.y * vdt.y; Rb.m[5] += gpn.y * vdt.z;
      Rb.m[6] += gpn.z * vdt.x; Rb.m[7] += gpn.z * vdt.y; Rb.m[8] += gpn.z * vdt.z;
      const V3 posr = expMapRot_vjp(r, Rb);
      const V3 velw = mdl.dt * expMapRot_vjp(mdl.dt * wv, Eb);
      const V3 vell = mdl.dt * tmul(R, gpn);
      yq[o] = posr.x; yq[o + 1] = posr.y; yq[o + 2] = posr.z; yq[o + 3] = gpn.x; yq[o + 4] = gpn.y; yq[o + 5] = gpn.z;
      yv[o] = velw.x; yv[o + 1] = velw.y; yv[o + 2] = velw.z; yv[o + 3] = vell.x; yv[o + 4] = vell.y; yv[o + 5] = vell.z;
    }
  }
  w.sync();
  // ---- lane = bouncing row i: t = a_i . y, |a_i|^2, and its column of G ----
  double tq = 0.0, tv = 0.0, nrm2 = 0.0;
  double gcol[MAXR];
#pragma unroll
  for (int k = 0; k < MAXR; k++) gcol[k] = 0.0;
  if (bouncing) {
    for (int d = 0; d < n; d++) {
      const double a = dn[lay.aall + d * MAX_ROWS + row];
      tq = fma(a, yq[d], tq); tv = fma(a, yv[d], tv); nrm2 = fma(a, a, nrm2);
    }
  }
#pragma unroll 1
  for (int k = 0; k < MAXR; k += 3) {                 // candidates: the normal rows
    if (!((bmask >> k) & 1u)) continue;
    double dotik = 0.0;
    if (bouncing) for (int d = 0; d < n; d++) dotik = fma(dn[lay.aall + d * MAX_ROWS + row], dn[lay.aall + d * MAX_ROWS + k], dotik);
#pragma unroll
    for (int kk = 0; kk < MAXR; kk++) if (kk == k) gcol[kk] = bouncing ? dotik * dotik : 0.0;     // G[k][i] (symmetric)
  }
  coopPinv(w, gcol, S, rmPop(bmask));
  const double cRow = coopPinvApply<DevWave, false>(w, S, bouncing ? eRow + nrm2 : 0.0, 0);
  // ---- (X - I) y = -A_b (c o t)   (lane = DOF) ----
  w.sync();
  if (ln < MAXR) { S.vec[1][ln] = bouncing ? cRow * tq : 0.0; S.vec[2][ln] = bouncing ? cRow * tv : 0.0; }
  w.sync();
  if (ln < n) {
    double dq = 0.0, dv = 0.0;
#pragma unroll 1
    for (int k = 0; k < MAXR; k += 3) {
      if (!((bmask >> k) & 1u)) continue;
      const double a = dn[lay.aall + ln * MAX_ROWS + k];
      dq = fma(-a, S.vec[1][k], dq); dv = fma(-a, S.vec[2][k], dv);
    }
    lws[(int64_t)(LB_QX + ln) * B + b] += dq;
    lws[(int64_t)(LB_VX + ln) * B + b] = dv;
  }
}

// Contact rows, one world per wavefront, lane = LCP row: per-row wrench, b = -J^T V(v_pre), the constraint-force column A_c[:, row]
// (DCC::getConstraintForces), one unit-impulse test per lane (BodyNode::updateBiasImpulse / updateVelocityChangeFD,
// BodyNode.cpp:2117-2215; GenericJoint.hpp:2482-2498, 2607-2613, 2713-2725) giving the column M^-1 J^T e_row ("massed")
// and the row of the Delassus matrix A (BoxedLcpConstraintSolver.cpp:250-320: entries of later contacts computed,
// earlier ones mirrored).
// Everything is carried in the WORLD frame: a prologue (lane = body) moves S, AI*S and the twist at v_pre of every body
// to the world frame once; then the contact wrench F is the same 6-vector at every body, impulses add up the chain and
// velocity changes pass down it without transforms, and an entry of A is F_col . (dV_A - dV_B).  Only the free-joint root
// is solved in its body frame like in abaSweeps.
//   lds doubles: Fw[rows][6] x 2  Sw[nb][6]  AISw[nb][6]  Vw[nb][6]  acc[nb][6][rows]  psi[nb]  free[nFree][54]  contact bodies
// WPW = 2 (24-row build, models of at most 32 device bodies): TWO worlds per wavefront - lanes 0..31 work for the first, lanes 32..63 for
// the second (lane within the half = body in the lane = body phases, LCP row in the others).  With 24 rows per world a wavefront of one world
// runs on 23 of its 64 lanes (profiles/r03 fp64_flops.json: 22.8 active lanes); the serial loops below go over the BODIES of the model,
// which the two worlds share, so the second world rides along in the idle lanes: half the wavefronts for the same chain length.  The
// model's topology (parent, joint type, collider -> body, ancestor masks) is read from the lanes of the first half by both.
template <int WPW>
__global__ __launch_bounds__(64) NBL_WAVES(NBL_W_ROWS) void k_contact_rows_coop(DevModel mdl, const DevBody* __restrict__ bodies,
                                                          const DevContactModel* __restrict__ cm, int64_t B,
                                                          double* __restrict__ saved, SavedLayout lay,
                                                          const double* __restrict__ ws) {
  static_assert(WPW == 1 || (WPW == 2 && MAX_ROWS <= 32), "two worlds per wavefront: 32 lanes each");
  extern __shared__ __attribute__((aligned(16))) double ldsRows[];
  const int nb = mdl.nb;
  constexpr int RW = WPW * MAX_ROWS;                // LCP rows of the wavefront's worlds side by side
  const DevWave w;
  const int ln = w.lane();
  const int h = WPW == 2 ? (ln >> 5) : 0;           // which of the wavefront's worlds this lane works for
  const int l = WPW == 2 ? (ln & 31) : ln;          // lane within that world
  const int hl = WPW == 2 ? (h << 5) : 0;           // first lane of the half
  double* Fs = ldsRows;                             // the row's wrench about the frame origin of body A's tree ...
  double* FsB = Fs + 6 * RW;                        // ... and of body B's tree
  double* Sw = FsB + 6 * RW + 6 * nb * h;           // (per world from here on)
  double* AISw = FsB + 6 * RW + 6 * nb * WPW + 6 * nb * h;
  double* Vw = FsB + 6 * RW + 12 * nb * WPW + 6 * nb * h;
  double* acc = FsB + 6 * RW + 18 * nb * WPW;       // [body][6][RW]
  // what the serial body loops below read per body, staged once (a global load inside those loops is waited for per body):
  // psi of the 1-DOF joints; per free joint its LDL^T (21), articulated inertia (21) and world transform (12); the two
  // bodies of every contact
  double* psiL = acc + 6 * nb * RW + nb * h;
  double* freeAll = acc + 6 * nb * RW + nb * WPW;   // [WPW][nFree][54]
  double* freeL = freeAll + 54 * mdl.nFree * h;
  int* cbody = reinterpret_cast<int*>(freeAll + 54 * mdl.nFree * WPW) + 2 * MAX_CONTACTS * h;   // [WPW][2][MAX_CONTACTS]
  NBL_PHASE(32);
  const int64_t bw0 = mdl.b0 + (int64_t)WPW * coopWorld(blockIdx.x, gridDim.x);
  if (bw0 >= mdl.b1) return;
  const bool valid = bw0 + h < mdl.b1;              // (an odd number of worlds: the last wavefront's second half idles)
  const int64_t b = valid ? bw0 + h : bw0;
  Ctx c = makeCtx(mdl, bodies, nullptr, const_cast<double*>(ws), B, b, saved, &lay);
  double* dn = denseOf(saved, lay, B, b);
  auto ld6 = [](const double* base) -> V6 { double a[6]; for (int e = 0; e < 6; e++) a[e] = base[e]; return fromArr(a); };
  auto st6 = [](double* base, V6 x) { double a[6]; toArr(x, a); for (int e = 0; e < 6; e++) base[e] = a[e]; };
  // ---- every global load of the prologue in ONE batch: none depends on another (lanes beyond nb / beyond the rows in use
  //      read body 0 / stale contact slots - valid memory, unused values), so their latency is paid once ----
  const double ncD = svAt(saved, lay.nc, B, b);
  const int bl = l < nb ? l : 0;
  const DevBody& bdL = bodies[bl];
  // topology of body `l` in the lane's registers: the serial body loops below fetch it with v_readlane (no memory access)
  const int myParent = l < nb ? bdL.parent : -1, myJtype = l < nb ? bdL.jtype : 0;
  const int myDofOff = l < nb ? bdL.dofOff : 0, myFreeIdx = l < nb ? bdL.freeIdx : -1;
  // The "world frame" of the spatial quantities has its origin at the root of each body's tree instead of the world's (one pure translation
  // per tree: within a tree one wrench still serves every body; a row between two trees has one wrench per side).  Moments about a far
  // origin would blur A's singular structure with the square of the distance and flip the rank decisions of the solver.
  const int myRoot = l < nb ? bdL.root : 0;
  const V3 myOrigin = ldTAt(c, myRoot, WS_TW).p;
  T12 TWl = ldTAt(c, bl, WS_TW);
  TWl.p = TWl.p - myOrigin;
  const V6 vtwL = ldV6(c, bl, WS_VTW), aisL = ldV6(c, bl, WS_AIS), SL = cV6(bdL.S);
  const double psiMine = wsAt(c, bl, WS_PSI);
  const int myBoxBody = cm->boxes[l < MAX_BOXES ? l : 0].body;       // collider -> body and body -> ancestor mask tables,
  const uint64_t myAnc = cm->ancestors[bl];                           // looked up with ds_bpermute below
  const int row = l < MAX_ROWS ? l : 0;
  const int rw = h * MAX_ROWS + row;                                  // this lane's row among the wavefront's rows
  const int ci = row / 3, kk = row % 3;
  const int r0 = lay.contacts + ci * CR_SIZE;
  const V3 p = mk3(svAt(saved, r0 + CR_POINT, B, b), svAt(saved, r0 + CR_POINT + 1, B, b), svAt(saved, r0 + CR_POINT + 2, B, b));
  const V3 nrm = mk3(svAt(saved, r0 + CR_NORMAL, B, b), svAt(saved, r0 + CR_NORMAL + 1, B, b), svAt(saved, r0 + CR_NORMAL + 2, B, b));
  const int bxA = (int)svAt(saved, r0 + CR_BOXA, B, b), bxB = (int)svAt(saved, r0 + CR_BOXB, B, b);
  const bool isLim = (int)svAt(saved, r0 + CR_TYPE, B, b) == CT_LIMIT;   // a joint-limit row (model_dev.hpp): unit impulse on a DOF
  const double limSigma = svAt(saved, r0 + CR_EA_FIXED + 1, B, b);
  const int nC = valid ? (int)ncD : 0;
  const int m = 3 * nC;
  if (WPW == 1 ? m == 0 : w.ballot(m > 0) == 0ull) return;
  {
    uint64_t fm = w.ballot(myFreeIdx >= 0);          // the (few) free-joint bodies
    if (WPW == 2) fm &= 0xffffffffull;               // (the second half holds the same topology)
    while (fm) {
      const int fb = __builtin_ctzll(fm);
      fm &= fm - 1;
      const int fIdx = w.bcastI(myFreeIdx, fb);
#pragma unroll
      for (int hh = 0; hh < WPW; hh++) {             // lanes 0..53 copy the block of world hh
        if (ln < 54) {
          const int64_t bh = (bw0 + hh < mdl.b1) ? bw0 + hh : bw0;
          const Ctx ch = makeCtx(mdl, bodies, nullptr, const_cast<double*>(ws), B, bh, saved, &lay);
          const int slot = ln < 21 ? WS_PSI + ln : (ln < 42 ? WS_AI + (ln - 21) : WS_TW + (ln - 42));
          // a free joint is the root of its tree: in the frame of its own origin its world transform has no translation
          freeAll[54 * (mdl.nFree * hh + fIdx) + ln] = ln >= 51 ? 0.0 : wsAt(ch, fb, slot);
        }
      }
    }
  }
  // ---- lane = body: world-frame joint axis, AI*S and twist at v_pre ----
  if (l < nb) {
    st6(Vw + 6 * l, AdT(TWl, vtwL));
    if (myJtype != JT_FREE) {
      st6(Sw + 6 * l, AdT(TWl, SL));
      st6(AISw + 6 * l, dAdInvT(TWl, aisL));
      psiL[l] = psiMine;
    }
  }
  NBL_PHASE(33);
  const bool on = l < m;
  auto accAt = [&](int body, int e) -> double& { return acc[(body * 6 + e) * RW + rw]; };
  auto ldAcc = [&](int body) -> V6 { double a[6]; for (int e = 0; e < 6; e++) a[e] = accAt(body, e); return fromArr(a); };
  auto stAcc = [&](int body, V6 x) { double a[6]; toArr(x, a); for (int e = 0; e < 6; e++) accAt(body, e) = a[e]; };
  // ---- this row's wrench and the two bodies it acts on ----
  V3 t1, t2;
  tangentBasis(nrm, t1, t2);
  // A frictionless contact (mu = min(mu_A, mu_B) <= DART_FRICTION_COEFF_THRESHOLD = 1e-3) has ONE row in the reference
  // (ContactConstraint.cpp:107-118, 229: mIsFrictionOn false -> dim 1).  Here it keeps its three row slots and the two tangent
  // rows are EMPTY: zero wrench -> zero row / column of A, b = 0, zero A_c column, bounds 0.  Every stage leaves such rows at
  // x = 0 and they add exact zeros to the sums of the other rows, so the result equals the reference's one-row problem.
  // (rows beyond the contacts in use read stale record slots: clamp the collider indices before using them as addresses)
  const double muRow = isLim ? 0.0 : fmin(cm->boxes[(unsigned)bxA < (unsigned)MAX_BOXES ? bxA : 0].mu, cm->boxes[(unsigned)bxB < (unsigned)MAX_BOXES ? bxB : 0].mu);
  const V3 dirOn = kk == 0 ? nrm : (kk == 1 ? t1 : t2);
  const V3 dir = (kk != 0 && !(muRow > 1e-3)) ? mk3(0.0, 0.0, 0.0) : dirOn;
  const int bAc = w.shflI(myBoxBody, bxA), bBc = w.shflI(myBoxBody, bxB);
  const int bA = isLim ? bxA - CR_BODY_CODE - 1 : bAc, bB = isLim ? bxB - CR_BODY_CODE - 1 : bBc;
  // wrench of a unit impulse along dir at p (on A; minus it on B), about the origin of A's tree and about the origin of B's
  // (the origins are per WORLD: read from the body lanes of this lane's own half)
  const int lA = hl + (bA < 0 ? 0 : bA), lB = hl + (bB < 0 ? 0 : bB);
  const V3 oA = mk3(w.shfl(myOrigin.x, lA), w.shfl(myOrigin.y, lA), w.shfl(myOrigin.z, lA));
  const V3 oB = mk3(w.shfl(myOrigin.x, lB), w.shfl(myOrigin.y, lB), w.shfl(myOrigin.z, lB));
  V6 F = mk6(cross(p - oA, dir), dir), FB = mk6(cross(p - oB, dir), dir);
  if (cm->nLimitDofs > 0) {
    // joint-limit row: the generalized unit impulse sigma e_d is the wrench pair (+F on the joint's child body, -F on its parent) with
    // S_d . F = sigma, F = sigma S_d / |S_d|^2 (S_d the joint's world-frame axis; child and parent share their tree's origin): the
    // relative velocity F . (V_child - V_parent) is sigma qdot_d, the response F . (dV_child - dV_parent) is sigma d(qdot_d), and above
    // the parent the two wrenches cancel.  (JointLimitConstraint::applyUnitImpulse / getVelocityChange, JointLimitConstraint.cpp:293-349)
    w.sync();                                           // Sw of the prologue
    if (isLim) {
      const V6 Sd = ld6(Sw + 6 * (on ? bA : 0));
      const double s2 = dot(Sd, Sd);
      F = kk == 0 ? (limSigma / s2) * Sd : zero6();
      FB = F;
    }
  }
  const int ancLoA = w.shflI((int)(uint32_t)myAnc, bA), ancHiA = w.shflI((int)(uint32_t)(myAnc >> 32), bA);
  const int ancLoB = w.shflI((int)(uint32_t)myAnc, bB), ancHiB = w.shflI((int)(uint32_t)(myAnc >> 32), bB);
  const uint64_t mA = bA >= 0 ? ((uint64_t)(uint32_t)ancHiA << 32) | (uint32_t)ancLoA : 0ull;
  const uint64_t mB = bB >= 0 ? ((uint64_t)(uint32_t)ancHiB << 32) | (uint32_t)ancLoB : 0ull;
  if (on) {
    st6(Fs + 6 * rw, F); st6(FsB + 6 * rw, FB);
    if (kk == 0) { cbody[ci] = bA; cbody[MAX_CONTACTS + ci] = bB; }
  }
  w.sync();
  NBL_PHASE(34);
  if (on) {
    // b = -J^T V: relative velocity of the contact point pair along dir (getRelVelocity)
    double rel = 0;
    if (bA >= 0) rel -= dot(F, ld6(Vw + 6 * bA));
    if (bB >= 0) rel += dot(FB, ld6(Vw + 6 * bB));
    if (kk == 0) {
      // "Bouncing" (ContactConstraint.cpp:393-441 / 470-512).  A: penetration correction (off by default, ConstraintSolver.cpp:69):
      // (depth - allowance) * ERP / dt, capped (DART_ERROR_ALLOWANCE 0, DART_ERP 0.01, DART_MAX_ERV 1e-3).  B: restitution, e = e_A e_B:
      // the contact bounces when e > 1e-3 and e * (approach speed) > 0.1 and then the larger of the two velocities is used, the
      // restitution one capped at 100.  The coefficient of the contacts that bounced (ContactConstraint::getCoefficientOfRestitution,
      // 0 otherwise) goes to the record for the backward pass.
      double bouncing = 0.0;
      if (cm->penetrationCorrection && !isLim) {
        double bv = svAt(saved, r0 + CR_DEPTH, B, b) - 0.0;
        if (bv < 0.0) bv = 0.0;
        else { bv *= 0.01 * (1.0 / mdl.dt); if (bv > 1e-3) bv = 1e-3; }
        bouncing = bv;
      }
      const double eR = isLim ? 0.0 : cm->boxes[(unsigned)bxA < (unsigned)MAX_BOXES ? bxA : 0].restitution * cm->boxes[(unsigned)bxB < (unsigned)MAX_BOXES ? bxB : 0].restitution;
      double coeff = 0.0;
      if (eR > 1e-3) {
        const double rv = rel * eR;
        if (rv > 1e-1) {
          coeff = eR;
          if (rv > bouncing) bouncing = rv > 1e+2 ? 1e+2 : rv;
        }
      }
      rel += bouncing;
      svAt(saved, lay.rest + ci, B, b) = coeff;
    }
    svAt(saved, lay.b + row, B, b) = rel;
  }
  // (the serial loops over the model's bodies run for every lane of the wavefront: topology by v_readlane from the first half's lanes;
  //  rows that are off only keep their lanes in step)
  {
    // constraint forces in joint space (DCC::getConstraintForces): A_c[i] = sigma_i s_i . F
    for (int i = 0; i < nb; i++) {
      const int jt = w.bcastI(myJtype, i), dofOff = w.bcastI(myDofOff, i);
      const int fIdx = w.bcastI(myFreeIdx, i);
      if (on) {
        const bool pa = (mA >> i) & 1ull, pb = (mB >> i) & 1ull;
        const double mult = (pa && pb) ? 0.0 : (pa ? 1.0 : (pb ? -1.0 : 0.0));
        const V6 Fi = pb ? FB : F;                       // the wrench about the origin of body i's tree (pa && pb: mult = 0)
        if (jt != JT_FREE) dn[lay.aall + dofOff * MAX_ROWS + row] = mult * dot(ld6(Sw + 6 * i), Fi);
        else {
          double v6[6];
          toArr(dAdT(cT(bodies[i].Tcj), dAdT(cT(freeL + 54 * fIdx + 42), Fi)), v6);
          for (int e = 0; e < 6; e++) dn[lay.aall + (dofOff + e) * MAX_ROWS + row] = mult * v6[e];
        }
        for (int e = 0; e < 6; e++) accAt(i, e) = 0.0;
      }
    }
    NBL_PHASE(35);
    // ---- unit-impulse test of this row.  leaf -> root: bias impulses along the two ancestor chains (world wrenches) ----
    const uint64_t chain = on ? (mA | mB) : 0ull;
    for (int i = nb - 1; i >= 0; i--) {
      const int jt = w.bcastI(myJtype, i), par = w.bcastI(myParent, i);
      if (!((chain >> i) & 1ull)) continue;
      V6 Bi = ldAcc(i);
      if (i == bA) Bi = Bi - F;
      if (i == bB) Bi = Bi + FB;
      stAcc(i, Bi);
      if (jt != JT_FREE && par >= 0) {
        const double uimp = -dot(ld6(Sw + 6 * i), Bi);
        stAcc(par, ldAcc(par) + Bi + (psiL[i] * uimp) * ld6(AISw + 6 * i));
      }
    }
    NBL_PHASE(36);
    // root -> leaf: velocity changes of every body (world twists), joint-space response
    for (int i = 0; i < nb; i++) {
      const int jt = w.bcastI(myJtype, i), par = w.bcastI(myParent, i), dofOff = w.bcastI(myDofOff, i);
      const int fIdx = w.bcastI(myFreeIdx, i);
      if (!on) continue;
      const V6 X = par >= 0 ? ldAcc(par) : zero6();
      const V6 Bi = ((chain >> i) & 1ull) ? ldAcc(i) : zero6();
      if (jt != JT_FREE) {
        const V6 S = ld6(Sw + 6 * i);
        const double dq = psiL[i] * (-dot(S, Bi) - dot(ld6(AISw + 6 * i), X));
        stAcc(i, X + dq * S);
        dn[lay.massed + dofOff * MAX_ROWS + row] = dq;
      } else {
        // the free-joint root in its body frame
        const double* fl = freeL + 54 * fIdx;
        const T12 Tcj = cT(bodies[i].Tcj), TW = cT(fl + 42);
        LDL6 f;
        for (int e = 0; e < 15; e++) f.l[e] = fl[e];
        for (int e = 0; e < 6; e++) f.d[e] = fl[15 + e];
        S6 AIb;
        for (int e = 0; e < 21; e++) AIb.a[e] = fl[21 + e];
        const V6 Xb = AdInvT(TW, X);
        double r[6], u[6], pj[6];
        toArr(dAdT(Tcj, dAdT(TW, Bi)), u);
        toArr(dAdT(Tcj, mul(AIb, Xb)), pj);
        for (int e = 0; e < 6; e++) r[e] = -u[e] - pj[e];
        ldl6Solve(f, r);
        stAcc(i, AdT(TW, Xb + AdT(Tcj, fromArr(r))));
        for (int e = 0; e < 6; e++) dn[lay.massed + (dofOff + e) * MAX_ROWS + row] = r[e];
      }
    }
    NBL_PHASE(37);
    // row of A: relative-velocity response at every row of the contacts c2 >= ci, mirrored into the earlier rows
    if (on) {
      for (int c2 = ci; c2 < nC; c2++) {
        const int b2A = cbody[c2], b2B = cbody[MAX_CONTACTS + c2];
        const V6 dVA = b2A >= 0 ? ldAcc(b2A) : zero6(), dVB = b2B >= 0 ? ldAcc(b2B) : zero6();
        for (int k2 = 0; k2 < 3; k2++) {
          const int col = 3 * c2 + k2;
          const double val = dot(ld6(Fs + 6 * (h * MAX_ROWS + col)), dVA) - dot(ld6(FsB + 6 * (h * MAX_ROWS + col)), dVB);
          dn[lay.A + row * MAX_ROWS + col] = val;
          if (c2 > ci) dn[lay.A + col * MAX_ROWS + row] = val;
        }
      }
    }
    NBL_PHASE(38);
  }
}

// Tree part of the contact adjoint, one world per wavefront (the header of contact_backward.hip derives the terms).  Everything is carried in the WORLD frame, where transmitted wrenches add up
// over subtrees without transforms and ad* is equivariant (dAdT(T, dad(V, F)) = dad(AdInvT(T, V), dAdT(T, F))):
//   1a lane = joint-rate field f (9: lambda1, v_pre, p1..3, s1..3, w): world twists FW[i][f] = FW[parent][f] + Ad(TW_i) S_i rate
//   1b lane = (body, field): local wrench G_i (twist in the body frame), moved to the world frame -> TF[i][f]
//   2  lane = LCP row (24): a walk up the ancestor chains would add, for every body l between a contact body and the root,
//          add_l = term - sgn dad(T_end - tw(parent l), F_w),   tw(body) = sum_e cf_e FW[body][e]
//      which is bilinear: summed over rows,  xi[l] = C[l] + sum_e dad(FW[parent l][e], Phi_e[l])  with
//          C   = sum over the rows whose contact body lies in the subtree of l of (term - sgn dad(T_end, F_w))
//          Phi_e = the same subtree sum of sgn cf_e F_w.
//      Each row adds 54 numbers at its two contact bodies (deterministic reduction over the three rows of a contact).
//   3  lane = component: ONE leaf->root pass forms the subtree sums of D = [C, Phi] and of TF (the transmitted wrenches of
//      the reverse Newton-Euler pass at v = 0 for the four mass-matrix pairs (lambda1, w), (s_k, p_k))
//   4  lane = body: xi_W = C + sum_e dad(FW[par][e], Phi_e) - sum_pairs (dad(FW[par][adj], TF[acc]) + dad(FW[par][acc], TF[adj])),
//      projected on the joint (applyHt) -> the position cotangent LB_QX
// lds doubles: FW[nb][9][6] D[nb][54] { tmp[54][24] | TF[nb][9][6] }
template <bool CAPS>
__global__ __launch_bounds__(64) NBL_WAVES(NBL_W_BWDB) void k_bwd_contact_b_coop(DevModel mdl, const DevBody* __restrict__ bodies,
                                                           const DevContactModel* __restrict__ cm, int64_t B,
                                                           double* __restrict__ saved, SavedLayout lay,
                                                           const double* __restrict__ ws, double* __restrict__ lws) {
  extern __shared__ __attribute__((aligned(16))) double ldsB[];
  const DevWave w;
  const int ln = w.lane();
  NBL_PHASE(48);
  const int64_t b = mdl.b0 + coopWorld(blockIdx.x, gridDim.x);
  if (b >= mdl.b1) return;
  if (lws[(int64_t)LB_FLAG * B + b] == 0.0) return;
  const int nb = mdl.nb;
  double* FW = ldsB;
  double* D = FW + nb * 54;
  double* TF = D + nb * 54;      // written after the row phase: shares its storage with tmp
  double* tmp = TF;
  double* TWs = TF + (nb * 54 > 54 * MAX_ROWS ? nb * 54 : 54 * MAX_ROWS);   // [nb][12] world transforms
  int* cbody = reinterpret_cast<int*>(TWs + nb * 12);                        // [2][MAX_CONTACTS]
  Ctx c = makeCtx(mdl, bodies, nullptr, const_cast<double*>(ws), B, b, saved, &lay);
  LaneMem SV; SV.base = saved; SV.B = B; SV.b = b;
  const double* q = saved;
  auto ld6 = [](const double* base, int stride) -> V6 { double a[6]; for (int e = 0; e < 6; e++) a[e] = base[e * stride]; return fromArr(a); };
  auto st6 = [](double* base, int stride, V6 x) { double a[6]; toArr(x, a); for (int e = 0; e < 6; e++) base[e * stride] = a[e]; };
  const int myParent = ln < nb ? bodies[ln].parent : -1;   // topology in lane registers, fetched with v_readlane in the serial loops
  // ---- staging: world transforms of all bodies and the bodies of every contact (read many times below) ----
  const int m = 3 * (int)svAt(saved, lay.nc, B, b);
  const int nC = m / 3;
  // frame origin at body 0 (see k_contact_rows_coop): world transforms and every POSITION of the contact records are shifted
  const V3 worldOrigin = mk3(wsAt(c, 0, WS_TW + 9), wsAt(c, 0, WS_TW + 10), wsAt(c, 0, WS_TW + 11));
  for (int idx = ln; idx < nb * 12; idx += 64) {
    const int e = idx % 12;
    TWs[idx] = wsAt(c, idx / 12, WS_TW + e) - (e == 9 ? worldOrigin.x : (e == 10 ? worldOrigin.y : (e == 11 ? worldOrigin.z : 0.0)));
  }
  if (ln < nC) {
    const int q0 = lay.contacts + ln * CR_SIZE;
    cbody[ln] = crBodyOf(cm, (int)svAt(saved, q0 + CR_BOXA, B, b));
    cbody[MAX_CONTACTS + ln] = crBodyOf(cm, (int)svAt(saved, q0 + CR_BOXB, B, b));
  }
  for (int idx = ln; idx < nb * 54; idx += 64) D[idx] = 0.0;
  w.sync();
  NBL_PHASE(49);
  // ---- phase 1a: world twists of the nine joint-rate fields.  lane = (body, field): own joint twist in the world frame,
  //      then lane = (field, component): prefix sums down the tree (bodies are listed parents first) ----
  for (int item = ln; item < nb * 9; item += 64) {
    const int i = item / 9, f = item - 9 * i;
    const double* src; const int64_t stride = B;
    if (f == 0) src = lws + (int64_t)LB_LAM1 * B + b;
    else if (f == 1) src = saved + (int64_t)lay.vpre * B + b;
    else if (f <= 4) src = lws + (int64_t)(LB_P + (f - 2) * MAX_DOF_CONTACT) * B + b;
    else if (f <= 7) src = lws + (int64_t)(LB_S + (f - 5) * MAX_DOF_CONTACT) * B + b;
    else src = saved + (int64_t)lay.w * B + b;
    const DevBody& bd = bodies[i];
    V6 tw;
    if (bd.jtype == JT_FREE) {
      const int o = bd.dofOff;
      tw = AdT(cT(bd.Tcj), mk6(mk3(src[o * stride], src[(o + 1) * stride], src[(o + 2) * stride]),
                               mk3(src[(o + 3) * stride], src[(o + 4) * stride], src[(o + 5) * stride])));
    } else tw = src[bd.dofOff * stride] * cV6(bd.S);
    st6(FW + item * 6, 1, AdT(cT(TWs + 12 * i), tw));
  }
  w.sync();
  if (ln < 54) {
    for (int i = 1; i < nb; i++) {
      const int par = w.bcastI(myParent, i);
      if (par >= 0) FW[i * 54 + ln] += FW[par * 54 + ln];
    }
  }
  w.sync();
  NBL_PHASE(50);
  // ---- phase 2: per-row constants, side A then side B ----
  double cf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool any = false;
  ContactRec CR;
  CR.bA = -1; CR.bB = -1; CR.type = 0;
  V6 Fw = zero6(), TA = zero6(), TB = zero6();
  RowTerms RT;
  RT.vertexTerm = RT.faceTerm = RT.edgeTermA = RT.edgeTermB = zero6();
  if (ln < m) {
    const int row = ln, ci = row / 3, k = row % 3;
    for (int e = 0; e < 8; e++) { cf[e] = lws[(int64_t)(LB_COEF + row * 8 + e) * B + b]; any = any || cf[e] != 0.0; }
    if (any) {
      CR = loadContactRec<CAPS>(SV, lay, cm, ci);
      CR.p = CR.p - worldOrigin;
      if (CR.type >= CT_EDGE_EDGE) CR.eAP = CR.eAP - worldOrigin;                                        // edge A's point / the sphere centre (A's)
      if (CR.type == CT_EDGE_EDGE || CR.type == CT_SPHERE_SPHERE || CR.type >= CT_PIPE_SPHERE) CR.eBP = CR.eBP - worldOrigin;   // edge B's point / sphere B's centre / the pipe's fixed point
      const TangentFrame TF_ = tangentFrameOf(CR.nrm);
      const V3 d = k == 0 ? CR.nrm : (k == 1 ? TF_.t1 : TF_.t2);
      Fw = mk6(cross(CR.p, d), d);
      auto twistOf = [&](int body) -> V6 {   // world twist of `body` under the joint rates z_row
        V6 z = zero6();
        if (body < 0) return z;
        for (int e = 0; e < 8; e++) z = z + cf[e] * ld6(FW + (body * 9 + e) * 6, 1);
        return z;
      };
      TA = twistOf(CR.bA); TB = twistOf(CR.bB);
      RT = contactRowTerms<CAPS>(CR, TF_, k, d, TA - TB);
    }
  }
  NBL_PHASE(51);
  const bool aIsVertex = (CR.type == CT_VERTEX_FACE);
  for (int side = 0; side < 2; side++) {
    if (ln < MAX_ROWS) {
      const double sgn = side == 0 ? 1.0 : -1.0;
      const int start = side == 0 ? CR.bA : CR.bB;
      const bool vertexSide = (side == 0) == aIsVertex;
      V6 Cc = zero6();
      double sc = 0.0;
      if (any && start >= 0) {
        Cc = -sgn * dad(side == 0 ? TA : TB, Fw);
        if (CR.type == CT_VERTEX_FACE || CR.type == CT_FACE_VERTEX) Cc = Cc + (vertexSide ? RT.vertexTerm : RT.faceTerm);
        else if (CR.type >= CT_EDGE_EDGE) Cc = Cc + (side == 0 ? RT.edgeTermA : RT.edgeTermB);   // edge-edge and the sphere types
        sc = sgn;
      }
      double c6[6], f6[6];
      toArr(Cc, c6); toArr(Fw, f6);
      for (int e = 0; e < 6; e++) tmp[e * MAX_ROWS + ln] = c6[e];
      for (int e = 0; e < 8; e++) for (int x = 0; x < 6; x++) tmp[(6 + e * 6 + x) * MAX_ROWS + ln] = sc * cf[e] * f6[x];
    }
    w.sync();
    if (ln < 54) {
      for (int ci = 0; ci < nC; ci++) {
        const int st = cbody[side * MAX_CONTACTS + ci];
        if (st >= 0) D[st * 54 + ln] += (tmp[ln * MAX_ROWS + 3 * ci] + tmp[ln * MAX_ROWS + 3 * ci + 1]) + tmp[ln * MAX_ROWS + 3 * ci + 2];
      }
    }
    w.sync();
  }
  if (cm->selfCollision) {
    // third pass (models with self-collision only): what a DOF above both bodies of an edge-edge contact is owed (contactRowTerms), added
    // at the lowest common ancestor of the two bodies = the deepest body whose joint moves both (bodies are numbered parents first)
    if (ln < MAX_ROWS) {
      const bool both = any && CR.type == CT_EDGE_EDGE && CR.bA >= 0 && CR.bB >= 0;
      const V3 ca = both ? RT.commonAngular : mk3(0, 0, 0);
      tmp[0 * MAX_ROWS + ln] = ca.x; tmp[1 * MAX_ROWS + ln] = ca.y; tmp[2 * MAX_ROWS + ln] = ca.z;
    }
    w.sync();
    if (ln < 3) {
      for (int ci = 0; ci < nC; ci++) {
        const int bA = cbody[ci], bB = cbody[MAX_CONTACTS + ci];
        if (bA < 0 || bB < 0) continue;
        const uint64_t common = cm->ancestors[bA] & cm->ancestors[bB];
        if (common == 0ull) continue;
        const int lca = 63 - __builtin_clzll(common);
        D[lca * 54 + ln] += (tmp[ln * MAX_ROWS + 3 * ci] + tmp[ln * MAX_ROWS + 3 * ci + 1]) + tmp[ln * MAX_ROWS + 3 * ci + 2];
      }
    }
    w.sync();
  }
  NBL_PHASE(52);
  // ---- phase 1b (after the rows: TF takes over tmp's storage): local wrenches of the nine fields, world frame ----
  for (int item = ln; item < nb * 9; item += 64) {
    const int i = item / 9;
    const T12 TW = cT(TWs + 12 * i);
    const V6 twB = AdInvT(TW, ld6(FW + item * 6, 1));
    st6(TF + item * 6, 1, dAdInvT(TW, mul(cS6(bodies[i].G), twB)));
  }
  w.sync();
  NBL_PHASE(53);
  // ---- phase 3: subtree sums, leaf -> root (D and the transmitted wrenches) ----
  if (ln < 54) {
    for (int i = nb - 1; i >= 1; i--) {
      const int par = w.bcastI(myParent, i);
      if (par >= 0) { D[par * 54 + ln] += D[i * 54 + ln]; TF[par * 54 + ln] += TF[i * 54 + ln]; }
    }
  }
  w.sync();
  NBL_PHASE(54);
  // ---- phase 4 ----
  if (ln < nb) {
    const DevBody& bd = bodies[ln];
    const int i = (bd.jtype == JT_BALL || bd.jtype == JT_FREEC) ? ln - bd.ballComp : ln;   // a ball joint's (non-root free joint's) positions act through the first body of its triple (sextuple)
    const int par = bodies[i].parent;
    V6 xiW = ld6(D + i * 54, 1);
    if (par >= 0) {
      const double* FWp = FW + par * 54;
      for (int e = 0; e < 8; e++) xiW = xiW + dad(ld6(FWp + e * 6, 1), ld6(D + i * 54 + 6 + e * 6, 1));
      for (int k = 0; k < 4; k++) {
        const int ADJ = k == 0 ? 0 : 4 + k, ACC = k == 0 ? 8 : 1 + k;   // (lambda1, w), (s_k, p_k)
        xiW = xiW - dad(ld6(FWp + ADJ * 6, 1), ld6(TF + i * 54 + ACC * 6, 1)) - dad(ld6(FWp + ACC * 6, 1), ld6(TF + i * 54 + ADJ * 6, 1));
      }
    }
    double qb[6];
    applyHt(bd, q, B, b, dAdT(cT(TWs + 12 * i), xiW), qb);
    for (int k = 0; k < bd.ndof; k++) lws[(int64_t)(LB_QX + bd.dofOff + k) * B + b] = qb[k];
  }
  NBL_PHASE(55);
}

}  // namespace NBL_NS
