// gen_lcp_dev.hpp — the contact LCP of a world with ANY number of rows (up to MAX_ROWS of the general instantiation: 192 = 64 contacts):
// the solver cascade of BoxedLcpConstraintSolver::solveLcp (dart/constraint/BoxedLcpConstraintSolver.cpp:352-789) with the CGGM
// classification / standardisation (dart/neural/ConstrainedGroupGradientMatrices.cpp:218-339, 482-872), one world per WAVEFRONT, every
// matrix in the world's slice of HBM (L2-resident) and every per-row quantity in an array of `GenRows` (LDS).
//
// Why a second statement of the algorithm next to coop_dev.hpp / coop_dantzig_dev.hpp: those are written around lane = LCP row with a
// compile-time row count <= 64 (register-resident columns, unrolled factorisations, LDS tiles) - that is what makes the 24-row metric
// path fast and what caps it.  The reference has no cap (ConstraintSolver.cpp:563-606 keeps every contact of every pair; its own
// data/skel/test/box_stacking.skel is a ten-cube tower: 40 contacts, 120 rows, in one constrained group).  This file trades the speed
// for generality: loops over rows instead of lanes, lanes share a loop by striding through it (`for (i = lane; i < n; i += lanes)`), the
// sequential parts (the Dantzig pivoting driver, whose value is being BIT-IDENTICAL to the reference's dSolveLCP; the row order of
// Gauss-Seidel) run on lane 0.  A model gets it only when it asks for more than 16 contact slots (nimble_amd_dispatch.cpp).
//
// Everything is a template over a wave policy W (lane(), lanes(), sync(), maxAll, minAllI, sumAll, anyAll): GenWaveDev on the GPU
// (gen_contact.hip), a ONE-lane policy on the host (tests/host_shim/gen_shim.cpp) under which this file is plain sequential code that the
// CPU tests run against the oracle and against the reference's own dSolveLCP at every size up to 192 rows.
#pragma once
#include "lcp_dev.hpp"

namespace NBL_NS {

// developer instrumentation of the general solve kernel (-DNBL_GEN_TIMING, tools/gen_timing.py): cycles per phase summed over the worlds
#if defined(NBL_GEN_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_genStat[32];
#endif
#if defined(NBL_GEN_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define GEN_T0() long long genT = clock64()
#define GEN_T(k) do { const long long n_ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_genStat[k], (unsigned long long)(n_ - genT)); genT = n_; } while (0)
#define GEN_CNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_genStat[k], 1ull); } while (0)
#define GEN_L0() long long genL = clock64()
#define GEN_L(k) do { const long long n_ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_genStat[k], (unsigned long long)(n_ - genL)); genL = n_; } while (0)
#else
#define GEN_T0() do { } while (0)
#define GEN_T(k) do { } while (0)
#define GEN_CNT(k) do { } while (0)
#define GEN_L0() do { } while (0)
#define GEN_L(k) do { } while (0)
#endif

constexpr int GR = MAXR;            // rows of the arrays below: the instantiation's cap (192 / 384)
// The leading dimension of every scratch matrix AND of the record's dense blocks is a property of the MODEL, not of the build: its rows
// (3 x max_contacts) rounded up to a multiple of 8 - GenRows::ld, SavedLayout::ldr.  A tower of five cubes (28 slots, 88 rows) then holds
// 62 kB per matrix instead of the cap's 295 kB.
__host__ __device__ inline int genLeadingDim(int maxContacts) { const int r = (3 * maxContacts + 7) & ~7; return r < 8 ? 8 : (r > MAXR ? MAXR : r); }

// per-row data of one world's LCP (LDS on the device).  The arrays are carved out of ONE pool whose size follows the rows the MODEL can hold
// (round 6: they were members of GR = 192 / 384 entries each - 24.6 kB of LDS per world for a model of 72 rows, five worlds per CU;
// genRowsCarve(pool, cap) with cap = the model's rows: 8.3 kB, and the registers become the limit).  Every lane holds its own copy of the
// descriptor (pointers, m, ld, anyLim: wave-uniform values); scal / iscal are the shared scalars lane 0 writes for the others.
struct GenRows {
  int m;                            // rows in use (3 per contact slot)
  int ld;                           // leading dimension of the world's scratch matrices and of the record's dense blocks (genLeadingDim)
  int cap;                          // entries of every array below
  int anyLim;
  double *Bv, *mu, *colNorm;
  double *X, *X0, *E;
  double *t0, *t1, *t2, *t3;        // scratch vectors
  double* invd;
  double* scal;                     // 8 broadcast scalars
  int *cls, *fp, *perm, *gid;
  int* iscal;                       // 8
  unsigned char *fric, *lim, *neg, *rowOn, *on, *done, *in, *pad_;      // (pad_: stage 0's guess rows)
};
constexpr int GEN_ROWS_MINCAP = NBL_MAXC > 64 ? NBL_MAXC : 64;      // (perm doubles as 64 skeleton labels, t0 as 2 x MAX_CONTACTS ints)
__host__ __device__ inline int genRowsCap(int rows) { const int c = (rows + 7) & ~7; return c < GEN_ROWS_MINCAP ? GEN_ROWS_MINCAP : c; }
// doubles (8-byte units) of the pool of a GenRows of `cap` entries: 11 double arrays + 8 scalars, 4 int arrays + 8 ints, 8 byte arrays
__host__ __device__ inline size_t genRowsDoubles(int cap) { return (size_t)11 * cap + 8 + ((size_t)4 * cap + 8 + 1) / 2 + (size_t)cap; }
__host__ __device__ inline void genRowsCarve(GenRows& R, double* pool, int cap) {
  R.cap = cap; R.m = 0; R.ld = 0; R.anyLim = 0;
  double* d = pool;
  R.Bv = d; d += cap; R.mu = d; d += cap; R.colNorm = d; d += cap; R.X = d; d += cap; R.X0 = d; d += cap; R.E = d; d += cap;
  R.t0 = d; d += cap; R.t1 = d; d += cap; R.t2 = d; d += cap; R.t3 = d; d += cap; R.invd = d; d += cap; R.scal = d; d += 8;
  int* i = reinterpret_cast<int*>(d);
  R.cls = i; i += cap; R.fp = i; i += cap; R.perm = i; i += cap; R.gid = i; i += cap; R.iscal = i; i += 8;
  if ((4 * cap + 8) & 1) i += 1;
  unsigned char* u = reinterpret_cast<unsigned char*>(i);
  R.fric = u; u += cap; R.lim = u; u += cap; R.neg = u; u += cap; R.rowOn = u; u += cap; R.on = u; u += cap; R.done = u; u += cap; R.in = u; u += cap; R.pad_ = u;
}

// scratch of one world (HBM): GEN_NMAT matrices of ld x ld doubles + 16 vectors of ld (four matrices + the vectors in the step; the fifth
// matrix: the self-test's problem)
constexpr int GEN_NMAT = 5;
__host__ __device__ inline size_t genScratchDoubles(int ld) { return (size_t)GEN_NMAT * ld * ld + (size_t)16 * ld; }

// ---- sequential sums with their operands fetched FOUR steps at a time -------------------------------------------------------------------
// acc = fma(a(i), b(i), acc) for i = from .. to - 1 IN THAT ORDER (the bits of the plain loop).  Written as a plain loop the compiler
// neither unrolls it (run-time bounds) nor moves the loads of step i + 1 above the multiply-add of step i: every step then waits for
// its own loads - an LDS or L2 round trip per term (the ISA of round 6's first build: s_waitcnt vmcnt(0) lgkmcnt(0) in every iteration,
// 11 instructions per term).  Here the loads of four steps are independent statements before the four dependent multiply-adds.
template <class FA, class FB>
DEV double genFmaSeq(int from, int to, double acc, FA a, FB b) {
  int i = from;
  for (; i + 3 < to; i += 4) {
    const double a0 = a(i), a1 = a(i + 1), a2 = a(i + 2), a3 = a(i + 3);
    const double b0 = b(i), b1 = b(i + 1), b2 = b(i + 2), b3 = b(i + 3);
    acc = fma(a0, b0, acc); acc = fma(a1, b1, acc); acc = fma(a2, b2, acc); acc = fma(a3, b3, acc);
  }
  for (; i < to; i++) acc = fma(a(i), b(i), acc);
  return acc;
}
// y(i) = fma(c, x(i), y(i)) for i = from .. to - 1 (independent steps), four at a time: loads, multiply-adds, stores
template <class FX, class FY>
DEV void genAxpy4(int from, int to, double c, FX x, FY y) {
  int i = from;
  for (; i + 3 < to; i += 4) {
    const double x0 = x(i), x1 = x(i + 1), x2 = x(i + 2), x3 = x(i + 3);
    double& r0 = y(i); double& r1 = y(i + 1); double& r2 = y(i + 2); double& r3 = y(i + 3);
    const double y0 = r0, y1 = r1, y2 = r2, y3 = r3;
    r0 = fma(c, x0, y0); r1 = fma(c, x1, y1); r2 = fma(c, x2, y2); r3 = fma(c, x3, y3);
  }
  for (; i < to; i++) { double& r = y(i); r = fma(c, x(i), r); }
}

// ---- dense helpers: lanes stride through the rows / columns, A symmetric with leading dimension lda ----------------------------------
// y_r = sum_j A[j][r] x_j over the rows that are on (y = 0 on the others); x is masked by `on` as well
template <class W>
DEV void genAx(const W& w, const double* A, int lda, const GenRows& R, const double* x, double* y) {
  const int m = R.m;
  for (int r = w.lane(); r < m; r += w.lanes()) {
    double s = 0.0;
    if (R.on[r]) {
      int j = 0;
      for (; j + 3 < m; j += 4) {      // (operands of four terms first, then the four steps in order: genFmaSeq with the row mask)
        const double a0 = A[(size_t)j * lda + r], a1 = A[(size_t)(j + 1) * lda + r], a2 = A[(size_t)(j + 2) * lda + r], a3 = A[(size_t)(j + 3) * lda + r];
        const double x0 = x[j], x1 = x[j + 1], x2 = x[j + 2], x3 = x[j + 3];
        const bool o0 = R.on[j] != 0, o1 = R.on[j + 1] != 0, o2 = R.on[j + 2] != 0, o3 = R.on[j + 3] != 0;
        if (o0) s = fma(a0, x0, s);
        if (o1) s = fma(a1, x1, s);
        if (o2) s = fma(a2, x2, s);
        if (o3) s = fma(a3, x3, s);
      }
      for (; j < m; j++) if (R.on[j]) s = fma(A[(size_t)j * lda + r], x[j], s);
    }
    y[r] = s;
  }
  w.sync();
}

// LCPUtils::isLCPSolutionValid (LCPUtils.cpp:12-80) on the rows that are on; v: scratch of m doubles.  Uniform result.
template <class W>
DEV bool genValid(const W& w, const double* A, int lda, const GenRows& R, const double* X, bool ignoreFriction, double cfm, double* v) {
  const double tol = 1e-5;
  genAx(w, A, lda, R, X, v);
  bool bad = false;
  for (int r = w.lane(); r < R.m; r += w.lanes()) {
    if (!R.on[r]) continue;
    const double x = X[r];
    const double vr = -R.Bv[r] + cfm * x + v[r];
    double upper = R.fric[r] ? R.mu[r] : INFINITY, lower = R.fric[r] ? -R.mu[r] : 0.0;
    if (R.fric[r]) {
      if (ignoreFriction) { if (x != 0.0) bad = true; continue; }
      const double xn = R.on[R.fp[r]] ? X[R.fp[r]] : 0.0;
      upper *= xn; lower *= xn;
    }
    if (fabs(lower) < tol && fabs(upper) < tol && fabs(x) < tol) {}
    else if (fabs(x - lower) < tol) { if (vr < -tol) bad = true; }
    else if (fabs(x - upper) < tol) { if (vr > tol) bad = true; }
    else if (x > lower && x < upper) { if (fabs(vr) > tol) bad = true; }
    else bad = true;
  }
  const bool any = w.anyAll(bad);
  w.sync();
  return !any;
}

struct GenClasses { int nc, nu; };

// CGGM::constructMatrices classification (CGGM.cpp:535-713) of the rows that are on -> R.cls, R.E
template <class W>
DEV GenClasses genClassify(const W& w, GenRows& R, const double* X, bool ignoreFriction) {
  const double TH = 1e-6, tie = 1e-5;
  const int m = R.m;
  // pass 1: clamping or not (a friction row looks at the impulse of its normal row, not at its class)
  for (int r = w.lane(); r < m; r += w.lanes()) {
    int cls = RC_NOT_CLAMPING;
    bool inElse = false;
    const double x = R.on[r] ? X[r] : 0.0;
    const double xn = R.on[R.fp[r]] ? X[R.fp[r]] : 0.0;
    const double hi = R.fric[r] ? R.mu[r] : INFINITY, lo = R.fric[r] ? -R.mu[r] : 0.0;
    double upper = hi, lower = lo;
    if (R.fric[r]) { upper *= xn; lower *= xn; }
    if (R.on[r] && !(R.colNorm[r] < 1e-9)) {
      if (fabs(x) < TH) {
        if (R.fric[r] && !(fabs(xn) < TH) && !ignoreFriction) cls = RC_CLAMPING;
      } else if ((x > lower + tie && x < upper - tie) || (lower - x > 1e-2 || x - upper > 1e-2)) cls = RC_CLAMPING;
      else inElse = true;
    }
    R.cls[r] = cls;
    R.in[r] = inElse ? 1 : 0;
    R.E[r] = 0.0;
  }
  w.sync();
  // pass 2: a friction row on its bound rides on its (clamping) normal row
  for (int r = w.lane(); r < m; r += w.lanes()) {
    if (!R.in[r] || !R.fric[r]) continue;
    const int fp = R.fp[r];
    const double xn = R.on[fp] ? X[fp] : 0.0;
    if (fabs(xn) > 1e-9 && R.colNorm[fp] > 1e-9 && R.cls[fp] == RC_CLAMPING) {
      R.cls[r] = RC_UPPER_BOUND;
      const double ub = xn * R.mu[r], lb = -xn * R.mu[r];
      R.E[r] = (fabs(X[r] - ub) < fabs(X[r] - lb)) ? R.mu[r] : -R.mu[r];
    }
  }
  w.sync();
  int nc = 0, nu = 0;
  for (int r = w.lane(); r < m; r += w.lanes()) { nc += R.cls[r] == RC_CLAMPING; nu += R.cls[r] == RC_UPPER_BOUND; }
  GenClasses K;
  K.nc = (int)w.sumAll((double)nc); K.nu = (int)w.sumAll((double)nu);
  return K;
}

// Q[i][s] = A[i][s] + [s normal] sum_{u = s+1, s+2 upper-bound} E[u] A[i][u] + cfm [i == s] for clamping i and s, zero elsewhere
// (coopBuildQ of coop_dev.hpp with loops instead of lanes) -> M (row-major, leading dimension R.ld)
template <class W>
DEV void genBuildQ(const W& w, const double* A, int lda, const GenRows& R, const GenClasses& K, double cfm, double* M, const double* cfmRow = nullptr, int ldM = 0) {
  const int ld = ldM > 0 ? ldM : R.ld;      // leading dimension of M (default: that of the scratch matrices - the model's rows, rounded up)
  const int m = R.m;
  for (int s = w.lane(); s < m; s += w.lanes()) {
    const bool colOn = R.cls[s] == RC_CLAMPING;
    double e1 = 0.0, e2 = 0.0;
    if (K.nu > 0 && !R.fric[s] && s + 2 < m) {
      if (R.cls[s + 1] == RC_UPPER_BOUND) e1 = R.E[s + 1];
      if (R.cls[s + 2] == RC_UPPER_BOUND) e2 = R.E[s + 2];
    }
    // (A is symmetric: A[i][s] = A[s][i]; rows that are off are inert.)  The entries of A a row needs are fetched for FOUR rows before the
    // first of them is used (every one an L2 round trip; behind the branches of the plain loop they came one by one: genFmaSeq)
    const bool three = K.nu > 0 && !R.fric[s] && s + 2 < m;
    const int s1 = s + 1 < m ? s + 1 : s, s2 = s + 2 < m ? s + 2 : s;
    const bool onS = R.on[s] != 0, onS1 = R.on[s1] != 0, onS2 = R.on[s2] != 0, limS = R.lim[s] != 0;
    auto qOf = [&](int i, double as, double as1, double as2) -> double {
      double q = 0.0;
      if (colOn && R.cls[i] == RC_CLAMPING) {
        const bool oi = R.on[i] != 0;
        q = (onS && oi) ? as : 0.0;
        if (three) q = fma(e2, (onS2 && oi) ? as2 : 0.0, fma(e1, (onS1 && oi) ? as1 : 0.0, q));
        // a joint-limit constraint has no constraint-force column in the reference's Q = A_c^T M^-1 (A_c + A_ub E) (DCC.cpp:51-99)
        if (K.nu > 0 && (limS || R.lim[i])) q = 0.0;
        if (i == s) q += cfmRow ? cfmRow[s] : cfm;     // (cfmRow: one constant per row, its constrained group's)
      }
      return q;
    };
    int i = 0;
    for (; i + 3 < m; i += 4) {
      const double* r0 = A + (size_t)i * lda; const double* r1 = r0 + lda; const double* r2 = r1 + lda; const double* r3 = r2 + lda;
      const double a0 = r0[s], a1 = r1[s], a2 = r2[s], a3 = r3[s];
      double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
      if (K.nu > 0) { b0 = r0[s1]; b1 = r1[s1]; b2 = r2[s1]; b3 = r3[s1]; c0 = r0[s2]; c1 = r1[s2]; c2 = r2[s2]; c3 = r3[s2]; }
      const double q0 = qOf(i, a0, b0, c0), q1 = qOf(i + 1, a1, b1, c1), q2 = qOf(i + 2, a2, b2, c2), q3 = qOf(i + 3, a3, b3, c3);
      M[(size_t)i * ld + s] = q0; M[(size_t)(i + 1) * ld + s] = q1; M[(size_t)(i + 2) * ld + s] = q2; M[(size_t)(i + 3) * ld + s] = q3;
    }
    for (; i < m; i++) {
      const double* r0 = A + (size_t)i * lda;
      M[(size_t)i * ld + s] = qOf(i, r0[s], K.nu > 0 ? r0[s1] : 0.0, K.nu > 0 ? r0[s2] : 0.0);
    }
  }
  w.sync();
}

// P <- pseudo-inverse of the m x m matrix M (row-major, masked rows / columns zero; destroyed); G, T: scratch matrices.  cTrue = number
// of unmasked columns (Eigen's `size` in the rank threshold eps * size * |R_00| of completeOrthogonalDecomposition, CGGM.cpp:280,
// LCPUtils.cpp:113).  Column-pivoted Householder QR M Pi = H [R1 R2; 0], G = H^T carried along; full rank: Q^+ = Pi R1^-1 G1; rank
// deficient: R = R1 [I W], W = R1^-1 R2, and the minimum-norm solution of R u = g is u = [I; W^T] (I + W W^T)^-1 R1^-1 g - the same
// algebra as coopPinvImpl (coop_dev.hpp), which explains why that is as accurate as the second Householder pass.  Returns the rank.
// Tfast / TfastDoubles: memory the lanes share FAST (LDS on the device) that is free for the duration of the call; the r x r matrix
// I + W W^T and its Cholesky factor live there (packed, leading dimension r) when r^2 fits.  Its factorisation is a chain of 2 r barriers
// each behind a store: through HBM scratch every one of them waited for the store to land (two flat feet: r = 12, ~100 k cycles per call).
template <class W>
DEV int genPinv(const W& w, GenRows& R, double* M, double* G, double* T, double* P, int m, int cTrue, bool symPsd = false, int ldMG = 0,
                double* Tfast = nullptr, int TfastDoubles = 0) {
  const int ld = R.ld;              // leading dimension of T and P (the scratch matrices: the model's rows, rounded up)
  const int lm = ldMG > 0 ? ldMG : ld;   // ... and of M and G: the caller may hand in a packed pair in fast memory (genPinvFast)
  const int ln = w.lane(), nl = w.lanes();
  GEN_L0(); GEN_CNT(29);
  for (int j = ln; j < m; j += nl) { R.done[j] = 0; for (int i = 0; i < m; i++) G[(size_t)i * lm + j] = (i == j) ? 1.0 : 0.0; }
  w.sync();
  // Rank threshold: the reference's eps * size * |R_00| (CGGM.cpp:280, LCPUtils.cpp:113).  For a SYMMETRIC positive semi-definite matrix
  // (A on the guess rows; Q with no friction row on its bound) 64 x that, the policy of the 24- / 48-row builds' Cholesky route
  // (coopPinvSymImpl, coop_dev.hpp): on an exactly singular Q the trailing pivot is pure round-off, a few eps - close enough to
  // eps * size that this factorisation's round-off passes it where the reference's does not (soak seeds 240049 / 243083 on the
  // general build: a 3 x 3 Q of rank 2 inverted at rank 3, gradients of 1e13); 64 x keeps every round-off pivot out and differs from
  // the reference only where its own answer is round-off times 1e12
  const double thr = (symPsd ? 64.0 : 1.0) * 2.220446049250313e-16 * cTrue;
  double best0 = 0.0;
  int rank = 0;
  double* V = R.t3;
  for (int k = 0; k < m; k++) {
    // remaining squared column norms, pivot = the largest (lowest index among equals)
    double myBest = -1.0;
    int myCol = 0x7fffffff;
    for (int j = ln; j < m; j += nl) {
      if (R.done[j]) continue;
      double s0 = 0.0, s1 = 0.0;
      int i = k;
      for (; i + 3 < m; i += 4) {
        const double a = M[(size_t)i * lm + j], b = M[(size_t)(i + 1) * lm + j], c = M[(size_t)(i + 2) * lm + j], d = M[(size_t)(i + 3) * lm + j];
        s0 = fma(a, a, s0); s1 = fma(b, b, s1); s0 = fma(c, c, s0); s1 = fma(d, d, s1);
      }
      for (; i + 1 < m; i += 2) { const double a = M[(size_t)i * lm + j], b = M[(size_t)(i + 1) * lm + j]; s0 = fma(a, a, s0); s1 = fma(b, b, s1); }
      if (i < m) { const double a = M[(size_t)i * lm + j]; s0 = fma(a, a, s0); }
      const double nrm = s0 + s1;
      if (nrm > myBest) { myBest = nrm; myCol = j; }
    }
    const double best = w.maxAll(myBest);
    if (k == 0) best0 = best;
    if (!(best > thr * thr * best0) || !(best > 0.0)) break;
    const int p = w.minAllI(myBest == best ? myCol : 0x7fffffff);
    // the reflector of column p: V[i] = x_i (unscaled), v = x - alpha e_k scaled so that v_k = 1
    for (int i = k + ln; i < m; i += nl) V[i] = M[(size_t)i * lm + p];
    w.sync();
    const double akk = V[k];
    const double below = genFmaSeq(k + 1, m, 0.0, [&](int i) { return V[i]; }, [&](int i) { return V[i]; });
    const double normx = sqrt(fma(akk, akk, below));
    const double alpha = akk > 0 ? -normx : normx;
    const double vk = akk - alpha;
    const double vnorm2 = fma(vk, vk, below);
    const double vinv = 1.0 / vk;
    const double tau = 2.0 * vk * vk / vnorm2;
    // H = I - tau v v^T applied to the columns still in play and to the carried block
    for (int jj = ln; jj < 2 * m; jj += nl) {
      const bool carried = jj >= m;
      const int j = carried ? jj - m : jj;
      double* Mc = carried ? G : M;
      if (!carried && (R.done[j] || j == p)) continue;
      double d = genFmaSeq(k + 1, m, 0.0, [&](int i) { return V[i]; }, [&](int i) { return Mc[(size_t)i * lm + j]; });
      d = fma(vinv, d, Mc[(size_t)k * lm + j]) * tau;
      Mc[(size_t)k * lm + j] -= d;
      const double dv = d * vinv;
      genAxpy4(k + 1, m, -dv, [&](int i) { return V[i]; }, [&](int i) -> double& { return Mc[(size_t)i * lm + j]; });
    }
    w.sync();
    if (ln == 0) { M[(size_t)k * lm + p] = alpha; R.done[p] = 1; R.perm[k] = p; }
    for (int i = k + 1 + ln; i < m; i += nl) M[(size_t)i * lm + p] = 0.0;
    w.sync();
    rank = k + 1;
  }
  const int r = rank;
  GEN_L(26);
  if (ln == 0) {   // columns never chosen (dependent or masked) take the remaining pivot positions in index order
    int pos = r;
    for (int j = 0; j < m; j++) if (!R.done[j]) R.perm[pos++] = j;
  }
  w.sync();
  if (r == 0) {
    for (int j = ln; j < m; j += nl) for (int i = 0; i < m; i++) P[(size_t)i * ld + j] = 0.0;
    w.sync();
    return 0;
  }
  for (int k = ln; k < r; k += nl) R.invd[k] = 1.0 / M[(size_t)k * lm + R.perm[k]];
  w.sync();
  // x = R1^-1 (column) in place, R1[i][k] = M[i][perm[k]]: the columns of G1 (right-hand sides) and, rank deficient, those of R2
  const int nExtra = r >= cTrue ? 0 : m - r;
  for (int jj = ln; jj < m + nExtra; jj += nl) {
    double* Mc = jj < m ? G : M;
    const int j = jj < m ? jj : R.perm[r + (jj - m)];
    for (int kk = r - 1; kk >= 0; kk--) {
      const int pk = R.perm[kk];
      const double y = Mc[(size_t)kk * lm + j] * R.invd[kk];
      Mc[(size_t)kk * lm + j] = y;
      genAxpy4(0, kk, y, [&](int i) { return -M[(size_t)i * lm + pk]; }, [&](int i) -> double& { return Mc[(size_t)i * lm + j]; });
    }
  }
  w.sync();
  GEN_L(27);
  if (r >= cTrue) {
    for (int j = ln; j < m; j += nl) {
      for (int kk = 0; kk < r; kk++) P[(size_t)R.perm[kk] * ld + j] = G[(size_t)kk * lm + j];
      for (int pp = r; pp < m; pp++) P[(size_t)R.perm[pp] * ld + j] = 0.0;
    }
    w.sync();
    GEN_L(28);
    return r;
  }
  GEN_CNT(30);
  // S = I + W W^T (r x r) -> T, W[i][t] = M[i][perm[r + t]]
  const int nw = m - r;
  int lt = ld;
  if (Tfast && r * r <= TfastDoubles) { T = Tfast; lt = r; }
  for (int e = ln; e < r * r; e += nl) {
    const int a = e / r, b = e - a * r;
    double s = (a == b) ? 1.0 : 0.0;
    s = genFmaSeq(0, nw, s, [&](int t) { return M[(size_t)a * lm + R.perm[r + t]]; }, [&](int t) { return M[(size_t)b * lm + R.perm[r + t]]; });
    T[(size_t)a * lt + b] = s;
  }
  w.sync();
  // Cholesky S = L L^T in place (lower triangle of T), lanes = rows below the pivot
  for (int k = 0; k < r; k++) {
    if (ln == 0) {
      double s = T[(size_t)k * lt + k];
      s = genFmaSeq(0, k, s, [&](int i) { return -T[(size_t)k * lt + i]; }, [&](int i) { return T[(size_t)k * lt + i]; });
      const double lkk = sqrt(s);
      T[(size_t)k * lt + k] = lkk;
      R.scal[0] = 1.0 / lkk;
    }
    w.sync();
    const double inv = R.scal[0];
    for (int a = k + 1 + ln; a < r; a += nl) {
      double s = T[(size_t)a * lt + k];
      s = genFmaSeq(0, k, s, [&](int i) { return -T[(size_t)a * lt + i]; }, [&](int i) { return T[(size_t)k * lt + i]; });
      T[(size_t)a * lt + k] = s * inv;
    }
    w.sync();
  }
  // z = S^-1 x for the columns of G1 (in place), then Q^+ = Pi [z; W^T z]
  for (int j = ln; j < m; j += nl) {
    for (int k = 0; k < r; k++) {
      double s = G[(size_t)k * lm + j];
      s = genFmaSeq(0, k, s, [&](int i) { return -T[(size_t)k * lt + i]; }, [&](int i) { return G[(size_t)i * lm + j]; });
      G[(size_t)k * lm + j] = s / T[(size_t)k * lt + k];
    }
    for (int k = r - 1; k >= 0; k--) {
      double s = G[(size_t)k * lm + j];
      s = genFmaSeq(k + 1, r, s, [&](int i) { return -T[(size_t)i * lt + k]; }, [&](int i) { return G[(size_t)i * lm + j]; });
      G[(size_t)k * lm + j] = s / T[(size_t)k * lt + k];
    }
    for (int k = 0; k < r; k++) P[(size_t)R.perm[k] * ld + j] = G[(size_t)k * lm + j];
    for (int t = 0; t < nw; t++) {
      const int c = R.perm[r + t];
      double s = 0.0;
      s = genFmaSeq(0, r, s, [&](int i) { return M[(size_t)i * lm + c]; }, [&](int i) { return G[(size_t)i * lm + j]; });
      P[(size_t)c * ld + j] = s;
    }
  }
  w.sync();
  GEN_L(28);
  return r;
}

// y_i = sum_k P[i][k] x_k (TRANS: P[k][i]) for i < m
template <class W, bool TRANS>
DEV void genPinvApply(const W& w, const double* P, int ld, int m, const double* x, double* y) {
  for (int i = w.lane(); i < m; i += w.lanes()) {
    double s = 0.0;
    s = genFmaSeq(0, m, s, [&](int k) { return TRANS ? P[(size_t)k * ld + i] : P[(size_t)i * ld + k]; }, [&](int k) { return x[k]; });
    y[i] = s;
  }
  w.sync();
}

// the world's scratch matrices
struct GenScratch {
  double* mat[GEN_NMAT];    // M, G, T, P, and one more for the cascade's problem / factor (ld x ld each)
  double* vec;              // 16 x ld doubles
  int ld;
  // A small pool the lanes share FAST (LDS on the device; NULL: none): GEN_FAST_MATS matrices of fastN x fastN and 20 vectors of fastN
  // doubles.  A problem of at most fastN rows - eight or ten contacts of a model that asked for many more slots - runs its Dantzig driver
  // and its Gauss-Seidel sweeps there instead of in HBM scratch, whose latency (the scratch of a launch does not fit the L2) every one of
  // their dependent steps waited for.  Placement only: the arithmetic does not change.
  double* fast = nullptr;
  int fastN = 0;
  bool vecFast = false;     // `vec` lives in fast memory (LDS)
  int vecDoubles = 0;       // ... and has this many doubles (>= 16 ld)
  int fastMats = 0;         // matrices of fastN x fastN in the pool (the Dantzig driver wants GEN_FAST_MATS + its vectors, Gauss-Seidel one)
};
// The working pair of genPinv (M, the matrix being factorised, and G, the carried block: what every Householder step reads and writes) for
// a problem of m rows: packed (leading dimension m) in the cascade's 16 scratch vectors when those live in fast memory (vecFast: LDS on the
// device, genSolveVecDoubles) and 2 m^2 doubles fit there (eight contacts: always).  Otherwise the scratch matrices.
struct GenPinvPair { double* M; double* G; int ld; };
DEV GenPinvPair genPinvPair(const GenScratch& S, int m) {
  GenPinvPair p;
  if (S.vecFast && (size_t)2 * m * m <= (size_t)S.vecDoubles) { p.M = S.vec; p.G = S.vec + (size_t)m * m; p.ld = m; }
  else { p.M = S.mat[0]; p.G = S.mat[1]; p.ld = S.ld; }
  return p;
}
constexpr int GEN_FAST_N = 32;
constexpr int GEN_FAST_MATS = 3;
constexpr int GEN_FAST_DOUBLES = GEN_FAST_MATS * GEN_FAST_N * GEN_FAST_N + 20 * GEN_FAST_N;

// CGGM::constructMatrices + opportunisticallyStandardizeResults as a loop (coopStandardizeLoop of coop_dev.hpp).  X in: the solver's x
// (R.X), out: the last accepted solution.  guessValid: S.mat[3] holds the pseudo-inverse of A restricted to the rows R.in0 (stage 0's
// guess).  Returns whether the results are standardised; pinvValid: S.mat[3] is Q^+ of the classification in R.cls.
template <class W>
DEV bool genStandardizeLoop(const W& w, const double* A, int lda, GenRows& R, const GenScratch& S, double cfm, bool ignoreFriction,
                            const unsigned char* guessRows, bool& pinvValid, GenClasses& K) {
  const int m = R.m;
  double* X = R.X;
  double* fc = R.t0;
  double* newX = R.t1;
  bool ok = false;
  GEN_L0();
  for (int iter = 0; iter < m + 1; iter++) {
    GEN_CNT(24);
    K = genClassify(w, R, X, ignoreFriction);
    GEN_L(19);
    if (K.nc == 0) {
      pinvValid = false;
      for (int r = w.lane(); r < m; r += w.lanes()) newX[r] = 0.0;
      w.sync();
      ok = genValid(w, A, lda, R, newX, ignoreFriction, cfm, R.t2);
      if (ok) { for (int r = w.lane(); r < m; r += w.lanes()) X[r] = 0.0; w.sync(); }
      break;
    }
    bool sameAsGuess = false;
    if (iter == 0 && K.nu == 0 && guessRows) {
      bool diff = false;
      for (int r = w.lane(); r < m; r += w.lanes()) if ((R.cls[r] == RC_CLAMPING) != (guessRows[r] != 0)) diff = true;
      sameAsGuess = !w.anyAll(diff);
    }
    if (sameAsGuess) {
      for (int r = w.lane(); r < m; r += w.lanes()) fc[r] = X[r];
      w.sync();
    } else {
      const GenPinvPair pp = genPinvPair(S, m);
      genBuildQ(w, A, lda, R, K, cfm, pp.M, nullptr, pp.ld);
      GEN_L(20); GEN_CNT(25);
      genPinv(w, R, pp.M, pp.G, S.mat[2], S.mat[3], m, K.nc, K.nu == 0, pp.ld, R.t0, 3 * R.cap);      // (t0 .. t2: free here, contiguous)
      for (int r = w.lane(); r < m; r += w.lanes()) R.t2[r] = R.cls[r] == RC_CLAMPING ? R.Bv[r] : 0.0;
      w.sync();
      GEN_L(21);
      genPinvApply<W, false>(w, S.mat[3], R.ld, m, R.t2, fc);
      pinvValid = true;
    }
    bool newlyNot = false;
    for (int r = w.lane(); r < m; r += w.lanes()) {
      double nx = 0.0;
      if (R.cls[r] == RC_CLAMPING) {
        nx = fc[r];
        if (fabs(nx) < 1e-6 && fabs(X[r]) > 1e-6 && !R.fric[r]) newlyNot = true;
      } else if (R.cls[r] == RC_UPPER_BOUND) {
        const double om = X[R.fp[r]] / X[r];
        const double clean = (fabs(om - R.mu[r]) < fabs(om + R.mu[r])) ? R.mu[r] : -R.mu[r];
        nx = fc[R.fp[r]] * clean;
      }
      newX[r] = nx;
    }
    w.sync();
    const bool again = w.anyAll(newlyNot);
    GEN_L(22);
    const bool valid_ = genValid(w, A, lda, R, newX, ignoreFriction, cfm, R.t2);
    GEN_L(23);
    if (!valid_) { ok = false; break; }
    for (int r = w.lane(); r < m; r += w.lanes()) X[r] = newX[r];
    w.sync();
    ok = true;
    if (!again) break;
    pinvValid = false;   // X moved on: K and S.mat[3] belong to the previous iterate until the next pass refactorises (matters when the loop runs out)
  }
  return ok;
}

// LCPUtils::guessSolution (when there is no matching warm start) + the standardisation loop: stage 0 of the solver cascade
// (BoxedLcpConstraintSolver.cpp:380-460).  R.X in: the warm start (haveCache), out: the solution; R.X0: the pre-solve x.
template <class W>
DEV bool genStage0(const W& w, const double* A, int lda, GenRows& R, const GenScratch& S, bool haveCache, bool& pinvValid, GenClasses& K) {
  const int ld = R.ld;              // leading dimension of the scratch matrices (the model's rows, rounded up)
  const int m = R.m;
  pinvValid = false;
  bool haveGuess = false;
  unsigned char* in0 = R.pad_;
  GEN_L0();
  if (haveCache) {
    for (int r = w.lane(); r < m; r += w.lanes()) { if (!R.on[r]) R.X[r] = 0.0; in0[r] = 0; }
    w.sync();
  } else {
    // (the empty tangent rows of frictionless contacts are not rows of the reference's problem; a negated joint-limit row: the reference tests ITS b > 0)
    int cnt = 0;
    for (int r = w.lane(); r < m; r += w.lanes()) {
      const bool in = R.on[r] && (R.fric[r] ? R.mu[r] != 0.0 : (R.neg[r] ? R.Bv[r] < 0 : R.Bv[r] > 0));
      in0[r] = in ? 1 : 0;
      cnt += in;
      R.X[r] = 0.0;
    }
    w.sync();
    const int nIn = (int)w.sumAll((double)cnt);
    if (nIn > 0) {
      const GenPinvPair pp = genPinvPair(S, m);
      double* M = pp.M;
      for (int s = w.lane(); s < m; s += w.lanes())
        for (int i = 0; i < m; i++) M[(size_t)i * pp.ld + s] = (in0[s] && in0[i]) ? A[(size_t)i * lda + s] : 0.0;
      w.sync();
      GEN_L(16);
      genPinv(w, R, M, pp.G, S.mat[2], S.mat[3], m, nIn, true, pp.ld, R.t0, 3 * R.cap);
      GEN_L(17);          // A restricted to the guess rows: symmetric positive semi-definite
      for (int r = w.lane(); r < m; r += w.lanes()) R.t2[r] = in0[r] ? R.Bv[r] : 0.0;
      w.sync();
      genPinvApply<W, false>(w, S.mat[3], R.ld, m, R.t2, R.t0);
      for (int r = w.lane(); r < m; r += w.lanes()) R.X[r] = in0[r] ? R.t0[r] : 0.0;
      w.sync();
      haveGuess = true;
      pinvValid = true;   // of A restricted to the guess rows; stays valid only if the first classification agrees
    }
  }
  for (int r = w.lane(); r < m; r += w.lanes()) R.X0[r] = R.X[r];
  w.sync();
  GEN_L(18);
  const bool ok = genStandardizeLoop(w, A, lda, R, S, 0.0, false, haveGuess ? in0 : nullptr, pinvValid, K);
  pinvValid = ok && pinvValid;
  return ok;
}

// ---- stages 1-3: the reduced problems ---------------------------------------------------------------------------------------------------
struct GenProblem {          // compacted boxed LCP (arrays in the world's scratch vectors), matrix n x n with leading dimension ld
  int n, ld;
  double *A, *x, *b, *lo, *hi;
  int *findex, *mapTo;      // mapTo[original row] = column of the problem (-1: dropped)
};

// A (+ cfm on the diagonal) restricted to the rows the reference's problem has: rows that are on, without the empty tangent rows of
// frictionless contacts (ContactConstraint dimension 1).  Lane 0 compacts the indices, the lanes copy.
template <class W>
DEV void genLoadProblem(const W& w, const double* A, int lda, GenRows& R, double cfmDiag, const double* x0, GenProblem& P) {
  const int ld = R.ld;              // leading dimension of the scratch matrices (the model's rows, rounded up)
  const int m = R.m;
  if (w.lane() == 0) {
    int n = 0;
    for (int r = 0; r < m; r++) {
      const bool keep = R.on[r] && !(R.fric[r] && R.mu[r] == 0.0);
      P.mapTo[r] = keep ? n : -1;
      if (keep) R.perm[n++] = r;
    }
    for (int r = 0; r < m; r++) {
      const int c = P.mapTo[r];
      if (c < 0) continue;
      P.x[c] = x0[r]; P.b[c] = R.Bv[r];
      P.lo[c] = R.fric[r] ? -R.mu[r] : 0.0; P.hi[c] = R.fric[r] ? R.mu[r] : INFINITY;
      P.findex[c] = R.fric[r] ? P.mapTo[R.fp[r]] : -1;
    }
    R.iscal[0] = n;
  }
  w.sync();
  const int n = R.iscal[0];
  P.n = n; P.ld = ld;
  for (int j = w.lane(); j < n; j += w.lanes()) {
    const int sj = R.perm[j];
    int i = 0;
    for (; i + 3 < n; i += 4) {      // (four entries of the record's A before the four stores: genFmaSeq)
      const int p0 = R.perm[i], p1 = R.perm[i + 1], p2 = R.perm[i + 2], p3 = R.perm[i + 3];
      const double a0 = A[(size_t)p0 * lda + sj], a1 = A[(size_t)p1 * lda + sj], a2 = A[(size_t)p2 * lda + sj], a3 = A[(size_t)p3 * lda + sj];
      P.A[(size_t)i * ld + j] = a0 + (i == j ? cfmDiag : 0.0); P.A[(size_t)(i + 1) * ld + j] = a1 + (i + 1 == j ? cfmDiag : 0.0);
      P.A[(size_t)(i + 2) * ld + j] = a2 + (i + 2 == j ? cfmDiag : 0.0); P.A[(size_t)(i + 3) * ld + j] = a3 + (i + 3 == j ? cfmDiag : 0.0);
    }
    for (; i < n; i++) P.A[(size_t)i * ld + j] = A[(size_t)R.perm[i] * lda + sj] + (i == j ? cfmDiag : 0.0);
  }
  w.sync();
}

// delete row + column `col` (lane 0)
DEV void genRemoveRowCol(GenProblem& P, int col) {
  const int ld = P.ld;              // leading dimension of the scratch matrices (the model's rows, rounded up)
  const int n = P.n;
  for (int i = 0; i < n; i++) {
    if (i == col) continue;
    const int ni = i > col ? i - 1 : i;
    for (int j = 0; j < n; j++) {
      if (j == col) continue;
      const int nj = j > col ? j - 1 : j;
      P.A[(size_t)ni * ld + nj] = P.A[(size_t)i * ld + j];     // rows / columns move up-left: reads stay ahead of writes
    }
  }
  for (int i = col; i + 1 < n; i++) { P.x[i] = P.x[i + 1]; P.b[i] = P.b[i + 1]; P.lo[i] = P.lo[i + 1]; P.hi[i] = P.hi[i + 1]; P.findex[i] = P.findex[i + 1]; }
  P.n = n - 1;
}

// LCPUtils::reduce (LCPUtils.cpp:144-201, mergeLCPColumns :346-449): merge near-identical columns (squared distance < 1e-4, |b_a - b_b| <
// 1e-4, same findex / hi / lo).  mOrig: rows of the world (for mapTo).  The SEARCH for the first pair (a, b) in the reference's order - a
// ascending, then b - is shared by the lanes (round 6; on lane 0 alone it was 276 pairs x four dependent loads from HBM scratch for the
// metric worlds, where nothing merges): for every a the lanes test the columns b > a side by side - the cheap conditions first, the n-term
// column distance only where they hold - and the lowest b that passes wins.  The merge itself (rare) stays with lane 0.
template <class W>
DEV void genLcpReduce(const W& w, GenRows& R, GenProblem& P, int mOrig) {
  const int ld = R.ld;              // leading dimension of the scratch matrices (the model's rows, rounded up)
  const double TH = 1e-4;
  const int ln = w.lane(), nl = w.lanes();
  for (;;) {
    const int n = P.n;
    int ma = -1, mb = -1;
    for (int a = 0; a < n - 1; a++) {
      const double ba = P.b[a], ha = P.hi[a], la = P.lo[a];
      const int fa = P.findex[a];
      int mine = 0x7fffffff;
      for (int b = a + 1 + ln; b < n; b += nl) {
        if (!(fabs(ba - P.b[b]) < TH && fa == P.findex[b] && ha == P.hi[b] && la == P.lo[b])) continue;
        double d2 = 0.0;
        for (int r = 0; r < n; r++) { const double d = P.A[(size_t)r * ld + a] - P.A[(size_t)r * ld + b]; d2 += d * d; }
        if (d2 < TH) { mine = b; break; }       // (a lane visits its columns in ascending order: its first hit is its lowest)
      }
      if (!w.anyAll(mine != 0x7fffffff)) continue;
      ma = a; mb = w.minAllI(mine);
      break;
    }
    if (ma < 0) break;
    w.sync();
    if (ln == 0) {
      for (int r = 0; r < n; r++) P.A[(size_t)r * ld + ma] *= 2.0;
      for (int i = 0; i < n; i++) {
        if (P.findex[i] == mb) P.findex[i] = ma;
        else if (P.findex[i] > mb) P.findex[i] -= 1;
      }
      genRemoveRowCol(P, mb);
      for (int o = 0; o < mOrig; o++) {
        if (P.mapTo[o] == mb) P.mapTo[o] = ma;
        else if (P.mapTo[o] > mb) P.mapTo[o] -= 1;
      }
    }
    P.n = n - 1;            // (every lane holds its own copy of the descriptor)
    w.sync();
  }
}

// LCPUtils::removeFriction (LCPUtils.cpp:208-247): drop every row with findex != -1.  What removing them one by one from the last one
// down leaves - the kept rows / columns in their order, findex of a kept row unchanged (-1), mapTo of a dropped row -1 - as ONE gather
// through the scratch matrix T (round 6; one by one on lane 0 it was sixteen passes over the matrix in HBM scratch for eight contacts:
// 1.2 M cycles, more than the Gauss-Seidel sweeps it prepares).  Pure data movement.
template <class W>
DEV void genLcpRemoveFriction(const W& w, GenRows& R, GenProblem& P, int mOrig, double* T) {
  const int ld = R.ld;
  const int ln = w.lane(), nl = w.lanes();
  const int n = P.n;
  int* src = R.perm;                // new row -> old row
  int* newOf = R.cls;               // old row -> new row or -1 (the classes are not in use between the stages)
  if (ln == 0) {
    int k = 0;
    for (int i = 0; i < n; i++) { const bool keep = P.findex[i] == -1; newOf[i] = keep ? k : -1; if (keep) src[k++] = i; }
    R.iscal[3] = k;
  }
  w.sync();
  const int nNew = R.iscal[3];
  for (int o = ln; o < mOrig; o += nl) { const int c = P.mapTo[o]; if (c >= 0) P.mapTo[o] = newOf[c]; }
  for (int idx = ln; idx < nNew * nNew; idx += nl) { const int i = idx / nNew, j = idx - i * nNew; T[(size_t)i * ld + j] = P.A[(size_t)src[i] * ld + src[j]]; }
  // (the row data through the four scratch vectors of the rows: gathered, barrier, written back)
  for (int k = ln; k < nNew; k += nl) { const int sk = src[k]; R.t0[k] = P.x[sk]; R.t1[k] = P.b[sk]; R.t2[k] = P.lo[sk]; R.t3[k] = P.hi[sk]; }
  w.sync();
  for (int idx = ln; idx < nNew * nNew; idx += nl) { const int i = idx / nNew, j = idx - i * nNew; P.A[(size_t)i * ld + j] = T[(size_t)i * ld + j]; }
  for (int k = ln; k < nNew; k += nl) { P.x[k] = R.t0[k]; P.b[k] = R.t1[k]; P.lo[k] = R.t2[k]; P.hi[k] = R.t3[k]; P.findex[k] = -1; }
  P.n = nNew;
  w.sync();
}

// one Gauss-Seidel row: x <- clamp(x + r) between the bounds (a friction row: +- mu times the impulse of its normal row, read from xs);
// returns the change.  FIRST: the first sweep's test |d| > 1e-6, later |d| > 1e-3 |x| where |x| > 1e-9 (the reference divides; the product
// form is what the lane = row builds test).  zeroRow: "x[i] = 0; continue" - the row leaves the problem, the others see x_i go to 0.
DEV double genPgsClamp(const double* xs, double xo, double r, double hi, double lo, int fi, bool first, bool zeroRow, double& nxOut, bool& moved) {
  const double dxTh = 1e-6, relTol = 1e-3, epsDiv = 1e-9;
  double nx = 0.0;
  if (!zeroRow) {
    nx = xo + r;
    if (fi >= 0) { hi = hi * xs[fi]; lo = -hi; }
    nx = nx > hi ? hi : (nx < lo ? lo : nx);
  }
  const double d = nx - xo;
  if (!zeroRow && (first ? fabs(d) > dxTh : (fabs(nx) > epsDiv && fabs(d) > relTol * fabs(nx)))) moved = true;
  nxOut = nx;
  return d;
}
// The sweeps with every lane holding its NTT rows (row j = lane + lanes * t) in registers: residual, x, bounds, findex; the entries
// a'_.i of a step (row i of AT) are fetched one step ahead.  The only shared traffic of a step is x_i (written by its owner, read by the
// friction rows that follow it) and the broadcast of the change.  R.t0 = x, R.t1 = r, R.t2 / R.t3 = bounds, R.cls = findex, R.perm = the
// rows in the problem (`no` of them), R.in = rows left out: set up by genPgs.  x comes back in R.t0.
template <int NTT, class W>
DEV bool genPgsHeld(const W& w, GenRows& R, int n, int no, const double* AT, int atLd) {
  const int maxIteration = 30;
  const int ln = w.lane(), nl = w.lanes();
  double* xs = R.t0;
  const int* order = R.perm;
  double cn[NTT], rr[NTT], xx[NTT], hh[NTT], ll[NTT];
  int ff[NTT];
#pragma unroll
  for (int t = 0; t < NTT; t++) {
    const int j = ln + nl * t; const bool in = j < n;
    rr[t] = in ? R.t1[j] : 0.0; xx[t] = in ? xs[j] : 0.0; ll[t] = in ? R.t2[j] : 0.0; hh[t] = in ? R.t3[j] : 0.0; ff[t] = in ? R.cls[j] : -1; cn[t] = 0.0;
  }
  auto fetch = [&](int i) {
    if (i < 0) return;
    const double* col = AT + (size_t)i * atLd;
#pragma unroll
    for (int t = 0; t < NTT; t++) { const int j = ln + nl * t; cn[t] = j < n ? col[j] : 0.0; }
  };
  auto rowStep = [&](int i, int iNext, bool first, bool zeroRow, bool& moved) {
    const int owner = NTT == 1 ? i : i % nl, slot = NTT == 1 ? 0 : i / nl;
    double cc[NTT];
#pragma unroll
    for (int t = 0; t < NTT; t++) cc[t] = cn[t];
    fetch(iNext);
    double d = 0.0;
    if (ln == owner) {
      double xo = 0.0, r = 0.0, hi = 0.0, lo = 0.0; int fi = -1;
#pragma unroll
      for (int t = 0; t < NTT; t++) if (t == slot) { xo = xx[t]; r = rr[t]; hi = hh[t]; lo = ll[t]; fi = ff[t]; }
      double nx;
      d = genPgsClamp(xs, xo, r, hi, lo, fi, first, zeroRow, nx, moved);
#pragma unroll
      for (int t = 0; t < NTT; t++) if (t == slot) xx[t] = nx;
      xs[i] = nx;
    }
    d = w.bcast(d, owner);
#pragma unroll
    for (int t = 0; t < NTT; t++) rr[t] = fma(-cc[t], d, rr[t]);
    w.fence();
  };
  bool moved = false;
  fetch(n > 0 ? 0 : -1);
  for (int i = 0; i < n; ++i) rowStep(i, i + 1 < n ? i + 1 : -1, true, R.in[i] != 0, moved);
  bool possible = !w.anyAll(moved);
  if (!possible) {
    for (int iter = 1; iter < maxIteration; ++iter) {
      moved = false;
      fetch(no > 0 ? order[0] : -1);
      int iCur = no > 0 ? order[0] : -1;
      for (int t = 0; t < no; t++) {
        const int iNext = t + 1 < no ? order[t + 1] : -1;
        rowStep(iCur, iNext, false, false, moved);
        iCur = iNext;
      }
      possible = !w.anyAll(moved);
      if (possible) break;
    }
  }
  return possible;
}

// ... and for a problem of at most as many rows as the wave has lanes (one row per lane: the usual case) NOTHING of a step goes through
// shared memory but the column of AT: the lane of a friction row takes the impulse of its normal row from a second broadcast (xf), lane t holds the t-th row of the sweep order and hands the next row on through the broadcast primitive.
// AT_LDS: AT is in LDS (else HBM scratch) - the device reads it through a pointer of that address space: a generic (flat) load counts
// on both memory counters and every wait for one waits for all of them, the prefetch included.
template <bool AT_LDS, class W>
DEV bool genPgsHeld1(const W& w, GenRows& R, int n, int no, const double* AT, int atLd) {
  const int maxIteration = 30;
  const double dxTh = 1e-6, relTol = 1e-3, epsDiv = 1e-9;
  const int ln = w.lane();
  const bool in = ln < n;
  double r = in ? R.t1[ln] : 0.0, x = in ? R.t0[ln] : 0.0;
  const double lo = in ? R.t2[ln] : 0.0, hi = in ? R.t3[ln] : 0.0;
  const int fi = in ? R.cls[ln] : -1;
  double xf = fi >= 0 ? R.t0[fi] : 1.0;                       // the impulse of the normal row a friction row hangs on
  const int myOrder = ln < no ? R.perm[ln] : 0;               // lane t: the t-th row of the later sweeps
  const bool zeroMe = in && R.in[ln] != 0;
  const bool identity = no == n;                              // nothing left out: the later sweeps visit 0 .. n-1 like the first one
  // What a step costs on a wavefront that is alone on its SIMD is its instruction count (tools/dbg/lat_probe.hip: ~5.5 cycles each,
  // whatever they are), so the loop is written for few of them: the entry a'_.i of a step is an UNCONDITIONAL load (lanes beyond the
  // problem read their clamped neighbour's, rows beyond the sweep row 0's: never used), fetched four steps ahead (a generic-address
  // load: AT may be LDS or HBM scratch).
  const double* colBase = AT + (in ? ln : 0);
  const unsigned ldu = (unsigned)atLd;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const double __attribute__((address_space(3)))* LdsPtr;
  typedef const double __attribute__((address_space(1)))* GlobalPtr;
  auto colOf = [&](int i) -> double {
    const unsigned off = (unsigned)(i < 0 ? 0 : i) * ldu;
    if constexpr (AT_LDS) return ((LdsPtr)colBase)[off];
    else return ((GlobalPtr)colBase)[off];
  };
#else
  auto colOf = [&](int i) -> double { return colBase[(unsigned)(i < 0 ? 0 : i) * ldu]; };
#endif
  // A step WITHOUT a branch: every lane forms the clamped candidate of ITS row (the same instructions whether one lane runs them or all
  // do), the owner's is taken with a select, the change and the new x_i come back through two broadcasts.  The bounds of a friction row
  // (h = mu x_normal, l = -h) are registers, refreshed by the lanes that hang on row i.  The convergence test of a row needs only what
  // its lane holds after its step (its last change and its x): it runs ONCE per sweep, for all rows at once (sweepMoved).
  double h = hi, l = lo;
  if (fi >= 0) { h = hi * xf; l = -h; }
  // ... and the change of a row is x after the sweep minus x before it (a row is visited once per sweep: the same subtraction); which
  // rows have friction rows hanging on them is a wave-uniform bit mask, so that the steps of all the others skip the refresh of the bounds.
  unsigned long long followed = 0ull;
  for (int i = 0; i < n; i++) if (w.anyAll(fi == i)) followed |= 1ull << i;
  double xStart = x;
  auto rowStep = [&](int i, double cc, bool first) {
    double nx = x + r;
    nx = nx > h ? h : (nx < l ? l : nx);
    if (first) nx = zeroMe ? 0.0 : nx;        // (rows left out of the problem: set to zero in the first sweep, never visited again)
    const double dAll = nx - x;
    x = ln == i ? nx : x;
    const double d = w.bcast(dAll, i);
    r = fma(-cc, d, r);
    if ((followed >> i) & 1ull) {
      const double xi = w.bcast(nx, i);        // (the new x_i itself: x_old + (x_new - x_old) is not always x_new)
      const double hn = hi * xi;
      const bool hangs = fi == i;
      h = hangs ? hn : h;
      l = hangs ? -hn : l;
    }
  };
  auto sweepMoved = [&](bool first) -> bool {
    const double myD = x - xStart;
    xStart = x;
    const bool big = first ? fabs(myD) > dxTh : (fabs(x) > epsDiv && fabs(myD) > relTol * fabs(x));
    return w.anyAll(in && !zeroMe && big);
  };
  // a sweep over `cnt` rows, four rows per trip; ORDERED: the t-th row is lane t's myOrder, else t itself
  auto sweep = [&](int cnt, bool first, bool ordered) {
    auto rowAt = [&](int t) -> int { return t < cnt ? (ordered ? w.bcastI(myOrder, t) : t) : -1; };
    int i0 = rowAt(0), i1 = rowAt(1), i2 = rowAt(2), i3 = rowAt(3);
    double c0 = colOf(i0), c1 = colOf(i1), c2 = colOf(i2), c3 = colOf(i3);
    for (int t0 = 0; t0 < cnt; t0 += 4) {
      const int j0 = rowAt(t0 + 4), j1 = rowAt(t0 + 5), j2 = rowAt(t0 + 6), j3 = rowAt(t0 + 7);
      const double n0 = colOf(j0), n1 = colOf(j1), n2 = colOf(j2), n3 = colOf(j3);
      rowStep(i0, c0, first);
      if (i1 >= 0) rowStep(i1, c1, first);
      if (i2 >= 0) rowStep(i2, c2, first);
      if (i3 >= 0) rowStep(i3, c3, first);
      i0 = j0; i1 = j1; i2 = j2; i3 = j3; c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
  };
  sweep(n, true, false);
  bool possible = !sweepMoved(true);
  if (!possible) {
    for (int iter = 1; iter < maxIteration; ++iter) {
      sweep(no, false, !identity);
      possible = !sweepMoved(false);
      if (possible) break;
    }
  }
  if (in) R.t0[ln] = x;
  w.fence();
  return possible;
}

// PgsBoxedLcpSolver::solve (PgsBoxedLcpSolver.cpp:79-268), Option(30, 1e-6, 1e-3, 1e-9, false).  Gauss-Seidel is sequential over the
// rows: up to 30 sweeps x n row steps are one dependent chain and its length is the cost.  RESIDUAL form, like the lane = row builds
// (coopPgs, coop_dantzig_dev.hpp): every row j keeps the scaled residual r_j = b'_j - sum_i a'_ji x_i of ITS row (a'_j = row j of A over
// a_jj, as the reference scales the rows after its first sweep; b'_j = b_j / a_jj).  The step of row i is  x_i <- clamp(x_i + r_i)  by
// the lane that owns row i, ONE broadcast of the change d, and r_j -= a'_ji d for every row j (lane-strided; the entries a'_.i of a
// step are one contiguous row of AT, the scaled matrix transposed, built once per solve in the scratch matrix the caller lends).  x and
// r live in the rows' scratch vectors (LDS on the device).  Round 5 formed the reference's 23-term dot product per row step with the
// lanes sharing the sum - two barriers and a tree reduction per step, x and A in HBM scratch: 1500 cycles per step, 2.6 M cycles per
// world that reaches the fallback stages.  The two orders agree to round-off; the clamps, the convergence tests (the reference's
// division form) and the iteration cap are the reference's.  Rows with a_ii < eps are set to 0 once and left alone.  Uniform result.
template <class W>
DEV bool genPgs(const W& w, GenRows& R, GenProblem& P, double* AT, int atLd, bool atLds = false) {
  GEN_T0();
  const int ld = R.ld;              // leading dimension of the scratch matrices (the model's rows, rounded up); atLd: that of AT
  const int n = P.n;
  const int maxIteration = 30;
  const double dxTh = 1e-6, relTol = 1e-3, epsDiv = 1e-9;
  const int ln = w.lane(), nl = w.lanes();
  double* xs = R.t0;                // x
  double* rs = R.t1;                // scaled residuals
  double* los = R.t2;               // lower bound (or, friction rows: unused) / upper bound (friction rows: mu)
  double* his = R.t3;
  double* inv = R.E;                // 1 / a_jj (1 for the rows left out)
  int* fidx = R.cls;                // findex (the classes are not in use between the stages)
  int* order = R.perm;
  for (int j = ln; j < n; j += nl) {
    const double ajj = P.A[(size_t)j * ld + j];
    inv[j] = ajj < epsDiv ? 1.0 : 1.0 / ajj;
    R.in[j] = ajj < epsDiv ? 1 : 0;                     // (left out of the problem)
    xs[j] = P.x[j]; los[j] = P.lo[j]; his[j] = P.hi[j]; fidx[j] = P.findex[j];
  }
  if (ln == 0) {
    int no = 0;
    for (int i = 0; i < n; i++) if (!(P.A[(size_t)i * ld + i] < epsDiv)) order[no++] = i;
    R.iscal[4] = no;
  }
  w.sync();
  const int no = R.iscal[4];
  for (int idx = ln; idx < n * n; idx += nl) { const int i = idx / n, j = idx - i * n; AT[(size_t)i * atLd + j] = P.A[(size_t)j * ld + i] * inv[j]; }
  w.sync();
  for (int j = ln; j < n; j += nl) {
    double r0 = P.b[j] * inv[j], r1 = 0.0;
    int i = 0;
    for (; i + 1 < n; i += 2) { r0 = fma(-AT[(size_t)i * atLd + j], xs[i], r0); r1 = fma(-AT[(size_t)(i + 1) * atLd + j], xs[i + 1], r1); }
    if (i < n) r0 = fma(-AT[(size_t)i * atLd + j], xs[i], r0);
    rs[j] = r0 + r1;
  }
  w.sync();
  GEN_T(13);
  // When the wavefront covers the rows with at most NT rows per lane (always on the device), a lane HOLDS its rows in registers: genPgsHeld
  // (one row per lane when the problem has at most as many rows as the wave has lanes - the usual case).  Otherwise (the one-lane host
  // policy) everything is read in place.  Same arithmetic.
  constexpr int NT = (GR + 63) / 64;
  bool possible;
#if defined(__HIP_DEVICE_COMPILE__)
  // the address space the pointer IS in decides which typed load reads it, not the caller's word: a global load of an LDS address is a
  // memory aperture violation (seen on worlds of 20 contacts when the flag travelled through the inlined callers: tools/dbg/r06_tower.py)
  atLds = __builtin_amdgcn_is_shared(AT);
#endif
  if (n <= nl) possible = atLds ? genPgsHeld1<true>(w, R, n, no, AT, atLd) : genPgsHeld1<false>(w, R, n, no, AT, atLd);
  else if (nl * NT >= n) possible = genPgsHeld<NT>(w, R, n, no, AT, atLd);
  else {
    auto rowStep = [&](int i, bool first, bool zeroRow, bool& moved) {
      const int owner = i % nl;
      double d = 0.0;
      if (ln == owner) {
        double nx;
        d = genPgsClamp(xs, xs[i], rs[i], his[i], los[i], fidx[i], first, zeroRow, nx, moved);
        xs[i] = nx;
      }
      d = w.bcast(d, owner);
      const double* col = AT + (size_t)i * atLd;
      for (int j = ln; j < n; j += nl) rs[j] = fma(-col[j], d, rs[j]);
      w.fence();
    };
    bool moved = false;
    for (int i = 0; i < n; ++i) rowStep(i, true, R.in[i] != 0, moved);
    possible = !w.anyAll(moved);
    if (!possible) {
      for (int iter = 1; iter < maxIteration; ++iter) {
        moved = false;
        for (int t = 0; t < no; t++) rowStep(order[t], false, false, moved);
        possible = !w.anyAll(moved);
        if (possible) break;
      }
    }
  }
  w.sync();
  GEN_T(14);
  for (int j = ln; j < n; j += nl) P.x[j] = xs[j];
  w.sync();
  return possible;
}

}  // namespace NBL_NS
