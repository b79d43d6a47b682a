// tests/host_shim/lane_lcp_statement.hpp — TEST INFRASTRUCTURE.  The one-world-per-lane STATEMENT of stage 0 of the LCP solver
// cascade (guessSolution, the CGGM classification / standardisation loop, isLCPSolutionValid, a column-pivoted Householder COD):
// plain sequential code, the way the reference states the algorithm.  The product's wavefront-cooperative code (coop_dev.hpp)
// is unit-tested against it on the host (tests/test_coop_host.py).  It was the product's slow-path code in round 1; the kernels
// that ran it are gone.
#pragma once
#include "lcp_dev.hpp"

namespace NBL_NS {

// ---- factorisation of a c x c matrix stored row-major with leading dimension ld at offset off ----
struct CodFactor {
  int c, rank, ld, offQR, offChol;  // offChol: rank x rank Cholesky factor of R1 R1^T (ld = MAXR)
  int perm[MAXR];
  double tau[MAXR];
};

DEV void codFactor(const LaneMem& w, CodFactor& f) {
  const int c = f.c, ld = f.ld, o = f.offQR;
  for (int j = 0; j < c; j++) f.perm[j] = j;
  double maxPivot = 0.0;
  double diag[MAXR], cn[MAXR], cn0[MAXR];
  // squared column norms, downdated as the factorisation proceeds (recomputed when cancellation bites)
  for (int j = 0; j < c; j++) {
    double s = 0;
    for (int i = 0; i < c; i++) { double a = w.at(o + i * ld + j); s += a * a; }
    cn[j] = s; cn0[j] = s;
  }
  for (int k = 0; k < c; k++) {
    int piv = k;
    double best = -1.0;
    for (int j = k; j < c; j++) {
      if (cn[j] < 1e-8 * cn0[j] || cn[j] < 0) {
        double s = 0;
        for (int i = k; i < c; i++) { double a = w.at(o + i * ld + j); s += a * a; }
        cn[j] = s; cn0[j] = s;
      }
      if (cn[j] > best) { best = cn[j]; piv = j; }
    }
    if (piv != k) {
      for (int i = 0; i < c; i++) { double t = w.at(o + i * ld + k); w.at(o + i * ld + k) = w.at(o + i * ld + piv); w.at(o + i * ld + piv) = t; }
      int t = f.perm[k]; f.perm[k] = f.perm[piv]; f.perm[piv] = t;
      double tn = cn[k]; cn[k] = cn[piv]; cn[piv] = tn;
      tn = cn0[k]; cn0[k] = cn0[piv]; cn0[piv] = tn;
    }
    // exact norm of the pivot column below the diagonal
    double akk = w.at(o + k * ld + k);
    double below = 0;
    for (int i = k + 1; i < c; i++) { double a = w.at(o + i * ld + k); below += a * a; }
    double normx = sqrt(akk * akk + below);
    if (normx == 0.0) { f.tau[k] = 0; diag[k] = 0; continue; }
    double alpha = akk > 0 ? -normx : normx;
    // v = x - alpha e_k, stored scaled so that v_k = 1
    double vk = akk - alpha;
    double vnorm2 = vk * vk + below;
    f.tau[k] = 2.0 * vk * vk / vnorm2;  // H = I - tau v v^T with v_k = 1
    double inv = 1.0 / vk;
    for (int i = k + 1; i < c; i++) w.at(o + i * ld + k) *= inv;
    w.at(o + k * ld + k) = alpha;
    for (int j = k + 1; j < c; j++) {
      double d = w.at(o + k * ld + j);
      for (int i = k + 1; i < c; i++) d += w.at(o + i * ld + k) * w.at(o + i * ld + j);
      d *= f.tau[k];
      double rkj = w.at(o + k * ld + j) - d;
      w.at(o + k * ld + j) = rkj;
      for (int i = k + 1; i < c; i++) w.at(o + i * ld + j) -= d * w.at(o + i * ld + k);
      cn[j] -= rkj * rkj;
    }
    diag[k] = alpha;
    maxPivot = fmax(maxPivot, fabs(alpha));
  }
  const double thresh = 2.220446049250313e-16 * c * maxPivot;
  int r = 0;
  for (int k = 0; k < c; k++) { if (fabs(diag[k]) > thresh) r++; else break; }
  f.rank = r;
  if (r < c && r > 0) {
    // Cholesky of R1 R1^T (r x r), lower factor at offChol with leading dimension MAXR
    const int oc = f.offChol;
    for (int i = 0; i < r; i++)
      for (int j = 0; j <= i; j++) {
        double s = 0;
        for (int k = i; k < c; k++) s += w.at(o + i * ld + k) * w.at(o + j * ld + k);  // R upper: row i starts at col i >= j
        for (int k = 0; k < j; k++) s -= w.at(oc + i * MAXR + k) * w.at(oc + j * MAXR + k);
        if (i == j) w.at(oc + i * MAXR + i) = sqrt(s > 0 ? s : 0.0);
        else w.at(oc + i * MAXR + j) = s / w.at(oc + j * MAXR + j);
      }
  }
}

// apply Q^T = H_{c-1} ... H_0 to a vector in place
DEV void codApplyQt(const LaneMem& w, const CodFactor& f, double* v) {
  const int c = f.c, ld = f.ld, o = f.offQR;
  for (int k = 0; k < c; k++) {
    if (f.tau[k] == 0) continue;
    double d = v[k];
    for (int i = k + 1; i < c; i++) d += w.at(o + i * ld + k) * v[i];
    d *= f.tau[k];
    v[k] -= d;
    for (int i = k + 1; i < c; i++) v[i] -= d * w.at(o + i * ld + k);
  }
}
DEV void codApplyQ(const LaneMem& w, const CodFactor& f, double* v) {
  const int c = f.c, ld = f.ld, o = f.offQR;
  for (int k = c - 1; k >= 0; k--) {
    if (f.tau[k] == 0) continue;
    double d = v[k];
    for (int i = k + 1; i < c; i++) d += w.at(o + i * ld + k) * v[i];
    d *= f.tau[k];
    v[k] -= d;
    for (int i = k + 1; i < c; i++) v[i] -= d * w.at(o + i * ld + k);
  }
}
DEV void cholSolve(const LaneMem& w, int oc, int r, double* z) {
  for (int i = 0; i < r; i++) { double s = z[i]; for (int k = 0; k < i; k++) s -= w.at(oc + i * MAXR + k) * z[k]; z[i] = s / w.at(oc + i * MAXR + i); }
  for (int i = r - 1; i >= 0; i--) { double s = z[i]; for (int k = i + 1; k < r; k++) s -= w.at(oc + k * MAXR + i) * z[k]; z[i] = s / w.at(oc + i * MAXR + i); }
}
// x = A^+ b   (minimum-norm least squares).  b is destroyed.
DEV void codSolve(const LaneMem& w, const CodFactor& f, double* b, double* x) {
  const int c = f.c, ld = f.ld, o = f.offQR, r = f.rank;
  for (int i = 0; i < c; i++) x[i] = 0;
  if (r == 0) return;
  codApplyQt(w, f, b);
  double y[MAXR];
  if (r == c) {
    for (int i = c - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < c; k++) s -= w.at(o + i * ld + k) * y[k]; y[i] = s / w.at(o + i * ld + i); }
  } else {
    cholSolve(w, f.offChol, r, b);  // z = (R1 R1^T)^-1 c1 in b[0..r)
    for (int k = 0; k < c; k++) { double s = 0; for (int i = 0; i < r && i <= k; i++) s += w.at(o + i * ld + k) * b[i]; y[k] = s; }
  }
  for (int k = 0; k < c; k++) x[f.perm[k]] = y[k];
}
// x = (A^+)^T y = (A^T)^+ y.  y is destroyed.
DEV void codSolveT(const LaneMem& w, const CodFactor& f, double* yv, double* x) {
  const int c = f.c, ld = f.ld, o = f.offQR, r = f.rank;
  for (int i = 0; i < c; i++) x[i] = 0;
  if (r == 0) return;
  double t[MAXR], u[MAXR];
  for (int k = 0; k < c; k++) t[k] = yv[f.perm[k]];
  if (r == c) {
    // R^T u = t  (forward substitution), out = Q [u]
    for (int i = 0; i < c; i++) { double s = t[i]; for (int k = 0; k < i; k++) s -= w.at(o + k * ld + i) * u[k]; u[i] = s / w.at(o + i * ld + i); }
  } else {
    for (int i = 0; i < r; i++) { double s = 0; for (int k = i; k < c; k++) s += w.at(o + i * ld + k) * t[k]; u[i] = s; }
    cholSolve(w, f.offChol, r, u);
    for (int i = r; i < c; i++) u[i] = 0;
  }
  for (int i = 0; i < c; i++) x[i] = u[i];
  codApplyQ(w, f, x);
}

// ---- LCP description: frictional contacts, 3 rows each: [normal, t1, t2]; lo/hi/findex implied ----
struct LcpView {
  LaneMem mem;
  int offA;   // m x m, leading dimension MAXR
  int m;
  double mu[MAXR / 3];
  DEV double A(int r, int c) const { return mem.at(offA + r * MAXR + c); }
  DEV int findex(int r) const { return (r % 3) == 0 ? -1 : r - (r % 3); }
  DEV double hi(int r) const { return (r % 3) == 0 ? INFINITY : mu[r / 3]; }
  DEV double lo(int r) const { return (r % 3) == 0 ? 0.0 : -mu[r / 3]; }
};

DEV bool lcpValid(const LcpView& L, const double* X, const double* Bv, bool ignoreFriction, double cfmDiag) {
  const double tol = 1e-5;
  for (int i = 0; i < L.m; i++) {
    double v = -Bv[i] + cfmDiag * X[i];
    for (int j = 0; j < L.m; j++) v += L.A(i, j) * X[j];
    double upper = L.hi(i), lower = L.lo(i);
    const int fi = L.findex(i);
    if (fi != -1) {
      if (ignoreFriction) { if (X[i] != 0) return false; continue; }
      upper *= X[fi];
      lower *= X[fi];
    }
    if (fabs(lower) < tol && fabs(upper) < tol && fabs(X[i]) < tol) {
    } else if (fabs(X[i] - lower) < tol) { if (v < -tol) return false; }
    else if (fabs(X[i] - upper) < tol) { if (v > tol) return false; }
    else if (X[i] > lower && X[i] < upper) { if (fabs(v) > tol) return false; }
    else return false;
  }
  return true;
}

struct Classes {
  int cls[MAXR], cidx[MAXR], uidx[MAXR];
  double E[MAXR];  // for upper-bound rows: the multiple (hi or lo) of their normal row
  int nc, nu;
};

// CGGM::constructMatrices classification (CGGM.cpp:535-713)
DEV void classify(const LcpView& L, const double* X, const double* colNorm, bool ignoreFriction, Classes& K) {
  const double TH = 1e-6;
  K.nc = 0; K.nu = 0;
  for (int j = 0; j < L.m; j++) {
    K.cls[j] = RC_NOT_CLAMPING; K.cidx[j] = -1; K.uidx[j] = -1; K.E[j] = 0;
    if (colNorm[j] < 1e-9) continue;
    const int fp = L.findex(j);
    double upper = L.hi(j), lower = L.lo(j);
    if (fp != -1) { upper *= X[fp]; lower *= X[fp]; }
    if (fabs(X[j]) < TH) {
      if (fp != -1 && !(fabs(X[fp]) < TH) && !ignoreFriction) { K.cls[j] = RC_CLAMPING; K.cidx[j] = K.nc++; }
      continue;
    }
    const double tie = 1e-5;
    if ((X[j] > lower + tie && X[j] < upper - tie) || (lower - X[j] > 1e-2 || X[j] - upper > 1e-2)) {
      K.cls[j] = RC_CLAMPING; K.cidx[j] = K.nc++;
    } else if (fp != -1 && fabs(X[fp]) > 1e-9 && colNorm[fp] > 1e-9 && K.cls[fp] == RC_CLAMPING) {
      K.cls[j] = RC_UPPER_BOUND; K.uidx[j] = K.nu++;
      const double ub = X[fp] * L.hi(j), lb = X[fp] * L.lo(j);
      K.E[j] = (fabs(X[j] - ub) < fabs(X[j] - lb)) ? L.hi(j) : L.lo(j);
    }
  }
}

// Q = clamping block of A (+ A[:,ub] E) + cfm I   written to offQ (ld = MAXR); bc = clamping entries of b
DEV void buildQ(const LcpView& L, const Classes& K, double cfm, const LaneMem& out, int offQ, const double* Bv, double* bc) {
  for (int r = 0; r < L.m; r++) {
    if (K.cls[r] != RC_CLAMPING) continue;
    bc[K.cidx[r]] = Bv[r];
    for (int c = 0; c < L.m; c++) {
      if (K.cls[c] != RC_CLAMPING) continue;
      double q = L.A(r, c);
      if (K.nu > 0 && (c % 3) == 0) {
        // upper-bound friction rows of this contact ride on its normal force: A_ub E
        for (int u = c + 1; u < c + 3 && u < L.m; u++)
          if (K.cls[u] == RC_UPPER_BOUND) q += K.E[u] * L.A(r, u);
      }
      if (r == c) q += cfm;
      out.at(offQ + K.cidx[r] * MAXR + K.cidx[c]) = q;
    }
  }
}

// ---- shared pieces of the stage-0 kernel and the cascade kernel ----
// CGGM::constructMatrices + opportunisticallyStandardizeResults as a loop (the reference recurses while normal
// rows drop out of the clamping set, CGGM.cpp:321-332).  On return K holds the last classification and X the
// last accepted solution; returns whether the results are standardised (valid least-squares solution).
DEV bool standardizeLoop(const LcpView& V, const LaneMem& L, CodFactor& F, double* X, const double* Bv, const double* colNorm,
                         double cfm, bool ignoreFriction, RowMask guessMask, Classes& K) {
  const int m = V.m;
  bool ok = false;
  for (int iter = 0; iter < MAXR + 1; iter++) {
    classify(V, X, colNorm, ignoreFriction, K);
    if (K.nc == 0) {
      double zero[MAXR];
      for (int r = 0; r < m; r++) zero[r] = 0;
      ok = lcpValid(V, zero, Bv, ignoreFriction, cfm);
      if (ok) for (int r = 0; r < m; r++) X[r] = 0;
      break;
    }
    double bc[MAXR], fc[MAXR], newX[MAXR], origFc[MAXR];
    RowMask clampMask = 0;
    for (int r = 0; r < m; r++) if (K.cls[r] == RC_CLAMPING) { origFc[K.cidx[r]] = X[r]; clampMask |= RM1 << r; }
    if (iter == 0 && K.nu == 0 && guessMask != 0 && clampMask == guessMask) {
      // the clamping set is exactly the guess's set: Q and b_c are the system just solved, f_c = that solution
      for (int i = 0; i < K.nc; i++) fc[i] = origFc[i];
    } else {
      buildQ(V, K, cfm, L, 0, Bv, bc);
      F.c = K.nc;
      codFactor(L, F);
      codSolve(L, F, bc, fc);
    }
    bool newlyNot = false;
    for (int i = 0; i < m; i++) {
      newX[i] = 0;
      if (K.cls[i] == RC_CLAMPING) {
        newX[i] = fc[K.cidx[i]];
        if (fabs(newX[i]) < 1e-6 && fabs(X[i]) > 1e-6 && (i % 3) == 0) newlyNot = true;
      } else if (K.cls[i] == RC_UPPER_BOUND) {
        const int fp = i - (i % 3);
        double om = origFc[K.cidx[fp]] / X[i];
        double clean = (fabs(om - V.hi(i)) < fabs(om - V.lo(i))) ? V.hi(i) : V.lo(i);
        newX[i] = fc[K.cidx[fp]] * clean;
      }
    }
    if (!lcpValid(V, newX, Bv, ignoreFriction, cfm)) { ok = false; break; }
    for (int i = 0; i < m; i++) X[i] = newX[i];
    ok = true;
    if (!newlyNot) break;
  }
  return ok;
}

// Stage 0 of the solver cascade for one world (BoxedLcpConstraintSolver.cpp:380-460): X is the warm start when
// haveCache, else LCPUtils::guessSolution (LCPUtils.cpp:86-140) is computed into it; X0 receives the pre-solve x
// (mXBackup, what the PGS fallback starts from); then the standardisation loop.  L: scratch for 2 x MAXR x MAXR doubles.
DEV bool laneStage0(const LcpView& V, const LaneMem& L, bool haveCache, double* X, double* X0, const double* Bv,
                    const double* colNorm, Classes& K) {
  const int m = V.m;
  CodFactor F;
  F.ld = MAXR; F.offQR = 0; F.offChol = MAXR * MAXR;
  RowMask guessMask = 0;   // rows of the guess's clamping set; its factorisation can be reused by the first standardisation
  if (!haveCache) {
    int idx[MAXR], nc = 0;
    for (int r = 0; r < m; r++) if ((r % 3) != 0 || Bv[r] > 0) { idx[nc++] = r; guessMask |= RM1 << r; }
    for (int r = 0; r < m; r++) X[r] = 0;
    if (nc > 0) {
      double rhs[MAXR], sol[MAXR];
      for (int i = 0; i < nc; i++) { rhs[i] = Bv[idx[i]]; for (int j = 0; j < nc; j++) L.at(i * MAXR + j) = V.A(idx[i], idx[j]); }
      F.c = nc;
      codFactor(L, F);
      codSolve(L, F, rhs, sol);
      for (int i = 0; i < nc; i++) X[idx[i]] = sol[i];
    }
  }
  for (int r = 0; r < m; r++) X0[r] = X[r];
  return standardizeLoop(V, L, F, X, Bv, colNorm, 0.0, false, guessMask, K);
}

}  // namespace NBL_NS
