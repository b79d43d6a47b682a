// coop_dantzig_dev.hpp — stages 1-3 of the LCP solver cascade with ONE WORLD PER WAVEFRONT (lane = LCP row).
//
// k_contact_cascade runs these stages one world per lane; its single dependent instruction stream per world (a fresh
// LDL^T of A(C,C) per pivot, index vectors in scratch) costs 4-7 ms as soon as ONE world of a batch leaves stage 0.
// Here the wavefront of a failed world shares the work: lane k owns position k of the Dantzig driver's permuted problem
// (x, w, b, lo, hi, dx, dw, state, findex, p in registers), the permuted symmetric matrix and the LDL^T factor live in
// LDS, factorisation / triangular solves / products are lane-parallel, step-length events are found by a wave arg-min
// with the reference's scan order as the tie-break.  Same mathematics and event order as dantzig_dev.hpp (which restates
// dart/external/odelcpsolver/lcp.cpp:780-1113 and is pinned against the reference's own dSolveLCP on the host).
#pragma once
#include "coop_dev.hpp"

namespace nbl {

struct CascadeLds {
  double A[MAXR * CLD];    // reduced problem; for Dantzig: symmetrised from the lower triangle, rows/columns in driver order
  double L[MAXR * CLD];    // LDL^T of A(C,C): unit lower factor below the diagonal, D on it
  double v[4][MAXR];       // broadcast vectors
  int iv[2][MAXR];
};

// one problem row per lane (lanes >= n idle); findex refers to reduced row indices
struct CoopLcpRow {
  double x, b, lo, hi;
  int findex;
};

// The Dantzig driver.  In: the reduced problem (n rows) with its matrix in C.A (only the lower triangle is meaningful,
// lcp.cpp:138-140) and one row per lane in `row`.  Out: row.x = solution in the ORIGINAL reduced order; returns false on
// early termination (s <= 0), like dantzigSolve.  Returns 1 (solved), 0 (early termination) or -1 (a NaN step length:
// the one-world-per-lane code would carry the NaN into x and its caller would reset x and flag the world).
template <class W>
DEV int coopDantzig(const W& w, CascadeLds& C, int n, CoopLcpRow& row) {
  const int ln = w.lane();
  const bool on = ln < n;
  const int me = on ? ln : 0;
  // ---- symmetrise in place: A[u][v] (u < v) <- A[v][u]; lane = column v ----
  if (on) for (int u = 0; u < n; u++) if (u < ln) C.A[u * CLD + ln] = C.A[ln * CLD + u];
  w.sync();
  double x = 0.0, ww = 0.0, b = on ? row.b : 0.0, lo = on ? row.lo : 0.0, hi = on ? row.hi : 0.0, dx = 0.0, dw = 0.0;
  int st = 0, fidx = on ? row.findex : -1, p = me;
  auto swapProblem = [&](int i1, int i2) {   // uniform arguments
    if (i1 == i2) return;
    // rows, then columns of the permuted matrix
    if (on) { const double t = C.A[i1 * CLD + ln]; C.A[i1 * CLD + ln] = C.A[i2 * CLD + ln]; C.A[i2 * CLD + ln] = t; }
    w.sync();
    if (on) { const double t = C.A[ln * CLD + i1]; C.A[ln * CLD + i1] = C.A[ln * CLD + i2]; C.A[ln * CLD + i2] = t; }
    w.sync();
    const int src = ln == i1 ? i2 : (ln == i2 ? i1 : ln);
    x = w.shfl(x, src); b = w.shfl(b, src); ww = w.shfl(ww, src); lo = w.shfl(lo, src); hi = w.shfl(hi, src);
    p = w.shflI(p, src); st = w.shflI(st, src); fidx = w.shflI(fidx, src);
  };
  int nC = 0, nN = 0;
  // contact problems have no unbounded rows (nub = 0); every findex row goes to the end (lcp.cpp:487-498)
  {
    int atEnd = 0;
    for (int k = n - 1; k >= 0; k--) {
      const int fk = w.shflI(fidx, k);
      if (fk >= 0) { swapProblem(k, n - 1 - atEnd); atEnd++; }
    }
  }
  // dx(C) = -dir * A(C,C)^-1 A(C,i) via a fresh LDL^T (no pivoting, like dFactorLDLT), all lane-parallel
  auto solve1 = [&](int i, int dir) {
    if (nC == 0) return;
    const bool inC = ln < nC;
    if (inC) for (int j = 0; j < nC; j++) if (j <= ln) C.L[ln * CLD + j] = C.A[ln * CLD + j];
    w.sync();
    for (int k = 0; k < nC; k++) {
      const double dk = C.L[k * CLD + k];
      const double aik = (inC && ln > k) ? C.L[ln * CLD + k] : 0.0;
      const double lik = aik / dk;
      if (inC && ln > k) for (int j = k + 1; j < nC; j++) if (j <= ln) C.L[ln * CLD + j] -= lik * C.L[j * CLD + k];
      w.sync();
      if (inC && ln > k) C.L[ln * CLD + k] = lik;
      w.sync();
    }
    // L y = A(C, i);  y /= D;  L^T z = y
    double y = inC ? C.A[ln * CLD + i] : 0.0;
    for (int k = 0; k < nC; k++) {
      const double yk = w.shfl(y, k);
      if (inC && ln > k) y -= C.L[ln * CLD + k] * yk;
    }
    if (inC) y /= C.L[ln * CLD + ln];
    for (int k = nC - 1; k >= 0; k--) {
      const double zk = w.shfl(y, k);
      if (inC && ln < k) y -= C.L[k * CLD + ln] * zk;
    }
    dx = inC ? (dir > 0 ? -y : y) : dx;
  };
  bool hitFirstFriction = false;
  for (int i = 0; i < n; ++i) {
    const int fi = w.shflI(fidx, i);
    if (!hitFirstFriction && fi >= 0) {
      // un[p[j]] = x[j]; bounds of the friction rows frozen from the solved normals
      if (on) C.v[0][p] = x;
      w.sync();
      if (on && ln >= i) {
        const double wfk = C.v[0][fidx >= 0 ? fidx : 0];
        if (wfk == 0) { hi = 0; lo = 0; }
        else { hi = fabs(hi * wfk); lo = -hi; }
      }
      w.sync();
      hitFirstFriction = true;
    }
    // w[i] = A(i, C+N) x - b[i]
    if (on) C.v[0][ln] = x;
    w.sync();
    {
      double s = -b;
      for (int j = 0; j < nC + nN; j++) s += C.A[me * CLD + j] * C.v[0][j];
      if (ln == i) ww = s;
    }
    w.sync();
    const double wi0 = w.shfl(ww, i), loi = w.shfl(lo, i), hii = w.shfl(hi, i);
    if (loi == 0 && wi0 >= 0) { if (ln == i) st = 0; nN++; }
    else if (hii == 0 && wi0 <= 0) { if (ln == i) st = 1; nN++; }
    else if (wi0 == 0) { swapProblem(nC, i); nC++; }
    else {
      for (;;) {
        const double wi = w.shfl(ww, i);
        const int dir = (wi <= 0) ? 1 : -1;
        const double dirf = dir;
        solve1(i, dir);
        // dw(N) = A(N,C) dx(C) + dir A(N,i);  dw[i] likewise
        if (on) C.v[1][ln] = dx;
        w.sync();
        {
          double s = 0;
          for (int j = 0; j < nC; j++) s += C.A[me * CLD + j] * C.v[1][j];
          const bool inN = ln >= nC && ln < nC + nN;
          if (inN || ln == i) dw = s + dirf * C.A[me * CLD + i];
        }
        w.sync();
        // step length: first minimum in the reference's scan order (i's own events, N rows, C rows)
        double s = INFINITY;
        int cmd = 0, order = 1 << 20;
        if (ln == i) {
          s = -ww / dw; cmd = 1; order = 0;
          if (dir > 0) { if (hi < INFINITY) { const double s2 = (hi - x) * dirf; if (s2 < s) { s = s2; cmd = 3; } } }
          else { if (lo > -INFINITY) { const double s2 = (lo - x) * dirf; if (s2 < s) { s = s2; cmd = 2; } } }
        } else if (ln >= nC && ln < nC + nN) {
          if ((st == 0) ? dw < 0 : dw > 0) {
            if (!(lo == 0 && hi == 0)) { s = -ww / dw; cmd = 4; order = 1 + (ln - nC); }
          }
        } else if (ln < nC) {
          if (dx < 0 && lo > -INFINITY) { s = (lo - x) / dx; cmd = 5; order = 100 + 2 * ln; }
          if (dx > 0 && hi < INFINITY) { s = (hi - x) / dx; cmd = 6; order = 101 + 2 * ln; }
        }
        // the reference keeps a candidate only if it is STRICTLY smaller than the running minimum, which starts at lane
        // i's value: arg-min over (s, order)
        const double sOwn = w.shfl(s, i);
        if (sOwn != sOwn) { row.x = 0.0; return -1; }
        const double sMin = -w.maxAll(cmd != 0 ? -s : -INFINITY);
        const uint64_t tie = w.ballot(cmd != 0 && s == sMin);
        if (tie == 0ull) { row.x = 0.0; return -1; }
        // smallest order among the ties: lane i (order 0) < N lanes by position < C lanes by position
        int best;
        {
          const uint64_t mi = tie & (1ull << i);
          const uint64_t maskN = nN > 0 ? (((1ull << nN) - 1ull) << nC) : 0ull;
          const uint64_t maskC = nC > 0 ? ((1ull << nC) - 1ull) : 0ull;
          if (mi) best = i;
          else if (tie & maskN) best = __builtin_ctzll(tie & maskN);
          else best = __builtin_ctzll((tie & maskC) ? (tie & maskC) : tie);
        }
        const int cmdB = w.shflI(cmd, best);
        if (sMin <= 0.0) { row.x = 0.0; return 0; }   // earlyTermination (the caller always has the PGS fallback, BoxedLcpConstraintSolver.cpp:463)
        const int si = best;
        // apply the step
        if (ln < nC) x += sMin * dx;
        if (ln == i) x += sMin * dirf;
        if (ln >= nC && ln < nC + nN) ww += sMin * dw;
        if (ln == i) ww += sMin * dw;
        switch (cmdB) {
          case 1: if (ln == i) ww = 0; swapProblem(nC, i); nC++; break;
          case 2: if (ln == i) { x = lo; st = 0; } nN++; break;
          case 3: if (ln == i) { x = hi; st = 1; } nN++; break;
          case 4: if (ln == si) ww = 0; swapProblem(nC, si); nN--; nC++; break;
          case 5: if (ln == si) { x = lo; st = 0; } swapProblem(si, nC - 1); nN++; nC--; break;
          case 6: if (ln == si) { x = hi; st = 1; } swapProblem(si, nC - 1); nN++; nC--; break;
        }
        if (cmdB <= 3) break;
      }
    }
  }
  // back to the original (reduced) order: P.x[p[j]] = x[j]
  if (on) C.v[0][p] = x;
  w.sync();
  row.x = on ? C.v[0][ln] : 0.0;
  w.sync();
  return 1;
}

// ---- LCPUtils::reduce / removeFriction (LCPUtils.cpp:144-247, 346-520), lane = reduced row / column ----
// delete row + column `col` of the n x n problem; lanes >= col take the row data of their right neighbour
template <class W>
DEV void coopRemoveRow(const W& w, CascadeLds& C, int n, int col, CoopLcpRow& row) {
  const int ln = w.lane();
  if (ln < n) for (int j = col; j + 1 < n; j++) C.A[ln * CLD + j] = C.A[ln * CLD + j + 1];     // columns left, lane = row
  w.sync();
  if (ln < n) for (int i = col; i + 1 < n; i++) C.A[i * CLD + ln] = C.A[(i + 1) * CLD + ln];   // rows up, lane = column
  w.sync();
  const double x1 = w.shfl(row.x, ln + 1), b1 = w.shfl(row.b, ln + 1), l1 = w.shfl(row.lo, ln + 1), h1 = w.shfl(row.hi, ln + 1);
  const int f1 = w.shflI(row.findex, ln + 1);
  if (ln >= col && ln + 1 < n) { row.x = x1; row.b = b1; row.lo = l1; row.hi = h1; row.findex = f1; }
}

// merge near-identical columns (squared distance < 1e-4, |b_a - b_b| < 1e-4, same findex / hi / lo).  mapTo: this lane's
// ORIGINAL row -> reduced column.  Returns the reduced size.
template <class W>
DEV int coopLcpReduce(const W& w, CascadeLds& C, int n, CoopLcpRow& row, int& mapTo) {
  const double TH = 1e-4;
  const int ln = w.lane();
  for (;;) {
    int ma = -1, mb = -1;
    for (int a = 0; a < n - 1; a++) {
      double d2 = 0;
      const int bcol = ln < n ? ln : 0;
      for (int r = 0; r < n; r++) { const double d = C.A[r * CLD + a] - C.A[r * CLD + bcol]; d2 += d * d; }
      const double ba = w.shfl(row.b, a), ha = w.shfl(row.hi, a), la = w.shfl(row.lo, a);
      const int fa = w.shflI(row.findex, a);
      const bool match = ln > a && ln < n && d2 < TH && fabs(ba - row.b) < TH && fa == row.findex && ha == row.hi && la == row.lo;
      const uint64_t mm = w.ballot(match);
      if (mm) { ma = a; mb = __builtin_ctzll(mm); break; }
    }
    if (ma < 0) break;
    // mergeLCPColumns(colA = ma, colB = mb): column A doubled, row/column B deleted, findex remapped
    if (ln < n) C.A[ln * CLD + ma] *= 2.0;
    if (row.findex == mb) row.findex = ma;
    else if (row.findex > mb) row.findex -= 1;
    w.sync();
    coopRemoveRow(w, C, n, mb, row);
    n -= 1;
    if (mapTo == mb) mapTo = ma;
    else if (mapTo > mb) mapTo -= 1;
  }
  return n;
}

// drop every friction row (from the last one down)
template <class W>
DEV int coopLcpRemoveFriction(const W& w, CascadeLds& C, int n, CoopLcpRow& row, int& mapTo) {
  for (int i = n - 1; i >= 0; i--) {
    const int fi = w.shflI(row.findex, i);
    if (fi == -1) continue;
    if (row.findex > i) row.findex -= 1;
    coopRemoveRow(w, C, n, i, row);
    n -= 1;
    if (mapTo == i) mapTo = -1;
    else if (mapTo > i) mapTo -= 1;
  }
  return n;
}

// ---- PgsBoxedLcpSolver::solve (PgsBoxedLcpSolver.cpp:79-268), Option(30, 1e-6, 1e-3, 1e-9, false) ----
// Gauss-Seidel is sequential over the rows; the row whose turn it is works (same summation order as pgsSolve), x lives in
// LDS.  A is modified (rows normalised), like the reference.  row.x in: start, out: result.
template <class W>
DEV bool coopPgs(const W& w, CascadeLds& C, int n, CoopLcpRow& row) {
  const int maxIteration = 30;
  const double dxTh = 1e-6, relTol = 1e-3, epsDiv = 1e-9;
  const int ln = w.lane();
  const bool on = ln < n;
  double* X = C.v[0];
  if (on) X[ln] = row.x;
  w.sync();
  const double aii = on ? C.A[ln * CLD + ln] : 1.0;
  const bool inOrder = on && !(aii < epsDiv);
  double bb = row.b;
  bool possible = true;
  for (int i = 0; i < n; ++i) {
    if (ln == i) {
      if (!inOrder) X[i] = 0.0;
      else {
        const double old_x = X[i];
        double new_x = bb;
        for (int j = 0; j < i; ++j) new_x -= C.A[i * CLD + j] * X[j];
        for (int j = i + 1; j < n; ++j) new_x -= C.A[i * CLD + j] * X[j];
        new_x /= aii;
        double xi;
        if (row.findex >= 0) {
          const double hi_tmp = row.hi * X[row.findex], lo_tmp = -hi_tmp;
          xi = new_x > hi_tmp ? hi_tmp : (new_x < lo_tmp ? lo_tmp : new_x);
        } else xi = new_x > row.hi ? row.hi : (new_x < row.lo ? row.lo : new_x);
        X[i] = xi;
        if (fabs(xi - old_x) > dxTh) possible = false;
      }
    }
    w.sync();
  }
  if (w.ballot(!possible) == 0ull) { row.x = on ? X[ln] : 0.0; w.sync(); return true; }
  if (inOrder) {
    const double dummy = 1.0 / aii;
    bb *= dummy;
    for (int j = 0; j < n; ++j) C.A[ln * CLD + j] *= dummy;
  }
  w.sync();
  bool done = false;
  for (int iter = 1; iter < maxIteration; ++iter) {
    possible = true;
    for (int i = 0; i < n; ++i) {
      if (ln == i && inOrder) {
        double new_x = bb;
        const double old_x = X[i];
        for (int j = 0; j < i; j++) new_x -= C.A[i * CLD + j] * X[j];
        for (int j = i + 1; j < n; j++) new_x -= C.A[i * CLD + j] * X[j];
        double xi;
        if (row.findex >= 0) {
          const double hi_tmp = row.hi * X[row.findex], lo_tmp = -hi_tmp;
          xi = new_x > hi_tmp ? hi_tmp : (new_x < lo_tmp ? lo_tmp : new_x);
        } else xi = new_x > row.hi ? row.hi : (new_x < row.lo ? row.lo : new_x);
        X[i] = xi;
        if (fabs(xi) > epsDiv && fabs((xi - old_x) / xi) > relTol) possible = false;
      }
      w.sync();
    }
    if (w.ballot(!possible) == 0ull) { done = true; break; }
  }
  row.x = on ? X[ln] : 0.0;
  w.sync();
  return done;
}

// ---- stages 1-3 of BoxedLcpConstraintSolver::solveLcp (:461-677) + registration / standardisation (:718-736) ----
struct CoopCascadeOut {
  double X;          // impulses of this lane's row
  CoopClasses K;
  double cfm;
  uint32_t st;       // NBL_ST_* bits to OR into the world's status
  bool pinvValid;
};

template <class W>
DEV void coopCascade(const W& w, CoopLds& S, CascadeLds& C, const CoopRow& R, double X0, double fallbackCfm, CoopCascadeOut& out) {
  const int ln = w.lane();
  const int m = R.m;
  CoopLcpRow row;
  int mapTo = -1;
  auto loadProblem = [&](double cfmDiag, double x0) {
    if (ln < m) for (int j = 0; j < m; j++) C.A[ln * CLD + j] = R.a(j) + (ln == j ? cfmDiag : 0.0);   // A is symmetric: row = column
    row.x = x0; row.b = R.Bv;
    row.lo = R.fric ? -R.mu : 0.0; row.hi = R.fric ? R.mu : INFINITY; row.findex = R.fric ? R.fp : -1;
    mapTo = ln < m ? ln : -1;
    w.sync();
  };
  auto mapped = [&](double xred, int nred) -> double {   // X[o] = x_reduced[mapTo[o]]
    w.sync();
    if (ln < nred) C.v[3][ln] = xred;
    w.sync();
    return (ln < m && mapTo >= 0) ? C.v[3][mapTo] : 0.0;
  };
  auto hasNan = [&](double x) -> bool { return w.ballot(ln < m && x != x) != 0ull; };
  uint32_t st = 0;
  bool success = false, ignoreFriction = false;
  double cfm = 0.0, X = X0;
  // ---- stage 1: reduce + Dantzig with early termination (:461-522) ----
  loadProblem(0.0, X0);
  int nr = coopLcpReduce(w, C, m, row, mapTo);
  const int rc = coopDantzig(w, C, nr, row);
  if (rc == 1) {
    X = mapped(row.x, nr);
    success = coopValid(w, S, R, X, false, 0.0, 1);
    if (success) st |= 0x4u;
  }
  if (rc < 0 || hasNan(X)) { success = false; X = 0.0; st |= 0x40u; }
  if (!success) {
    // ---- stage 2: CFM + PGS from the pre-solve x (:539-597) ----
    cfm = fallbackCfm;
    loadProblem(cfm, X0);
    nr = coopLcpReduce(w, C, m, row, mapTo);
    if (coopPgs(w, C, nr, row)) {
      X = mapped(row.x, nr);
      success = coopValid(w, S, R, X, false, cfm, 1);
      if (success) st |= 0x8u;
    }
  }
  if (!success) {
    // ---- stage 3: drop friction, PGS from zero (:606-677) ----
    ignoreFriction = true;
    loadProblem(cfm, X0);
    nr = coopLcpRemoveFriction(w, C, m, row, mapTo);
    row.x = 0.0;
    const bool ok3 = coopPgs(w, C, nr, row);
    X = mapped(row.x, nr);
    st |= 0x10u;
    if (!ok3) st |= 0x20u;
  }
  if (hasNan(X)) { X = 0.0; st |= 0x40u; }
  // ---- register the fresh solution, classify, standardise (:718-736) ----
  bool pinvValid = false;
  const bool std = coopStandardizeLoop(w, S, R, X, cfm, ignoreFriction, 0u, pinvValid, out.K);
  if (std) st |= 0x100u;
  out.X = X; out.cfm = cfm; out.st = st; out.pinvValid = std && pinvValid;
}

}  // namespace nbl
