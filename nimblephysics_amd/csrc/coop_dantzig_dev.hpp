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

template <int I> struct IntTag { static constexpr int value = I; };

struct CascadeLds {
  double A[MAXR * CLD];    // reduced problem; for Dantzig: symmetrised from the lower triangle, rows/columns in driver order
  double L[MAXR * CLD];    // LDL^T of a permutation of A(C,C): unit lower factor by rows (the reference's m_L), pivots in d[]
  double v[4][MAXR];       // broadcast vectors
  double d[MAXR];          // reciprocal pivots of the factor (the reference's m_d)
  int iv[2][MAXR];
};

// one problem row per lane (lanes >= n idle); findex refers to reduced row indices
struct CoopLcpRow {
  double x, b, lo, hi;
  int findex;
};

// The Dantzig driver: dSolveLCP (dart/external/odelcpsolver/lcp.cpp:780-1113) with nub = 0, earlyTermination = true, restated
// OPERATION BY OPERATION: the factor of A(C,C) is the reference's L / d (unit lower factor by rows, RECIPROCAL pivots) kept in
// the reference's own row order through the index vector C[] (lcp.cpp:100-108); a row entering C appends ell / Dell to it
// (transfer_i_to_C / transfer_i_from_N_to_C, lcp.cpp:520-600), a row leaving C down-dates it with dLDLTRemove -> dLDLTAddTL
// (lcp.cpp:603-650, matrix.cpp:286-426) - no refactorisation anywhere.  Every floating-point expression keeps the reference's
// association: the blocked accumulation order of dSolveL1 / dSolveL1T (fastlsolve.cpp, fastltsolve.cpp: a row of a full
// 4-block sums the columns left of its block into Z, x = b - Z, then subtracts the in-block terms one by one; tail rows sum
// everything into Z), dDot's running sum from 0 (fastdot.cpp), products and sums never fused (contraction is switched off for
// this function: the reference build has no FMA).  On identical inputs the result is therefore BIT-identical to the
// reference's, success flag included, also where A(C,C) is singular and the s <= 0 exit is decided by round-off
// (tests/test_coop_host.py pins it against oracle/_ref on the host emulation, tests/test_gpu_lcp_selftest.py on the GPU).
//
// Lane k owns POSITION k of the permuted problem (x, w, b, lo, hi, dx, dw, state, findex, p) and, for the factor, ROW k of
// L / entry k of d / C[k].  In: the reduced problem (n rows) with its matrix in C.A (only the lower triangle is meaningful,
// lcp.cpp:138-140) and one row per lane in `row`.  Out: row.x = solution in the ORIGINAL reduced order.  Returns 1 (solved),
// 0 (early termination, s <= 0) or -1 (a NaN step length of the driving row: the reference would carry the NaN into x and
// its caller resets x and flags the world, BoxedLcpConstraintSolver.cpp:500-520).
template <class W>
DEV int coopDantzig(const W& w, CascadeLds& C, int n, CoopLcpRow& row) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const int ln = w.lane();
  const bool on = ln < n;
  const int me = on ? ln : 0;
  // ---- symmetrise in place: A[u][v] (u < v) <- A[v][u]; lane = column v ----
  if (on) for (int u = 0; u < n; u++) if (u < ln) C.A[u * CLD + ln] = C.A[ln * CLD + u];
  w.sync();
  double x = 0.0, ww = 0.0, b = on ? row.b : 0.0, lo = on ? row.lo : 0.0, hi = on ? row.hi : 0.0, dx = 0.0, dw = 0.0;
  int st = 0, fidx = on ? row.findex : -1, p = me;
  int Cv = 0;                      // C[ln]: position of the problem row that factor row ln belongs to (lanes < nC)
  double ell = 0.0, Dell = 0.0;    // lanes < nC, factor order
  int nC = 0, nN = 0;
  auto swapProblem = [&](int i1, int i2) {   // uniform arguments
    if (i1 == i2) return;
    // rows, then columns of the permuted matrix
    if (on) { const double t = C.A[i1 * CLD + ln]; C.A[i1 * CLD + ln] = C.A[i2 * CLD + ln]; C.A[i2 * CLD + ln] = t; }
    w.sync();
    if (on) { const double t = C.A[ln * CLD + i1]; C.A[ln * CLD + i1] = C.A[ln * CLD + i2]; C.A[ln * CLD + i2] = t; }
    w.sync();
    // two-lane exchange with readlanes (i1, i2 uniform)
    auto xd = [&](double& v) { const double a = w.bcast(v, i1), c2 = w.bcast(v, i2); v = ln == i1 ? c2 : (ln == i2 ? a : v); };
    auto xi = [&](int& v) { const int a = w.bcastI(v, i1), c2 = w.bcastI(v, i2); v = ln == i1 ? c2 : (ln == i2 ? a : v); };
    xd(x); xd(b); xd(ww); xd(lo); xd(hi); xi(p); xi(st); xi(fidx);
  };
  // contact problems have no unbounded rows (nub = 0); every findex row goes to the end (lcp.cpp:487-498)
  {
    int atEnd = 0;
    for (int k = n - 1; k >= 0; k--) {
      const int fk = w.bcastI(fidx, k);
      if (fk >= 0) { swapProblem(k, n - 1 - atEnd); atEnd++; }
    }
  }
  // dDot over lanes [from, to): the running sum from 0 in lane order (fastdot.cpp); every lane gets the result
  auto seqSum = [&](double prod, int from, int to) -> double {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < MAXR; k++) { const double pk = w.bcast(prod, k); s = (k >= from && k < to) ? s + pk : s; }
    return s;
  };
  // dSolveL1 (fastlsolve.cpp): L y = rhs over the factor rows, lane = row
  auto solveL1 = [&](double rhs) -> double {
    const bool act = ln < nC;
    const int nb4 = nC & ~3;
    const int i0 = (ln < nb4) ? (ln & ~3) : ln;     // columns < i0 go into Z, the rest is subtracted term by term
    double lrow[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; k++) lrow[k] = C.L[me * CLD + k];
    double Z = 0.0, y = rhs;
#pragma unroll
    for (int k = 0; k < MAXR; k++) {
      if (k >= nC) break;
      if (k == i0) y = rhs - Z;
      const double xk = w.bcast(y, k);
      const double t = lrow[k] * xk;
      if (act && k < ln) { if (k < i0) Z = Z + t; else y = y - t; }
    }
    return y;
  };
  // dSolveL1T (fastltsolve.cpp): L^T y = rhs, the same blocking on the reversed index
  auto solveL1T = [&](double rhs) -> double {
    const bool act = ln < nC;
    const int jr = nC - 1 - ln;
    const int nb4 = nC & ~3;
    const int i0 = (jr < nb4) ? (jr & ~3) : jr;
    double Z = 0.0, y = rhs;
#pragma unroll 1
    for (int kr = 0; kr < nC; kr++) {
      const int k = nC - 1 - kr;
      if (kr == i0) y = rhs - Z;
      const double xk = w.bcast(y, k);
      const double t = C.L[k * CLD + me] * xk;
      if (act && kr < jr) { if (kr < i0) Z = Z + t; else y = y - t; }
    }
    return y;
  };
  // Dell = L^-1 A(i, C[.]), ell = Dell * d      (first half of dLCP::solve1, lcp.cpp:700-730)
  auto solveEll = [&](int i) {
    const double rhs = ln < nC ? C.A[i * CLD + Cv] : 0.0;
    Dell = solveL1(rhs);
    ell = ln < nC ? Dell * C.d[me] : 0.0;
  };
  auto solve1 = [&](int i, int dir) {
    if (nC == 0) return;
    solveEll(i);
    const double t = solveL1T(ell);
    w.sync();
    if (ln < nC) C.v[1][Cv] = dir > 0 ? -t : t;      // a[C[j]] = -/+ tmp[j]
    w.sync();
    dx = ln < nC ? C.v[1][me] : dx;
  };
  // the row at position i (with ell / Dell of the last solveEll(i)) becomes factor row nC   (transfer_i_to_C, lcp.cpp:520-553)
  auto appendFactorRow = [&](int i) {
    const double aii = C.A[i * CLD + i];
    if (nC > 0) {
      if (ln < nC) C.L[nC * CLD + ln] = ell;
      const double dot = seqSum(ell * Dell, 0, nC);
      if (ln == 0) C.d[nC] = 1.0 / (aii - dot);
    } else if (ln == 0) C.d[0] = 1.0 / aii;
    w.sync();
  };
  // dLDLTAddTL (matrix.cpp:286-359) on the trailing block [r, n2) of the factor with the vector a (lane = factor row)
  auto ldltAddTL = [&](int r, int n2, double a) {
    const int nn = n2 - r;
    if (nn < 2) return;
    const bool act = ln >= r && ln < n2;
    const int pl = ln - r;
    const double SQ = 0.70710678118654752440;   // M_SQRT1_2
    double W1 = (act && pl >= 1) ? a * SQ : 0.0, W2 = W1;
    const double a0 = w.bcast(a, r);
    const double W11 = (0.5 * a0 + 1.0) * SQ, W21 = (0.5 * a0 - 1.0) * SQ;
    double alpha1 = 1.0, alpha2 = 1.0;
    {
      double dee = C.d[r];
      double alphanew = alpha1 + (W11 * W11) * dee;
      dee /= alphanew;
      const double gamma1 = W11 * dee;
      dee *= alpha1;
      alpha1 = alphanew;
      alphanew = alpha2 - (W21 * W21) * dee;
      dee /= alphanew;
      alpha2 = alphanew;
      const double k1 = 1.0 - W21 * gamma1;
      const double k2 = W21 * gamma1 * W11 - W21;
      if (act && pl >= 1) {
        const double Wp = W1, el = C.L[me * CLD + r];
        W1 = Wp - W11 * el;
        W2 = k1 * Wp + k2 * el;
      }
    }
#pragma unroll 1
    for (int j = 1; j < nn; j++) {
      const double k1 = w.bcast(W1, r + j), k2 = w.bcast(W2, r + j);
      double dee = C.d[r + j];
      double alphanew = alpha1 + (k1 * k1) * dee;
      dee /= alphanew;
      const double gamma1 = k1 * dee;
      dee *= alpha1;
      alpha1 = alphanew;
      alphanew = alpha2 - (k2 * k2) * dee;
      dee /= alphanew;
      const double gamma2 = k2 * dee;
      dee *= alpha2;
      w.sync();
      if (ln == 0) C.d[r + j] = dee;
      alpha2 = alphanew;
      if (act && pl > j) {
        double el = C.L[me * CLD + r + j];
        double Wp = W1 - k1 * el;
        el += gamma1 * Wp;
        W1 = Wp;
        Wp = W2 - k2 * el;
        el -= gamma2 * Wp;
        W2 = Wp;
        C.L[me * CLD + r + j] = el;
      }
    }
    w.sync();
  };
  // dLDLTRemove (matrix.cpp:374-426): factor row / column r leaves the n2-row factor
  auto ldltRemove = [&](int n2, int r) {
    if (r == n2 - 1) return;    // deleting the last row / column is easy
    const bool act = ln >= r && ln < n2;
    const int Cr = w.bcastI(Cv, r);
    const double ga = C.A[(act ? Cv : 0) * CLD + Cr];    // GETA(p[r + i], p[r]) (the permuted matrix is kept symmetric)
    double a;
    if (r == 0) a = -ga;
    else {
      if (ln < r) C.v[2][ln] = C.L[r * CLD + ln] / C.d[ln];   // t[i] = L[r][i] / d[i]
      w.sync();
      double s = 0.0;
#pragma unroll 1
      for (int k = 0; k < r; k++) s = s + C.L[me * CLD + k] * C.v[2][k];    // dDot(L[r + i], t, r)
      a = s - ga;
    }
    if (ln == r) a += 1.0;
    ldltAddTL(r, n2, act ? a : 0.0);
    // dRemoveRowCol: snip row / column r out of L and d
    double lrow[MAXR];
    const int src = ln >= r ? ln + 1 : ln;
    const int srcc = src < MAXR ? src : MAXR - 1;
#pragma unroll
    for (int k = 0; k < MAXR; k++) lrow[k] = C.L[srcc * CLD + (k >= r ? (k + 1 < MAXR ? k + 1 : MAXR - 1) : k)];
    const double dsrc = C.d[srcc];
    w.sync();
    if (ln < n2 - 1) {
#pragma unroll
      for (int k = 0; k < MAXR; k++) C.L[ln * CLD + k] = lrow[k];
      C.d[ln] = dsrc;
    }
    w.sync();
  };
  // transfer_i_from_C_to_N (lcp.cpp:603-650): position i leaves C
  auto removeFromC = [&](int i) {
    const int j = __builtin_ctzll(w.ballot(ln < nC && Cv == i));
    ldltRemove(nC, j);
    const int k = __builtin_ctzll(w.ballot(ln < nC && Cv == nC - 1));
    if (ln == k) Cv = i;
    const int cNext = w.shflI(Cv, ln + 1);
    if (ln >= j) Cv = cNext;
    swapProblem(i, nC - 1);
    nN++; nC--;
  };
  bool hitFirstFriction = false;
  for (int i = 0; i < n; ++i) {
    const int fi = w.bcastI(fidx, i);
    if (!hitFirstFriction && fi >= 0) {
      // un[p[j]] = x[j]; bounds of the friction rows frozen from the solved normals (lcp.cpp:856-873)
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
    // w[i] = A(i,C) x(C) + A(i,N) x(N) - b[i]: two running sums (lcp.cpp:877)
    {
      const double pr = on ? C.A[i * CLD + me] * x : 0.0;
      const double s = seqSum(pr, 0, nC) + seqSum(pr, nC, nC + nN) - w.bcast(b, i);
      if (ln == i) ww = s;
    }
    const double wi0 = w.bcast(ww, i), loi = w.bcast(lo, i), hii = w.bcast(hi, i);
    if (loi == 0 && wi0 >= 0) { if (ln == i) st = 0; nN++; }
    else if (hii == 0 && wi0 <= 0) { if (ln == i) st = 1; nN++; }
    else if (wi0 == 0) {
      if (nC > 0) solveEll(i);                 // solve1(delta_x, i, 0, only_transfer)
      appendFactorRow(i); swapProblem(nC, i); if (ln == nC) Cv = nC; nC++;
    } else {
      for (;;) {
        const double wi = w.bcast(ww, i);
        const int dir = (wi <= 0) ? 1 : -1;
        const double dirf = dir;
        solve1(i, dir);
        // dw(N) = A(N,C) dx(C) +/- A(i,N);  dw[i] = A(i,C) dx(C) + A(i,i) dirf   (lcp.cpp:926-928)
        {
          double arow[MAXR];
#pragma unroll
          for (int j = 0; j < MAXR; j++) arow[j] = C.A[me * CLD + j];
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < MAXR; j++) { const double pr = arow[j] * w.bcast(dx, j); s = (j < nC) ? s + pr : s; }
          const bool inN = ln >= nC && ln < nC + nN;
          const double ai = C.A[me * CLD + i];
          if (inN) dw = dir > 0 ? s + ai : s - ai;
          if (ln == i) dw = s + ai * dirf;
        }
        // step length: first minimum in the reference's scan order (i's own events, N rows, C rows)
        double s = INFINITY;
        int cmd = 0;
        if (ln == i) {
          s = -ww / dw; cmd = 1;
          if (dir > 0) { if (hi < INFINITY) { const double s2 = (hi - x) * dirf; if (s2 < s) { s = s2; cmd = 3; } } }
          else { if (lo > -INFINITY) { const double s2 = (lo - x) * dirf; if (s2 < s) { s = s2; cmd = 2; } } }
        } else if (ln >= nC && ln < nC + nN) {
          if ((st == 0) ? dw < 0 : dw > 0) {
            if (!(lo == 0 && hi == 0)) { s = -ww / dw; cmd = 4; }
          }
        } else if (ln < nC) {
          if (dx < 0 && lo > -INFINITY) { s = (lo - x) / dx; cmd = 5; }
          if (dx > 0 && hi < INFINITY) { s = (hi - x) / dx; cmd = 6; }
        }
        // the reference keeps a candidate only if it is STRICTLY smaller than the running minimum, which starts at lane
        // i's value: arg-min over (s, scan position)
        const double sOwn = w.bcast(s, i);
        if (sOwn != sOwn) { row.x = 0.0; return -1; }
        const double sMin = -w.maxAll(cmd != 0 ? -s : -INFINITY);
        const uint64_t tie = w.ballot(cmd != 0 && s == sMin);
        if (tie == 0ull) { row.x = 0.0; return -1; }
        // first of the ties in scan order: lane i, then the N lanes by position, then the C lanes by position
        int best;
        {
          const uint64_t mi = tie & (1ull << i);
          const uint64_t maskN = nN > 0 ? (((1ull << nN) - 1ull) << nC) : 0ull;
          const uint64_t maskC = nC > 0 ? ((1ull << nC) - 1ull) : 0ull;
          if (mi) best = i;
          else if (tie & maskN) best = __builtin_ctzll(tie & maskN);
          else best = __builtin_ctzll((tie & maskC) ? (tie & maskC) : tie);
        }
        const int cmdB = w.bcastI(cmd, best);
        if (sMin <= 0.0) { row.x = 0.0; return 0; }   // earlyTermination (the caller always has the PGS fallback, BoxedLcpConstraintSolver.cpp:463)
        const int si = best;
        // apply the step (lcp.cpp:1031-1036)
        if (ln < nC) x = x + sMin * dx;
        if (ln == i) x = x + sMin * dirf;
        if (ln >= nC && ln < nC + nN) ww = ww + sMin * dw;
        if (ln == i) ww = ww + sMin * dw;
        switch (cmdB) {
          case 1: if (ln == i) ww = 0; appendFactorRow(i); swapProblem(nC, i); if (ln == nC) Cv = nC; nC++; break;   // ell / Dell of solve1(i)
          case 2: if (ln == i) { x = lo; st = 0; } nN++; break;
          case 3: if (ln == i) { x = hi; st = 1; } nN++; break;
          case 4:                                                                                                  // transfer_i_from_N_to_C
            if (ln == si) ww = 0;
            if (nC > 0) solveEll(si);
            appendFactorRow(si); swapProblem(nC, si); if (ln == nC) Cv = nC; nN--; nC++; break;
          case 5: if (ln == si) { x = lo; st = 0; } removeFromC(si); break;
          case 6: if (ln == si) { x = hi; st = 1; } removeFromC(si); break;
        }
        if (cmdB <= 3) break;
      }
    }
  }
  // back to the original (reduced) order: P.x[p[j]] = x[j]
  w.sync();
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
#pragma unroll
      for (int r = 0; r < MAXR; r++) {   // unconditional reads (in bounds), rows >= n discarded by the select
        const double dr = C.A[r * CLD + a] - C.A[r * CLD + bcol];
        const double d = (r < n) ? dr : 0.0;
        d2 += d * d;
      }
      const double ba = w.bcast(row.b, a), ha = w.bcast(row.hi, a), la = w.bcast(row.lo, a);
      const int fa = w.bcastI(row.findex, a);
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
    const int fi = w.bcastI(row.findex, i);
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
// Gauss-Seidel is sequential over the rows.  Lane i keeps its row of A in registers and every lane a replica of x, so the
// row whose turn it is forms its sum from registers in the reference's order (no LDS, no barrier: ~720 row steps are a
// dependent chain and their latency is the cost) and the new x_i is broadcast with a readlane.  A is modified (rows
// normalised) only in registers.  row.x in: start, out: result.
template <class W>
DEV bool coopPgs(const W& w, CascadeLds& C, int n, CoopLcpRow& row) {
  const int maxIteration = 30;
  const double dxTh = 1e-6, relTol = 1e-3, epsDiv = 1e-9;
  const int ln = w.lane();
  const bool on = ln < n;
  const int me = on ? ln : 0;
  // Residual form of the projected Gauss-Seidel sweep.  Every lane keeps ITS row of A, scaled by 1 / a_ii (the reference divides
  // in its first sweep and pre-scales the rows for the later ones, PgsBoxedLcpSolver.cpp:150-200), and the scaled residual
  // r = b' - sum_k a'_k x_k of its row.  The step of row i is then  x_i <- clamp(x_i + r_i)  on lane i, one broadcast of the
  // change, and one FMA per lane (A is symmetric: lane j's a'_j[i] is its share of column i).  The reference's own order - a
  // 23-term dot product per row - would be a 23-deep chain of dependent FMAs on one lane, 720 times per stage; the two
  // orders agree to round-off, the clamps, the convergence tests and the iteration cap are the reference's.
  double arow[MAXR];
#pragma unroll
  for (int j = 0; j < MAXR; j++) { const double av = C.A[me * CLD + j]; arow[j] = (on && j < n) ? av : 0.0; }
  const double aii = on ? C.A[me * CLD + me] : 1.0;
  const bool inOrder = on && !(aii < epsDiv);          // rows with a_ii ~ 0 are set to 0 once and then left alone
  const double sc = inOrder ? 1.0 / aii : 1.0;
#pragma unroll
  for (int j = 0; j < MAXR; j++) arow[j] *= sc;
  double xOwn = on ? row.x : 0.0;
  double r0 = row.b * sc, r1 = 0.0;
#pragma unroll
  for (int j = 0; j < MAXR; j += 2) {
    r0 = fma(-arow[j], j < n ? w.bcast(xOwn, j) : 0.0, r0);
    r1 = fma(-arow[j + 1], j + 1 < n ? w.bcast(xOwn, j + 1) : 0.0, r1);
  }
  double r = r0 + r1;
  // friction rows follow the current impulse of their normal row: every lane keeps it up to date itself
  const int fiOwn = on ? row.findex : -1;
  double xfOwn = w.shfl(xOwn, fiOwn >= 0 ? fiOwn : 0);
  bool bad = false;
  // One Gauss-Seidel row step for row i (compile-time i), branch-free: EVERY lane evaluates the update of its own row from its
  // own (x, r, bounds) - only lane i's result is kept, broadcast, and folded into everybody's residual.  (A divergent
  // `if (lane == i)` costs EXEC bookkeeping and a scalar branch per row step, 720 times per stage.)
  auto rowStep = [&](auto iTag, bool first) {
    constexpr int i = decltype(iTag)::value;
    if (i >= n) return;
    const double old_x = xOwn;
    const double new_x = old_x + r;
    const double hi_tmp = fiOwn >= 0 ? row.hi * xfOwn : row.hi, lo_tmp = fiOwn >= 0 ? -hi_tmp : row.lo;
    double xi = fmin(fmax(new_x, lo_tmp), hi_tmp);   // = the reference's nested comparisons for lo <= hi (two ops, no VCC round trip)
    if (!inOrder) xi = first ? 0.0 : old_x;
    const double dOwn = xi - old_x;
    const bool badOwn = inOrder && (first ? fabs(dOwn) > dxTh : (fabs(xi) > epsDiv && fabs(dOwn) > relTol * fabs(xi)));
    const bool mineNow = ln == i;
    bad = bad || (mineNow && badOwn);
    const double delta = w.bcast(dOwn, i);
    xOwn = mineNow ? xi : xOwn;
    xfOwn = fiOwn == i ? xfOwn + delta : xfOwn;
    r = fma(-arow[i], delta, r);
  };
  auto sweep = [&](bool first) {
    rowStep(IntTag<0>{}, first); rowStep(IntTag<1>{}, first); rowStep(IntTag<2>{}, first); rowStep(IntTag<3>{}, first);
    rowStep(IntTag<4>{}, first); rowStep(IntTag<5>{}, first); rowStep(IntTag<6>{}, first); rowStep(IntTag<7>{}, first);
    rowStep(IntTag<8>{}, first); rowStep(IntTag<9>{}, first); rowStep(IntTag<10>{}, first); rowStep(IntTag<11>{}, first);
    rowStep(IntTag<12>{}, first); rowStep(IntTag<13>{}, first); rowStep(IntTag<14>{}, first); rowStep(IntTag<15>{}, first);
    rowStep(IntTag<16>{}, first); rowStep(IntTag<17>{}, first); rowStep(IntTag<18>{}, first); rowStep(IntTag<19>{}, first);
    rowStep(IntTag<20>{}, first); rowStep(IntTag<21>{}, first); rowStep(IntTag<22>{}, first); rowStep(IntTag<23>{}, first);
  };
  sweep(true);
  if (w.ballot(bad) == 0ull) { row.x = xOwn; return true; }
  bool done = false;
#pragma unroll 1
  for (int iter = 1; iter < maxIteration; ++iter) {
    bad = false;
    sweep(false);
    if (w.ballot(bad) == 0ull) { done = true; break; }
  }
  row.x = xOwn;
  return done;
}

// ---- stages 1-3 of BoxedLcpConstraintSolver::solveLcp (:461-677) + registration / standardisation (:718-736) ----
struct CoopCascadeOut {
  double X;          // impulses of this lane's row
  CoopClasses K;
  double cfm;
  uint32_t st;       // NBL_ST_* bits to OR into the world's status
  bool pinvValid;
#ifdef NBL_CASCADE_TIMING
  long long t[8];    // cycle stamps: start, after reduce, Dantzig, validity, stage 2, stage 3, standardise
  int iters;
#endif
};

template <class W>
DEV void coopCascade(const W& w, CoopLds& S, CascadeLds& C, const CoopRow& R, double X0, double fallbackCfm, CoopCascadeOut& out) {
  const int ln = w.lane();
  const int m = R.m;
  CoopLcpRow row;
  int mapTo = -1;
  auto loadProblem = [&](double cfmDiag, double x0) {
    {
      // all 24 loads of the lane's column in flight together (in a rolled loop every load was waited for before the next)
      double col[MAXR];
#pragma unroll
      for (int j = 0; j < MAXR; j++) col[j] = R.a(j);
      if (ln < m) {
#pragma unroll
        for (int j = 0; j < MAXR; j++) if (j < m) C.A[ln * CLD + j] = col[j] + (ln == j ? cfmDiag : 0.0);   // A is symmetric: row = column
      }
    }
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
#ifdef NBL_CASCADE_TIMING
  out.t[0] = clock64();
#endif
  loadProblem(0.0, X0);
  int nr = coopLcpReduce(w, C, m, row, mapTo);
#ifdef NBL_CASCADE_TIMING
  out.t[1] = clock64();
#endif
  const int rc = coopDantzig(w, C, nr, row);
#ifdef NBL_CASCADE_TIMING
  out.t[2] = clock64();
#endif
  if (rc == 1) {
    X = mapped(row.x, nr);
    success = coopValid(w, S, R, X, false, 0.0, 1);
    if (success) st |= 0x4u;
  }
  if (rc < 0 || hasNan(X)) { success = false; X = 0.0; st |= 0x40u; }
#ifdef NBL_CASCADE_TIMING
  out.t[3] = clock64();
#endif
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
#ifdef NBL_CASCADE_TIMING
  out.t[4] = clock64();
#endif
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
#ifdef NBL_CASCADE_TIMING
  out.t[5] = clock64();
#endif
  // ---- register the fresh solution, classify, standardise (:718-736) ----
  bool pinvValid = false;
  const bool std = coopStandardizeLoop(w, S, R, X, cfm, ignoreFriction, 0u, pinvValid, out.K);
  if (std) st |= 0x100u;
  out.X = X; out.cfm = cfm; out.st = st; out.pinvValid = std && pinvValid;
#ifdef NBL_CASCADE_TIMING
  out.t[6] = clock64();
#endif
}

}  // namespace nbl
