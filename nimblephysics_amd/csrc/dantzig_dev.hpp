// dantzig_dev.hpp — per-lane boxed-LCP solvers for the lanes whose warm start is not already a solution
// (stages 1-3 of BoxedLcpConstraintSolver::solveLcp, dart/constraint/BoxedLcpConstraintSolver.cpp:461-677):
//
//   * lcpReduce / lcpRemoveFriction   LCPUtils::reduce / removeFriction (LCPUtils.cpp:144-247, 346-520)
//   * dantzigSolve                    the ODE Dantzig driver the reference calls (dart/external/odelcpsolver/lcp.cpp:780-1113):
//                                     same driving order (friction rows moved to the end, normals solved first, friction
//                                     bounds frozen once from the solved normals), same step-length events and
//                                     tie-breaking order over the permuted N and C sets, early termination on s <= 0.
//                                     The incrementally updated LDL^T of A(C,C) (solve1 / dLDLTAddTL / dLDLTRemove) is
//                                     replaced by a fresh LDL^T of the current A(C,C): identical mathematics, and the
//                                     problems here have at most 24 rows.  Like the reference only the LOWER triangle
//                                     of A is referenced (lcp.cpp:138-140), which matters after column merging.
//   * pgsSolve                        PgsBoxedLcpSolver::solve (PgsBoxedLcpSolver.cpp:79-268), Option(30, 1e-6, 1e-3, 1e-9, false)
//
// Everything is written against a small memory accessor (`at(index)`) so the same code runs per lane on the
// device (LDS / lane-interleaved HBM) and, in the unit tests, on the host against the reference's own dSolveLCP.
#pragma once
#include "lcp_dev.hpp"

namespace nbl {

// Reduced LCP: matrix (n x n, leading dimension MAXR) behind `mem`, vectors in private arrays.
struct RedLcp {
  int n;
  double x[MAXR], b[MAXR], lo[MAXR], hi[MAXR];
  int findex[MAXR];
  int mapTo[MAXR];   // original row -> reduced column (mapOut is a 0/1 matrix, one 1 per row), -1 = dropped
  int nOrig;
};

template <class Mem>
DEV void redRemove(const Mem& mem, int offA, RedLcp& P, int col) {  // delete row+column `col`
  const int n = P.n;
  for (int i = 0; i < n; i++) {
    if (i == col) continue;
    const int ni = i > col ? i - 1 : i;
    for (int j = 0; j < n; j++) {
      if (j == col) continue;
      const int nj = j > col ? j - 1 : j;
      mem.at(offA + ni * MAXR + nj) = mem.at(offA + i * MAXR + j);  // rows/cols move up-left: reads stay ahead of writes
    }
  }
  for (int i = col; i + 1 < n; i++) { P.x[i] = P.x[i + 1]; P.b[i] = P.b[i + 1]; P.lo[i] = P.lo[i + 1]; P.hi[i] = P.hi[i + 1]; P.findex[i] = P.findex[i + 1]; }
  P.n = n - 1;
}

// LCPUtils::reduce: merge near-identical columns (squared distance < 1e-4, |b_a - b_b| < 1e-4, same findex/hi/lo)
template <class Mem>
DEV void lcpReduce(const Mem& mem, int offA, RedLcp& P) {
  const double TH = 1e-4;
  for (;;) {
    const int n = P.n;
    int ma = -1, mb = -1;
    for (int a = 0; a < n - 1 && ma < 0; a++)
      for (int b = a + 1; b < n; b++) {
        double d2 = 0;
        for (int r = 0; r < n; r++) { double d = mem.at(offA + r * MAXR + a) - mem.at(offA + r * MAXR + b); d2 += d * d; }
        if (d2 < TH && fabs(P.b[a] - P.b[b]) < TH && P.findex[a] == P.findex[b] && P.hi[a] == P.hi[b] && P.lo[a] == P.lo[b]) { ma = a; mb = b; break; }
      }
    if (ma < 0) break;
    // mergeLCPColumns(colA = ma, colB = mb): column A doubled, row/column B deleted, findex remapped
    for (int r = 0; r < n; r++) mem.at(offA + r * MAXR + ma) *= 2.0;
    for (int i = 0; i < n; i++) {
      if (P.findex[i] == mb) P.findex[i] = ma;
      else if (P.findex[i] > mb) P.findex[i] -= 1;
    }
    redRemove(mem, offA, P, mb);
    for (int o = 0; o < P.nOrig; o++) {
      if (P.mapTo[o] == mb) P.mapTo[o] = ma;
      else if (P.mapTo[o] > mb) P.mapTo[o] -= 1;
    }
  }
}

// LCPUtils::removeFriction: drop every row with findex != -1 (from the last one down)
template <class Mem>
DEV void lcpRemoveFriction(const Mem& mem, int offA, RedLcp& P) {
  for (int i = P.n - 1; i >= 0; i--) {
    if (P.findex[i] == -1) continue;
    for (int k = 0; k < P.n; k++) if (P.findex[k] > i) P.findex[k] -= 1;
    redRemove(mem, offA, P, i);
    for (int o = 0; o < P.nOrig; o++) {
      if (P.mapTo[o] == i) P.mapTo[o] = -1;
      else if (P.mapTo[o] > i) P.mapTo[o] -= 1;
    }
  }
}

// ---- PGS ----  (A is modified: rows normalised, like the reference)
template <class Mem>
DEV bool pgsSolve(const Mem& mem, int offA, RedLcp& P) {
  const int n = P.n;
  const int maxIteration = 30;
  const double dxTh = 1e-6, relTol = 1e-3, epsDiv = 1e-9;
  int order[MAXR], no = 0;
  bool possible = true;
  for (int i = 0; i < n; ++i) {
    const double aii = mem.at(offA + i * MAXR + i);
    if (aii < epsDiv) { P.x[i] = 0.0; continue; }
    order[no++] = i;
    const double old_x = P.x[i];
    double new_x = P.b[i];
    for (int j = 0; j < i; ++j) new_x -= mem.at(offA + i * MAXR + j) * P.x[j];
    for (int j = i + 1; j < n; ++j) new_x -= mem.at(offA + i * MAXR + j) * P.x[j];
    new_x /= aii;
    if (P.findex[i] >= 0) {
      const double hi_tmp = P.hi[i] * P.x[P.findex[i]], lo_tmp = -hi_tmp;
      P.x[i] = new_x > hi_tmp ? hi_tmp : (new_x < lo_tmp ? lo_tmp : new_x);
    } else P.x[i] = new_x > P.hi[i] ? P.hi[i] : (new_x < P.lo[i] ? P.lo[i] : new_x);
    if (possible && fabs(P.x[i] - old_x) > dxTh) possible = false;
  }
  if (possible) return true;
  for (int t = 0; t < no; t++) {
    const int idx = order[t];
    const double dummy = 1.0 / mem.at(offA + idx * MAXR + idx);
    P.b[idx] *= dummy;
    for (int j = 0; j < n; ++j) mem.at(offA + idx * MAXR + j) *= dummy;
  }
  for (int iter = 1; iter < maxIteration; ++iter) {
    possible = true;
    for (int t = 0; t < no; t++) {
      const int idx = order[t];
      double new_x = P.b[idx];
      const double old_x = P.x[idx];
      for (int j = 0; j < idx; j++) new_x -= mem.at(offA + idx * MAXR + j) * P.x[j];
      for (int j = idx + 1; j < n; j++) new_x -= mem.at(offA + idx * MAXR + j) * P.x[j];
      if (P.findex[idx] >= 0) {
        const double hi_tmp = P.hi[idx] * P.x[P.findex[idx]], lo_tmp = -hi_tmp;
        P.x[idx] = new_x > hi_tmp ? hi_tmp : (new_x < lo_tmp ? lo_tmp : new_x);
      } else P.x[idx] = new_x > P.hi[idx] ? P.hi[idx] : (new_x < P.lo[idx] ? P.lo[idx] : new_x);
      if (possible && fabs(P.x[idx]) > epsDiv) {
        if (fabs((P.x[idx] - old_x) / P.x[idx]) > relTol) possible = false;
      }
    }
    if (possible) break;
  }
  return possible;
}

// ---- Dantzig ----
// Solves in place: P.x receives the solution (indexed by reduced row), returns false on early termination.
// `offL`: n x n scratch (leading dimension MAXR) for the LDL^T factor of A(C,C).
template <class Mem>
DEV bool dantzigSolve(const Mem& mem, int offA, int offL, RedLcp& P) {
  const int n = P.n;
  int p[MAXR];
  double x[MAXR], w[MAXR], b[MAXR], lo[MAXR], hi[MAXR], dx[MAXR], dw[MAXR];
  bool state[MAXR];
  int fidx[MAXR];
  for (int k = 0; k < n; k++) { p[k] = k; x[k] = 0; w[k] = 0; b[k] = P.b[k]; lo[k] = P.lo[k]; hi[k] = P.hi[k]; state[k] = false; fidx[k] = P.findex[k]; dx[k] = 0; dw[k] = 0; }
  // only the lower triangle of the (reduced) matrix is referenced
  auto Aperm = [&](int i, int j) -> double {
    const int u = p[i], v = p[j];
    return u >= v ? mem.at(offA + u * MAXR + v) : mem.at(offA + v * MAXR + u);
  };
  auto swapProblem = [&](int i1, int i2) {
    if (i1 == i2) return;
    double t;
    int ti;
    bool tb;
    t = x[i1]; x[i1] = x[i2]; x[i2] = t;
    t = b[i1]; b[i1] = b[i2]; b[i2] = t;
    t = w[i1]; w[i1] = w[i2]; w[i2] = t;
    t = lo[i1]; lo[i1] = lo[i2]; lo[i2] = t;
    t = hi[i1]; hi[i1] = hi[i2]; hi[i2] = t;
    ti = p[i1]; p[i1] = p[i2]; p[i2] = ti;
    tb = state[i1]; state[i1] = state[i2]; state[i2] = tb;
    ti = fidx[i1]; fidx[i1] = fidx[i2]; fidx[i2] = ti;
  };
  int nC = 0, nN = 0;
  // contact problems have no unbounded rows (nub = 0); every findex row goes to the end (lcp.cpp:487-498)
  {
    int atEnd = 0;
    for (int k = n - 1; k >= 0; k--)
      if (fidx[k] >= 0) { swapProblem(k, n - 1 - atEnd); atEnd++; }
  }
  // dx(C) = -dir * A(C,C)^-1 A(C,i)  via a fresh LDL^T (no pivoting, like dFactorLDLT)
  auto solve1 = [&](int i, int dir) {
    if (nC == 0) return;
    for (int r = 0; r < nC; r++)
      for (int c = 0; c <= r; c++) {
        double s = Aperm(r, c);
        for (int k = 0; k < c; k++) s -= mem.at(offL + r * MAXR + k) * mem.at(offL + c * MAXR + k) * mem.at(offL + k * MAXR + k);
        if (r == c) mem.at(offL + r * MAXR + r) = s;                 // D on the diagonal
        else mem.at(offL + r * MAXR + c) = s / mem.at(offL + c * MAXR + c);
      }
    double y[MAXR];
    for (int r = 0; r < nC; r++) { double s = Aperm(r, i); for (int k = 0; k < r; k++) s -= mem.at(offL + r * MAXR + k) * y[k]; y[r] = s; }
    for (int r = 0; r < nC; r++) y[r] /= mem.at(offL + r * MAXR + r);
    for (int r = nC - 1; r >= 0; r--) { double s = y[r]; for (int k = r + 1; k < nC; k++) s -= mem.at(offL + k * MAXR + r) * y[k]; y[r] = s; }
    for (int r = 0; r < nC; r++) dx[r] = dir > 0 ? -y[r] : y[r];
  };
  bool hitFirstFriction = false;
  for (int i = 0; i < n; ++i) {
    if (!hitFirstFriction && fidx[i] >= 0) {
      double un[MAXR];
      for (int j = 0; j < n; ++j) un[p[j]] = x[j];
      for (int k = i; k < n; ++k) {
        const double wfk = un[fidx[k]];
        if (wfk == 0) { hi[k] = 0; lo[k] = 0; }
        else { hi[k] = fabs(hi[k] * wfk); lo[k] = -hi[k]; }
      }
      hitFirstFriction = true;
    }
    {
      double s = -b[i];
      for (int j = 0; j < nC + nN; j++) s += Aperm(i, j) * x[j];
      w[i] = s;
    }
    if (lo[i] == 0 && w[i] >= 0) { nN++; state[i] = false; }
    else if (hi[i] == 0 && w[i] <= 0) { nN++; state[i] = true; }
    else if (w[i] == 0) { swapProblem(nC, i); nC++; }
    else {
      for (;;) {
        const int dir = (w[i] <= 0) ? 1 : -1;
        const double dirf = dir;
        solve1(i, dir);
        for (int k = 0; k < nN; k++) {
          const int r = nC + k;
          double s = 0;
          for (int j = 0; j < nC; j++) s += Aperm(r, j) * dx[j];
          dw[r] = s + dirf * Aperm(r, i);
        }
        {
          double s = 0;
          for (int j = 0; j < nC; j++) s += Aperm(i, j) * dx[j];
          dw[i] = s + Aperm(i, i) * dirf;
        }
        int cmd = 1, si = 0;
        double s = -w[i] / dw[i];
        if (dir > 0) {
          if (hi[i] < INFINITY) { double s2 = (hi[i] - x[i]) * dirf; if (s2 < s) { s = s2; cmd = 3; } }
        } else {
          if (lo[i] > -INFINITY) { double s2 = (lo[i] - x[i]) * dirf; if (s2 < s) { s = s2; cmd = 2; } }
        }
        for (int k = 0; k < nN; ++k) {
          const int r = nC + k;
          if (!state[r] ? dw[r] < 0 : dw[r] > 0) {
            if (lo[r] == 0 && hi[r] == 0) continue;
            double s2 = -w[r] / dw[r];
            if (s2 < s) { s = s2; cmd = 4; si = r; }
          }
        }
        for (int k = 0; k < nC; ++k) {
          if (dx[k] < 0 && lo[k] > -INFINITY) { double s2 = (lo[k] - x[k]) / dx[k]; if (s2 < s) { s = s2; cmd = 5; si = k; } }
          if (dx[k] > 0 && hi[k] < INFINITY) { double s2 = (hi[k] - x[k]) / dx[k]; if (s2 < s) { s = s2; cmd = 6; si = k; } }
        }
        if (s <= 0.0) return false;   // earlyTermination (the caller always has the PGS fallback, BoxedLcpConstraintSolver.cpp:463)
        for (int k = 0; k < nC; k++) x[k] += s * dx[k];
        x[i] += s * dirf;
        for (int k = 0; k < nN; k++) w[nC + k] += s * dw[nC + k];
        w[i] += s * dw[i];
        switch (cmd) {
          case 1: w[i] = 0; swapProblem(nC, i); nC++; break;
          case 2: x[i] = lo[i]; state[i] = false; nN++; break;
          case 3: x[i] = hi[i]; state[i] = true; nN++; break;
          case 4: w[si] = 0; swapProblem(nC, si); nN--; nC++; break;
          case 5: x[si] = lo[si]; state[si] = false; swapProblem(si, nC - 1); nN++; nC--; break;
          case 6: x[si] = hi[si]; state[si] = true; swapProblem(si, nC - 1); nN++; nC--; break;
        }
        if (cmd <= 3) break;
      }
    }
  }
  for (int j = 0; j < n; ++j) P.x[p[j]] = x[j];
  return true;
}

}  // namespace nbl
