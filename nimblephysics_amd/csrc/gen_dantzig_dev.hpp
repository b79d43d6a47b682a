// gen_dantzig_dev.hpp — the Dantzig boxed-LCP driver for problems of ANY size, and stages 1-3 of the solver cascade around it.
//
// genDantzig restates dSolveLCP (dart/external/odelcpsolver/lcp.cpp:780-1113, the dLCP object :330-780, with dLDLTAddTL / dLDLTRemove /
// dRemoveRowCol of matrix.cpp:286-460, dSolveL1 / dSolveL1T of fastlsolve.cpp / fastltsolve.cpp and dDot of fastdot.cpp) operation by
// operation, as ONE sequential instruction stream (lane 0 of the world's wavefront): the same driving order, the same index sets and
// swaps, the factor L / d of A(C,C) in the reference's own row order with rows appended (ell / Dell) and removed (down-dating), the
// blocked order of the additions in the two triangular solves, no fused multiply-adds - so that x and the success flag are BIT-IDENTICAL
// to the reference's solver on identical inputs, rank-deficient problems included (tests/test_gen_host.py against the reference's
// compiled dSolveLCP up to 192 rows; tests/test_gpu_general.py on the device).  The wavefront-cooperative statement of the same driver
// (coop_dantzig_dev.hpp) is what the 24- and 48-row builds run; it owes its speed to lane = row and cannot go past 64 rows.
// Contact problems have no unbounded rows (nub = 0: the driver's initial factorisation does not occur).
// Attribution: the algorithm restated here derives from the Open Dynamics Engine (ODE), Copyright (C) 2001-2003 Russell L. Smith, which the
// reference vendors under ODE's BSD-style licence (dart/external/odelcpsolver/, dart/collision/dart/DARTCollide.cpp); this file is an
// independent restatement for another execution model - ODE's arithmetic order and, where the bit-for-bit tests need them recognisable,
// its identifiers are kept on purpose.
#pragma once
#include "gen_lcp_dev.hpp"

namespace NBL_NS {

struct GenDantzigMem {      // all in the world's HBM scratch
  int ld;                   // leading dimension of A (and of the scratch matrix of the removals)
  int ldL;                  // ... of L: the factor may live packed in fast memory (genCarve)
  double* A;                // n x n (leading dimension ld): the problem's matrix; symmetrised and permuted in place
  double* L;                // factor rows
  double *d, *x, *w, *b, *lo, *hi, *dx, *dw, *ell, *Dell, *tmp, *tvec, *W1, *W2;
  int *p, *C, *findex, *state;
};

// returns 1 solved, 0 early termination (s <= 0), -1 NaN step
DEV int genDantzigSeq(const GenDantzigMem& M, int n, double* xOut) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  double* A = M.A; double* L = M.L;
  auto AA = [&](int i, int j) -> double& { return A[(size_t)i * M.ld + j]; };
  auto LL = [&](int i, int j) -> double& { return L[(size_t)i * M.ldL + j]; };
  // dDot (fastdot.cpp): the running sum from 0 in index order
  auto dotRows = [&](const double* a, const double* b, int cnt) -> double { double s = 0.0; for (int k = 0; k < cnt; k++) s = s + a[k] * b[k]; return s; };
  int nC = 0, nN = 0;
  for (int k = 0; k < n; k++) { M.p[k] = k; M.x[k] = 0.0; M.w[k] = 0.0; M.state[k] = 0; M.dx[k] = 0.0; M.dw[k] = 0.0; }
  // the reference only references the LOWER triangle of A (lcp.cpp:138-140, which matters after column merging): symmetrise from it once,
  // then keep the permuted matrix symmetric
  for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) AA(i, j) = AA(j, i);
  auto swapProblem = [&](int i1, int i2) {
    if (i1 == i2) return;
    for (int k = 0; k < n; k++) { const double t = AA(i1, k); AA(i1, k) = AA(i2, k); AA(i2, k) = t; }
    for (int k = 0; k < n; k++) { const double t = AA(k, i1); AA(k, i1) = AA(k, i2); AA(k, i2) = t; }
    double t;
    t = M.x[i1]; M.x[i1] = M.x[i2]; M.x[i2] = t;
    t = M.b[i1]; M.b[i1] = M.b[i2]; M.b[i2] = t;
    t = M.w[i1]; M.w[i1] = M.w[i2]; M.w[i2] = t;
    t = M.lo[i1]; M.lo[i1] = M.lo[i2]; M.lo[i2] = t;
    t = M.hi[i1]; M.hi[i1] = M.hi[i2]; M.hi[i2] = t;
    int ti;
    ti = M.p[i1]; M.p[i1] = M.p[i2]; M.p[i2] = ti;
    ti = M.state[i1]; M.state[i1] = M.state[i2]; M.state[i2] = ti;
    ti = M.findex[i1]; M.findex[i1] = M.findex[i2]; M.findex[i2] = ti;
  };
  // every findex row goes to the end (lcp.cpp:487-498)
  {
    int atEnd = 0;
    for (int k = n - 1; k >= 0; k--)
      if (M.findex[k] >= 0) { swapProblem(k, n - 1 - atEnd); atEnd++; }
  }
  // dSolveL1 (fastlsolve.cpp): L y = B in place, blocks of four rows: Z = the sum over the columns before the block (index order),
  // then the block's own columns subtracted one by one
  auto solveL1 = [&](double* B, int cnt) {
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {
      double Z0 = 0.0, Z1 = 0.0, Z2 = 0.0, Z3 = 0.0;
      for (int j = 0; j < i; j++) {
        const double q = B[j];
        Z0 = Z0 + LL(i, j) * q; Z1 = Z1 + LL(i + 1, j) * q; Z2 = Z2 + LL(i + 2, j) * q; Z3 = Z3 + LL(i + 3, j) * q;
      }
      const double y0 = B[i] - Z0;
      B[i] = y0;
      const double y1 = B[i + 1] - Z1 - LL(i + 1, i) * y0;
      B[i + 1] = y1;
      const double y2 = B[i + 2] - Z2 - LL(i + 2, i) * y0 - LL(i + 2, i + 1) * y1;
      B[i + 2] = y2;
      const double y3 = B[i + 3] - Z3 - LL(i + 3, i) * y0 - LL(i + 3, i + 1) * y1 - LL(i + 3, i + 2) * y2;
      B[i + 3] = y3;
    }
    for (; i < cnt; i++) {
      double Z = 0.0;
      for (int j = 0; j < i; j++) Z = Z + LL(i, j) * B[j];
      B[i] = B[i] - Z;
    }
  };
  // dSolveL1T (fastltsolve.cpp): L^T y = B in place - the same blocking on the reversed index (rr = cnt - 1 - row)
  auto solveL1T = [&](double* B, int cnt) {
    auto Lt = [&](int rr, int jj) -> double { return LL(cnt - 1 - jj, cnt - 1 - rr); };   // L^T between reversed indices, jj < rr
    auto Bv = [&](int rr) -> double& { return B[cnt - 1 - rr]; };
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {
      double Z0 = 0.0, Z1 = 0.0, Z2 = 0.0, Z3 = 0.0;
      for (int j = 0; j < i; j++) {
        const double q = Bv(j);
        Z0 = Z0 + Lt(i, j) * q; Z1 = Z1 + Lt(i + 1, j) * q; Z2 = Z2 + Lt(i + 2, j) * q; Z3 = Z3 + Lt(i + 3, j) * q;
      }
      const double y0 = Bv(i) - Z0;
      Bv(i) = y0;
      const double y1 = Bv(i + 1) - Z1 - Lt(i + 1, i) * y0;
      Bv(i + 1) = y1;
      const double y2 = Bv(i + 2) - Z2 - Lt(i + 2, i) * y0 - Lt(i + 2, i + 1) * y1;
      Bv(i + 2) = y2;
      const double y3 = Bv(i + 3) - Z3 - Lt(i + 3, i) * y0 - Lt(i + 3, i + 1) * y1 - Lt(i + 3, i + 2) * y2;
      Bv(i + 3) = y3;
    }
    for (; i < cnt; i++) {
      double Z = 0.0;
      for (int j = 0; j < i; j++) Z = Z + Lt(i, j) * Bv(j);
      Bv(i) = Bv(i) - Z;
    }
  };
  // Dell = L^-1 A(i, C[.]), ell = Dell * d      (first half of dLCP::solve1, lcp.cpp:700-730)
  auto solveEll = [&](int i) {
    for (int j = 0; j < nC; j++) M.Dell[j] = AA(i, M.C[j]);
    solveL1(M.Dell, nC);
    for (int j = 0; j < nC; j++) M.ell[j] = M.Dell[j] * M.d[j];
  };
  auto solve1 = [&](int i, int dir) {
    if (nC == 0) return;
    solveEll(i);
    for (int j = 0; j < nC; j++) M.tmp[j] = M.ell[j];
    solveL1T(M.tmp, nC);
    if (dir > 0) for (int j = 0; j < nC; j++) M.dx[M.C[j]] = -M.tmp[j];
    else for (int j = 0; j < nC; j++) M.dx[M.C[j]] = M.tmp[j];
  };
  // the row at position i (with ell / Dell of the last solveEll(i)) becomes factor row nC   (transfer_i_to_C, lcp.cpp:520-553)
  auto appendFactorRow = [&](int i) {
    if (nC > 0) {
      for (int j = 0; j < nC; j++) LL(nC, j) = M.ell[j];
      M.d[nC] = 1.0 / (AA(i, i) - dotRows(M.ell, M.Dell, nC));
    } else M.d[0] = 1.0 / AA(i, i);
  };
  // dLDLTAddTL (matrix.cpp:286-359) on the trailing block of the factor that starts at row / column r0 (n2 rows) with the vector a
  auto ldltAddTL = [&](int r0, int n2, const double* a) {
    if (n2 < 2) return;
    const double SQ = 0.70710678118654752440;   // M_SQRT1_2
    double* W1 = M.W1; double* W2 = M.W2;
    W1[0] = 0.0; W2[0] = 0.0;
    for (int j = 1; j < n2; j++) W1[j] = W2[j] = a[j] * SQ;
    const double W11 = (0.5 * a[0] + 1.0) * SQ, W21 = (0.5 * a[0] - 1.0) * SQ;
    double alpha1 = 1.0, alpha2 = 1.0;
    {
      double dee = M.d[r0];
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
      for (int pp = 1; pp < n2; pp++) {
        const double Wp = W1[pp];
        const double el = LL(r0 + pp, r0);
        W1[pp] = Wp - W11 * el;
        W2[pp] = k1 * Wp + k2 * el;
      }
    }
    for (int j = 1; j < n2; j++) {
      const double k1 = W1[j], k2 = W2[j];
      double dee = M.d[r0 + j];
      double alphanew = alpha1 + (k1 * k1) * dee;
      dee /= alphanew;
      const double gamma1 = k1 * dee;
      dee *= alpha1;
      alpha1 = alphanew;
      alphanew = alpha2 - (k2 * k2) * dee;
      dee /= alphanew;
      const double gamma2 = k2 * dee;
      dee *= alpha2;
      M.d[r0 + j] = dee;
      alpha2 = alphanew;
      for (int pp = j + 1; pp < n2; pp++) {
        double el = LL(r0 + pp, r0 + j);
        double Wp = W1[pp] - k1 * el;
        el += gamma1 * Wp;
        W1[pp] = Wp;
        Wp = W2[pp] - k2 * el;
        el -= gamma2 * Wp;
        W2[pp] = Wp;
        LL(r0 + pp, r0 + j) = el;
      }
    }
  };
  // dLDLTRemove (matrix.cpp:374-426): factor row / column r leaves the n2-row factor; then dRemoveRowCol snips it out of L and d
  auto ldltRemove = [&](int n2, int r) {
    if (r == n2 - 1) return;    // deleting the last row / column is easy
    double* a = M.tvec + n2;    // (the reference's tmp layout: t[0..r), a = t + r; any two disjoint buffers do)
    if (r == 0) {
      const int p0 = M.C[0];
      for (int i = 0; i < n2; i++) a[i] = -AA(M.C[i], p0);      // GETA(p[i], p[0]) (the permuted matrix is kept symmetric)
      a[0] += 1.0;
      ldltAddTL(0, n2, a);
    } else {
      double* t = M.tvec;
      for (int i = 0; i < r; i++) t[i] = LL(r, i) / M.d[i];
      const int pr = M.C[r];
      for (int i = 0; i < n2 - r; i++) a[i] = dotRows(&LL(r + i, 0), t, r) - AA(M.C[r + i], pr);
      a[0] += 1.0;
      ldltAddTL(r, n2 - r, a);
    }
    // dRemoveRowCol(L, n2, r) + memmove of d
    for (int i = 0; i < n2 - 1; i++) {
      const int si = i >= r ? i + 1 : i;
      for (int j = 0; j < n2 - 1; j++) {
        const int sj = j >= r ? j + 1 : j;
        if (j <= i) LL(i, j) = LL(si, sj);   // (lower triangle; sources are never above-left of their targets' unread neighbours: si >= i, sj >= j)
      }
    }
    for (int i = r; i + 1 < n2; i++) M.d[i] = M.d[i + 1];
  };
  // transfer_i_from_C_to_N (lcp.cpp:603-650): position i leaves C
  auto removeFromC = [&](int i) {
    int last_idx = -1;
    int j = 0;
    for (; j < nC; ++j) {
      if (M.C[j] == nC - 1) last_idx = j;
      if (M.C[j] == i) {
        ldltRemove(nC, j);
        int k;
        if (last_idx == -1) {
          for (k = j + 1; k < nC; ++k) if (M.C[k] == nC - 1) break;
        } else k = last_idx;
        M.C[k] = M.C[j];
        for (int q = j; q + 1 < nC; q++) M.C[q] = M.C[q + 1];
        break;
      }
    }
    swapProblem(i, nC - 1);
    nN++; nC--;
  };
  bool hitFirstFriction = false;
  for (int i = 0; i < n; ++i) {
    if (!hitFirstFriction && M.findex[i] >= 0) {
      // un[p[j]] = x[j]; bounds of the friction rows frozen from the solved normals (lcp.cpp:856-873)
      for (int j = 0; j < n; ++j) M.dw[M.p[j]] = M.x[j];
      for (int k = i; k < n; ++k) {
        const double wfk = M.dw[M.findex[k]];
        if (wfk == 0) { M.hi[k] = 0; M.lo[k] = 0; }
        else { M.hi[k] = fabs(M.hi[k] * wfk); M.lo[k] = -M.hi[k]; }
      }
      hitFirstFriction = true;
    }
    // w[i] = A(i,C) x(C) + A(i,N) x(N) - b[i]: two running sums (lcp.cpp:877)
    M.w[i] = dotRows(&AA(i, 0), M.x, nC) + dotRows(&AA(i, nC), M.x + nC, nN) - M.b[i];
    if (M.lo[i] == 0 && M.w[i] >= 0) { nN++; M.state[i] = 0; }
    else if (M.hi[i] == 0 && M.w[i] <= 0) { nN++; M.state[i] = 1; }
    else if (M.w[i] == 0) {
      if (nC > 0) solveEll(i);                 // solve1(delta_x, i, 0, only_transfer)
      appendFactorRow(i); swapProblem(nC, i); M.C[nC] = nC; nC++;
    } else {
      for (;;) {
        const int dir = (M.w[i] <= 0) ? 1 : -1;
        const double dirf = dir;
        solve1(i, dir);
        // dw(N) = A(N,C) dx(C) +/- A(i,N);  dw[i] = A(i,C) dx(C) + A(i,i) dirf   (lcp.cpp:926-928)
        for (int k = 0; k < nN; k++) M.dw[nC + k] = dotRows(&AA(nC + k, 0), M.dx, nC);
        if (dir > 0) for (int k = 0; k < nN; k++) M.dw[nC + k] += AA(i, nC + k);
        else for (int k = 0; k < nN; k++) M.dw[nC + k] -= AA(i, nC + k);
        M.dw[i] = dotRows(&AA(i, 0), M.dx, nC) + AA(i, i) * dirf;
        // step length: the first minimum in the reference's scan order (lcp.cpp:938-998)
        int cmd = 1, si = 0;
        double s = -M.w[i] / M.dw[i];
        if (dir > 0) {
          if (M.hi[i] < INFINITY) { const double s2 = (M.hi[i] - M.x[i]) * dirf; if (s2 < s) { s = s2; cmd = 3; } }
        } else {
          if (M.lo[i] > -INFINITY) { const double s2 = (M.lo[i] - M.x[i]) * dirf; if (s2 < s) { s = s2; cmd = 2; } }
        }
        for (int k = 0; k < nN; ++k) {
          const int r = nC + k;
          if (!M.state[r] ? M.dw[r] < 0 : M.dw[r] > 0) {
            if (M.lo[r] == 0 && M.hi[r] == 0) continue;
            const double s2 = -M.w[r] / M.dw[r];
            if (s2 < s) { s = s2; cmd = 4; si = r; }
          }
        }
        for (int k = 0; k < nC; ++k) {
          if (M.dx[k] < 0 && M.lo[k] > -INFINITY) { const double s2 = (M.lo[k] - M.x[k]) / M.dx[k]; if (s2 < s) { s = s2; cmd = 5; si = k; } }
          if (M.dx[k] > 0 && M.hi[k] < INFINITY) { const double s2 = (M.hi[k] - M.x[k]) / M.dx[k]; if (s2 < s) { s = s2; cmd = 6; si = k; } }
        }
        if (s != s) return -1;      // (the wavefront-cooperative driver reports a NaN step the same way: coopDantzig)
        if (s <= 0.0) return 0;     // earlyTermination (the caller always has the PGS fallback, BoxedLcpConstraintSolver.cpp:463)
        // apply the step (lcp.cpp:1031-1036)
        for (int k = 0; k < nC; k++) M.x[k] += s * M.dx[k];
        M.x[i] += s * dirf;
        for (int k = 0; k < nN; k++) M.w[nC + k] += s * M.dw[nC + k];
        M.w[i] += s * M.dw[i];
        switch (cmd) {
          case 1: M.w[i] = 0; appendFactorRow(i); swapProblem(nC, i); M.C[nC] = nC; nC++; break;   // ell / Dell of solve1(i)
          case 2: M.x[i] = M.lo[i]; M.state[i] = 0; nN++; break;
          case 3: M.x[i] = M.hi[i]; M.state[i] = 1; nN++; break;
          case 4:                                                                                  // transfer_i_from_N_to_C
            M.w[si] = 0;
            if (nC > 0) solveEll(si);
            appendFactorRow(si); swapProblem(nC, si); M.C[nC] = nC; nN--; nC++; break;
          case 5: M.x[si] = M.lo[si]; M.state[si] = 0; removeFromC(si); break;
          case 6: M.x[si] = M.hi[si]; M.state[si] = 1; removeFromC(si); break;
        }
        if (cmd <= 3) break;
      }
    }
  }
  for (int j = 0; j < n; ++j) xOut[M.p[j]] = M.x[j];    // unpermute
  return 1;
}

// ---- the same driver with the wavefront sharing the work (round 6) ---------------------------------------------------------------------
// genDantzigSeq above is ONE instruction stream: on the device it ran on lane 0 of a 64-lane wavefront with its matrices in HBM scratch
// (3.8 ms per 1024-world launch on the metric worlds: the general build's whole step was 9 % of the 24-row build's).  genDantzigPar is the
// same algorithm, the same arithmetic IN THE SAME ORDER PER NUMBER - every floating-point expression of the sequential statement is formed
// by exactly one lane, with the same operands, association and rounding - with the loops that have independent iterations strided over
// the lanes of the policy W and the loops that are one chain of dependent additions (dDot's running sum, the column order of dSolveL1 /
// dSolveL1T) cut where the chain allows it:
//   * dDot: the products are independent (formed in parallel, rounded, parked in `prod`), the running sum is one chain (every lane adds
//     the parked products in index order and gets the same bits - no broadcast needed);
//   * dSolveL1 / dSolveL1T: the reference sums, for a row of a 4-block, the columns before the block in column order (Z), forms
//     y = B - Z and subtracts the block's own columns one by one; single rows sum everything into Z.  Column by column 4-block: ONE lane
//     solves the block's 4 x 4 corner (ten dependent operations), then EVERY later row adds its four terms to its Z, in column order -
//     the additions of a row happen in exactly the reference's order, only rows no longer wait for each other;
//   * products A(N,C) dx(C): a lane per row (each a complete dDot in index order); step-length scan: a lane-strided arg-min with the
//     scan position as the tie-break (the reference keeps the FIRST minimum: strict <); dLDLTAddTL: the scalar recurrence on every lane,
//     the rows below strided over the lanes; dRemoveRowCol through a second matrix (an in-place shift by lanes would race).
// Under a one-lane policy (tests/host_shim: HostWave1) this text IS a sequential program; tests/test_gen_host.py pins it against the
// reference's compiled dSolveLCP with one lane and with several (threads + barriers), tests/test_gpu_general.py on the device.
// prod / zacc / w1 / w2: four scratch vectors of n doubles the lanes share (LDS on the device); T: a scratch matrix (ld x ld).
template <class W>
DEV int genDantzigPar(const W& wv, const GenDantzigMem& M, int n, double* xOut, double* prod, double* zacc, double* W1, double* W2, double* T) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const int ln = wv.lane(), NL = wv.lanes();
  double* A = M.A; double* L = M.L;
  const size_t ld = (size_t)M.ld, ldL = (size_t)M.ldL;
  auto AA = [&](int i, int j) -> double& { return A[(size_t)i * ld + j]; };
  auto LL = [&](int i, int j) -> double& { return L[(size_t)i * ldL + j]; };
  // the running sum of v[0 .. cnt) from +0.0 in index order (fastdot.cpp) - formed by every lane: the same bits everywhere
  // (four terms fetched before the four dependent additions - gen_lcp_dev.hpp::genFmaSeq says why: a plain loop waits for every load)
  auto runSum = [&](const double* v, int cnt) -> double {
    double s = 0.0;
    int k = 0;
    for (; k + 3 < cnt; k += 4) { const double v0 = v[k], v1 = v[k + 1], v2 = v[k + 2], v3 = v[k + 3]; s = s + v0; s = s + v1; s = s + v2; s = s + v3; }
    for (; k < cnt; k++) s = s + v[k];
    return s;
  };
  // dDot(a, b, cnt), uniform result.  (prod is free again when the call returns.)
  auto dotAll = [&](const double* a, const double* b, int cnt) -> double {
    for (int k = ln; k < cnt; k += NL) prod[k] = a[k] * b[k];
    wv.sync();
    const double s = runSum(prod, cnt);
    wv.sync();
    return s;
  };
  int nC = 0, nN = 0;
  for (int k = ln; k < n; k += NL) { M.p[k] = k; M.x[k] = 0.0; M.w[k] = 0.0; M.state[k] = 0; M.dx[k] = 0.0; M.dw[k] = 0.0; }
  // symmetrise from the lower triangle (lcp.cpp:138-140)
  for (int idx = ln; idx < n * n; idx += NL) { const int i = idx / n, j = idx - i * n; if (j > i) AA(i, j) = AA(j, i); }
  wv.sync();
  auto swapProblem = [&](int i1, int i2) {     // (uniform arguments)
    if (i1 == i2) return;
    for (int k = ln; k < n; k += NL) { const double t = AA(i1, k); AA(i1, k) = AA(i2, k); AA(i2, k) = t; }
    wv.sync();
    for (int k = ln; k < n; k += NL) { const double t = AA(k, i1); AA(k, i1) = AA(k, i2); AA(k, i2) = t; }
    if (ln == 0) {
      double t;
      t = M.x[i1]; M.x[i1] = M.x[i2]; M.x[i2] = t;
      t = M.b[i1]; M.b[i1] = M.b[i2]; M.b[i2] = t;
      t = M.w[i1]; M.w[i1] = M.w[i2]; M.w[i2] = t;
      t = M.lo[i1]; M.lo[i1] = M.lo[i2]; M.lo[i2] = t;
      t = M.hi[i1]; M.hi[i1] = M.hi[i2]; M.hi[i2] = t;
      int ti;
      ti = M.p[i1]; M.p[i1] = M.p[i2]; M.p[i2] = ti;
      ti = M.state[i1]; M.state[i1] = M.state[i2]; M.state[i2] = ti;
      ti = M.findex[i1]; M.findex[i1] = M.findex[i2]; M.findex[i2] = ti;
    }
    wv.sync();
  };
  // every findex row goes to the end (lcp.cpp:487-498)
  {
    int atEnd = 0;
    for (int k = n - 1; k >= 0; k--)
      if (M.findex[k] >= 0) { swapProblem(k, n - 1 - atEnd); atEnd++; }
  }
  // dSolveL1 / dSolveL1T on B (cnt rows).  REV: the reversed index of dSolveL1T (row rr <-> entry cnt - 1 - rr, L^T between reversed indices).
  auto solveTri = [&](double* B, int cnt, bool REV) {
    auto Lt = [&](int rr, int jj) -> double { return REV ? LL(cnt - 1 - jj, cnt - 1 - rr) : LL(rr, jj); };   // jj < rr
    auto Bv = [&](int rr) -> double& { return B[REV ? cnt - 1 - rr : rr]; };
    const int nb4 = cnt & ~3;
    for (int r = ln; r < cnt; r += NL) zacc[r] = 0.0;
    wv.sync();
    for (int jb = 0; jb < nb4; jb += 4) {
      if (ln == 0) {      // the block's own corner: y = B - Z, then its columns one by one
        const double y0 = Bv(jb) - zacc[jb];
        Bv(jb) = y0;
        const double y1 = Bv(jb + 1) - zacc[jb + 1] - Lt(jb + 1, jb) * y0;
        Bv(jb + 1) = y1;
        const double y2 = Bv(jb + 2) - zacc[jb + 2] - Lt(jb + 2, jb) * y0 - Lt(jb + 2, jb + 1) * y1;
        Bv(jb + 2) = y2;
        const double y3 = Bv(jb + 3) - zacc[jb + 3] - Lt(jb + 3, jb) * y0 - Lt(jb + 3, jb + 1) * y1 - Lt(jb + 3, jb + 2) * y2;
        Bv(jb + 3) = y3;
      }
      wv.sync();
      if (jb + 4 < cnt) {
        const double q0 = Bv(jb), q1 = Bv(jb + 1), q2 = Bv(jb + 2), q3 = Bv(jb + 3);
        for (int r = jb + 4 + ln; r < cnt; r += NL) {      // every later row: its Z takes the four terms in column order
          double z = zacc[r];
          z = z + Lt(r, jb) * q0; z = z + Lt(r, jb + 1) * q1; z = z + Lt(r, jb + 2) * q2; z = z + Lt(r, jb + 3) * q3;
          zacc[r] = z;
        }
        wv.sync();
      }
    }
    for (int j = nb4; j < cnt; j++) {                       // single rows: everything before them went into Z
      if (ln == 0) Bv(j) = Bv(j) - zacc[j];
      wv.sync();
      if (j + 1 < cnt) {
        const double q = Bv(j);
        for (int r = j + 1 + ln; r < cnt; r += NL) zacc[r] = zacc[r] + Lt(r, j) * q;
        wv.sync();
      }
    }
  };
  // Dell = L^-1 A(i, C[.]), ell = Dell * d      (first half of dLCP::solve1, lcp.cpp:700-730)
  auto solveEll = [&](int i) {
    for (int j = ln; j < nC; j += NL) M.Dell[j] = AA(i, M.C[j]);
    wv.sync();
    solveTri(M.Dell, nC, false);
    for (int j = ln; j < nC; j += NL) M.ell[j] = M.Dell[j] * M.d[j];
    wv.sync();
  };
  auto solve1 = [&](int i, int dir) {
    if (nC == 0) return;
    solveEll(i);
    for (int j = ln; j < nC; j += NL) M.tmp[j] = M.ell[j];
    wv.sync();
    solveTri(M.tmp, nC, true);
    if (dir > 0) { for (int j = ln; j < nC; j += NL) M.dx[M.C[j]] = -M.tmp[j]; }
    else { for (int j = ln; j < nC; j += NL) M.dx[M.C[j]] = M.tmp[j]; }
    wv.sync();
  };
  // the row at position i (with ell / Dell of the last solveEll(i)) becomes factor row nC   (transfer_i_to_C, lcp.cpp:520-553)
  auto appendFactorRow = [&](int i) {
    if (nC > 0) {
      for (int j = ln; j < nC; j += NL) LL(nC, j) = M.ell[j];
      const double dot = dotAll(M.ell, M.Dell, nC);
      if (ln == 0) M.d[nC] = 1.0 / (AA(i, i) - dot);
    } else if (ln == 0) M.d[0] = 1.0 / AA(i, i);
    wv.sync();
  };
  // dLDLTAddTL (matrix.cpp:286-359) on the trailing block of the factor that starts at row / column r0 (n2 rows) with the vector a
  auto ldltAddTL = [&](int r0, int n2, const double* a) {
    if (n2 < 2) return;
    const double SQ = 0.70710678118654752440;   // M_SQRT1_2
    for (int j = ln; j < n2; j += NL) { const double v = j == 0 ? 0.0 : a[j] * SQ; W1[j] = v; W2[j] = v; }
    const double a0 = a[0];
    wv.sync();
    const double W11 = (0.5 * a0 + 1.0) * SQ, W21 = (0.5 * a0 - 1.0) * SQ;
    double alpha1 = 1.0, alpha2 = 1.0;
    {
      double dee = M.d[r0];
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
      for (int pp = 1 + ln; pp < n2; pp += NL) {
        const double Wp = W1[pp];
        const double el = LL(r0 + pp, r0);
        W1[pp] = Wp - W11 * el;
        W2[pp] = k1 * Wp + k2 * el;
      }
      wv.sync();
    }
    for (int j = 1; j < n2; j++) {
      const double k1 = W1[j], k2 = W2[j];
      double dee = M.d[r0 + j];
      double alphanew = alpha1 + (k1 * k1) * dee;
      dee /= alphanew;
      const double gamma1 = k1 * dee;
      dee *= alpha1;
      alpha1 = alphanew;
      alphanew = alpha2 - (k2 * k2) * dee;
      dee /= alphanew;
      const double gamma2 = k2 * dee;
      dee *= alpha2;
      alpha2 = alphanew;
      wv.sync();                               // (every lane has read d[r0 + j] and W1[j] / W2[j] before they change)
      if (ln == 0) M.d[r0 + j] = dee;
      for (int pp = j + 1 + ln; pp < n2; pp += NL) {
        double el = LL(r0 + pp, r0 + j);
        double Wp = W1[pp] - k1 * el;
        el += gamma1 * Wp;
        W1[pp] = Wp;
        Wp = W2[pp] - k2 * el;
        el -= gamma2 * Wp;
        W2[pp] = Wp;
        LL(r0 + pp, r0 + j) = el;
      }
      wv.sync();
    }
  };
  // dLDLTRemove (matrix.cpp:374-426): factor row / column r leaves the n2-row factor; then dRemoveRowCol snips it out of L and d
  auto ldltRemove = [&](int n2, int r) {
    if (r == n2 - 1) return;    // deleting the last row / column is easy
    double* a = M.tvec + n2;
    if (r == 0) {
      const int p0 = M.C[0];
      for (int i = ln; i < n2; i += NL) { double v = -AA(M.C[i], p0); if (i == 0) v += 1.0; a[i] = v; }      // GETA(p[i], p[0])
      wv.sync();
      ldltAddTL(0, n2, a);
    } else {
      double* t = M.tvec;
      for (int i = ln; i < r; i += NL) t[i] = LL(r, i) / M.d[i];
      wv.sync();
      const int pr = M.C[r];
      for (int i = ln; i < n2 - r; i += NL) {
        double sdot = 0.0;
        {                                                                         // dDot(L[r + i], t, r)
          int k = 0;
          for (; k + 3 < r; k += 4) {
            const double l0 = LL(r + i, k), l1 = LL(r + i, k + 1), l2 = LL(r + i, k + 2), l3 = LL(r + i, k + 3);
            const double t0 = t[k], t1 = t[k + 1], t2 = t[k + 2], t3 = t[k + 3];
            sdot = sdot + l0 * t0; sdot = sdot + l1 * t1; sdot = sdot + l2 * t2; sdot = sdot + l3 * t3;
          }
          for (; k < r; k++) sdot = sdot + LL(r + i, k) * t[k];
        }
        double v = sdot - AA(M.C[r + i], pr);
        if (i == 0) v += 1.0;
        a[i] = v;
      }
      wv.sync();
      ldltAddTL(r, n2 - r, a);
    }
    // dRemoveRowCol(L, n2, r): rows below r move up, columns right of r move left - through T (rows above r do not change: j <= i < r)
    for (int i = r; i < n2 - 1; i++)
      for (int j = ln; j <= i; j += NL) T[(size_t)i * ld + j] = LL(i + 1, j >= r ? j + 1 : j);
    double dnext = 0.0;
    wv.sync();
    for (int i = r; i < n2 - 1; i++)
      for (int j = ln; j <= i; j += NL) LL(i, j) = T[(size_t)i * ld + j];
    // (d: every entry from r on takes its right neighbour - read, barrier, write; chunks of NL entries in ascending order, so a chunk's
    //  sources are read before any lane overwrites them)
    for (int i0 = r; i0 < n2 - 1; i0 += NL) {
      const int i = i0 + ln;
      if (i < n2 - 1) dnext = M.d[i + 1];
      wv.sync();
      if (i < n2 - 1) M.d[i] = dnext;
      wv.sync();
    }
    wv.sync();
  };
  // transfer_i_from_C_to_N (lcp.cpp:603-650): position i leaves C.  (C holds every position 0 .. nC-1 once: the reference's search for
  // nC - 1 before or after i's entry finds THE entry that holds it.)
  auto removeFromC = [&](int i) {
    int jmine = 0x7fffffff, kmine = 0x7fffffff;
    for (int q = ln; q < nC; q += NL) { const int c = M.C[q]; if (c == i) jmine = q; if (c == nC - 1) kmine = q; }
    const int j = wv.minAllI(jmine), k = wv.minAllI(kmine);
    ldltRemove(nC, j);
    if (ln == 0) M.C[k] = M.C[j];
    wv.sync();
    for (int q0 = j; q0 + 1 < nC; q0 += NL) {     // C[q] = C[q + 1] from j on (read, barrier, write; ascending chunks)
      const int q = q0 + ln;
      int cn = 0;
      if (q + 1 < nC) cn = M.C[q + 1];
      wv.sync();
      if (q + 1 < nC) M.C[q] = cn;
      wv.sync();
    }
    swapProblem(i, nC - 1);
    nN++; nC--;
  };
  bool hitFirstFriction = false;
  for (int i = 0; i < n; ++i) {
    if (!hitFirstFriction && M.findex[i] >= 0) {
      // un[p[j]] = x[j]; bounds of the friction rows frozen from the solved normals (lcp.cpp:856-873)
      for (int j = ln; j < n; j += NL) M.dw[M.p[j]] = M.x[j];
      wv.sync();
      for (int k = i + ln; k < n; k += NL) {
        const double wfk = M.dw[M.findex[k]];
        if (wfk == 0) { M.hi[k] = 0; M.lo[k] = 0; }
        else { M.hi[k] = fabs(M.hi[k] * wfk); M.lo[k] = -M.hi[k]; }
      }
      wv.sync();
      hitFirstFriction = true;
    }
    // w[i] = A(i,C) x(C) + A(i,N) x(N) - b[i]: two running sums (lcp.cpp:877)
    double wi;
    {
      for (int k = ln; k < nC + nN; k += NL) prod[k] = AA(i, k) * M.x[k];
      wv.sync();
      const double sC = runSum(prod, nC), sN = runSum(prod + nC, nN);
      wi = sC + sN - M.b[i];
      wv.sync();
      if (ln == 0) M.w[i] = wi;
    }
    const double loI = M.lo[i], hiI = M.hi[i];
    if (loI == 0 && wi >= 0) { nN++; if (ln == 0) M.state[i] = 0; wv.sync(); }
    else if (hiI == 0 && wi <= 0) { nN++; if (ln == 0) M.state[i] = 1; wv.sync(); }
    else if (wi == 0) {
      wv.sync();
      if (nC > 0) solveEll(i);                 // solve1(delta_x, i, 0, only_transfer)
      appendFactorRow(i); swapProblem(nC, i); if (ln == 0) M.C[nC] = nC; wv.sync(); nC++;
    } else {
      wv.sync();
      for (;;) {
        const double wcur = M.w[i];
        const int dir = (wcur <= 0) ? 1 : -1;
        const double dirf = dir;
        solve1(i, dir);
        // dw(N) = A(N,C) dx(C) +/- A(i,N);  dw[i] = A(i,C) dx(C) + A(i,i) dirf   (lcp.cpp:926-928): a lane per row
        for (int k = ln; k <= nN; k += NL) {
          const int row = k < nN ? nC + k : i;
          double sdot = 0.0;
          for (int q = 0; q < nC; q++) sdot = sdot + AA(row, q) * M.dx[q];
          if (k < nN) M.dw[row] = dir > 0 ? sdot + AA(i, row) : sdot - AA(i, row);
          else M.dw[i] = sdot + AA(i, i) * dirf;
        }
        wv.sync();
        // step length: the first minimum in the reference's scan order (lcp.cpp:938-998): the driving row's own events (position 0), the N
        // rows (1 + k), the C rows (1 + nN + k)
        const double xI = M.x[i];
        int cmd0 = 1;
        double s0 = -wcur / M.dw[i];
        if (dir > 0) {
          if (hiI < INFINITY) { const double s2 = (hiI - xI) * dirf; if (s2 < s0) { s0 = s2; cmd0 = 3; } }
        } else {
          if (loI > -INFINITY) { const double s2 = (loI - xI) * dirf; if (s2 < s0) { s0 = s2; cmd0 = 2; } }
        }
        if (s0 != s0) return -1;      // (the reference carries the NaN through its scans: nothing compares below it)
        double best = INFINITY;
        int bestPos = 0x7fffffff;
        if (ln == 0) { best = s0; bestPos = 0; }
        for (int k = ln; k < nN; k += NL) {
          const int r = nC + k;
          const double dwr = M.dw[r];
          if (!M.state[r] ? dwr < 0 : dwr > 0) {
            if (M.lo[r] == 0 && M.hi[r] == 0) continue;
            const double s2 = -M.w[r] / dwr;
            if (s2 < best) { best = s2; bestPos = 1 + k; }
          }
        }
        for (int k = ln; k < nC; k += NL) {
          const double dxk = M.dx[k];
          if (dxk < 0 && M.lo[k] > -INFINITY) { const double s2 = (M.lo[k] - M.x[k]) / dxk; if (s2 < best) { best = s2; bestPos = 1 + nN + k; } }
          if (dxk > 0 && M.hi[k] < INFINITY) { const double s2 = (M.hi[k] - M.x[k]) / dxk; if (s2 < best) { best = s2; bestPos = 1 + nN + k; } }
        }
        // (a lane visits its positions in ascending order with a strict <: it holds its FIRST minimum; lane 0's includes position 0)
        const double s = -wv.maxAll(-best);
        const int pos = wv.minAllI(best == s ? bestPos : 0x7fffffff);
        int cmd, si = 0;
        if (pos == 0) cmd = cmd0;
        else if (pos <= nN) { cmd = 4; si = nC + pos - 1; }
        else { si = pos - 1 - nN; cmd = M.dx[si] < 0 ? 5 : 6; }
        if (s <= 0.0) return 0;     // earlyTermination (the caller always has the PGS fallback, BoxedLcpConstraintSolver.cpp:463)
        // apply the step (lcp.cpp:1031-1036)
        for (int k = ln; k < nC; k += NL) M.x[k] += s * M.dx[k];
        for (int k = ln; k < nN; k += NL) M.w[nC + k] += s * M.dw[nC + k];
        wv.sync();
        if (ln == 0) { M.x[i] += s * dirf; M.w[i] += s * M.dw[i]; }
        wv.sync();
        switch (cmd) {
          case 1: if (ln == 0) M.w[i] = 0; wv.sync(); appendFactorRow(i); swapProblem(nC, i); if (ln == 0) M.C[nC] = nC; wv.sync(); nC++; break;   // ell / Dell of solve1(i)
          case 2: if (ln == 0) { M.x[i] = M.lo[i]; M.state[i] = 0; } wv.sync(); nN++; break;
          case 3: if (ln == 0) { M.x[i] = M.hi[i]; M.state[i] = 1; } wv.sync(); nN++; break;
          case 4:                                                                                  // transfer_i_from_N_to_C
            if (ln == 0) M.w[si] = 0;
            wv.sync();
            if (nC > 0) solveEll(si);
            appendFactorRow(si); swapProblem(nC, si); if (ln == 0) M.C[nC] = nC; wv.sync(); nN--; nC++; break;
          case 5: if (ln == 0) { M.x[si] = M.lo[si]; M.state[si] = 0; } wv.sync(); removeFromC(si); break;
          case 6: if (ln == 0) { M.x[si] = M.hi[si]; M.state[si] = 1; } wv.sync(); removeFromC(si); break;
        }
        if (cmd <= 3) break;
      }
    }
  }
  for (int j = ln; j < n; j += NL) xOut[M.p[j]] = M.x[j];    // unpermute
  wv.sync();
  return 1;
}

// ---- stages 1-3 of BoxedLcpConstraintSolver::solveLcp (:461-677) ------------------------------------------------------------------------
constexpr int GS_SOLVED = 1;   // the solver reported success (Dantzig: no early termination; PGS: converged)
constexpr int GS_VALID = 2;    // ... and isLCPSolutionValid accepted it
constexpr int GS_NAN = 4;      // Dantzig: NaN step length

// The stride of the cascade's scratch vectors: the model's rows in HBM scratch; in the fast pool (LDS) the WORLD's rows rounded up to 8 -
// sixteen vectors of an eight-contact world then take 384 of the pool's 1152 doubles whatever the model's slot count is, and the rest
// holds the Dantzig factor (genCarve) or the matrix of the Gauss-Seidel sweeps (genPgsAT).
DEV int genVecStride(const GenScratch& S, int m) { return S.vecFast ? ((m + 7) & ~7) : S.ld; }
// carve the problem arrays and the Dantzig arrays out of the world's scratch (mat[4]: the problem's matrix, mat[1]: the factor - or, for a
// world small enough, the factor PACKED (leading dimension m) behind the vectors in the fast pool: the two triangular solves and the
// factor updates of every pivot read and write it through barriers, and in HBM scratch each of those waited for memory - 21 k cycles per
// pivot against the 24-row build's 8 k).  m: the world's rows.
DEV void genCarve(const GenScratch& S, GenProblem& P, GenDantzigMem& D, int m) {
  double* v = S.vec;
  const int g = genVecStride(S, m);
  P.ld = S.ld; D.ld = S.ld;
  P.A = S.mat[4]; P.x = v; P.b = v + g; P.lo = v + 2 * g; P.hi = v + 3 * g;
  P.findex = reinterpret_cast<int*>(v + 4 * g); P.mapTo = P.findex + g;
  D.A = S.mat[4]; D.L = S.mat[1]; D.ldL = S.ld;
  if (S.vecFast && (size_t)16 * g + (size_t)m * m <= (size_t)S.vecDoubles) { D.L = v + 16 * g; D.ldL = m; }
  // (v + 5 g: the candidate of a stage, genCascade; the Dantzig driver's vectors come last: Gauss-Seidel takes their place, genPgsAT)
  D.d = v + 6 * g; D.x = v + 7 * g; D.w = v + 8 * g; D.dx = v + 9 * g; D.dw = v + 10 * g; D.ell = v + 11 * g; D.Dell = v + 12 * g;
  D.tmp = v + 13 * g; D.tvec = S.mat[2]; D.W1 = S.mat[2] + 2 * S.ld; D.W2 = S.mat[2] + 3 * S.ld;
  D.b = P.b; D.lo = P.lo; D.hi = P.hi; D.findex = P.findex;
  D.p = reinterpret_cast<int*>(v + 14 * g); D.C = D.p + g; D.state = reinterpret_cast<int*>(v + 15 * g);
}
// The scaled, transposed matrix of the Gauss-Seidel sweeps: packed in the part of the fast vector region the Dantzig driver's vectors
// occupy (v + 6 g to the end: they are dead while Gauss-Seidel runs) when the problem is small enough, else the scratch matrix.
struct GenPgsAT { double* AT; int ld; bool lds; };
DEV GenPgsAT genPgsAT(const GenScratch& S, int n, int m) {
  GenPgsAT a;
  const int g = genVecStride(S, m);
  if (S.vecFast && (size_t)n * n + (size_t)6 * g <= (size_t)S.vecDoubles) { a.AT = S.vec + 6 * g; a.ld = n; a.lds = true; }
  else if (S.fast && S.fastMats >= 1 && n <= S.fastN) { a.AT = S.fast; a.ld = S.fastN; a.lds = true; }
  else { a.AT = S.mat[1]; a.ld = S.ld; a.lds = false; }
  return a;
}

// X[o] = x_reduced[mapTo[o]] -> out (rows that are off: 0)
template <class W>
DEV void genMapOut(const W& w, const GenRows& R, const GenProblem& P, const double* xred, double* out) {
  for (int r = w.lane(); r < R.m; r += w.lanes()) out[r] = P.mapTo[r] >= 0 ? xred[P.mapTo[r]] : 0.0;
  w.sync();
}

// stage 1: reduce + Dantzig with early termination (:461-522).  out: the candidate (m rows), returns GS_* flags
template <class W>
DEV int genStage1(const W& w, const double* A, int lda, GenRows& R, const GenScratch& S, double* out) {
  GenProblem P; GenDantzigMem D;
  GEN_T0();
  genCarve(S, P, D, R.m);
  genLoadProblem(w, A, lda, R, 0.0, R.X0, P);
  genLcpReduce(w, R, P, R.m);
  GEN_T(4);
  int rc;
  if (S.fast && S.fastMats >= GEN_FAST_MATS && P.n <= S.fastN) {
    // a small problem: matrix, factor, the removal's scratch matrix and every vector of the driver in the fast pool
    const int N = S.fastN, n = P.n;
    GenDantzigMem F;
    double* v = S.fast + (size_t)GEN_FAST_MATS * N * N;
    F.ld = N; F.ldL = N; F.A = S.fast; F.L = S.fast + (size_t)N * N;
    F.d = v; F.x = v + N; F.w = v + 2 * N; F.dx = v + 3 * N; F.dw = v + 4 * N; F.ell = v + 5 * N; F.Dell = v + 6 * N; F.tmp = v + 7 * N;
    F.b = v + 8 * N; F.lo = v + 9 * N; F.hi = v + 10 * N; F.tvec = v + 11 * N;                    // (tvec: 2 N)
    F.W1 = nullptr; F.W2 = nullptr;
    F.p = reinterpret_cast<int*>(v + 13 * N); F.C = F.p + N; F.state = reinterpret_cast<int*>(v + 14 * N); F.findex = F.state + N;
    for (int idx = w.lane(); idx < n * n; idx += w.lanes()) { const int i = idx / n, j = idx - i * n; F.A[(size_t)i * N + j] = P.A[(size_t)i * P.ld + j]; }
    for (int k = w.lane(); k < n; k += w.lanes()) { F.b[k] = P.b[k]; F.lo[k] = P.lo[k]; F.hi[k] = P.hi[k]; F.findex[k] = P.findex[k]; }
    w.sync();
    rc = genDantzigPar(w, F, n, P.x, R.t0, R.t1, R.t2, R.t3, S.fast + (size_t)2 * N * N);
  } else {
    rc = genDantzigPar(w, D, P.n, P.x, R.t0, R.t1, R.t2, R.t3, S.mat[0]);
  }
  GEN_T(5);
  int flags = 0;
  for (int r = w.lane(); r < R.m; r += w.lanes()) out[r] = 0.0;
  w.sync();
  if (rc == 1) {
    genMapOut(w, R, P, P.x, out);
    flags = GS_SOLVED | (genValid(w, A, lda, R, out, false, 0.0, R.t2) ? GS_VALID : 0);
  } else if (rc < 0) flags = GS_NAN;
  GEN_T(6);
  return flags;
}
// stage 2: CFM + reduce + PGS from the pre-solve x (:539-597)
template <class W>
DEV int genStage2(const W& w, const double* A, int lda, GenRows& R, const GenScratch& S, double cfm, double* out) {
  GenProblem P; GenDantzigMem D;
  genCarve(S, P, D, R.m);
  genLoadProblem(w, A, lda, R, cfm, R.X0, P);
  genLcpReduce(w, R, P, R.m);
  int flags = 0;
  for (int r = w.lane(); r < R.m; r += w.lanes()) out[r] = 0.0;
  w.sync();
  const GenPgsAT at = genPgsAT(S, P.n, R.m);
  if (genPgs(w, R, P, at.AT, at.ld, at.lds)) {
    genMapOut(w, R, P, P.x, out);
    flags = GS_SOLVED | (genValid(w, A, lda, R, out, false, cfm, R.t2) ? GS_VALID : 0);
  }
  return flags;
}
// stage 3: drop friction, PGS from zero (:606-677); its result is used whatever the solver says
template <class W>
DEV int genStage3(const W& w, const double* A, int lda, GenRows& R, const GenScratch& S, double cfm, double* out) {
  GenProblem P; GenDantzigMem D;
  genCarve(S, P, D, R.m);
  genLoadProblem(w, A, lda, R, cfm, R.X0, P);
  genLcpRemoveFriction(w, R, P, R.m, S.mat[1]);
  for (int c = w.lane(); c < P.n; c += w.lanes()) P.x[c] = 0.0;
  w.sync();
  const GenPgsAT at = genPgsAT(S, P.n, R.m);
  const bool ok3 = genPgs(w, R, P, at.AT, at.ld, at.lds);
  genMapOut(w, R, P, P.x, out);
  return ok3 ? GS_SOLVED : 0;
}

// The cascade for the rows that are on: the stages in the reference's order of preference (a later stage only runs when the earlier ones
// did not deliver), then registration, classification and standardisation of the chosen solution (:718-736).  R.X0: the pre-solve x;
// out: R.X (impulses), R.cls / R.E, cfmOut, st (NBL_ST_* bits), pinvValid (S.mat[3] is Q^+ of the classification).
template <class W>
DEV void genCascade(const W& w, const double* A, int lda, GenRows& R, const GenScratch& S, double fallbackCfm, double& cfmOut, uint32_t& stOut,
                    bool& pinvValid, GenClasses& K) {
  const int m = R.m;
  double* cand = S.vec + 5 * genVecStride(S, m);
  auto hasNan = [&](const double* x) -> bool { bool b = false; for (int r = w.lane(); r < m; r += w.lanes()) if (x[r] != x[r]) b = true; return w.anyAll(b); };
  auto take = [&](const double* x) { for (int r = w.lane(); r < m; r += w.lanes()) R.X[r] = R.on[r] ? x[r] : 0.0; w.sync(); };
  bool success = false, ignoreFriction = false;
  uint32_t st = 0;
  double cfm = 0.0;
  take(R.X0);
  GEN_CNT(11);
  const int f1 = genStage1(w, A, lda, R, S, cand);
  GEN_T0();
  if (f1 & GS_SOLVED) {
    take(cand);
    success = (f1 & GS_VALID) != 0;
    if (success) st |= 0x4u;
  }
  if ((f1 & GS_NAN) || hasNan(R.X)) { success = false; for (int r = w.lane(); r < m; r += w.lanes()) R.X[r] = 0.0; w.sync(); st |= 0x40u; }
  if (!success) {
    cfm = fallbackCfm;
    const int f2 = genStage2(w, A, lda, R, S, fallbackCfm, cand);
    if (f2 & GS_SOLVED) {
      take(cand);
      success = (f2 & GS_VALID) != 0;
      if (success) st |= 0x8u;
    }
  }
  GEN_T(7);
  if (!success) {
    ignoreFriction = true;
    const int f3 = genStage3(w, A, lda, R, S, fallbackCfm, cand);
    take(cand);
    st |= 0x10u;
    if (!(f3 & GS_SOLVED)) st |= 0x20u;
  }
  if (hasNan(R.X)) { for (int r = w.lane(); r < m; r += w.lanes()) R.X[r] = 0.0; w.sync(); st |= 0x40u; }
  pinvValid = false;
  GEN_T(8);
  const bool std = genStandardizeLoop(w, A, lda, R, S, cfm, ignoreFriction, nullptr, pinvValid, K);
  GEN_T(9);
  if (std) st |= 0x100u;
  cfmOut = cfm; stOut = st;
}

}  // namespace NBL_NS
