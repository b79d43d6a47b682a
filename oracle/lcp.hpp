// oracle/lcp.hpp — TEST INFRASTRUCTURE ONLY.
//
// Boxed-LCP utilities restated from dart/constraint/LCPUtils.cpp and
// dart/constraint/PgsBoxedLcpSolver.cpp, a complete-orthogonal-decomposition least-squares solve
// standing in for Eigen's `completeOrthogonalDecomposition().solve()` (Eigen is not vendored in
// the reference tree), and a hook to the REAL vendored Dantzig solver: when oracle/_ref/libodelcp_ref.so
// exists (built from /root/reference/dart/external/odelcpsolver/*.cpp where they lie, see
// oracle/ref_build.py) stage 1 of the cascade calls the reference's own dSolveLCP.
// Attribution: the algorithm restated here derives from the Open Dynamics Engine (ODE), Copyright (C) 2001-2003 Russell L. Smith, which the
// reference vendors under ODE's BSD-style licence (dart/external/odelcpsolver/, dart/collision/dart/DARTCollide.cpp); this file is an
// independent restatement for another execution model - ODE's arithmetic order and, where the bit-for-bit tests need them recognisable,
// its identifiers are kept on purpose.
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <string>

#include "spatial.hpp"

namespace nbo {

// ---- Eigen::completeOrthogonalDecomposition().solve(b): minimum-norm least-squares solution ----
// Column-pivoted Householder QR, rank = #{ |R_kk| > eps * min(rows, cols) * max|R_kk| } (Eigen's default
// threshold), then the minimum-norm solution of the rank-r triangular system.
inline VecX codSolve(const MatX& Ain, const VecX& b, int* rankOut = nullptr) {
  const int m = Ain.r, n = Ain.c;
  VecX x(n, 0.0);
  if (m == 0 || n == 0) { if (rankOut) *rankOut = 0; return x; }
  MatX A = Ain;
  VecX c = b;
  std::vector<int> perm(n);
  for (int j = 0; j < n; j++) perm[j] = j;
  const int kmax = std::min(m, n);
  VecX colNorm(n, 0.0);
  for (int j = 0; j < n; j++) { s_t s = 0; for (int i = 0; i < m; i++) s += A(i, j) * A(i, j); colNorm[j] = s; }
  s_t maxPivot = 0;
  VecX diag(kmax, 0.0);
  for (int k = 0; k < kmax; k++) {
    // pivot: largest remaining column norm (recomputed exactly; sizes here are <= 24)
    int piv = k; s_t best = -1;
    for (int j = k; j < n; j++) { s_t s = 0; for (int i = k; i < m; i++) s += A(i, j) * A(i, j); colNorm[j] = s; if (s > best) { best = s; piv = j; } }
    if (piv != k) { for (int i = 0; i < m; i++) std::swap(A(i, k), A(i, piv)); std::swap(perm[k], perm[piv]); }
    // Householder on column k
    s_t normx = std::sqrt(best > 0 ? best : 0);
    if (normx == 0) { diag[k] = 0; continue; }
    s_t alpha = A(k, k) > 0 ? -normx : normx;
    VecX v(m, 0.0);
    for (int i = k; i < m; i++) v[i] = A(i, k);
    v[k] -= alpha;
    s_t vnorm2 = 0; for (int i = k; i < m; i++) vnorm2 += v[i] * v[i];
    if (vnorm2 > 0) {
      for (int j = k; j < n; j++) {
        s_t d = 0; for (int i = k; i < m; i++) d += v[i] * A(i, j);
        d = 2 * d / vnorm2;
        for (int i = k; i < m; i++) A(i, j) -= d * v[i];
      }
      s_t d = 0; for (int i = k; i < m; i++) d += v[i] * c[i];
      d = 2 * d / vnorm2;
      for (int i = k; i < m; i++) c[i] -= d * v[i];
    }
    diag[k] = A(k, k);
    maxPivot = std::max(maxPivot, std::fabs(diag[k]));
  }
  const s_t thresh = 2.220446049250313e-16 * kmax * maxPivot;
  int r = 0;
  for (int k = 0; k < kmax; k++) if (std::fabs(diag[k]) > thresh) r++; else break;
  if (rankOut) *rankOut = r;
  if (r == 0) return x;
  // R1 = A[0:r, 0:n] (upper trapezoidal), minimum-norm y with R1 y = c[0:r]:  y = R1^T (R1 R1^T)^-1 c1
  MatX RRt(r, r);
  for (int i = 0; i < r; i++)
    for (int j = 0; j < r; j++) { s_t s = 0; for (int k = std::max(i, j); k < n; k++) s += A(i, k) * A(j, k); RRt(i, j) = s; }
  MatX RRtInv;
  VecX y(n, 0.0);
  if (r == n) {  // full rank square/tall: plain back substitution
    for (int i = r - 1; i >= 0; i--) { s_t s = c[i]; for (int k = i + 1; k < n; k++) s -= A(i, k) * y[k]; y[i] = s / A(i, i); }
  } else {
    spdInverse(RRt, RRtInv);
    VecX z(r, 0.0);
    for (int i = 0; i < r; i++) { s_t s = 0; for (int j = 0; j < r; j++) s += RRtInv(i, j) * c[j]; z[i] = s; }
    for (int k = 0; k < n; k++) { s_t s = 0; for (int i = 0; i < r && i <= k; i++) s += A(i, k) * z[i]; y[k] = s; }
  }
  for (int k = 0; k < n; k++) x[perm[k]] = y[k];
  return x;
}

// ---- LCPUtils::isLCPSolutionValid (LCPUtils.cpp:12-80) ----
inline bool isLCPSolutionValid(const MatX& A, const VecX& X, const VecX& B, const VecX& Hi, const VecX& Lo,
                               const std::vector<int>& FIndex, bool ignoreFrictionIndices) {
  const int n = (int)X.size();
  VecX v = matvec(A, X);
  for (int i = 0; i < n; i++) v[i] -= B[i];
  for (int i = 0; i < n; i++) {
    s_t upperLimit = Hi[i], lowerLimit = Lo[i];
    if (FIndex[i] != -1) {
      if (ignoreFrictionIndices) { if (X[i] != 0) return false; continue; }
      upperLimit *= X[FIndex[i]];
      lowerLimit *= X[FIndex[i]];
    }
    const s_t tol = 1e-5;
    if (std::fabs(lowerLimit) < tol && std::fabs(upperLimit) < tol && std::fabs(X[i]) < tol) {
    } else if (std::fabs(X[i] - lowerLimit) < tol) { if (v[i] < -tol) return false; }
    else if (std::fabs(X[i] - upperLimit) < tol) { if (v[i] > tol) return false; }
    else if (X[i] > lowerLimit && X[i] < upperLimit) { if (std::fabs(v[i]) > tol) return false; }
    else return false;
  }
  return true;
}

// ---- LCPUtils::guessSolution (LCPUtils.cpp:86-140) ----
inline VecX guessSolution(const MatX& A, const VecX& B, const std::vector<int>& FIndex) {
  const int n = (int)B.size();
  std::vector<int> cl;
  for (int i = 0; i < n; i++) {
    if (FIndex[i] == -1) { if (B[i] > 0) cl.push_back(i); }
    else cl.push_back(i);
  }
  const int nc = (int)cl.size();
  if (nc == n) return codSolve(A, B);
  if (nc == 0) return VecX(n, 0.0);
  MatX rA(nc, nc);
  VecX rB(nc);
  for (int r = 0; r < nc; r++) { rB[r] = B[cl[r]]; for (int c = 0; c < nc; c++) rA(r, c) = A(cl[r], cl[c]); }
  VecX rX = codSolve(rA, rB);
  VecX full(n, 0.0);
  for (int i = 0; i < nc; i++) full[cl[i]] = rX[i];
  return full;
}

struct LcpProblem {
  MatX A;
  VecX x, b, hi, lo;
  std::vector<int> findex;
};

// LCPUtils::mergeLCPColumns (:346-440) / dropLCPColumn (:447-520) / reduce (:144-201) / removeFriction (:208-247)
inline void mergeColumns(int colA, int colB, LcpProblem& p, MatX& mapOut) {
  const int n = p.A.c;
  MatX newACols(n, n - 1), newMap(mapOut.r, n - 1);
  LcpProblem q;
  q.x.assign(n - 1, 0); q.b.assign(n - 1, 0); q.hi.assign(n - 1, 0); q.lo.assign(n - 1, 0); q.findex.assign(n - 1, 0);
  for (int i = 0; i < n; i++) {
    if (i == colB) { for (int r = 0; r < mapOut.r; r++) newMap(r, colA) += mapOut(r, i); continue; }
    int ni = i > colB ? i - 1 : i;
    for (int r = 0; r < n; r++) newACols(r, ni) = p.A(r, i) * (i == colA ? 2.0 : 1.0);
    q.x[ni] = p.x[i]; q.b[ni] = p.b[i]; q.hi[ni] = p.hi[i]; q.lo[ni] = p.lo[i];
    if (p.findex[i] < colB) q.findex[ni] = p.findex[i];
    else if (p.findex[i] == colB) q.findex[ni] = colA;
    else q.findex[ni] = p.findex[i] - 1;
    for (int r = 0; r < mapOut.r; r++) newMap(r, ni) += mapOut(r, i);
  }
  q.A = MatX(n - 1, n - 1);
  for (int i = 0; i < n; i++) {
    if (i == colB) continue;
    int ni = i > colB ? i - 1 : i;
    for (int c = 0; c < n - 1; c++) q.A(ni, c) = newACols(i, c);
  }
  p = q;
  mapOut = newMap;
}
inline void dropColumn(int col, LcpProblem& p, MatX& mapOut) {
  const int n = p.A.c;
  MatX newACols(n, n - 1), newMap(mapOut.r, n - 1);
  LcpProblem q;
  q.x.assign(n - 1, 0); q.b.assign(n - 1, 0); q.hi.assign(n - 1, 0); q.lo.assign(n - 1, 0); q.findex.assign(n - 1, 0);
  for (int i = 0; i < n; i++) {
    if (i == col) continue;
    int ni = i > col ? i - 1 : i;
    for (int r = 0; r < n; r++) newACols(r, ni) = p.A(r, i);
    q.x[ni] = p.x[i]; q.b[ni] = p.b[i]; q.hi[ni] = p.hi[i]; q.lo[ni] = p.lo[i];
    if (p.findex[i] < col) q.findex[ni] = p.findex[i];
    else if (p.findex[i] > col) q.findex[ni] = p.findex[i] - 1;
    for (int r = 0; r < mapOut.r; r++) newMap(r, ni) += mapOut(r, i);
  }
  q.A = MatX(n - 1, n - 1);
  for (int i = 0; i < n; i++) {
    if (i == col) continue;
    int ni = i > col ? i - 1 : i;
    for (int c = 0; c < n - 1; c++) q.A(ni, c) = newACols(i, c);
  }
  p = q;
  mapOut = newMap;
}
inline MatX reduceLcp(LcpProblem& p) {
  MatX mapOut = identityX(p.A.r);
  const s_t MERGE_THRESHOLD = 1e-4;
  while (true) {
    const int n = p.A.c;
    bool found = false;
    for (int a = 0; a < n - 1 && !found; a++)
      for (int b = a + 1; b < n; b++) {
        s_t d2 = 0;
        for (int r = 0; r < n; r++) { s_t d = p.A(r, a) - p.A(r, b); d2 += d * d; }
        if (d2 < MERGE_THRESHOLD && std::fabs(p.b[a] - p.b[b]) < MERGE_THRESHOLD && p.findex[a] == p.findex[b] &&
            p.hi[a] == p.hi[b] && p.lo[a] == p.lo[b]) {
          found = true;
          mergeColumns(a, b, p, mapOut);
          break;
        }
      }
    if (!found) break;
  }
  return mapOut;
}
inline MatX removeFrictionLcp(LcpProblem& p) {
  MatX mapOut = identityX(p.A.r);
  std::vector<int> fi = p.findex;
  for (int i = (int)fi.size() - 1; i >= 0; i--)
    if (fi[i] != -1) dropColumn(i, p, mapOut);
  return mapOut;
}

// ---- PgsBoxedLcpSolver::solve (PgsBoxedLcpSolver.cpp:79-268) with Option(30, 1e-6, 1e-3, 1e-9, false) ----
inline bool pgsSolve(LcpProblem& p, int maxIteration = 30, s_t deltaXThreshold = 1e-6, s_t relTol = 1e-3, s_t epsDiv = 1e-9) {
  const int n = (int)p.x.size();
  MatX& A = p.A;
  VecX &x = p.x, &b = p.b, &lo = p.lo, &hi = p.hi;
  std::vector<int>& findex = p.findex;
  std::vector<int> order;
  bool possibleToTerminate = true;
  for (int i = 0; i < n; ++i) {
    if (A(i, i) < epsDiv) { x[i] = 0.0; continue; }
    order.push_back(i);
    const s_t old_x = x[i];
    s_t new_x = b[i];
    for (int j = 0; j < i; ++j) new_x -= A(i, j) * x[j];
    for (int j = i + 1; j < n; ++j) new_x -= A(i, j) * x[j];
    new_x /= A(i, i);
    if (findex[i] >= 0) {
      const s_t hi_tmp = hi[i] * x[findex[i]], lo_tmp = -hi_tmp;
      x[i] = new_x > hi_tmp ? hi_tmp : (new_x < lo_tmp ? lo_tmp : new_x);
    } else {
      x[i] = new_x > hi[i] ? hi[i] : (new_x < lo[i] ? lo[i] : new_x);
    }
    if (possibleToTerminate && std::fabs(x[i] - old_x) > deltaXThreshold) possibleToTerminate = false;
  }
  if (possibleToTerminate) return true;
  for (int index : order) {
    const s_t dummy = 1.0 / A(index, index);
    b[index] *= dummy;
    for (int j = 0; j < n; ++j) A(index, j) *= dummy;
  }
  for (int iter = 1; iter < maxIteration; ++iter) {
    possibleToTerminate = true;
    for (int index : order) {
      s_t new_x = b[index];
      const s_t old_x = x[index];
      for (int j = 0; j < index; j++) new_x -= A(index, j) * x[j];
      for (int j = index + 1; j < n; j++) new_x -= A(index, j) * x[j];
      if (findex[index] >= 0) {
        const s_t hi_tmp = hi[index] * x[findex[index]], lo_tmp = -hi_tmp;
        x[index] = new_x > hi_tmp ? hi_tmp : (new_x < lo_tmp ? lo_tmp : new_x);
      } else {
        x[index] = new_x > hi[index] ? hi[index] : (new_x < lo[index] ? lo[index] : new_x);
      }
      if (possibleToTerminate && std::fabs(x[index]) > epsDiv) {
        if (std::fabs((x[index] - old_x) / x[index]) > relTol) possibleToTerminate = false;
      }
    }
    if (possibleToTerminate) break;
  }
  return possibleToTerminate;
}

// ---- Dantzig: the reference's own dSolveLCP (dart/external/odelcpsolver/lcp.cpp:780-1113) ----
typedef int (*RefDantzigFn)(int, double*, double*, double*, double*, double*, int*, int);
inline RefDantzigFn loadRefDantzig() {
  Dl_info info;
  std::string dir = ".";
  if (dladdr((void*)&loadRefDantzig, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t s = p.find_last_of('/');
    if (s != std::string::npos) dir = p.substr(0, s);
  }
  const char* env = getenv("NBO_REF_DIR");
  std::string path = (env ? std::string(env) : dir + "/_ref") + "/libodelcp_ref.so";
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  return h ? (RefDantzigFn)dlsym(h, "nbo_ref_dantzig") : nullptr;
}
inline RefDantzigFn refDantzig() {
  static RefDantzigFn fn = loadRefDantzig();  // thread-safe one-time initialisation
  return fn;
}
inline int dPAD(int a) { return (a > 1) ? (((a - 1) | 3) + 1) : a; }
// returns 1 success, 0 failure, -1 solver unavailable (oracle/_ref not built)
inline int dantzigSolve(LcpProblem& p, bool earlyTermination) {
  RefDantzigFn fn = refDantzig();
  if (!fn) return -1;
  const int n = (int)p.x.size();
  if (n == 0) return 1;
  const int nskip = dPAD(n);
  std::vector<double> A((size_t)n * nskip, 0.0);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A[(size_t)i * nskip + j] = p.A(i, j);
  int ok = fn(n, A.data(), p.x.data(), p.b.data(), p.lo.data(), p.hi.data(), p.findex.data(), earlyTermination ? 1 : 0);
  return ok ? 1 : 0;
}

}  // namespace nbo
