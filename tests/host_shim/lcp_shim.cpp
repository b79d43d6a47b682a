// Host build of the product's per-lane LCP device code (test harness only).
#include "dantzig_dev.hpp"

using namespace nbl;

struct HostMem {
  double* base;
  double& at(int idx) const { return base[idx]; }
};

static void load(RedLcp& P, HostMem& M, int n, const double* A, const double* x, const double* b, const double* lo,
                 const double* hi, const int32_t* findex) {
  P.n = n; P.nOrig = n;
  for (int i = 0; i < n; i++) {
    P.x[i] = x[i]; P.b[i] = b[i]; P.lo[i] = lo[i]; P.hi[i] = hi[i]; P.findex[i] = findex[i]; P.mapTo[i] = i;
    for (int j = 0; j < n; j++) M.at(i * MAXR + j) = A[i * n + j];
  }
}

extern "C" {
int shim_dantzig(int n, const double* A, double* x, const double* b, const double* lo, const double* hi, const int32_t* findex) {
  static thread_local double bufA[MAXR * MAXR], bufL[MAXR * MAXR * 2];
  HostMem M{bufL};   // offA = 0, offL = MAXR*MAXR
  RedLcp P;
  HostMem MA{bufL};
  load(P, MA, n, A, x, b, lo, hi, findex);
  (void)bufA;
  bool ok = dantzigSolve(M, 0, MAXR * MAXR, P);
  for (int i = 0; i < n; i++) x[i] = P.x[i];
  return ok ? 1 : 0;
}
int shim_pgs(int n, const double* A, double* x, const double* b, const double* lo, const double* hi, const int32_t* findex) {
  static thread_local double buf[MAXR * MAXR];
  HostMem M{buf};
  RedLcp P;
  load(P, M, n, A, x, b, lo, hi, findex);
  bool ok = pgsSolve(M, 0, P);
  for (int i = 0; i < n; i++) x[i] = P.x[i];
  return ok ? 1 : 0;
}
// reduce (removeFriction = 0) or removeFriction (= 1); outputs the reduced problem and mapTo
int shim_reduce(int n, const double* A, const double* x, const double* b, const double* lo, const double* hi, const int32_t* findex,
                int removeFriction, double* Ar, double* xr, double* br, double* lor, double* hir, int32_t* fr, int32_t* mapTo) {
  static thread_local double buf[MAXR * MAXR];
  HostMem M{buf};
  RedLcp P;
  load(P, M, n, A, x, b, lo, hi, findex);
  if (removeFriction) lcpRemoveFriction(M, 0, P); else lcpReduce(M, 0, P);
  for (int i = 0; i < P.n; i++) {
    xr[i] = P.x[i]; br[i] = P.b[i]; lor[i] = P.lo[i]; hir[i] = P.hi[i]; fr[i] = P.findex[i];
    for (int j = 0; j < P.n; j++) Ar[i * P.n + j] = M.at(i * MAXR + j);
  }
  for (int i = 0; i < n; i++) mapTo[i] = P.mapTo[i];
  return P.n;
}
int shim_cod_solve(int c, const double* A, const double* b, double* x, int transpose) {
  static thread_local double buf[2 * MAXR * MAXR];
  HostMem Mh{buf};
  LaneMem M; M.base = buf; M.B = 1; M.b = 0;
  (void)Mh;
  for (int i = 0; i < c; i++) for (int j = 0; j < c; j++) M.at(i * MAXR + j) = A[i * c + j];
  CodFactor F; F.ld = MAXR; F.offQR = 0; F.offChol = MAXR * MAXR; F.c = c;
  codFactor(M, F);
  double rhs[MAXR];
  for (int i = 0; i < c; i++) rhs[i] = b[i];
  if (transpose) codSolveT(M, F, rhs, x); else codSolve(M, F, rhs, x);
  return F.rank;
}
}
