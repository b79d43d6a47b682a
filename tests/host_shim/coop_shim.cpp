// Host build of the product's wave-cooperative LCP code (coop_dev.hpp) on the thread-per-lane wave emulation,
// next to the one-world-per-lane statement of the same algorithm (lane_lcp_statement.hpp).  Test harness only.
#include "coop_dev.hpp"
#include "coop_dantzig_dev.hpp"
#include "lane_lcp_statement.hpp"
#include "wave_emu.hpp"

using namespace NBL_NS;

extern "C" {
// (every "24" below reads MAXR: the shim is built twice, for the 24-row and - -DNBL_MAXC=16 - the 48-row instantiation of the device code)
int shim_rows() { return MAXR; }
// Q: 24 x 24 row-major (masked rows/columns zero), P out 24 x 24 row-major; returns rank
int shim_coop_pinv(const double* Q, int cTrue, double* Pout) {
  static CoopLds S;
  int rank = -1;
  emuRunWave([&](const EmuWave& w) {
    double a[MAXR];
    const int ln = w.lane();
    for (int i = 0; i < MAXR; i++) a[i] = ln < MAXR ? Q[i * MAXR + ln] : 0.0;
    const int r = coopPinv(w, a, S, cTrue);
    if (ln == 0) rank = r;
  });
  for (int i = 0; i < MAXR; i++) for (int j = 0; j < MAXR; j++) Pout[i * MAXR + j] = S.P[i * CLD + j];
  return rank;
}

// the symmetric positive semi-definite route (coopPinvSym): same arguments
int shim_coop_pinv_sym(const double* Q, int cTrue, double* Pout) {
  static CoopLds S;
  int rank = -1;
  emuRunWave([&](const EmuWave& w) {
    double a[MAXR];
    const int ln = w.lane();
    for (int i = 0; i < MAXR; i++) a[i] = ln < MAXR ? Q[i * MAXR + ln] : 0.0;
    const int r = coopPinvSym(w, a, S, cTrue);
    if (ln == 0) rank = r;
  });
  for (int i = 0; i < MAXR; i++) for (int j = 0; j < MAXR; j++) Pout[i * MAXR + j] = S.P[i * CLD + j];
  return rank;
}

static void fillRow(CoopRow& R, int ln, int m, const double* A, const double* b, const double* mu) {
  R.m = m; R.fric = (ln % 3) != 0; R.fp = ln < MAXR ? ln - (ln % 3) : 0;
  R.mu = ln < m ? mu[ln / 3] : 0.0; R.Bv = ln < m ? b[ln] : 0.0;
  R.on = ln < m;
  R.Acol = A + (ln < m ? ln : 0);
  double cn = 0;
  for (int i = 0; i < MAXR; i++) cn += R.a(i) * R.a(i);
  R.colNorm = cn;
}

// A: 24 x 24 row-major (m x m used), b[24], mu[8]; outputs per row.  Returns ok | pinvValid << 1.
int shim_coop_stage0(int m, const double* A, const double* b, const double* mu, int haveCache, const double* xcache,
                     double* X, double* X0, int* cls, double* E, double* Pout) {
  static CoopLds S;
  int ret = 0;
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopRow R;
    fillRow(R, ln, m, A, b, mu);
    CoopStage0 out;
    coopStage0(w, S, R, haveCache != 0, ln < m ? xcache[ln] : 0.0, out);
    if (ln < MAXR) { X[ln] = out.X; X0[ln] = out.X0; cls[ln] = out.K.cls; E[ln] = out.K.E; }
    if (ln == 0) ret = (out.ok ? 1 : 0) | (out.pinvValid ? 2 : 0);
  });
  for (int i = 0; i < MAXR; i++) for (int j = 0; j < MAXR; j++) Pout[i * MAXR + j] = S.P[i * CLD + j];
  return ret;
}

// stages 1-3 + the order of preference + standardisation on the rows of `mask` only; outputs like shim_coop_cascade plus the
// standardised x and the row classes
int shim_coop_cascade_masked(int m, const double* A, const double* b, const double* mu, const double* x0, unsigned long long mask, double fallbackCfm,
                             double* X, double* cfmOut, double* Xstd, int* cls) {
  static CascadeLds C1;
  static PgsLds C2, C3;
  static CoopLds S;
  uint32_t stOut = 0;
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopRow R;
    fillRow(R, ln, m, A, b, mu);
    R.on = R.on && ((mask >> ln) & 1ull);
    const double X0 = R.on ? x0[ln] : 0.0;
    CoopStageResult r1, r2, r3;
    coopCascadeStage1(w, C1, R, X0, r1);
    coopCascadeStage2(w, C2, R, X0, fallbackCfm, r2);
    coopCascadeStage3(w, C3, R, X0, fallbackCfm, r3);
    double x, cfm;
    bool noFric;
    uint32_t st;
    coopCascadeChoose(w, m, X0, fallbackCfm, r1, r2, r3, x, cfm, noFric, st);
    if (ln < MAXR) X[ln] = x;
    CoopCascadeOut out;
    coopCascadeSelect(w, S, R, X0, fallbackCfm, r1, r2, r3, out);
    if (ln < MAXR) { Xstd[ln] = out.X; cls[ln] = out.K.cls; }
    if (ln == 0) { stOut = out.st; *cfmOut = cfm; }
  });
  return (int)stOut;
}

// stage 0 on the rows of `mask` only (the rows of a world's other constrained groups switched off, as the kernels run it)
int shim_coop_stage0_masked(int m, const double* A, const double* b, const double* mu, unsigned long long mask, double* X, double* X0, int* cls, double* E) {
  static CoopLds S;
  int ret = 0;
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopRow R;
    fillRow(R, ln, m, A, b, mu);
    R.on = R.on && ((mask >> ln) & 1ull);
    CoopStage0 out;
    coopStage0(w, S, R, false, 0.0, out);
    if (ln < MAXR) { X[ln] = out.X; X0[ln] = out.X0; cls[ln] = out.K.cls; E[ln] = out.K.E; }
    if (ln == 0) ret = (out.ok ? 1 : 0) | (out.pinvValid ? 2 : 0);
  });
  return ret;
}

// the same two with joint-limit rows (CoopRow::lim / neg / limMask as coopLoadRow<true> sets them): limMask = the limit rows, negMask = the
// ones carried negated (upper limits); A, b in the device's form (those rows and columns already negated)
int shim_coop_stage0_lim(int m, const double* A, const double* b, const double* mu, unsigned long long mask, unsigned long long limMask, unsigned long long negMask,
                         double* X, double* X0, int* cls, double* E) {
  static CoopLds S;
  int ret = 0;
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopRow R;
    fillRow(R, ln, m, A, b, mu);
    R.on = R.on && ((mask >> ln) & 1ull);
    R.lim = ((limMask >> ln) & 1ull); R.neg = ((negMask >> ln) & 1ull); R.limMask = (RowMask)limMask;
    CoopStage0 out;
    coopStage0(w, S, R, false, 0.0, out);
    if (ln < MAXR) { X[ln] = out.X; X0[ln] = out.X0; cls[ln] = out.K.cls; E[ln] = out.K.E; }
    if (ln == 0) ret = (out.ok ? 1 : 0) | (out.pinvValid ? 2 : 0);
  });
  return ret;
}
int shim_coop_cascade_lim(int m, const double* A, const double* b, const double* mu, const double* x0, unsigned long long mask, unsigned long long limMask,
                          unsigned long long negMask, double fallbackCfm, double* X, double* cfmOut, int* stages) {
  static CascadeLds C1;
  static PgsLds C2, C3;
  uint32_t stOut = 0;
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopRow R;
    fillRow(R, ln, m, A, b, mu);
    R.on = R.on && ((mask >> ln) & 1ull);
    R.lim = ((limMask >> ln) & 1ull); R.neg = ((negMask >> ln) & 1ull); R.limMask = (RowMask)limMask;
    const double X0 = R.on ? x0[ln] : 0.0;
    CoopStageResult r1, r2, r3;
    coopCascadeStage1(w, C1, R, X0, r1);
    coopCascadeStage2(w, C2, R, X0, fallbackCfm, r2);
    coopCascadeStage3(w, C3, R, X0, fallbackCfm, r3);
    double x, cfm;
    bool noFric;
    uint32_t st;
    coopCascadeChoose(w, m, X0, fallbackCfm, r1, r2, r3, x, cfm, noFric, st);
    if (ln < MAXR) X[ln] = x;
    if (ln == 0) { stOut = st; *cfmOut = cfm; stages[0] = r1.flags; stages[1] = r2.flags; stages[2] = r3.flags; }
  });
  return (int)stOut;
}

// the one-world-per-lane statement (laneStage0 of lcp_dev.hpp) on the same problem
int shim_lane_stage0(int m, const double* A, const double* b, const double* mu, int haveCache, const double* xcache,
                     double* X, double* X0, int* cls, double* E) {
  static thread_local double bufA[MAXR * MAXR], bufL[2 * MAXR * MAXR];
  LcpView V;
  V.mem.base = bufA; V.mem.B = 1; V.mem.b = 0; V.offA = 0; V.m = m;
  for (int i = 0; i < MAXR * MAXR; i++) bufA[i] = A[i];
  for (int c = 0; c < m / 3; c++) V.mu[c] = mu[c];
  LaneMem L; L.base = bufL; L.B = 1; L.b = 0;
  double colNorm[MAXR];
  for (int c = 0; c < m; c++) { double s = 0; for (int r = 0; r < m; r++) s += V.A(r, c) * V.A(r, c); colNorm[c] = s; }
  for (int r = 0; r < MAXR; r++) { X[r] = (haveCache && r < m) ? xcache[r] : 0.0; X0[r] = 0; cls[r] = 0; E[r] = 0; }
  Classes K;
  const bool ok = laneStage0(V, L, haveCache != 0, X, X0, b, colNorm, K);
  for (int r = 0; r < m; r++) { cls[r] = K.cls[r]; E[r] = K.E[r]; }
  return ok ? 1 : 0;
}

// cooperative Dantzig on an n-row problem (A n x n row-major, only its lower triangle is meaningful); returns 1 / 0 / -1
int shim_coop_dantzig(int n, const double* A, double* x, const double* b, const double* lo, const double* hi, const int32_t* findex) {
  static CascadeLds C;
  int ret = -2;
  for (int i = 0; i < MAXR * CLD; i++) { C.A[i] = 0; C.L[i] = 0; }
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) C.A[i * CLD + j] = A[i * n + j];
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopLcpRow row;
    row.x = 0; row.b = ln < n ? b[ln] : 0; row.lo = ln < n ? lo[ln] : 0; row.hi = ln < n ? hi[ln] : 0; row.findex = ln < n ? findex[ln] : -1;
    const int r = coopDantzig(w, C, n, row);
    if (ln < n) x[ln] = row.x;
    if (ln == 0) ret = r;
  });
  return ret;
}

static void loadRows(CascadeLds& C, int n, const double* A, int ln, CoopLcpRow& row, const double* x, const double* b, const double* lo,
                     const double* hi, const int32_t* findex) {
  if (ln == 0) { for (int i = 0; i < MAXR * CLD; i++) C.A[i] = 0; for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) C.A[i * CLD + j] = A[i * n + j]; }
  row.x = ln < n ? x[ln] : 0; row.b = ln < n ? b[ln] : 0; row.lo = ln < n ? lo[ln] : 0; row.hi = ln < n ? hi[ln] : 0; row.findex = ln < n ? findex[ln] : -1;
}
int shim_coop_pgs(int n, const double* A, double* x, const double* b, const double* lo, const double* hi, const int32_t* findex) {
  static CascadeLds C;
  int ret = -1;
  emuRunWave([&](const EmuWave& w) {
    CoopLcpRow row;
    loadRows(C, n, A, w.lane(), row, x, b, lo, hi, findex);
    w.sync();
    const bool ok = coopPgs(w, C, n, row);
    if (w.lane() < n) x[w.lane()] = row.x;
    if (w.lane() == 0) ret = ok ? 1 : 0;
  });
  return ret;
}
// reduce (removeFriction = 0) or removeFriction (= 1); outputs the reduced problem and mapTo
int shim_coop_reduce(int n, const double* A, const double* x, const double* b, const double* lo, const double* hi, const int32_t* findex,
                     int removeFriction, double* Ar, double* xr, double* br, double* lor, double* hir, int32_t* fr, int32_t* mapTo) {
  static CascadeLds C;
  int nr = -1;
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopLcpRow row;
    loadRows(C, n, A, ln, row, x, b, lo, hi, findex);
    w.sync();
    int mt = ln < n ? ln : -1;
    const int r = removeFriction ? coopLcpRemoveFriction(w, C, n, row, mt) : coopLcpReduce(w, C, n, row, mt);
    if (ln < r) { xr[ln] = row.x; br[ln] = row.b; lor[ln] = row.lo; hir[ln] = row.hi; fr[ln] = row.findex; }
    if (ln < n) mapTo[ln] = mt;
    if (ln == 0) nr = r;
  });
  for (int i = 0; i < nr; i++) for (int j = 0; j < nr; j++) Ar[i * nr + j] = C.A[i * CLD + j];
  return nr;
}

// stages 1-3 of the solver cascade exactly as k_contact_cascade_stages / _final run them (three independent stage functions, then
// the reference's order of preference), on one problem: A 24 x 24 row-major (m x m used), b[24], mu[8], x0[24] = the pre-solve x.
// Outputs the chosen x (before standardisation), the NBL_ST_* bits and the CFM the solver ended with.
int shim_coop_cascade(int m, const double* A, const double* b, const double* mu, const double* x0, double fallbackCfm, double* X,
                      double* cfmOut) {
  static CascadeLds C1;
  static PgsLds C2, C3;
  uint32_t stOut = 0;
  emuRunWave([&](const EmuWave& w) {
    const int ln = w.lane();
    CoopRow R;
    fillRow(R, ln, m, A, b, mu);
    const double X0 = ln < m ? x0[ln] : 0.0;
    CoopStageResult r1, r2, r3;
    coopCascadeStage1(w, C1, R, X0, r1);
    coopCascadeStage2(w, C2, R, X0, fallbackCfm, r2);
    coopCascadeStage3(w, C3, R, X0, fallbackCfm, r3);
    double x, cfm;
    bool noFric;
    uint32_t st;
    coopCascadeChoose(w, m, X0, fallbackCfm, r1, r2, r3, x, cfm, noFric, st);
    if (ln < MAXR) X[ln] = x;
    if (ln == 0) { stOut = st; *cfmOut = cfm; }
  });
  return (int)stOut;
}
}

// the exact reverse mode of the SO(3) / SE(3) position integration (spatial_dev.hpp), for the CPU test against finite differences
extern "C" void shim_so3_integration_vjp(const double* q, const double* w, double dt, const double* g, double* posT, double* velT) {
  V3 p, v;
  so3IntegrationVjp(mk3(q[0], q[1], q[2]), mk3(w[0], w[1], w[2]), dt, mk3(g[0], g[1], g[2]), p, v);
  posT[0] = p.x; posT[1] = p.y; posT[2] = p.z; velT[0] = v.x; velT[1] = v.y; velT[2] = v.z;
}
extern "C" void shim_so3_integrate(const double* q, const double* w, double dt, double* out) {
  const V3 r = logMap(mul(expMapRot(mk3(q[0], q[1], q[2])), expMapRot(dt * mk3(w[0], w[1], w[2]))));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
