// Host build of the product's GENERAL-m contact LCP code (gen_lcp_dev.hpp, gen_dantzig_dev.hpp: any number of rows up to 192) under a
// ONE-lane wave policy, where it is plain sequential code.  Test harness only (tests/test_gen_host.py).
#include "gen_lcp_dev.hpp"
#include "gen_dantzig_dev.hpp"

#include <cstring>
#include <vector>
#include <thread>
#include <pthread.h>

using namespace NBL_NS;

struct HostWave1 {
  int lane() const { return 0; }
  int lanes() const { return 1; }
  void sync() const {}
  double maxAll(double v) const { return v; }
  int minAllI(int v) const { return v; }
  double sumAll(double v) const { return v; }
  bool anyAll(bool b) const { return b; }
  double bcast(double v, int) const { return v; }
  int bcastI(int v, int) const { return v; }
  void fence() const {}
};

namespace {
// The leading dimension of a problem of m rows: the rows rounded up to a multiple of 8, like the library sizes the scratch and the record of a
// model (genLeadingDim) - every call exercises the run-time strides at its own size instead of at the cap.
// gshim_device_pool(rows): the worlds that follow are laid out like k_contact_solve_gen lays out a world of a MODEL of `rows` rows (0: off) -
// that leading dimension and the cascade's vectors in their own "fast" pool of max(16 rows, 1152) doubles (LDS on the device), which
// switches on the packed placements (genPinvPair, genPgsAT).  Placement only: every result must stay bit for bit what it was.
static int g_modelRows = 0;
static int shimLd(int m) { int r = (m + 7) & ~7; if (g_modelRows > r) r = (g_modelRows + 7) & ~7; return r < 8 ? 8 : (r > GR ? GR : r); }
struct World {
  GenRows R;
  std::vector<double> buf, rowsPool, vecPool;
  GenScratch S;
  int ld;
  explicit World(int m) : buf(genScratchDoubles(shimLd(m)), 0.0), rowsPool(genRowsDoubles(genRowsCap(shimLd(m))), 0.0), ld(shimLd(m)) {
    for (int k = 0; k < GEN_NMAT; k++) S.mat[k] = buf.data() + (size_t)k * ld * ld;
    S.vec = buf.data() + (size_t)GEN_NMAT * ld * ld;
    S.ld = ld;
    if (g_modelRows > 0) {
      const size_t nv = (size_t)16 * ld > 1152 ? (size_t)16 * ld : 1152;
      vecPool.assign(nv, 0.0);
      S.vec = vecPool.data(); S.vecFast = true; S.vecDoubles = (int)nv;
    }
    genRowsCarve(R, rowsPool.data(), genRowsCap(ld));      // (the rows' arrays sized by the problem, like the library sizes them by the model)
    R.ld = ld;
  }
};
// rows as the kernels set them up: three per contact, [normal, t1, t2]; mu per contact (mu <= 1e-3: frictionless, its tangent rows empty);
// lim / neg per row (joint-limit pseudo-contacts), mask = the rows of the constrained group at hand
void fillRows(GenRows& R, int m, const double* A, int GLD, const double* b, const double* mu, const unsigned char* mask, const unsigned char* lim, const unsigned char* neg) {
  R.m = m;
  R.anyLim = 0;
  for (int r = 0; r < m; r++) {
    R.fric[r] = (r % 3) != 0; R.fp[r] = r - (r % 3);
    double mr = mu[r / 3];
    if (lim && lim[r - (r % 3)]) mr = 0.0;
    if (!(mr > 1e-3)) mr = 0.0;
    R.mu[r] = mr; R.Bv[r] = b[r];
    R.lim[r] = lim ? (lim[r] && (r % 3) == 0) : 0; R.neg[r] = neg ? neg[r] : 0;
    if (R.lim[r]) R.anyLim = 1;
    R.rowOn[r] = 1; R.on[r] = mask ? mask[r] : 1;
    double cn = 0;
    for (int i = 0; i < m; i++) cn += (R.on[r] ? A[(size_t)i * GLD + r] : 0.0) * (R.on[r] ? A[(size_t)i * GLD + r] : 0.0);
    R.colNorm[r] = cn;
  }
}
}  // namespace

extern "C" {
int gshim_rows() { return GR; }
void gshim_device_pool(int rows) { g_modelRows = rows; }

// Q: m x m row-major (masked rows / columns zero), P out m x m row-major; returns the rank
int gshim_pinv(int m, const double* Q, int cTrue, double* Pout) {
  World Wd(m);
  const int GLD = Wd.ld;
  const HostWave1 w;
  Wd.R.m = m;
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Wd.S.mat[0][(size_t)i * GLD + j] = Q[(size_t)i * m + j];
  const int r = genPinv(w, Wd.R, Wd.S.mat[0], Wd.S.mat[1], Wd.S.mat[2], Wd.S.mat[3], m, cTrue);
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Pout[(size_t)i * m + j] = Wd.S.mat[3][(size_t)i * GLD + j];
  return r;
}

// the same with the symmetric-positive-semi-definite rank policy (64 x the reference's threshold, see genPinv)
int gshim_pinv_sym(int m, const double* Q, int cTrue, double* Pout) {
  World Wd(m);
  const int GLD = Wd.ld;
  const HostWave1 w;
  Wd.R.m = m;
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Wd.S.mat[0][(size_t)i * GLD + j] = Q[(size_t)i * m + j];
  const int r = genPinv(w, Wd.R, Wd.S.mat[0], Wd.S.mat[1], Wd.S.mat[2], Wd.S.mat[3], m, cTrue, true);
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Pout[(size_t)i * m + j] = Wd.S.mat[3][(size_t)i * GLD + j];
  return r;
}

// the Dantzig driver on an n-row boxed LCP with explicit bounds (row-major A); returns 1 / 0 / -1 like genDantzigSeq
int gshim_dantzig(int n, const double* A, const double* b, const double* lo, const double* hi, const int* findex, double* x) {
  World Wd(n);
  const int GLD = Wd.ld;
  GenProblem P; GenDantzigMem D;
  genCarve(Wd.S, P, D, n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) P.A[(size_t)i * GLD + j] = A[(size_t)i * n + j];
    P.b[i] = b[i]; P.lo[i] = lo[i]; P.hi[i] = hi[i]; P.findex[i] = findex[i]; P.x[i] = 0.0;
  }
  std::vector<double> xo(n, 0.0);
  const int rc = genDantzigSeq(D, n, xo.data());
  for (int i = 0; i < n; i++) x[i] = rc == 1 ? xo[i] : 0.0;
  return rc;
}

// A policy of SEVERAL lanes for the wave-shared driver (genDantzigPar): a thread per lane, sync() a barrier, the reductions through shared
// slots - so that the CPU suite runs the text the device runs with its loops really strided over lanes and its barriers really needed.
struct HostWaveShared {
  pthread_barrier_t bar;
  double dslot[64];
  int islot[64];
  int nl;
};
struct HostWaveN {
  HostWaveShared* sh;
  int ln;
  int lane() const { return ln; }
  int lanes() const { return sh->nl; }
  void sync() const { pthread_barrier_wait(&sh->bar); }
  double maxAll(double v) const {
    sh->dslot[ln] = v; sync();
    double m = sh->dslot[0];
    for (int i = 1; i < sh->nl; i++) m = fmax(m, sh->dslot[i]);
    sync();
    return m;
  }
  int minAllI(int v) const {
    sh->islot[ln] = v; sync();
    int m = sh->islot[0];
    for (int i = 1; i < sh->nl; i++) m = sh->islot[i] < m ? sh->islot[i] : m;
    sync();
    return m;
  }
  double sumAll(double v) const {
    sh->dslot[ln] = v; sync();
    double s = 0.0;
    for (int i = 0; i < sh->nl; i++) s += sh->dslot[i];
    sync();
    return s;
  }
  bool anyAll(bool b) const { return minAllI(b ? 0 : 1) == 0; }
  double bcast(double v, int src) const {
    sh->dslot[ln] = v; sync();
    const double r = sh->dslot[src];
    sync();
    return r;
  }
  int bcastI(int v, int src) const {
    sh->islot[ln] = v; sync();
    const int r = sh->islot[src];
    sync();
    return r;
  }
  void fence() const { sync(); }
};

// the wave-shared Dantzig driver (genDantzigPar) on `lanes` lanes (1: the one-lane policy, no threads); same interface as gshim_dantzig
int gshim_dantzig_par(int n, const double* A, const double* b, const double* lo, const double* hi, const int* findex, double* x, int lanes) {
  World Wd(n);
  const int GLD = Wd.ld;
  GenProblem P; GenDantzigMem D;
  genCarve(Wd.S, P, D, n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) P.A[(size_t)i * GLD + j] = A[(size_t)i * n + j];
    P.b[i] = b[i]; P.lo[i] = lo[i]; P.hi[i] = hi[i]; P.findex[i] = findex[i]; P.x[i] = 0.0;
  }
  std::vector<double> xo(n, 0.0), sb((size_t)4 * GR, 0.0);
  int rc = 0;
  if (lanes <= 1) {
    const HostWave1 w;
    rc = genDantzigPar(w, D, n, xo.data(), sb.data(), sb.data() + GR, sb.data() + 2 * GR, sb.data() + 3 * GR, Wd.S.mat[0]);
  } else {
    if (lanes > 64) lanes = 64;
    HostWaveShared sh;
    sh.nl = lanes;
    pthread_barrier_init(&sh.bar, nullptr, (unsigned)lanes);
    std::vector<int> rcs(lanes, 0);
    std::vector<std::thread> th;
    for (int l = 0; l < lanes; l++)
      th.emplace_back([&, l]() {
        const HostWaveN w{&sh, l};
        rcs[l] = genDantzigPar(w, D, n, xo.data(), sb.data(), sb.data() + GR, sb.data() + 2 * GR, sb.data() + 3 * GR, Wd.S.mat[0]);
      });
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&sh.bar);
    rc = rcs[0];
    for (int l = 1; l < lanes; l++) if (rcs[l] != rc) rc = -99;      // (the lanes must agree)
  }
  for (int i = 0; i < n; i++) x[i] = rc == 1 ? xo[i] : 0.0;
  return rc;
}

// stage 0 on the rows of `mask` (NULL: all): A m x m row-major, b[m], mu[m / 3]; outputs per row.  Returns ok | pinvValid << 1.
int gshim_stage0(int m, const double* A, const double* b, const double* mu, const unsigned char* mask, const unsigned char* lim, const unsigned char* neg,
                 int haveCache, const double* xcache, double* X, double* X0, int* cls, double* E, double* Pout) {
  World Wd(m);
  const int GLD = Wd.ld;
  const HostWave1 w;
  std::vector<double> Ap((size_t)GLD * GLD, 0.0);
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Ap[(size_t)i * GLD + j] = A[(size_t)i * m + j];
  fillRows(Wd.R, m, Ap.data(), GLD, b, mu, mask, lim, neg);
  for (int r = 0; r < m; r++) Wd.R.X[r] = (haveCache && Wd.R.on[r]) ? xcache[r] : 0.0;
  bool pinvValid = false;
  GenClasses K;
  const bool ok = genStage0(w, Ap.data(), GLD, Wd.R, Wd.S, haveCache != 0, pinvValid, K);
  for (int r = 0; r < m; r++) { X[r] = Wd.R.X[r]; X0[r] = Wd.R.X0[r]; cls[r] = Wd.R.cls[r]; E[r] = Wd.R.E[r]; }
  if (Pout) for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Pout[(size_t)i * m + j] = Wd.S.mat[3][(size_t)i * GLD + j];
  return (ok ? 1 : 0) | (pinvValid ? 2 : 0);
}

// stages 1-3 in the reference's order + standardisation on the rows of `mask`, from the pre-solve x `x0`; returns the status bits
int gshim_cascade(int m, const double* A, const double* b, const double* mu, const unsigned char* mask, const unsigned char* lim, const unsigned char* neg,
                  const double* x0, double fallbackCfm, double* X, double* cfmOut, int* cls) {
  World Wd(m);
  const int GLD = Wd.ld;
  const HostWave1 w;
  std::vector<double> Ap((size_t)GLD * GLD, 0.0);
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Ap[(size_t)i * GLD + j] = A[(size_t)i * m + j];
  fillRows(Wd.R, m, Ap.data(), GLD, b, mu, mask, lim, neg);
  for (int r = 0; r < m; r++) Wd.R.X0[r] = Wd.R.on[r] ? x0[r] : 0.0;
  double cfm = 0.0;
  uint32_t st = 0;
  bool pinvValid = false;
  GenClasses K;
  genCascade(w, Ap.data(), GLD, Wd.R, Wd.S, fallbackCfm, cfm, st, pinvValid, K);
  for (int r = 0; r < m; r++) { X[r] = Wd.R.X[r]; cls[r] = Wd.R.cls[r]; }
  *cfmOut = cfm;
  return (int)st;
}
}

extern "C" {
// one stage of the cascade alone (1, 2 or 3) on the rows of `mask`, from the pre-solve x: the RAW candidate and the GS_* flags
int gshim_stage(int stage, int m, const double* A, const double* b, const double* mu, const unsigned char* mask, const double* x0, double fallbackCfm, double* X) {
  World Wd(m);
  const int GLD = Wd.ld;
  const HostWave1 w;
  std::vector<double> Ap((size_t)GLD * GLD, 0.0);
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Ap[(size_t)i * GLD + j] = A[(size_t)i * m + j];
  fillRows(Wd.R, m, Ap.data(), GLD, b, mu, mask, nullptr, nullptr);
  for (int r = 0; r < m; r++) Wd.R.X0[r] = Wd.R.on[r] ? x0[r] : 0.0;
  std::vector<double> out(GR, 0.0);
  int flags = 0;
  if (stage == 1) flags = genStage1(w, Ap.data(), GLD, Wd.R, Wd.S, out.data());
  else if (stage == 2) flags = genStage2(w, Ap.data(), GLD, Wd.R, Wd.S, fallbackCfm, out.data());
  else flags = genStage3(w, Ap.data(), GLD, Wd.R, Wd.S, fallbackCfm, out.data());
  for (int r = 0; r < m; r++) X[r] = out[r];
  return flags;
}

// the same stage with the work shared by `lanes` lanes (threads + barriers: HostWaveN) - the text the device runs with its 64 lanes
int gshim_stage_lanes(int stage, int m, const double* A, const double* b, const double* mu, const unsigned char* mask, const double* x0, double fallbackCfm,
                      double* X, int lanes) {
  World Wd(m);
  const int GLD = Wd.ld;
  std::vector<double> Ap((size_t)GLD * GLD, 0.0);
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Ap[(size_t)i * GLD + j] = A[(size_t)i * m + j];
  fillRows(Wd.R, m, Ap.data(), GLD, b, mu, mask, nullptr, nullptr);
  for (int r = 0; r < m; r++) Wd.R.X0[r] = Wd.R.on[r] ? x0[r] : 0.0;
  std::vector<double> out(GR, 0.0);
  if (lanes > 64) lanes = 64;
  if (lanes < 2) lanes = 2;
  HostWaveShared sh;
  sh.nl = lanes;
  pthread_barrier_init(&sh.bar, nullptr, (unsigned)lanes);
  std::vector<int> fl(lanes, 0);
  std::vector<std::thread> th;
  for (int l = 0; l < lanes; l++)
    th.emplace_back([&, l]() {
      const HostWaveN w{&sh, l};
      if (stage == 1) fl[l] = genStage1(w, Ap.data(), GLD, Wd.R, Wd.S, out.data());
      else if (stage == 2) fl[l] = genStage2(w, Ap.data(), GLD, Wd.R, Wd.S, fallbackCfm, out.data());
      else fl[l] = genStage3(w, Ap.data(), GLD, Wd.R, Wd.S, fallbackCfm, out.data());
    });
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&sh.bar);
  for (int l = 1; l < lanes; l++) if (fl[l] != fl[0]) return -99;      // (the lanes must agree)
  for (int r = 0; r < m; r++) X[r] = out[r];
  return fl[0];
}
}
