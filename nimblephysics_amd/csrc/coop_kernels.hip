// coop_kernels.hip — one world per WAVEFRONT for the dense part of the contact stage (see coop_dev.hpp).
//   k_contact_solve_coop   stage 0 of the LCP cascade + v' = v_pre + M^-1 J^T x; worlds it cannot resolve go to the
//                          compacted slow path (k_contact_cascade, one world per lane) exactly like k_contact_solve
//   k_bwd_contact_a_coop   the dense (c x c) part of the contact adjoint (k_bwd_contact_a)
#include "coop_dev.hpp"
#include "coop_wave_dev.hpp"

namespace nbl {

DEV void coopLoadRow(CoopRow& R, int ln, int m, const double* __restrict__ saved, const double* __restrict__ dn,
                     const SavedLayout& lay, const DevContactModel* __restrict__ cm, int64_t B, int64_t b) {
  R.m = m;
  R.fric = (ln % 3) != 0;
  R.fp = ln < MAXR ? ln - (ln % 3) : 0;
  R.mu = 0.0; R.Bv = 0.0;
  const bool on = ln < m;
  if (on) {
    const int r0 = lay.contacts + (ln / 3) * CR_SIZE;
    const double muA = cm->boxes[(int)saved[(int64_t)(r0 + CR_BOXA) * B + b]].mu, muB = cm->boxes[(int)saved[(int64_t)(r0 + CR_BOXB) * B + b]].mu;
    R.mu = muA < muB ? muA : muB;
    R.Bv = saved[(int64_t)(lay.b + ln) * B + b];
  }
  double cn = 0.0;
#pragma unroll
  for (int i = 0; i < MAXR; i++) {
    R.acol[i] = (on && i < m) ? dn[lay.A + i * MAX_ROWS + ln] : 0.0;
    cn = fma(R.acol[i], R.acol[i], cn);
  }
  R.colNorm = cn;
}

__global__ __launch_bounds__(64) void k_contact_solve_coop(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B,
                                                           double* __restrict__ saved, SavedLayout lay,
                                                           const double* __restrict__ cacheIn, double* __restrict__ cacheOut,
                                                           double* __restrict__ next, uint32_t* __restrict__ status,
                                                           double* __restrict__ lws, int32_t* __restrict__ failList,
                                                           uint32_t* __restrict__ failCount) {
  __shared__ CoopLds S;
  const DevWave w;
  const int ln = w.lane();
  const int64_t b = coopWorld(blockIdx.x, gridDim.x);
  if (b >= B) return;
  const int n = mdl.n;
  const int nC = (int)svAt(saved, lay.nc, B, b);
  const int m = 3 * nC;
  double* nv = next + (int64_t)n * B;
  double* dn = denseOf(saved, lay, B, b);
  if (m == 0) {
    if (cacheOut && ln <= MAX_ROWS) cacheOut[(int64_t)ln * B + b] = 0.0;
    if (ln < MAX_ROWS) { svAt(saved, lay.x + ln, B, b) = 0.0; svAt(saved, lay.cls + ln, B, b) = 0.0; }
    if (ln == 0) { svAt(saved, lay.cfm, B, b) = 0.0; svAt(saved, lay.pflag, B, b) = 0.0; }
    if (ln < n) svAt(saved, lay.w + ln, B, b) = 0.0;
    return;
  }
  CoopRow R;
  coopLoadRow(R, ln, m, saved, dn, lay, cm, B, b);
  const bool haveCache = cacheIn && ((int)cacheIn[(int64_t)MAX_ROWS * B + b] == m);
  const double Xcache = (haveCache && ln < m) ? cacheIn[(int64_t)ln * B + b] : 0.0;
  CoopStage0 out;
  coopStage0(w, S, R, haveCache, Xcache, out);
  if (out.ok) {
    if (ln < MAX_ROWS) {
      svAt(saved, lay.x + ln, B, b) = out.X;
      svAt(saved, lay.cls + ln, B, b) = out.K.cls == RC_UPPER_BOUND ? (out.K.E > 0 ? 2.0 : -2.0) : (double)out.K.cls;
      if (cacheOut) cacheOut[(int64_t)ln * B + b] = out.X;
    }
    if (ln == MAX_ROWS && cacheOut) cacheOut[(int64_t)MAX_ROWS * B + b] = (double)m;
    if (ln == 0) { svAt(saved, lay.cfm, B, b) = 0.0; svAt(saved, lay.pflag, B, b) = out.pinvValid ? 1.0 : 0.0; }
    // v' = v_pre + M^-1 J^T x  (lane = DOF)
    if (ln < MAXR) S.vec[2][ln] = out.X;
    w.sync();
    if (ln < n) {
      double wd = 0.0;
#pragma unroll
      for (int r = 0; r < MAXR; r++) wd = fma(dn[lay.massed + ln * MAX_ROWS + r], S.vec[2][r], wd);
      svAt(saved, lay.w + ln, B, b) = wd;
      nv[(int64_t)ln * B + b] = svAt(saved, lay.vpre + ln, B, b) + wd;
    }
    if (out.pinvValid && ln < MAXR) {
#pragma unroll
      for (int i = 0; i < MAXR; i++) dn[lay.pinv + i * MAX_ROWS + ln] = S.P[i * CLD + ln];
    }
    if (ln == 0 && status) status[b] |= 0x2u | 0x100u;
  } else {
    // the pre-solve x (mXBackup) is what the PGS fallback starts from (BoxedLcpConstraintSolver.cpp:541-547)
    if (ln < MAX_ROWS) lws[(int64_t)(LW_JA + ln) * B + b] = out.X0;
    if (ln == 0) { const uint32_t slot = atomicAdd(failCount, 1u); failList[slot] = (int32_t)b; }
  }
}

}  // namespace nbl
