// coop_dev.hpp — wavefront-cooperative dense linear algebra for the contact LCP: ONE WORLD PER WAVEFRONT.
//
// A one-world-per-lane solver is a single long dependent instruction stream per wave (~6e5 instructions for a 24-row
// stage-0 solve) and a B = 4096 launch has only 64..256 of them for 1024 SIMDs.  Here a wavefront owns one world; lane j
// (< MAXR) owns column j / row j of the MAXR x MAXR system (MAXR = 24, or 48 in the 16-contact build), and the serial
// chain per wave drops to ~1e4 instructions while 4096 waves fill the chip.  With 24 rows, lanes 24..47 carry the identity
// through the Householder factorisation (they end up holding H^T explicitly, for free); with 48 rows the reflectors are
// recorded in LDS and played back on the identity in a second pass over the same lanes (coopQrReplay).
//
// Same mathematics as the plain sequential statement in tests/host_shim/lane_lcp_statement.hpp (test harness only: the
// host-testable statement the CPU tests compare this code with):
//   * column-pivoted Householder QR, rank by Eigen's threshold eps * size * |R_00| (CGGM.cpp:280, LCPUtils.cpp:113)
//   * complete orthogonal decomposition for rank-deficient systems: a second (unpivoted) Householder QR of R^T,
//     R = [T^T 0] Z^T, minimum-norm solution Z1 T^-T c1 (what Eigen's completeOrthogonalDecomposition().solve() returns)
//   * the result is assembled as the explicit pseudo-inverse Q^+ (24 x 24) in LDS: every later solve with Q or Q^T
//     (1 in the forward pass, 4 in the backward pass) is a 24-term dot product per lane.
// Sub-matrices are selected by MASKS, not compaction: a row/column that is not in the clamping set is zeroed, which
// makes the padded pseudo-inverse equal to the compacted one embedded in zeros; the rank threshold uses the true size.
//
// Register discipline: a lane's column lives in a[0..23] with compile-time indices only; the k loop is a real loop
// and the column is SHIFTED down one row per step (row k of the result goes to LDS), so the code stays small
// (fits the instruction cache) and nothing spills to scratch.
//
// The wave primitives come from a policy class W (lane(), maxAll(), ballot(), shfl(), sync()): DevWave on the GPU
// (coop_wave_dev.hpp), a thread-per-lane emulation in tests/host_shim for the CPU tests.
#pragma once
#include "lcp_dev.hpp"

#ifndef NBL_PHASE
#define NBL_PHASE(k) do { } while (0)   // developer phase stamps (model_dev.hpp), compiled out
#endif

namespace NBL_NS {

// A value the optimiser must treat as new at this point: keeps loop-invariant 24-entry constants (the identity columns of the
// carried block, ...) from being hoisted out of the standardisation loop and parked in 48 VGPRs across both factorisations.
DEV int opaqueI(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
  return x;
}

// Nothing is scheduled across this point (device; no code): bounds how far ahead the compiler hoists the LDS loads of the next dense
// product, i.e. how many registers the products of a kernel hold at once.
DEV void coopSchedFence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// The two pseudo-inverse routes are inlined into their callers in the 24-row build (measured: out of line costs 10 us per call there, see
// coopPinv below).  The 48-row build - whose job is to exist, not to set the headline - keeps them out of line: one copy per kernel
// instead of up to four fully unrolled 48-step factorisations (compile time and code size).
#if NBL_MAXC > 8 && defined(__HIP_DEVICE_COMPILE__)
#define DEV_PINV __device__ __noinline__
#else
#define DEV_PINV DEV
#endif

constexpr int CLD = MAXR + 1;  // odd leading dimension: row reads and column reads of the LDS matrices are both conflict-free
constexpr int MFMA_TILES = (MAXR + 15) / 16;   // 16 x 16 output tiles per dimension of a MAXR x MAXR product on v_mfma_f64_16x16x4_f64

// The block carried through a Householder factorisation (the identity, ending as H^T / Z^T) rides on lanes MAXR .. 2 MAXR - 1 of the same
// pass when they exist (24 rows); otherwise (48 rows) every reflector stays in LDS (CoopLds::refl) and coopQrReplay applies them to the
// carried columns afterwards, on lanes 0 .. MAXR - 1.
constexpr bool QR_CARRY_LANES = 2 * MAXR <= 64;

struct CoopLds {
  double R[MAXR * CLD];        // rows of R (lane order; perm maps pivot position -> lane), later rows of T
  double G[MAXR * CLD];        // rows of G = H^T
  double P[MAXR * CLD];        // rows of Z^T during the second factorisation, then the pseudo-inverse (row-major)
  double vbuf[2][MAXR + 2];    // reflector broadcast, double-buffered: [1..MAXR-1] v (v_0 = 1 implied), [MAXR] tau, [MAXR+1] 1 / v_k
  double refl[QR_CARRY_LANES ? 2 : MAXR * (MAXR + 2)];   // !QR_CARRY_LANES: the reflector of every step, same layout as vbuf
  double invd[MAXR];           // reciprocals of the diagonal of R / T
  double vec[4][MAXR];         // vector broadcast scratch
  int perm[MAXR];
  int pad[4];
};
static_assert(sizeof(CoopLds) % 16 == 0, "CoopLds must keep 16-byte alignment in an array");

// Householder QR of the matrix whose column j is a[] of lane j (< 24); lanes 24..47 carry extra columns that only
// receive the reflections.  PIVOT: column pivoting with the rank test, else the pivot of step k is lane k and exactly
// `steps` reflections are made.  Row k of the triangular factor goes to rowsOut[k * CLD + lane], row k of the carried
// block to carryOut[k * CLD + lane - 24].  a[] is consumed.  Returns the number of reflections made (the rank).
template <class W, bool PIVOT>
DEV int coopQr(const W& w, double (&a)[MAXR], CoopLds& S, double* rowsOut, double* carryOut, int steps, double thr2) {
  const int ln = w.lane();
  bool done = false;
  double best0 = 0.0;
  int rank = 0;
#pragma unroll 1
  for (int k = 0; k < steps; k++) {
    // four partial sums: a 23-deep chain of dependent FMAs would cost ~23 x the FMA latency with nothing to overlap
    double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
#pragma unroll
    for (int i = 1; i + 3 < MAXR; i += 4) {
      b0 = fma(a[i], a[i], b0); b1 = fma(a[i + 1], a[i + 1], b1); b2 = fma(a[i + 2], a[i + 2], b2); b3 = fma(a[i + 3], a[i + 3], b3);
    }
#pragma unroll
    for (int i = 1 + 4 * ((MAXR - 1) / 4); i < MAXR; i++) b0 = fma(a[i], a[i], b0);
    const double below = (b0 + b1) + (b2 + b3);
    const double nrm = fma(a[0], a[0], below);
    // Every lane prepares the reflector scalars of ITS column before the pivot is known: the sqrt -> divide chain
    // (~60 dependent instructions) then overlaps with the wave-wide arg-max below instead of following it on one lane.
    const double akk = a[0];
    const double normx = sqrt(nrm);
    const double alpha = akk > 0 ? -normx : normx;
    const double vk = akk - alpha;                   // v = x - alpha e_k, scaled so that v_k = 1
    const double vnorm2 = fma(vk, vk, below);
    const double inv = 1.0 / vk;
    const double tauMine = 2.0 * vk * vk / vnorm2;   // tau:  H = I - tau v v^T   (unused lanes may hold inf / NaN)
    int p = k;
    if (PIVOT) {
      const double cand = (ln < MAXR && !done) ? nrm : -1.0;
      const double best = w.maxAll(cand);
      if (k == 0) best0 = best;
      if (!(best > thr2 * best0) || !(best > 0.0)) break;
      p = __builtin_ctzll(w.ballot(cand == best));
    }
    // The pivot lane is on everybody's critical path (all lanes wait for its reflector at the sync), so it does the minimum:
    // it publishes its column UNSCALED plus the two scalars (tau, 1 / v_k); the scaling v = x / v_k is folded into each
    // lane's own dot product and update (two extra multiplies per lane instead of 23 on the pivot lane).  Its own column is
    // not zeroed either: a finished column is never a candidate again and its rows of the triangular factor are written as
    // zeros below (one select instead of 23 moves).
    const bool wasDone = done;
    double* vb = QR_CARRY_LANES ? S.vbuf[k & 1] : S.refl + k * (MAXR + 2);
    if (ln == p) {
      vb[MAXR] = tauMine;
      vb[MAXR + 1] = inv;
#pragma unroll
      for (int i = 1; i < MAXR; i++) vb[i] = a[i];
      a[0] = alpha;
      done = true;
      if (PIVOT) S.perm[k] = p;
    }
    w.sync();
    const double tau = vb[MAXR], vinv = vb[MAXR + 1];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int i = 1; i + 3 < MAXR; i += 4) {
      d0 = fma(vb[i], a[i], d0); d1 = fma(vb[i + 1], a[i + 1], d1); d2 = fma(vb[i + 2], a[i + 2], d2); d3 = fma(vb[i + 3], a[i + 3], d3);
    }
#pragma unroll
    for (int i = 1 + 4 * ((MAXR - 1) / 4); i < MAXR; i++) d0 = fma(vb[i], a[i], d0);
    double d = fma(vinv, (d0 + d1) + (d2 + d3), a[0]);
    d = (ln == p) ? 0.0 : d * tau;
    a[0] -= d;
    const double dv = d * vinv;
#pragma unroll
    for (int i = 1; i < MAXR; i++) a[i] = fma(-dv, vb[i], a[i]);   // v re-read from LDS: cheaper than 48 live VGPRs
    if (ln < MAXR) rowsOut[k * CLD + ln] = wasDone ? 0.0 : a[0];
    else if (QR_CARRY_LANES && ln < 2 * MAXR) carryOut[k * CLD + ln - MAXR] = a[0];
#pragma unroll
    for (int i = 0; i < MAXR - 1; i++) a[i] = a[i + 1];
    a[MAXR - 1] = 0.0;
    rank = k + 1;
  }
  if (PIVOT) {
    // columns never chosen (dependent or masked) take the remaining pivot positions in lane order
    const bool un = ln < MAXR && !done;
    const uint64_t um = w.ballot(un);
    if (un) S.perm[rank + __builtin_popcountll(um & ((1ull << ln) - 1ull))] = ln;
  }
  w.sync();
  return rank;
}

// !QR_CARRY_LANES: the `steps` reflectors the last coopQr left in S.refl, applied to the carried block whose column j is a[] of lane j
// (< MAXR) - the arithmetic of the carried lanes of coopQr, one pass later.  Row k of the result goes to carryOut[k * CLD + lane].
template <class W>
DEV void coopQrReplay(const W& w, double (&a)[MAXR], CoopLds& S, double* carryOut, int steps) {
  const int ln = w.lane();
#pragma unroll 1
  for (int k = 0; k < steps; k++) {
    const double* vb = S.refl + k * (MAXR + 2);
    const double tau = vb[MAXR], vinv = vb[MAXR + 1];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int i = 1; i + 3 < MAXR; i += 4) {
      d0 = fma(vb[i], a[i], d0); d1 = fma(vb[i + 1], a[i + 1], d1); d2 = fma(vb[i + 2], a[i + 2], d2); d3 = fma(vb[i + 3], a[i + 3], d3);
    }
#pragma unroll
    for (int i = 1 + 4 * ((MAXR - 1) / 4); i < MAXR; i++) d0 = fma(vb[i], a[i], d0);
    double d = fma(vinv, (d0 + d1) + (d2 + d3), a[0]);
    d = d * tau;
    a[0] -= d;
    const double dv = d * vinv;
#pragma unroll
    for (int i = 1; i < MAXR; i++) a[i] = fma(-dv, vb[i], a[i]);
    if (ln < MAXR) carryOut[k * CLD + ln] = a[0];
#pragma unroll
    for (int i = 0; i < MAXR - 1; i++) a[i] = a[i + 1];
    a[MAXR - 1] = 0.0;
  }
  w.sync();
}

// a[] <- column e of the identity (e out of range: zero)
DEV void coopIdentityColumn(double (&a)[MAXR], int eIn) {
  const int e = opaqueI(eIn);
#pragma unroll
  for (int i = 0; i < MAXR; i++) a[i] = (e == i) ? 1.0 : 0.0;
}

DEV double coopRsqrt(double x);
#if defined(NBL_CASCADE_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_pinvStat[8];   // developer counters of the Householder route (tools/cascade_timing.py): calls, cycles of the parts
#endif
#if defined(NBL_CASCADE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define PINV_T0() long long pvT = clock64()
#define PINV_ADD(k) do { const long long pvN = clock64(); if (ln == 0) atomicAdd(&g_pinvStat[k], (unsigned long long)(pvN - pvT)); pvT = pvN; } while (0)
#define PINV_CNT(k) do { if (ln == 0) atomicAdd(&g_pinvStat[k], 1ull); } while (0)
#else
#define PINV_T0() do { } while (0)
#define PINV_ADD(k) do { } while (0)
#define PINV_CNT(k) do { } while (0)
#endif

// S.P <- pseudo-inverse of the MAXR x MAXR matrix whose column j is a[] of lane j (< MAXR) (masked rows/columns zero);
// cTrue = number of unmasked columns (Eigen's `size` in the rank threshold).  Returns the rank.
template <class W>
DEV_PINV int coopPinvImpl(const W& w, double (&a)[MAXR], CoopLds& S, int cTrue) {
  const int ln = w.lane();
  if (QR_CARRY_LANES && ln >= MAXR) coopIdentityColumn(a, ln - MAXR);
  const double thr = 2.220446049250313e-16 * cTrue;
  PINV_T0();
  PINV_CNT(0);
  const int r = coopQr<W, true>(w, a, S, S.R, S.G, MAXR, thr * thr);
  PINV_ADD(1);
  if (!QR_CARRY_LANES && r > 0) {
    coopIdentityColumn(a, ln < MAXR ? ln : -1);
    coopQrReplay(w, a, S, S.G, r);
  }
  if (r == 0) {
#pragma unroll
    for (int i = 0; i < MAXR; i++) if (ln < MAXR) S.P[i * CLD + ln] = 0.0;
    w.sync();
    return 0;
  }
  double g[MAXR];
  if (r >= cTrue) {
    // full column rank on the unmasked columns (R2 = 0): column j of Q^+ is P R1^-1 G1[:, j], back substitution,
    // the current unknown always in g[0]
    if (ln < r) S.invd[ln] = 1.0 / S.R[ln * CLD + S.perm[ln]];
    w.sync();
    const int j = ln < MAXR ? ln : 0;
#pragma unroll
    for (int m = 0; m < MAXR; m++) g[m] = (m < r) ? S.G[(r - 1 - m) * CLD + j] : 0.0;
#pragma unroll 1
    for (int kk = r - 1; kk >= 0; kk--) {
      const int pk = S.perm[kk];
      const double yk = g[0] * S.invd[kk];
      if (ln < MAXR) S.P[pk * CLD + ln] = yk;
#pragma unroll
      for (int m = 1; m < MAXR; m++) {
        const int row = kk - m;
        const double rv = row >= 0 ? S.R[(row >= 0 ? row : 0) * CLD + pk] : 0.0;
        g[m - 1] = fma(-rv, yk, g[m]);
      }
      g[MAXR - 1] = 0.0;
    }
#pragma unroll 1
    for (int pp = r; pp < MAXR; pp++) if (ln < MAXR) S.P[S.perm[pp] * CLD + ln] = 0.0;
    w.sync();
    return r;
  }
  if constexpr (QR_CARRY_LANES != 0) {
    // rank deficient (two flat feet: rank 12 of up to 23 clamping rows): R = [R1 R2] = R1 [I W] in pivot order, W = R1^-1 R2, and the
    // minimum-norm solution of R u = g is u = [I; W^T] (I + W W^T)^-1 R1^-1 g  (R R^T = R1 (I + W W^T) R1^T).  The complete orthogonal
    // decomposition's second Householder pass over R^T - r more reflector steps, each a norm, a square root, a division and two sweeps
    // over 24 entries - becomes two triangular substitutions and the Cholesky factorisation of the r x r matrix I + W W^T, which column
    // pivoting keeps well conditioned (|W| <~ 1) whatever the condition of Q: the accuracy is that of the triangular solves with R1,
    // cond(Q) eps like the second pass (tests/test_coop_host.py).  Lane j < 24: column j of G1 (the first r rows of Q_h^T); lane 24 + t:
    // the remaining column perm[r + t] of R (masked columns are zero columns: their W is zero, their row of Q^+ comes out zero).
    const bool gl = ln < MAXR;
    const int tW = ln - MAXR;
    const bool wl = ln >= MAXR && tW < MAXR - r;
    if (ln < r) S.invd[ln] = 1.0 / S.R[ln * CLD + S.perm[ln]];
    const double* src = gl ? S.G : S.R;
    const int col = gl ? ln : (wl ? S.perm[r + tW] : 0);
#pragma unroll
    for (int m = 0; m < MAXR; m++) {
      const double v = src[(m < r ? m : 0) * CLD + col];
      g[m] = (m < r && (gl || wl)) ? v : 0.0;
    }
    w.sync();
    // x = R1^-1 (column): back substitution, R1[i][k] = S.R[i][perm[k]]   (reading the pivot columns and reciprocal pivots up front so
    // that the loads of R1 run ahead of the substitution costs 72 registers: measured slower, the kernels sit at 256)
#pragma unroll
    for (int k = MAXR - 1; k >= 0; k--) {
      if (k < r) {
        const int pk = S.perm[k];
        const double xk = g[k] * S.invd[k];
        g[k] = xk;
#pragma unroll
        for (int i = 0; i < k; i++) g[i] = fma(-S.R[i * CLD + pk], xk, g[i]);
      }
    }
    PINV_ADD(2);
    w.sync();    // every lane has its column of G1 / R in registers and is done with R1: the buffers are free
    if (ln >= MAXR && ln < 2 * MAXR) {
#pragma unroll
      for (int i = 0; i < MAXR; i++) S.G[i * CLD + tW] = g[i];       // W[i][t] (zero beyond the rank and beyond the remaining columns)
    }
    w.sync();
    // I + W W^T -> S.P (r x r; the identity is added when the Cholesky factorisation reads it)
#if defined(__HIP_DEVICE_COMPILE__)
    {
      typedef double v4d __attribute__((ext_vector_type(4)));
      const int li = ln & 15, lk = ln >> 4;
      const int ksteps = (MAXR - r + 3) >> 2;
#pragma unroll
      for (int tr = 0; tr < MFMA_TILES; tr++) {
#pragma unroll
        for (int tj = 0; tj < MFMA_TILES; tj++) {
          if (r > 16 * (tr > tj ? tr : tj)) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            const int mi = 16 * tr + li, ni = 16 * tj + li;
            const int mc = mi < MAXR ? mi : 0, nc2 = ni < MAXR ? ni : 0;
#pragma unroll
            for (int ks = 0; ks < MAXR / 4; ks++) {
              if (ks < ksteps) {
                const int kk = 4 * ks + lk;
                const double av = S.G[mc * CLD + kk], bv = S.G[nc2 * CLD + kk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(mi < r ? av : 0.0, ni < r ? bv : 0.0, acc, 0, 0, 0);
              }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int rr = 16 * tr + lk + 4 * q, cc = 16 * tj + li;
              if (rr < MAXR && cc < MAXR) S.P[rr * CLD + cc] = acc[q];
            }
          }
        }
      }
    }
#else
    if (gl) {
      for (int t2 = 0; t2 < MAXR; t2++) {
        double sum = 0.0;
        for (int jj = 0; jj < MAXR - r; jj++) sum += (t2 < r && ln < r) ? S.G[t2 * CLD + jj] * S.G[ln * CLD + jj] : 0.0;
        S.P[t2 * CLD + ln] = sum;
      }
    }
#endif
    w.sync();
    PINV_ADD(3);
    // I + W W^T = L L^T, lane t (< r) = row t of L; rows also in S.R for the broadcast, reciprocal diagonal in S.invd
    {
      const int row = gl ? ln : 0;
      double l2[MAXR];
#pragma unroll
      for (int i = 0; i < MAXR; i++) l2[i] = 0.0;
#pragma unroll
      for (int k = 0; k < MAXR; k++) {
        if (k < r) {
          double s0 = S.P[row * CLD + k] + (row == k ? 1.0 : 0.0), s1 = 0.0;
#pragma unroll
          for (int i = 0; i < k; i++) {
            const double lk2 = S.R[k * CLD + i];       // L[k][i], broadcast
            if (i & 1) s1 = fma(-l2[i], lk2, s1); else s0 = fma(-l2[i], lk2, s0);
          }
          const double sK = s0 + s1;
          const double inv = coopRsqrt(w.bcast(sK, k));
          const double v = (ln >= k && ln < r) ? sK * inv : 0.0;
          l2[k] = v;
          if (gl) S.R[row * CLD + k] = v;
          if (ln == 0) S.invd[k] = inv;
          w.sync();
        }
      }
    }
    PINV_ADD(4);
    // z = (L L^T)^-1 x for the columns of G1: forward with L, backward with L^T
#pragma unroll
    for (int k = 0; k < MAXR; k++) {
      if (k < r) {
        double s0 = g[k], s1 = 0.0;
#pragma unroll
        for (int i = 0; i < k; i++) {
          const double lk2 = S.R[k * CLD + i];
          if (i & 1) s1 = fma(-lk2, g[i], s1); else s0 = fma(-lk2, g[i], s0);
        }
        g[k] = (s0 + s1) * S.invd[k];
      }
    }
#pragma unroll
    for (int k = MAXR - 1; k >= 0; k--) {
      if (k < r) {
        const double wk = g[k] * S.invd[k];
        g[k] = wk;
#pragma unroll
        for (int i = 0; i < k; i++) g[i] = fma(-S.R[k * CLD + i], wk, g[i]);
      }
    }
    PINV_ADD(5);
    // column ln of Q^+ = P [z; W^T z]   (W^T z as a (24 - r) x r x 24 product on the matrix cores: 11 k -> 4 k cycles here, and 25 k more
    // elsewhere in the kernels that inline this - they sit at 256 registers and the accumulator tiles spill)
#pragma unroll
    for (int k = 0; k < MAXR; k++) if (k < r && gl) S.P[S.perm[k] * CLD + ln] = g[k];
#pragma unroll 1
    for (int tt = 0; tt < MAXR - r; tt++) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i4 = 0; i4 < MAXR; i4 += 4) {
        if (i4 < r) {
          s0 = fma(S.G[i4 * CLD + tt], g[i4], s0); s1 = fma(S.G[(i4 + 1) * CLD + tt], g[i4 + 1], s1);
          s0 = fma(S.G[(i4 + 2) * CLD + tt], g[i4 + 2], s0); s1 = fma(S.G[(i4 + 3) * CLD + tt], g[i4 + 3], s1);
        }
      }
      if (gl) S.P[S.perm[r + tt] * CLD + ln] = s0 + s1;
    }
    w.sync();
    PINV_ADD(6);
    return r;
  }
  // rank deficient: R = [R1 R2] (r x c, pivot order).  Second factorisation R^T = Z [T; 0]: lane i (< r) takes row i
  // of R as its column (entries in pivot order), lanes 24..47 carry the identity and end as Z^T.
  if (ln < MAXR) {
#pragma unroll
    for (int pp = 0; pp < MAXR; pp++) a[pp] = (ln < r) ? S.R[ln * CLD + S.perm[pp]] : 0.0;
  } else if (QR_CARRY_LANES) coopIdentityColumn(a, ln - MAXR);
  w.sync();   // every row of R is in registers before T overwrites the buffer
  coopQr<W, false>(w, a, S, S.R, S.P, r, 0.0);
  if (!QR_CARRY_LANES) {
    coopIdentityColumn(a, ln < MAXR ? ln : -1);
    coopQrReplay(w, a, S, S.P, r);
  }
  if (ln < r) S.invd[ln] = 1.0 / S.R[ln * CLD + ln];
  w.sync();
  // column j of Q^+ = P Z1 T^-T G1[:, j] in two passes, so that the 24 substitution registers and the 24 accumulators are never
  // live together: (1) forward substitution with T^T, w_k parked in this lane's own column of the G buffer (its entries are
  // all in registers by then), (2) y = Z1 w.
  const int j = ln < MAXR ? ln : 0;
#pragma unroll
  for (int m = 0; m < MAXR; m++) g[m] = (m < r) ? S.G[m * CLD + j] : 0.0;
#pragma unroll 1
  for (int k = 0; k < r; k++) {
    const double wk = g[0] * S.invd[k];
    if (ln < MAXR) S.G[k * CLD + j] = wk;
#pragma unroll
    for (int m = 1; m < MAXR; m++) {
      const int col = k + m;
      const double tv = col < r ? S.R[k * CLD + (col < MAXR ? col : 0)] : 0.0;      // T[k][col]
      g[m - 1] = fma(-tv, wk, g[m]);
    }
    g[MAXR - 1] = 0.0;
  }
  double y[MAXR];
#pragma unroll
  for (int pp = 0; pp < MAXR; pp++) y[pp] = 0.0;
#pragma unroll 1
  for (int k = 0; k < r; k++) {
    const double wk = S.G[k * CLD + j];
#pragma unroll
    for (int pp = 0; pp < MAXR; pp++) y[pp] = fma(S.P[k * CLD + pp], wk, y[pp]);   // Z[pp][k] = Z^T[k][pp]
  }
  w.sync();   // all reads of Z^T done before the buffer becomes Q^+
#pragma unroll
  for (int pp = 0; pp < MAXR; pp++) if (ln < MAXR) S.P[S.perm[pp] * CLD + ln] = y[pp];
  w.sync();
  return r;
}

// 1 / sqrt(x) of a wave-uniform positive x.  Device: the hardware estimate (v_rsq_f64) and two Newton steps - 10 dependent
// instructions instead of the ~35 of sqrt followed by a division; good to the last bit or two, which is all a factorisation needs.
DEV double coopRsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

// S.P <- pseudo-inverse of a SYMMETRIC POSITIVE SEMI-DEFINITE 24 x 24 matrix (masked rows / columns zero) whose column j (= row j) is
// a[] of lane j (< 24): A restricted to the guess rows, or Q = A(clamping, clamping) + cfm I when no friction row sits on its bound.
// Same contract as coopPinv (cTrue = number of unmasked columns, returns the rank), about a third of its instructions and of its
// dependent chain:
//   1. diagonally pivoted Cholesky  Q = G G^T,  G 24 x r (rows in lane order - nothing is permuted; rows in LDS); rank by the same
//      threshold as the reference's complete orthogonal decomposition, eps * size, on the pivots (the pivots of both factorisations
//      sit at the scale of the singular values of Q)
//   2. K = G^T G  (r x r, positive definite)                                   one GEMM: the matrix cores on the device
//   3. Cholesky K = L L^T, lane t = row t of L
//   4. W = G K^-1: lane j solves L y = g_j, L^T w = y for its row  (L broadcast from LDS, the right-hand side in registers)
//   5. Q^+ = W W^T = G (G^T G)^-2 G^T                                          one GEMM: the matrix cores on the device
// Accuracy: both Cholesky factorisations are backward stable and K has the condition number of Q on its range, so Q^+ carries
// cond(Q) eps like the QR route (measured on 1200 contact matrices of the metric, box-stack and soak distributions: rank equal to
// the COD's in every case, Q^+ b to 1e-9 at cond 3e6, median 2e-14).  Non-symmetric Q (a friction row on its bound folds its column
// into its normal's) stays with coopPinv.
// Code shape: fully unrolled over the factorisation step k, so that step k does exactly k multiply-adds per lane (the left-looking dot
// products and the substitutions are triangular: a rolled loop over fixed-length rows does 2-4 x the work) with every LDS offset a
// compile-time constant; a full-rank Q (every Q with the fallback CFM on its diagonal) skips steps 2-3: there W = G^-T, one
// substitution with the (row-permuted) triangular G itself.
template <class W>
DEV_PINV int coopPinvSymImpl(const W& w, CoopLds& S, int cTrue) {      // Q in S.R[i][j] (symmetric)
  const int ln = w.lane();
  const bool act = ln < MAXR;
  const int row = act ? ln : 0;
  double d = act ? S.R[row * CLD + row] : -1.0;     // remaining diagonal of this lane's row
  bool done = !act;
  double g[MAXR];                                   // this lane's row of G, later of W
#pragma unroll
  for (int i = 0; i < MAXR; i++) g[i] = 0.0;
  // Rank threshold.  The reference's complete orthogonal decomposition drops a column when |R_kk| <= eps * size * |R_00|; on an exactly
  // singular Q (four coplanar corners per foot, two contacts at one point) the quantity tested is pure round-off, a few eps, in
  // either factorisation - only a factor ~5-10 below that threshold, and the round-off of a down-dated Cholesky pivot is not the
  // round-off of a Householder column (soak seed 30168: the COD says rank 5, a pivot of 1.4e-15 passes eps * 6).  64 x the threshold
  // keeps every round-off pivot out; it differs from the reference's decision only for a Q whose trailing singular value sits between
  // 5e-15 and 3e-13 of the largest - where the reference's own solution is round-off times 1e12.
  const double thr = 64.0 * 2.220446049250313e-16 * cTrue;
  double best0 = 0.0;
  int r = 0;
  // ---- 1. pivoted Cholesky, left-looking: column k of G from column p of Q and the pivot row so far ----
#pragma unroll
  for (int k = 0; k < MAXR; k++) {
    const double cand = done ? -1.0 : d;
    const double best = w.maxAll(cand);
    if (k == 0) best0 = best;
    if (!(best > thr * best0) || !(best > 0.0)) break;
    const int p = __builtin_ctzll(w.ballot(cand == best));
    const double inv = coopRsqrt(best);              // overlaps with the dot product below
    double s0 = S.R[row * CLD + p], s1 = 0.0;
#pragma unroll
    for (int i = 0; i < k; i++) {
      const double pr = S.G[p * CLD + i];            // G[p][i], broadcast
      if (i & 1) s1 = fma(-g[i], pr, s1); else s0 = fma(-g[i], pr, s0);
    }
    const double gk = done ? 0.0 : (s0 + s1) * inv;  // rows already pivoted are exactly zero from here on
    g[k] = gk;
    d = fma(-gk, gk, d);
    if (ln == p) done = true;
    if (act) S.G[row * CLD + k] = gk;
    if (ln == 0) { S.perm[k] = p; S.invd[k] = inv; }
    w.sync();
    r = k + 1;
  }
  if (r == 0) {
#pragma unroll
    for (int i = 0; i < MAXR; i++) if (act) S.P[i * CLD + ln] = 0.0;
    w.sync();
    return 0;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double v4d __attribute__((ext_vector_type(4)));
  const int li = ln & 15, lk = ln >> 4;
#endif
  if (r >= cTrue) {
    // ---- full rank on the unmasked block: Q^-1 = G^-T G^-1, W = G^-T.  Row j of W = column j of G^-1 = the solution of G x = e_j;
    //      G's row perm[k] ends at column k, so x_k comes from that row: x_k = ([j == perm[k]] - sum_{i<k} G[perm[k]][i] x_i) / G[perm[k]][k]
    //      (1 / G[perm[k]][k] = the reciprocal square root of pivot k).  Masked lanes are no pivot: their row stays zero. ----
#pragma unroll
    for (int k = 0; k < MAXR; k++) {
      if (k < r) {
        const int pk = S.perm[k];
        double s0 = (ln == pk) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < k; i++) {
          const double gp = S.G[pk * CLD + i];
          if (i & 1) s1 = fma(-gp, g[i], s1); else s0 = fma(-gp, g[i], s0);
        }
        g[k] = (s0 + s1) * S.invd[k];                // (g[] is overwritten front to back: entries < k are x, entries >= k still G)
      }
    }
  } else {
    // ---- 2. K = G^T G -> S.P (r x r) ----
#if defined(__HIP_DEVICE_COMPILE__)
    // v_mfma_f64_16x16x4_f64: A operand lane l = A[l & 15][l >> 4], B operand lane l = B[l >> 4][l & 15], D register q of lane l =
    // D[(l >> 4) + 4 q][l & 15].  K[m][n] = sum_j G[j][m] G[j][n]: both operands read G[4 ks + lk][16 t + li].
#pragma unroll
    for (int tr = 0; tr < MFMA_TILES; tr++) {
#pragma unroll
      for (int tj = 0; tj < MFMA_TILES; tj++) {
        if (r > 16 * (tr > tj ? tr : tj)) {            // ranks up to 16 (two flat feet: 12) need one tile
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          const int mi = 16 * tr + li, ni = 16 * tj + li;
          const int mc = mi < MAXR ? mi : 0, nc2 = ni < MAXR ? ni : 0;
#pragma unroll
          for (int ks = 0; ks < MAXR / 4; ks++) {
            const int kk = 4 * ks + lk;
            const double av = S.G[kk * CLD + mc], bv = S.G[kk * CLD + nc2];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(mi < r ? av : 0.0, ni < r ? bv : 0.0, acc, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int rr = 16 * tr + lk + 4 * q, cc = 16 * tj + li;
            if (rr < MAXR && cc < MAXR) S.P[rr * CLD + cc] = acc[q];
          }
        }
      }
    }
#else
    if (act) {
      for (int t = 0; t < MAXR; t++) {
        double sum = 0.0;
        for (int j = 0; j < MAXR; j++) sum += (t < r && ln < r) ? S.G[j * CLD + t] * S.G[j * CLD + ln] : 0.0;
        S.P[t * CLD + ln] = sum;
      }
    }
#endif
    w.sync();
    // ---- 3. K = L L^T, lane t (< r) = row t of L; rows also in S.R for the broadcast, reciprocal diagonal in S.invd ----
    {
      double l2[MAXR];
#pragma unroll
      for (int i = 0; i < MAXR; i++) l2[i] = 0.0;
#pragma unroll
      for (int k = 0; k < MAXR; k++) {
        if (k < r) {
          double s0 = S.P[row * CLD + k], s1 = 0.0;
#pragma unroll
          for (int i = 0; i < k; i++) {
            const double lk2 = S.R[k * CLD + i];       // L[k][i], broadcast
            if (i & 1) s1 = fma(-l2[i], lk2, s1); else s0 = fma(-l2[i], lk2, s0);
          }
          const double sK = s0 + s1;
          const double inv = coopRsqrt(w.bcast(sK, k));
          const double v = (ln >= k && ln < r) ? sK * inv : 0.0;
          l2[k] = v;
          if (act) S.R[row * CLD + k] = v;
          if (ln == 0) S.invd[k] = inv;
          w.sync();
        }
      }
    }
    // ---- 4. this lane's row of W = G K^-1: forward with L, backward with L^T ----
#pragma unroll
    for (int k = 0; k < MAXR; k++) {
      if (k < r) {
        double s0 = g[k], s1 = 0.0;
#pragma unroll
        for (int i = 0; i < k; i++) {
          const double lk2 = S.R[k * CLD + i];
          if (i & 1) s1 = fma(-lk2, g[i], s1); else s0 = fma(-lk2, g[i], s0);
        }
        g[k] = (s0 + s1) * S.invd[k];
      }
    }
#pragma unroll
    for (int k = MAXR - 1; k >= 0; k--) {
      if (k < r) {
        const double wk = g[k] * S.invd[k];
        g[k] = wk;
#pragma unroll
        for (int i = 0; i < k; i++) g[i] = fma(-S.R[k * CLD + i], wk, g[i]);
      }
    }
  }
  w.sync();
#pragma unroll
  for (int t = 0; t < MAXR; t++) if (act) S.G[row * CLD + t] = g[t];      // W (columns >= r are zero)
  w.sync();
  // ---- 5. Q^+ = W W^T -> S.P ----
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const int ksteps = (r + 3) >> 2;
#pragma unroll
    for (int tr = 0; tr < MFMA_TILES; tr++) {
#pragma unroll
      for (int tj = 0; tj < MFMA_TILES; tj++) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        const int mi = 16 * tr + li, ni = 16 * tj + li;
        const int mc = mi < MAXR ? mi : 0, nc2 = ni < MAXR ? ni : 0;
#pragma unroll
        for (int ks = 0; ks < MAXR / 4; ks++) {
          if (ks < ksteps) {
            const int kk = 4 * ks + lk;
            const double av = S.G[mc * CLD + kk], bv = S.G[nc2 * CLD + kk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(mi < MAXR ? av : 0.0, ni < MAXR ? bv : 0.0, acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int rr = 16 * tr + lk + 4 * q, cc = 16 * tj + li;
          if (rr < MAXR && cc < MAXR) S.P[rr * CLD + cc] = acc[q];
        }
      }
    }
  }
#else
  if (act) {
    double out[MAXR];
    for (int c = 0; c < MAXR; c++) {
      double sum = 0.0;
      for (int t = 0; t < MAXR; t++) sum += S.G[ln * CLD + t] * S.G[c * CLD + t];
      out[c] = sum;
    }
    for (int c = 0; c < MAXR; c++) S.P[ln * CLD + c] = out[c];
  }
#endif
  w.sync();
  return r;
}

// S.P <- pseudo-inverse of the 24 x 24 matrix whose column j is a[] of lane j (< 24): Householder QR + complete orthogonal decomposition.
// (Both routes were also tried as OUT-OF-LINE device functions with the matrix passed through LDS - one copy per kernel instead of two
// to four inlined ones: the kernels keep their ~220 registers (the callee's count is the kernel's) and every call costs ~10 us of
// k_contact_solve_coop in callee-saved spills: 6.11 -> 5.84 M/s.  Inlined.)
template <class W>
DEV int coopPinv(const W& w, double (&a)[MAXR], CoopLds& S, int cTrue) { return coopPinvImpl(w, a, S, cTrue); }
// ... of a symmetric positive semi-definite matrix: two Cholesky factorisations
template <class W>
DEV int coopPinvSym(const W& w, double (&a)[MAXR], CoopLds& S, int cTrue) {
#ifdef NBL_NO_PINV_SYM       // A/B build (tools): every pseudo-inverse by the Householder route
  return coopPinvImpl(w, a, S, cTrue);
#endif
  const int ln = w.lane();
#pragma unroll
  for (int i = 0; i < MAXR; i++) if (ln < MAXR) S.R[i * CLD + ln] = a[i];      // Q into LDS: S.R[i][j] (column j by lane j; symmetric)
  w.sync();
  return coopPinvSymImpl(w, S, cTrue);
}

// y_lane = sum_k P[lane][k] x_k (TRANS: P[k][lane]) with x given one entry per lane (lanes >= 24 ignored)
template <class W, bool TRANS>
DEV double coopPinvApply(const W& w, CoopLds& S, double xLane, int slot) {
  const int ln = w.lane();
  if (ln < MAXR) S.vec[slot][ln] = xLane;
  w.sync();
  const int i = ln < MAXR ? ln : 0;
  double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;   // four partial sums instead of one 24-deep dependent chain
  // a REAL loop of three blocks of eight terms: fully unrolled, the 48 LDS loads of a product (and, hoisted, of the NEXT products of the
  // caller) are all issued up front - k_bwd_contact_a_coop, five such products in a row, sat at 256 VGPRs + 192 AGPRs
#pragma unroll 1
  for (int kb = 0; kb < MAXR; kb += 8) {
#pragma unroll
    for (int kq = 0; kq < 8; kq += 4) {
      const int k = kb + kq;
      y0 = fma(TRANS ? S.P[k * CLD + i] : S.P[i * CLD + k], S.vec[slot][k], y0);
      y1 = fma(TRANS ? S.P[(k + 1) * CLD + i] : S.P[i * CLD + k + 1], S.vec[slot][k + 1], y1);
      y2 = fma(TRANS ? S.P[(k + 2) * CLD + i] : S.P[i * CLD + k + 2], S.vec[slot][k + 2], y2);
      y3 = fma(TRANS ? S.P[(k + 3) * CLD + i] : S.P[i * CLD + k + 3], S.vec[slot][k + 3], y3);
    }
  }
  const double y = (y0 + y1) + (y2 + y3);
  return y;
}

// ---- per-row (lane = LCP row) pieces of LCPUtils / CGGM, see lcp_dev.hpp for the one-world-per-lane statement ----
struct CoopRow {
  int m;            // rows of this world's LCP (uniform)
  bool fric;        // this lane's row is a friction row
  int fp;           // lane of the normal row of this lane's contact
  double mu, Bv, colNorm;
  const double* Acol;  // &A[0][lane] (stride MAXR): column (= row, A is symmetric) `lane` of A.  Re-read where needed
                       // instead of being held in 48 VGPRs across the factorisations (it stays in L2).
  // joint-limit rows (model_dev.hpp, CT_LIMIT; only the general - MULTI - instantiation of the contact kernels sets these): this lane's row
  // is one / it is carried negated (upper limit) / the limit rows of the world (uniform)
  bool lim = false, neg = false;
  RowMask limMask = 0;
  bool on;             // this lane's row exists in the problem being solved: lane < m and, where a world has several
                       // constrained groups, its row belongs to the group at hand (rows that are off are inert everywhere)
  DEV double a(int i) const { return (on && i < m) ? Acol[i * MAXR] : 0.0; }
  // The same column through a pointer the optimiser cannot see through: without it the 24 loads are hoisted out of the
  // standardisation loop and kept in 48 VGPRs across both factorisations (which is exactly what re-reading is meant to avoid).
  DEV const double* fresh() const {
    const double* p = Acol;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(p));
#endif
    return p;
  }
  DEV double a(const double* p, int i) const { return (on && i < m) ? p[i * MAXR] : 0.0; }
};

// A x for this lane's row, x one entry per lane; vec: MAXR doubles of LDS scratch
template <class W>
DEV double coopAx(const W& w, double* vec, const CoopRow& R, double xLane) {
  const int ln = w.lane();
  if (ln < MAXR) vec[ln] = R.on ? xLane : 0.0;
  w.sync();
  double v0 = 0.0, v1 = 0.0;
  const double* Ac = R.fresh();
#pragma unroll 1
  for (int jb = 0; jb < MAXR; jb += 8) {             // eight loads of the column in flight per block (a real loop: see coopPinvApply)
#pragma unroll
    for (int jq = 0; jq < 8; jq += 2) { const int jx = jb + jq; v0 = fma(R.a(Ac, jx), vec[jx], v0); v1 = fma(R.a(Ac, jx + 1), vec[jx + 1], v1); }
  }
  return v0 + v1;
}

// LCPUtils::isLCPSolutionValid (LCPUtils.cpp:12-80), uniform result
template <class W>
DEV bool coopValid(const W& w, double* vec, const CoopRow& R, double X, bool ignoreFriction, double cfm) {
  const double tol = 1e-5;
  const int ln = w.lane();
  const double v = -R.Bv + cfm * X + coopAx(w, vec, R, X);
  const double Xn = w.shfl(X, R.fp);
  bool ok = true;
  if (R.on) {
    double upper = R.fric ? R.mu : INFINITY, lower = R.fric ? -R.mu : 0.0;
    bool skip = false;
    if (R.fric) {
      if (ignoreFriction) { ok = (X == 0.0); skip = true; }
      upper *= Xn; lower *= Xn;
    }
    if (!skip) {
      if (fabs(lower) < tol && fabs(upper) < tol && fabs(X) < tol) {}
      else if (fabs(X - lower) < tol) { if (v < -tol) ok = false; }
      else if (fabs(X - upper) < tol) { if (v > tol) ok = false; }
      else if (X > lower && X < upper) { if (fabs(v) > tol) ok = false; }
      else ok = false;
    }
  }
  return w.ballot(!ok) == 0ull;
}
template <class W>
DEV bool coopValid(const W& w, CoopLds& S, const CoopRow& R, double X, bool ignoreFriction, double cfm, int slot) {
  return coopValid(w, S.vec[slot], R, X, ignoreFriction, cfm);
}

struct CoopClasses {
  int cls;            // this lane's row
  double E;
  RowMask clampMask, ubMask;   // uniform
  int nc, nu;
#ifdef NBL_CASCADE_TIMING
  int dbgRank = -1;   // developer stamp: rank of the last Q factorised in the standardisation loop
#endif
};

// CGGM::constructMatrices classification (CGGM.cpp:535-713), lane = row
template <class W>
DEV void coopClassify(const W& w, const CoopRow& R, double X, bool ignoreFriction, CoopClasses& K) {
  const double TH = 1e-6, tie = 1e-5;
  const int ln = w.lane();
  const double Xn = w.shfl(X, R.fp);
  const double cnN = w.shfl(R.colNorm, R.fp);
  const double hi = R.fric ? R.mu : INFINITY, lo = R.fric ? -R.mu : 0.0;
  double upper = hi, lower = lo;
  if (R.fric) { upper *= Xn; lower *= Xn; }
  int cls = RC_NOT_CLAMPING;
  bool inElse = false;
  if (R.on && !(R.colNorm < 1e-9)) {
    if (fabs(X) < TH) {
      if (R.fric && !(fabs(Xn) < TH) && !ignoreFriction) cls = RC_CLAMPING;
    } else if ((X > lower + tie && X < upper - tie) || (lower - X > 1e-2 || X - upper > 1e-2)) {
      cls = RC_CLAMPING;
    } else {
      inElse = true;
    }
  }
  const uint64_t clampBits = w.ballot(cls == RC_CLAMPING);
  double E = 0.0;
  if (inElse && R.fric && fabs(Xn) > 1e-9 && cnN > 1e-9 && ((clampBits >> R.fp) & 1ull)) {
    cls = RC_UPPER_BOUND;
    const double ub = Xn * hi, lb = Xn * lo;
    E = (fabs(X - ub) < fabs(X - lb)) ? hi : lo;
  }
  K.cls = cls; K.E = E;
  K.clampMask = (RowMask)clampBits;
  K.ubMask = (RowMask)w.ballot(cls == RC_UPPER_BOUND);
  K.nc = rmPop(K.clampMask);
  K.nu = rmPop(K.ubMask);
}

// lane s (< 24): a[i] = Q[i][s] = A[i][s] + [s normal] sum_{u = s+1, s+2 upper-bound} E[u] A[i][u] + cfm [i == s]
// for clamping i and s, zero elsewhere (buildQ of lcp_dev.hpp with masks instead of compaction)
template <class W>
DEV void coopBuildQ(const W& w, CoopLds& S, const CoopRow& R, const CoopClasses& K, double cfm, double (&a)[MAXR]) {
  const int ln = w.lane();
  const bool colOn = ln < MAXR && K.cls == RC_CLAMPING;
  const double* Ac = R.fresh();
  double e1 = 0.0, e2 = 0.0;
  if (K.nu > 0) {
    // stage A in LDS so that a normal column can add its contact's upper-bound friction columns
#pragma unroll
    for (int i = 0; i < MAXR; i++) if (ln < MAXR) S.R[i * CLD + ln] = R.a(Ac, i);
    const double E1 = w.shfl(K.E, ln + 1), E2 = w.shfl(K.E, ln + 2);
    if (!R.fric && ln + 2 < MAXR) {
      if ((K.ubMask >> (ln + 1)) & 1u) e1 = E1;
      if ((K.ubMask >> (ln + 2)) & 1u) e2 = E2;
    }
    w.sync();
  }
  const int c1 = (ln + 1 < MAXR) ? ln + 1 : 0, c2 = (ln + 2 < MAXR) ? ln + 2 : 0;
#pragma unroll
  for (int i = 0; i < MAXR; i++) {
    double q = R.a(Ac, i);
    if (K.nu > 0) q = fma(e2, S.R[i * CLD + c2], fma(e1, S.R[i * CLD + c1], q));
    // The reference forms Q = A_c^T M^-1 (A_c + A_ub E) from constraint-force columns when a friction row sits on its bound, and a
    // joint-limit constraint has none (a DifferentiableContactConstraint without a contact: zero world force, DCC.cpp:51-99): its row
    // and column of Q are zero.  (Without upper-bound rows Q is the clamping block of A itself, CGGM.cpp:256-266, couplings included.)
    if (K.nu > 0 && (R.lim || ((R.limMask >> i) & 1u))) q = 0.0;
    if (i == ln) q += cfm;
    a[i] = (colOn && ((K.clampMask >> i) & 1u)) ? q : 0.0;
  }
  if (K.nu > 0) w.sync();   // reads of the staged A complete before the factorisation reuses the buffer
}

// Q^+ of the Q that coopBuildQ left in a[]: without upper-bound rows Q = A(clamping, clamping) + cfm I is symmetric positive
// semi-definite (the Cholesky route); a friction row on its bound folds its column into its normal's and Q is a general matrix (the
// Householder route).
template <class W>
DEV int coopPinvOfQ(const W& w, double (&a)[MAXR], CoopLds& S, const CoopClasses& K) {
  if (K.nu == 0) return coopPinvSym(w, a, S, K.nc);
  return coopPinv(w, a, S, K.nc);
}

struct CoopStage0 {
  double X, X0;       // solution / pre-solve x of this lane's row
  CoopClasses K;
  bool ok;            // standardised valid solution found (uniform)
  bool pinvValid;     // S.P is the pseudo-inverse of the Q of the final classification (uniform)
#ifdef NBL_CASCADE_TIMING
  long long tGuess; int nu, fast;   // developer stamps (tools/cascade_timing.py)
#endif
};

// CGGM::constructMatrices + opportunisticallyStandardizeResults as a loop (standardizeLoop of lcp_dev.hpp, lane = row).
// X in: the solver's x, out: the last accepted solution.  guessMask / pinvValid: the rows and state of a pseudo-inverse
// already in S.P (stage 0's guess), 0 / false otherwise.  Returns whether the results are standardised.
template <class W>
DEV bool coopStandardizeLoop(const W& w, CoopLds& S, const CoopRow& R, double& X, double cfm, bool ignoreFriction,
                             RowMask guessMask, bool& pinvValid, CoopClasses& K) {
  double a[MAXR];
  bool ok = false;
#pragma unroll 1
  for (int iter = 0; iter < MAXR + 1; iter++) {
    coopClassify(w, R, X, ignoreFriction, K);
    if (K.nc == 0) {
      pinvValid = false;
      ok = coopValid(w, S, R, 0.0, ignoreFriction, cfm, 1);
      if (ok) X = 0.0;
      break;
    }
    double fc;
    if (iter == 0 && K.nu == 0 && guessMask != 0 && K.clampMask == guessMask) fc = X;
    else {
      coopBuildQ(w, S, R, K, cfm, a);
#ifdef NBL_CASCADE_TIMING
      K.dbgRank = coopPinvOfQ(w, a, S, K);
#else
      coopPinvOfQ(w, a, S, K);
#endif
      fc = coopPinvApply<W, false>(w, S, K.cls == RC_CLAMPING ? R.Bv : 0.0, 0);
      pinvValid = true;
    }
    double newX = 0.0;
    bool newlyNot = false;
    const double fcN = w.shfl(fc, R.fp), Xn = w.shfl(X, R.fp);
    if (K.cls == RC_CLAMPING) {
      newX = fc;
      if (fabs(newX) < 1e-6 && fabs(X) > 1e-6 && !R.fric) newlyNot = true;
    } else if (K.cls == RC_UPPER_BOUND) {
      const double om = Xn / X;
      const double clean = (fabs(om - R.mu) < fabs(om + R.mu)) ? R.mu : -R.mu;
      newX = fcN * clean;
    }
    if (!coopValid(w, S, R, newX, ignoreFriction, cfm, 1)) { ok = false; break; }
    X = newX;
    ok = true;
    if (w.ballot(newlyNot) == 0ull) break;
    pinvValid = false;   // X moved on: K and S.P belong to the previous iterate until the next pass refactorises (matters when the loop runs out)
  }
  return ok;
}

// LCPUtils::guessSolution (when there is no matching warm start) + the standardisation loop: stage 0 of the solver cascade
// (BoxedLcpConstraintSolver.cpp:380-460), cfm = 0, friction kept.
template <class W>
DEV void coopStage0(const W& w, CoopLds& S, const CoopRow& R, bool haveCache, double Xcache, CoopStage0& out) {
  const int ln = w.lane();
  double X = 0.0;
  RowMask guessMask = 0;
  bool pinvValid = false;
  if (haveCache) X = R.on ? Xcache : 0.0;
  else {
    // (the empty tangent rows of frictionless contacts are not rows of the reference's problem; a negated joint-limit row: the reference
    // tests ITS b > 0)
    const bool in = R.on && (R.fric ? R.mu != 0.0 : (R.neg ? R.Bv < 0 : R.Bv > 0));
    guessMask = (RowMask)w.ballot(in);
    if (guessMask != 0) {
      double a[MAXR];
      const double* Ac = R.fresh();
#pragma unroll
      for (int i = 0; i < MAXR; i++) a[i] = (in && ((guessMask >> i) & 1u)) ? R.a(Ac, i) : 0.0;
      NBL_PHASE(43);
      coopPinvSym(w, a, S, rmPop(guessMask));      // A restricted to the guess rows: symmetric positive semi-definite
      NBL_PHASE(44);
      X = coopPinvApply<W, false>(w, S, in ? R.Bv : 0.0, 0);
      NBL_PHASE(45);
      if (!in) X = 0.0;
      pinvValid = true;   // of A restricted to guessMask; stays valid only if the first classification agrees
    }
  }
  out.X0 = X;
#if defined(NBL_CASCADE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  out.tGuess = clock64();
#endif
  CoopClasses K;
  const bool ok = coopStandardizeLoop(w, S, R, X, 0.0, false, guessMask, pinvValid, K);
#if defined(NBL_CASCADE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  out.nu = K.nu + 100 * K.nc + 10000 * (K.dbgRank > 0 ? K.dbgRank : 0); out.fast = (K.nu == 0 && guessMask != 0 && K.clampMask == guessMask) ? 1 : (K.dbgRank >= 0 && K.dbgRank < K.nc ? 2 : 0);
#endif
  NBL_PHASE(46);
  out.X = X; out.K = K; out.ok = ok; out.pinvValid = ok && pinvValid;
}

}  // namespace NBL_NS
