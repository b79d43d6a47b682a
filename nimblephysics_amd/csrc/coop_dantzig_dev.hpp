// coop_dantzig_dev.hpp — stages 1-3 of the LCP solver cascade with ONE WORLD PER WAVEFRONT (lane = LCP row).
//
// k_contact_cascade runs these stages one world per lane; its single dependent instruction stream per world (a fresh
// LDL^T of A(C,C) per pivot, index vectors in scratch) costs 4-7 ms as soon as ONE world of a batch leaves stage 0.
// Here the wavefront of a failed world shares the work: lane k owns position k of the Dantzig driver's permuted problem
// (x, w, b, lo, hi, dx, dw, state, findex, p in registers), the permuted symmetric matrix and the LDL^T factor live in
// LDS, factorisation / triangular solves / products are lane-parallel, step-length events are found by a wave arg-min
// with the reference's scan order as the tie-break.  Same mathematics and event order as dantzig_dev.hpp (which restates
// dart/external/odelcpsolver/lcp.cpp:780-1113 and is pinned against the reference's own dSolveLCP on the host).
// Attribution: the algorithm restated here derives from the Open Dynamics Engine (ODE), Copyright (C) 2001-2003 Russell L. Smith, which the
// reference vendors under ODE's BSD-style licence (dart/external/odelcpsolver/, dart/collision/dart/DARTCollide.cpp); this file is an
// independent restatement for another execution model - ODE's arithmetic order and, where the bit-for-bit tests need them recognisable,
// its identifiers are kept on purpose.
#pragma once
#include "coop_dev.hpp"

namespace NBL_NS {

template <int I> struct IntTag { static constexpr int value = I; };

// Values the optimiser must treat as produced HERE (in VGPRs on the device): keeps a block of LDS loads together and ahead of the
// dependent chain that consumes them instead of being sunk, one by one, into the predicated code that uses them.  One statement per
// eight values: a statement waits for its own operands only, so the 24 loads of a row stay in flight together.
DEV void coopPin24(double (&a)[MAXR]) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(MAXR % 8 == 0, "groups of eight");
#pragma unroll
  for (int g = 0; g < MAXR; g += 8)
    asm volatile("" : "+v"(a[g]), "+v"(a[g + 1]), "+v"(a[g + 2]), "+v"(a[g + 3]), "+v"(a[g + 4]), "+v"(a[g + 5]), "+v"(a[g + 6]), "+v"(a[g + 7]));
#endif
}

// Developer instrumentation of the Dantzig driver (tools/cascade_timing.py builds with -DNBL_CASCADE_TIMING): cycles per phase
// summed over all worlds.  Compiled out of the shipped library.
#if defined(NBL_CASCADE_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_dzStat[16];
__device__ unsigned long long g_dzStatSlow[16];   // the same sums over the SLOW solves only (more than NBL_DZ_SLOW cycles: the tail that a launch waits for)
#ifndef NBL_DZ_SLOW
#define NBL_DZ_SLOW 300000
#endif
#endif
#if defined(NBL_CASCADE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define DZ_T0() long long dzT = clock64(); long long dzAcc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DZ_ADD(k) do { const long long dzN = clock64(); dzAcc[k] += dzN - dzT; dzT = dzN; } while (0)
#define DZ_CNT(k) do { dzAcc[k] += 1; } while (0)
#define DZ_FLUSH() do { if (ln == 0) { long long dzTot = 0; for (int dzK = 0; dzK < 8; dzK++) dzTot += dzAcc[dzK];                      \
                                        for (int dzK = 0; dzK < 12; dzK++) atomicAdd(&g_dzStat[dzK], (unsigned long long)dzAcc[dzK]);           \
                                        if (dzTot > NBL_DZ_SLOW) for (int dzK = 0; dzK < 12; dzK++) atomicAdd(&g_dzStatSlow[dzK], (unsigned long long)dzAcc[dzK]); } } while (0)
#else
#define DZ_T0() do { } while (0)
#define DZ_ADD(k) do { } while (0)
#define DZ_CNT(k) do { } while (0)
#define DZ_FLUSH() do { } while (0)
#endif

// PGS / reduce only need the matrix and the broadcast vectors
struct PgsLds {
  double A[MAXR * CLD];
  double v[4][MAXR];
};

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

// ---- the two triangular solves of the Dantzig driver, one substitution step as hand-placed EXEC masks (device, factors of up to 32 rows) ----
// A step of dSolveL1 / dSolveL1T for ALL rows of the factor at once (lane = row): x_k, final on lane k, is broadcast; t = L[row][k] x_k; the
// rows of k's own 4-block that come after it subtract t from their running y, every later row adds t to its block sum Z, and a row
// switches from Z to y = rhs - Z when the step reaches the first column of its block (fastlsolve.cpp / fastltsolve.cpp: that order of
// the additions is what makes the result bit-identical to the reference's).  ONE accumulator per lane holds Z before the switch and y
// after it; which lanes do what at step k is a compile-time lane mask once the block grid is fixed - for dSolveL1 the grid starts at
// row 0, for dSolveL1T (reversed index) at row nC mod 4 - so a step is broadcast + multiply + two predicated adds under literal EXEC
// masks.  The host emulation and the 48-row build keep the compare / select form below.
//
// Round 6 (tools/dbg/lat_probe.hip on MI355X): a wavefront that is alone on its SIMD issues ONE instruction every ~5.5 cycles, whatever
// the instruction (s_mov, s_nop, v_readlane, v_add_f64 alike; a dependent v_add_f64 10) - what a step costs is its instruction COUNT: 9
// (pad, two v_readlane, v_mul, mask, v_add, mask, v_add, EXEC back to `all`, the mask the solve was entered with; round 5 also saved
// EXEC per step and cut the second mask to the rows of the factor - the accumulator of a lane beyond them is never read).  Measured at
// NO gain over this, each bit-identical on the device: the entries of a lane's own block negated so that ONE addition serves both kinds
// of row and EXEC is written once per step (the 24 sign flips per solve cost what the second addition and its mask cost); EXEC left
// narrowed between the statements of a block (the multiplication of the next step needs the lanes of both additions: a third mask).
// Every statement ends with EXEC = all, so the compiler's own instructions between two statements always see the full mask.
#if defined(__HIP_DEVICE_COMPILE__)
constexpr unsigned dzLanes(int from, int to) { return from >= to ? 0u : (unsigned)(((1ull << to) - 1ull) & ~((1ull << from) - 1ull)); }   // lanes [from, to)
template <unsigned SW>
DEV void dzSwitchStep(double& acc, double rhs, unsigned long long all) {          // acc = rhs - acc on the lanes SW
  asm volatile("s_mov_b64 exec, %[sw]\n\t"
               "v_add_f64 %[acc], %[rhs], -%[acc]\n\t"
               "s_mov_b64 exec, %[all]"
               : [acc] "+v"(acc)
               : [rhs] "v"(rhs), [sw] "n"(SW), [all] "s"(all));
}
template <unsigned YM, unsigned ZM>
DEV void dzSubstStep(double& acc, double l, double xk, unsigned long long all) {
  double t;
  if constexpr (YM != 0u && ZM != 0u) {
    asm volatile("v_mul_f64 %[t], %[l], %[xk]\n\t"
                 "s_mov_b64 exec, %[ym]\n\t"
                 "v_add_f64 %[acc], %[acc], -%[t]\n\t"
                 "s_mov_b64 exec, %[zm]\n\t"
                 "v_add_f64 %[acc], %[acc], %[t]\n\t"
                 "s_mov_b64 exec, %[all]"
                 : [acc] "+v"(acc), [t] "=&v"(t)
                 : [l] "v"(l), [xk] "s"(xk), [ym] "n"(YM), [zm] "n"(ZM), [all] "s"(all));
  } else if constexpr (YM != 0u) {
    asm volatile("v_mul_f64 %[t], %[l], %[xk]\n\t"
                 "s_mov_b64 exec, %[ym]\n\t"
                 "v_add_f64 %[acc], %[acc], -%[t]\n\t"
                 "s_mov_b64 exec, %[all]"
                 : [acc] "+v"(acc), [t] "=&v"(t)
                 : [l] "v"(l), [xk] "s"(xk), [ym] "n"(YM), [all] "s"(all));
  } else if constexpr (ZM != 0u) {
    asm volatile("v_mul_f64 %[t], %[l], %[xk]\n\t"
                 "s_mov_b64 exec, %[zm]\n\t"
                 "v_add_f64 %[acc], %[acc], %[t]\n\t"
                 "s_mov_b64 exec, %[all]"
                 : [acc] "+v"(acc), [t] "=&v"(t)
                 : [l] "v"(l), [xk] "s"(xk), [zm] "n"(ZM), [all] "s"(all));
  }
}
// dSolveL1: the complete 4-blocks (rows below nb4 = nC & ~3), then at most three single rows.  The guards nest - one test per block, and
// the first block that is not complete ends the walk with the single rows.
template <int K, class W>
DEV void dzL1BlockStep(const W& w, double& acc, double rhs, const double (&lrow)[MAXR], unsigned long long all) {
  if constexpr (K % 4 == 0) dzSwitchStep<dzLanes(K, K + 4)>(acc, rhs, all);
  const double xk = w.bcast(acc, K);
  dzSubstStep<dzLanes(K + 1, (K | 3) + 1), dzLanes((K | 3) + 1, 32)>(acc, lrow[K], xk, all);
}
template <int K, class W>
DEV void dzL1TailStep(const W& w, double& acc, double rhs, const double (&lrow)[MAXR], unsigned long long all) {
  dzSwitchStep<dzLanes(K, K + 1)>(acc, rhs, all);
  const double xk = w.bcast(acc, K);
  dzSubstStep<0u, dzLanes(K + 1, 32)>(acc, lrow[K], xk, all);
}
template <int J, class W>
DEV void dzL1Blocks(const W& w, double& acc, double rhs, const double (&lrow)[MAXR], int nC, int nb4, unsigned long long all) {
  if constexpr (4 * J < MAXR) {
    if (4 * J < nb4) {
      dzL1BlockStep<4 * J>(w, acc, rhs, lrow, all);
      dzL1BlockStep<4 * J + 1>(w, acc, rhs, lrow, all);
      dzL1BlockStep<4 * J + 2>(w, acc, rhs, lrow, all);
      dzL1BlockStep<4 * J + 3>(w, acc, rhs, lrow, all);
      dzL1Blocks<J + 1>(w, acc, rhs, lrow, nC, nb4, all);
    } else if (4 * J < nC) {
      dzL1TailStep<4 * J>(w, acc, rhs, lrow, all);
      if (4 * J + 1 < nC) {
        dzL1TailStep<4 * J + 1>(w, acc, rhs, lrow, all);
        if (4 * J + 2 < nC) dzL1TailStep<4 * J + 2>(w, acc, rhs, lrow, all);
      }
    }
  }
}
// dSolveL1T (K runs down), the block grid of the REVERSED index starting at lane RHO = nC mod 4: lanes [RHO + 4j, RHO + 4j + 4) are one
// block whose first reversed row is its LAST lane; the lanes below RHO are the single rows and come last.
template <int RHO, int K, class W>
DEV void dzL1TBlockStep(const W& w, double& acc, double rhs, const double (&lcol)[MAXR], unsigned long long all) {
  constexpr int B0 = RHO + 4 * ((K - RHO) / 4);                    // first lane of K's block
  if constexpr (K == B0 + 3) dzSwitchStep<dzLanes(B0, B0 + 4)>(acc, rhs, all);
  const double xk = w.bcast(acc, K);
  dzSubstStep<dzLanes(B0, K), dzLanes(0, B0)>(acc, lcol[K], xk, all);
}
template <int K, class W>
DEV void dzL1TTailStep(const W& w, double& acc, double rhs, const double (&lcol)[MAXR], unsigned long long all) {
  dzSwitchStep<dzLanes(K, K + 1)>(acc, rhs, all);
  const double xk = w.bcast(acc, K);
  dzSubstStep<0u, dzLanes(0, K)>(acc, lcol[K], xk, all);
}
// blocks from the top one down: block J covers lanes RHO + 4J .. RHO + 4J + 3 and exists when its last lane is below nC
template <int RHO, int J, class W>
DEV void dzL1TBlocks(const W& w, double& acc, double rhs, const double (&lcol)[MAXR], int nC, unsigned long long all) {
  if constexpr (J >= 0) {
    if constexpr (RHO + 4 * J + 3 < MAXR) {
      if (RHO + 4 * J + 3 < nC) {
        dzL1TBlockStep<RHO, RHO + 4 * J + 3>(w, acc, rhs, lcol, all);
        dzL1TBlockStep<RHO, RHO + 4 * J + 2>(w, acc, rhs, lcol, all);
        dzL1TBlockStep<RHO, RHO + 4 * J + 1>(w, acc, rhs, lcol, all);
        dzL1TBlockStep<RHO, RHO + 4 * J>(w, acc, rhs, lcol, all);
      }
    }
    dzL1TBlocks<RHO, J - 1>(w, acc, rhs, lcol, nC, all);
  } else {
    if constexpr (RHO >= 3) dzL1TTailStep<2>(w, acc, rhs, lcol, all);
    if constexpr (RHO >= 2) dzL1TTailStep<1>(w, acc, rhs, lcol, all);
    if constexpr (RHO >= 1) dzL1TTailStep<0>(w, acc, rhs, lcol, all);
  }
}
#endif

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
  DZ_T0();
  DZ_CNT(8);
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
    // the two lanes trade their values: every lane reads from its source lane (itself unless it is i1 / i2)
    const int src = ln == i1 ? i2 : (ln == i2 ? i1 : ln);
    x = w.shfl(x, src); b = w.shfl(b, src); ww = w.shfl(ww, src); lo = w.shfl(lo, src); hi = w.shfl(hi, src);
    p = w.shflI(p, src); st = w.shflI(st, src); fidx = w.shflI(fidx, src);
  };
  // Contact problems have no unbounded rows (nub = 0); every findex row goes to the end by the reference's sequence of swaps
  // (lcp.cpp:487-498).  The swaps are played on the index vectors only (p[j] = the row that ends at position j); the matrix is
  // symmetrised from its lower triangle (lcp.cpp:138-140) and permuted in ONE pass, the row data follows with one shuffle each.
  {
    // which position holds a friction row when the scan reaches it is known up front: a swap touches position k only at step k
    const uint64_t fricBits = w.ballot(on && fidx >= 0);
    int atEnd = 0;
    for (int k = n - 1; k >= 0; k--) {
      if ((fricBits >> k) & 1ull) {
        const int i2 = n - 1 - atEnd;
        if (k != i2) {       // lanes k and i2 trade p and findex: both lanes are wave-uniform, so four v_readlane and selects, no LDS round trip
          const int pk = w.bcastI(p, k), pi = w.bcastI(p, i2), fk = w.bcastI(fidx, k), f2 = w.bcastI(fidx, i2);
          const bool isK = ln == k, isI2 = ln == i2;
          p = isK ? pi : (isI2 ? pk : p);
          fidx = isK ? f2 : (isI2 ? fk : fidx);
        }
        atEnd++;
      }
    }
    double col[MAXR];    // column ln of the permuted matrix: A'[r][ln] = Asym[p_r][p_ln]
#pragma unroll
    for (int r = 0; r < MAXR; r++) {
      const int pr = w.bcastI(p, r < n ? r : 0);
      const int hiI = pr >= p ? pr : p, loI = pr >= p ? p : pr;          // ONE load through a selected address (a select between two
      const double av = C.A[hiI * CLD + loI];                            // loads compiles to two divergent branches with an LDS wait each)
      col[r] = (on && r < n) ? av : 0.0;
    }
    w.sync();
    // (one EXEC region for the 24 stores - a guard per row compiled to a branch per store; rows and columns beyond n get zeros, nobody reads them)
    if (ln < MAXR) {
#pragma unroll
      for (int r = 0; r < MAXR; r++) C.A[r * CLD + ln] = col[r];
    }
    b = w.shfl(b, p); lo = w.shfl(lo, p); hi = w.shfl(hi, p);
    w.sync();
  }
  // dDot over lanes [from, to): the running sum from +0.0 in lane order (fastdot.cpp); every lane gets the result.  The term of lane k
  // comes through v_readlane (k is wave-uniform: an SGPR lane select) straight into the scalar operand of the addition - no LDS round
  // trip (round 5: the products went through two LDS vectors, masked to their ranges, and every lane read all of them back: two waits of
  // an LDS latency per sum and 2 x 24 additions; 1330 cycles per w[i]).  Lanes outside [from, to) are never read, which is what adding
  // their +0.0 was: a running sum that starts at +0.0 is never -0.0, so leaving a +0.0 term out changes no bit.  Rolled loops, four terms
  // per trip: the eight readlanes of a trip do not depend on the sum and are issued ahead of the chain of dependent additions.
  auto seqRange = [&](double prod, int from, int to) -> double {
    double s = 0.0;
    int k = from;
#pragma unroll 1
    for (; k + 4 <= to; k += 4) {
      const double t0 = w.bcast(prod, k), t1 = w.bcast(prod, k + 1), t2 = w.bcast(prod, k + 2), t3 = w.bcast(prod, k + 3);
      s = s + t0; s = s + t1; s = s + t2; s = s + t3;
    }
#pragma unroll 1
    for (; k < to; k++) s = s + w.bcast(prod, k);
    return s;
  };
  // seqSum2: s1 = sum over [0, mid), s2 = sum over [mid, to).
#ifdef NBL_SEQSUM_LDS      // A/B switch (developer): round 5's form - the products through two LDS vectors, two interleaved chains of additions
  auto seqSum2 = [&](double prod, int mid, int to, double& s1, double& s2) {
    w.sync();
    if (ln < MAXR) { C.v[2][ln] = ln < mid ? prod : 0.0; C.v[3][ln] = (ln >= mid && ln < to) ? prod : 0.0; }
    w.sync();
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int k4 = 0; k4 < MAXR; k4 += 4) {
      if (k4 < to) {
#pragma unroll
        for (int k = k4; k < k4 + 4; k++) { s1 = s1 + C.v[2][k]; s2 = s2 + C.v[3][k]; }
      }
    }
  };
#else
  auto seqSum2 = [&](double prod, int mid, int to, double& s1, double& s2) {
    s1 = seqRange(prod, 0, mid);
    s2 = seqRange(prod, mid, to);
  };
#endif
#ifdef NBL_SEQSUM_LDS
  auto seqSum = [&](double prod, int from, int to) -> double { double s1, s2; seqSum2(prod, to, to, s1, s2); return s1; };
#else
  auto seqSum = [&](double prod, int from, int to) -> double { return seqRange(prod, from, to); };
#endif
  // dSolveL1 (fastlsolve.cpp): L y = rhs over the factor rows, lane = row.  Step k: y_k (final on lane k) is broadcast, every lane forms
  // its product with column k of its row UNCONDITIONALLY and only the one-instruction updates are predicated - written any other way
  // the compiler sinks the LDS loads of the factor into the divergent branches (an LDS round trip per step: 250 cycles instead of 40).
  auto solveL1 = [&](double rhs) -> double {
    const bool act = ln < nC;
    const int nb4 = nC & ~3;
    const int i0 = (ln < nb4) ? (ln & ~3) : ln;     // columns < i0 go into Z, the rest is subtracted term by term
    double lrow[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; k++) lrow[k] = C.L[me * CLD + k];
    coopPin24(lrow);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MAXR <= 32) {
      double acc = 0.0;
      const unsigned long long all = __builtin_amdgcn_read_exec();
      dzL1Blocks<0>(w, acc, rhs, lrow, nC, nb4, all);
      return ln < nC ? acc : rhs;
    }
#endif
    double Z = 0.0, y = rhs;
#pragma unroll
    for (int k = 0; k < MAXR; k++) {
      if (k < nC) {          // a guard, not a break: the loop must stay fully unrolled (lrow[] in registers, not in scratch)
        if (k == i0) y = rhs - Z;
        const double xk = w.bcast(y, k);
        const double t = lrow[k] * xk;
        if (act && k < i0) Z = Z + t;               // (k < i0 implies k < ln)
        if (k >= i0 && k < ln) y = y - t;           // (ln < nC for every lane that reaches here with k < nC ... and k < ln)
      }
    }
    return y;
  };
  // dSolveL1T (fastltsolve.cpp): L^T y = rhs, the same blocking on the reversed index
  auto solveL1T = [&](double rhs) -> double {
    const bool act = ln < nC;
    const int jr = nC - 1 - ln;
    const int nb4 = nC & ~3;
    const int i0 = (jr < nb4) ? (jr & ~3) : jr;
    double lcol[MAXR];   // this lane's column of L, fetched up front so that the substitution chain does not wait on LDS
#pragma unroll
    for (int k = 0; k < MAXR; k++) lcol[k] = C.L[k * CLD + me];
    coopPin24(lcol);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MAXR <= 32) {
      double acc = 0.0;
      const unsigned long long all = __builtin_amdgcn_read_exec();
      const int rho = nC & 3;
      constexpr int JTOP = MAXR / 4 - 1;
      if (rho == 0) dzL1TBlocks<0, JTOP>(w, acc, rhs, lcol, nC, all);
      else if (rho == 1) dzL1TBlocks<1, JTOP>(w, acc, rhs, lcol, nC, all);
      else if (rho == 2) dzL1TBlocks<2, JTOP>(w, acc, rhs, lcol, nC, all);
      else dzL1TBlocks<3, JTOP>(w, acc, rhs, lcol, nC, all);
      return ln < nC ? acc : rhs;
    }
#endif
    double Z = 0.0, y = rhs;
#pragma unroll
    for (int k = MAXR - 1; k >= 0; k--) {
      if (k < nC) {
        const int kr = nC - 1 - k;
        if (kr == i0) y = rhs - Z;
        const double xk = w.bcast(y, k);
        const double t = lcol[k] * xk;
        if (act && kr < i0) Z = Z + t;
        if (act && kr >= i0 && kr < jr) y = y - t;
      }
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
#ifdef NBL_DZ_COUNTS
  int dzIters = 0, dzRemovals = 0, dzTransfers = 0;
#endif
  DZ_ADD(0);   // setup
  for (int i = 0; i < n; ++i) {
    DZ_CNT(9);
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
      double sC, sN;
      seqSum2(pr, nC, nC + nN, sC, sN);
      const double s = sC + sN - w.bcast(b, i);
      if (ln == i) ww = s;
    }
    DZ_ADD(1);   // w[i]
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
        DZ_CNT(10);
#ifdef NBL_DZ_COUNTS
        dzIters++;
#endif
        solve1(i, dir);
        DZ_ADD(2);   // solve1
        // dw(N) = A(N,C) dx(C) +/- A(i,N);  dw[i] = A(i,C) dx(C) + A(i,i) dirf   (lcp.cpp:926-928)
        // The running sum of a row takes its terms in blocks of four behind ONE guard (a guard per term was a compare, a branch and a
        // wait per term: 10 instructions for one multiplication and one addition).  dx is taken as +0.0 beyond C, so the terms of the
        // last block that lie beyond C are +-0.0, and a running sum that starts at +0.0 is never -0.0: adding them changes no bit.
        {
          double arow[MAXR];
#pragma unroll
          for (int j = 0; j < MAXR; j++) arow[j] = C.A[me * CLD + j];
          const double ai = C.A[me * CLD + i];
          coopPin24(arow);
          const double dxz = ln < nC ? dx : 0.0;
          double s = 0.0;
#pragma unroll
          for (int j4 = 0; j4 < MAXR; j4 += 4) {
            if (j4 < nC) {
              const double x0 = w.bcast(dxz, j4), x1 = w.bcast(dxz, j4 + 1), x2 = w.bcast(dxz, j4 + 2), x3 = w.bcast(dxz, j4 + 3);
              const double p0 = arow[j4] * x0, p1 = arow[j4 + 1] * x1, p2 = arow[j4 + 2] * x2, p3 = arow[j4 + 3] * x3;
              s = s + p0; s = s + p1; s = s + p2; s = s + p3;
            }
          }
          const bool inN = ln >= nC && ln < nC + nN;
          if (inN) dw = dir > 0 ? s + ai : s - ai;
          if (ln == i) dw = s + ai * dirf;
        }
        // step length: first minimum in the reference's scan order (i's own events, N rows, C rows)
        DZ_ADD(3);   // dw
        // every lane's candidate is one quotient (lcp.cpp:938-998): -w / dw for the driving row and the N rows, (lo - x) / dx or
        // (hi - x) / dx for the C rows - formed with ONE division for the whole wave (the same operands, so the same bits).  (Which
        // event a lane stands for stays a nest of ifs: written as a chain of selects it measured 170 cycles per iteration SLOWER - the
        // EXEC regions skip what the selects compute for every lane.)
        const bool inN = ln >= nC && ln < nC + nN, inC = ln < nC, isI = ln == i;
        const bool dn = dx < 0;
        const double num = (isI || inN) ? -ww : ((dn ? lo : hi) - x);
        const double den = (isI || inN) ? dw : dx;
        const double q = num / den;
        double s = INFINITY;
        int cmd = 0;
        if (isI) {
          s = q; cmd = 1;
          if (dir > 0) { if (hi < INFINITY) { const double s2 = (hi - x) * dirf; if (s2 < s) { s = s2; cmd = 3; } } }
          else { if (lo > -INFINITY) { const double s2 = (lo - x) * dirf; if (s2 < s) { s = s2; cmd = 2; } } }
        } else if (inN) {
          if (((st == 0) ? dw < 0 : dw > 0) && !(lo == 0 && hi == 0)) { s = q; cmd = 4; }
        } else if (inC) {
          if (dn && lo > -INFINITY) { s = q; cmd = 5; }
          if (dx > 0 && hi < INFINITY) { s = q; cmd = 6; }
        }
        // the reference keeps a candidate only if it is STRICTLY smaller than the running minimum, which starts at lane
        // i's value: arg-min over (s, scan position)
        const double sOwn = w.bcast(s, i);
        if (sOwn != sOwn) { row.x = 0.0; DZ_FLUSH(); return -1; }
        const double sMin = w.template minRows<MAXR>(cmd != 0 ? s : INFINITY);
        const uint64_t tie = w.ballot(cmd != 0 && s == sMin);
        if (tie == 0ull) { row.x = 0.0; DZ_FLUSH(); return -1; }
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
        if (sMin <= 0.0) { row.x = 0.0; DZ_FLUSH(); return 0; }   // earlyTermination (the caller always has the PGS fallback, BoxedLcpConstraintSolver.cpp:463)
        const int si = best;
        DZ_ADD(4);   // step selection
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
        if (cmdB >= 5) DZ_CNT(11);
#ifdef NBL_DZ_COUNTS
        if (cmdB >= 5) dzRemovals++;
        if (cmdB == 4) dzTransfers++;
#endif
        if (cmdB >= 5) DZ_ADD(7);   // apply + a row leaving C (dLDLTRemove)
        else DZ_ADD(5);             // apply + a row entering C / N
        if (cmdB <= 3) break;
      }
    }
    DZ_ADD(6);
  }
  // back to the original (reduced) order: P.x[p[j]] = x[j]
  w.sync();
  if (on) C.v[0][p] = x;
  w.sync();
  row.x = on ? C.v[0][ln] : 0.0;
  w.sync();
  DZ_FLUSH();
#ifdef NBL_DZ_COUNTS      // developer build (tools/dantzig_bench.py): the iteration counts instead of the solution
  row.x = ln == 0 ? (double)dzIters : (ln == 1 ? (double)dzRemovals : (ln == 2 ? (double)dzTransfers : 0.0));
#endif
  return 1;
}

// ---- LCPUtils::reduce / removeFriction (LCPUtils.cpp:144-247, 346-520), lane = reduced row / column ----
// delete row + column `col` of the n x n problem; lanes >= col take the row data of their right neighbour
template <class W, class LDS>
DEV void coopRemoveRow(const W& w, LDS& C, int n, int col, CoopLcpRow& row) {
  const int ln = w.lane();
  if (ln < n) for (int j = col; j + 1 < n; j++) C.A[ln * CLD + j] = C.A[ln * CLD + j + 1];     // columns left, lane = row
  w.sync();
  if (ln < n) for (int i = col; i + 1 < n; i++) C.A[i * CLD + ln] = C.A[(i + 1) * CLD + ln];   // rows up, lane = column
  w.sync();
  const double x1 = w.shfl(row.x, ln + 1), b1 = w.shfl(row.b, ln + 1), l1 = w.shfl(row.lo, ln + 1), h1 = w.shfl(row.hi, ln + 1);
  const int f1 = w.shflI(row.findex, ln + 1);
  if (ln >= col && ln + 1 < n) { row.x = x1; row.b = b1; row.lo = l1; row.hi = h1; row.findex = f1; }
}

// Keep the rows / columns `keep` (bits < n) of the n-row problem, in their order: what removing the others one by one from the last one
// down (coopRemoveRow: two LDS passes over the matrix and five shuffles PER ROW - 16 times over for the friction rows of eight contacts)
// leaves, as one gather.  Pure data movement.  findex of a kept row follows its normal row (which is kept with it); mapTo (original row ->
// column of the problem) follows the columns, -1 for a row that is gone.  Returns the new size.
template <class W, class LDS>
DEV int coopKeepRows(const W& w, LDS& C, int n, uint64_t keep, CoopLcpRow& row, int& mapTo) {
  const int ln = w.lane();
  const int nNew = __builtin_popcountll(keep);
  auto newIdx = [&](int p) -> int { return __builtin_popcountll(keep & ((1ull << p) - 1ull)); };
  const bool kept = ln < n && ((keep >> ln) & 1ull);
  w.sync();
  if (kept) C.v[0][newIdx(ln)] = (double)ln;                    // new row -> old row
  w.sync();
  const int src = ln < nNew ? (int)C.v[0][ln] : 0;
  double col[MAXR];
  {
    uint64_t mm = keep;
#pragma unroll
    for (int j = 0; j < MAXR; j++) {
      if (j < nNew) {
        const int sj = __builtin_ctzll(mm);                     // (uniform)
        mm &= mm - 1ull;
        col[j] = C.A[src * CLD + sj];
      }
    }
  }
  w.sync();
  if (kept) { const int t = newIdx(ln); C.v[0][t] = row.x; C.v[1][t] = row.b; C.v[2][t] = row.lo; C.v[3][t] = row.hi; }
  const int fOld = w.shflI(row.findex, src);                    // (findex of the row this lane becomes)
  if (ln < nNew) {
#pragma unroll
    for (int j = 0; j < MAXR; j++) if (j < nNew) C.A[ln * CLD + j] = col[j];
  }
  w.sync();
  if (ln < nNew) {
    row.x = C.v[0][ln]; row.b = C.v[1][ln]; row.lo = C.v[2][ln]; row.hi = C.v[3][ln];
    row.findex = fOld >= 0 ? newIdx(fOld) : -1;
  }
  if (mapTo >= 0) mapTo = ((keep >> mapTo) & 1ull) ? newIdx(mapTo) : -1;
  w.sync();
  return nNew;
}

// merge near-identical columns (squared distance < 1e-4, |b_a - b_b| < 1e-4, same findex / hi / lo).  mapTo: this lane's
// ORIGINAL row -> reduced column.  Returns the reduced size.
template <class W, class LDS>
DEV int coopLcpReduce(const W& w, LDS& C, int n, CoopLcpRow& row, int& mapTo) {
  const double TH = 1e-4;
  const int ln = w.lane();
  for (;;) {
    int ma = -1, mb = -1;
    for (int a = 0; a < n - 1; a++) {
      // the cheap conditions first (equal findex / bounds, b within the threshold): they rule out most pairs, and the 24-term
      // column distance is only formed when some column b > a passes them (wave-uniform skip)
      const double ba = w.bcast(row.b, a), ha = w.bcast(row.hi, a), la = w.bcast(row.lo, a);
      const int fa = w.bcastI(row.findex, a);
      const bool cand = ln > a && ln < n && fabs(ba - row.b) < TH && fa == row.findex && ha == row.hi && la == row.lo;
      if (w.ballot(cand) == 0ull) continue;
      double d2 = 0;
      const int bcol = ln < n ? ln : 0;
#pragma unroll
      for (int r = 0; r < MAXR; r++) {   // unconditional reads (in bounds), rows >= n discarded by the select
        const double dr = C.A[r * CLD + a] - C.A[r * CLD + bcol];
        const double d = (r < n) ? dr : 0.0;
        d2 += d * d;
      }
      const uint64_t mm = w.ballot(cand && d2 < TH);
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
template <class W, class LDS>
DEV int coopLcpRemoveFriction(const W& w, LDS& C, int n, CoopLcpRow& row, int& mapTo) {
  const int ln = w.lane();
  return coopKeepRows(w, C, n, w.ballot(ln < n && row.findex == -1), row, mapTo);
}

// ---- PgsBoxedLcpSolver::solve (PgsBoxedLcpSolver.cpp:79-268), Option(30, 1e-6, 1e-3, 1e-9, false) ----
// Gauss-Seidel is sequential over the rows: up to 30 sweeps x 24 row steps are one dependent chain and its length is the cost.
// Residual form: every lane keeps ITS row of A scaled by 1 / a_ii (the reference divides in its first sweep and pre-scales the
// rows for the later ones, PgsBoxedLcpSolver.cpp:150-200) and the scaled residual r = b' - sum_k a'_k x_k of its row.  The step
// of row i is  x_i <- clamp(x_i + r_i)  on lane i, one broadcast of the change, one FMA per lane (A is symmetric: lane j's
// a'_j[i] is its share of column i).  The reference's own order - a 23-term dot product per row - would be a 23-deep chain on
// one lane, 720 times per stage; the two orders agree to round-off, the clamps, the convergence tests and the iteration cap are
// the reference's.
//
// The row step is written to compile to ~25 instructions (it was 131: selects on every lane, 64-bit mask bookkeeping, NaN
// canonicalisation around fmin / fmax, a row-count test per step):
//   * only lane i executes the update of row i (one EXEC region), no selects;
//   * bounds as products  lo = cL * xf, hi = cH * xf:  friction rows (cL, cH) = (-mu, mu) and xf = the current impulse of their
//     normal row; other rows (cL, cH) = (lo, hi) and xf = 1; rows left out (a_ii < eps, lanes >= n) cL = cH = 0, which pins them to
//     0 exactly like the reference's "x[i] = 0; continue";
//   * xf follows the normal row through the broadcast change, and only after NORMAL rows (a wave-uniform bit test);
//   * the sweep is instantiated for MAXR / 3, 2 MAXR / 3 or MAXR rows (24-row build: 8, 16, 24; the frictionless stage has 8).
// One row step = two hand-scheduled blocks around the broadcast (the two v_readlane of the change stay with the compiler: an asm
// operand cannot name the halves of a register pair).  Round 5's step was 27 instructions; this one is 17 (13 in a problem without
// friction rows):
//   * EXEC is not saved and restored around every block: the sweeps run in wave-uniform control flow, `all` (the EXEC mask on entry,
//     one SGPR pair for the whole solve) is put back where a block ends - v_readlane ignores EXEC;
//   * the convergence test of a row (5 instructions) is not part of its step: lane i keeps the change d_i of ITS row - its step runs under
//     EXEC = lane i, so the write lands in that lane only - and x_i is not touched again before the sweep ends, so ONE vector test after
//     the sweep judges all rows with the operands the reference's per-row test sees (pgsSweepBad);
//   * NOFRIC (stage 3: no friction row): every row has xf = 1, so lo = cL * 1 and hi = cH * 1 are cL and cH themselves (exact) and no
//     row follows another: no products, no follow step.
//   pgsOwnRow<I>:    on lane I only:  xi = min(max(x + r, cL xf), cH xf);  d = xi - x;  x = xi
//   pgsFollowRow<I>: on the lanes whose findex is I:  xf += ds  (ds = d of lane I, wave-uniform);  then on all lanes  r -= a'_I ds
#if defined(__HIP_DEVICE_COMPILE__)
#define NBL_PGS_OWN_NOFRIC                        \
        "s_mov_b64 exec, %[lane]\n\t"             \
        "v_add_f64 %[t0], %[x], %[r]\n\t"         \
        "v_max_f64 %[t0], %[t0], %[cL]\n\t"       \
        "v_min_f64 %[t0], %[t0], %[cH]\n\t"       \
        "v_add_f64 %[d], %[t0], -%[x]\n\t"        \
        "v_mov_b64 %[x], %[t0]\n\t"               \
        "s_mov_b64 exec, %[all]"
#define NBL_PGS_OWN_FRIC                          \
        "s_mov_b64 exec, %[lane]\n\t"             \
        "v_add_f64 %[t0], %[x], %[r]\n\t"         \
        "v_mul_f64 %[t1], %[cL], %[xf]\n\t"       \
        "v_max_f64 %[t0], %[t0], %[t1]\n\t"       \
        "v_mul_f64 %[t1], %[cH], %[xf]\n\t"       \
        "v_min_f64 %[t0], %[t0], %[t1]\n\t"       \
        "v_add_f64 %[d], %[t0], -%[x]\n\t"        \
        "v_mov_b64 %[x], %[t0]\n\t"               \
        "s_mov_b64 exec, %[all]"
// (the lane mask of row I: an inline constant for rows < 32; rows 32 .. 47 of the 16-contact build take it from an SGPR pair)
template <int I, bool NOFRIC>
DEV void pgsOwnRow(double& x, double r, double xf, double cL, double cH, double& d, unsigned long long all) {
  double t0, t1;
  if constexpr (I < 32) {
    if constexpr (NOFRIC)
      asm volatile(NBL_PGS_OWN_NOFRIC : [x] "+v"(x), [d] "+v"(d), [t0] "=&v"(t0)
                   : [r] "v"(r), [cL] "v"(cL), [cH] "v"(cH), [lane] "n"(1u << (I & 31)), [all] "s"(all));
    else
      asm volatile(NBL_PGS_OWN_FRIC : [x] "+v"(x), [d] "+v"(d), [t0] "=&v"(t0), [t1] "=&v"(t1)
                   : [r] "v"(r), [xf] "v"(xf), [cL] "v"(cL), [cH] "v"(cH), [lane] "n"(1u << (I & 31)), [all] "s"(all));
  } else {
    const unsigned long long laneMask = 1ull << I;
    if constexpr (NOFRIC)
      asm volatile(NBL_PGS_OWN_NOFRIC : [x] "+v"(x), [d] "+v"(d), [t0] "=&v"(t0)
                   : [r] "v"(r), [cL] "v"(cL), [cH] "v"(cH), [lane] "s"(laneMask), [all] "s"(all));
    else
      asm volatile(NBL_PGS_OWN_FRIC : [x] "+v"(x), [d] "+v"(d), [t0] "=&v"(t0), [t1] "=&v"(t1)
                   : [r] "v"(r), [xf] "v"(xf), [cL] "v"(cL), [cH] "v"(cH), [lane] "s"(laneMask), [all] "s"(all));
  }
}
// (ds comes from two v_readlane just ahead of the statement: a VALU read of an SGPR a VALU wrote needs two wait states - v_cmpx, which
//  does not read it, is one of them)
template <int I>
DEV void pgsFollowRow(double& xf, double& r, int fi, double ds, double narowI, unsigned long long all) {
  asm volatile("v_cmpx_eq_u32_e32 vcc, %[row], %[fi]\n\t"
               "s_nop 0\n\t"
               "v_add_f64 %[xf], %[xf], %[ds]\n\t"
               "s_mov_b64 exec, %[all]\n\t"
               "v_fmac_f64_e32 %[r], %[ds], %[na]"
               : [xf] "+v"(xf), [r] "+v"(r)
               : [fi] "v"(fi), [ds] "s"(ds), [na] "v"(narowI), [row] "n"(I), [all] "s"(all)
               : "vcc");
}
// the reference's per-row convergence test on all rows at once: FIRST sweep |d| > thr, later |x| > eps and |d| > rel |x|
template <bool FIRST>
DEV bool pgsSweepBad(double x, double d, double thrFirst, double relTol, double epsDiv) {
  unsigned long long bad;
  if constexpr (FIRST) {
    asm volatile("v_cmp_gt_f64_e64 %[bad], |%[d]|, %[thr]" : [bad] "=s"(bad) : [d] "v"(d), [thr] "v"(thrFirst));
  } else {
    double t2;
    unsigned long long sc;
    asm volatile("v_mul_f64 %[t2], |%[x]|, %[rel]\n\t"
                 "v_cmp_gt_f64_e64 %[sc], |%[x]|, %[eps]\n\t"
                 "v_cmp_gt_f64_e64 %[bad], |%[d]|, %[t2]\n\t"
                 "s_and_b64 %[bad], %[bad], %[sc]"
                 : [bad] "=&s"(bad), [sc] "=&s"(sc), [t2] "=&v"(t2)
                 : [x] "v"(x), [d] "v"(d), [rel] "v"(relTol), [eps] "v"(epsDiv)
                 : "scc");
  }
  return bad != 0ull;
}
#else
// host statements of the same steps (wave emulation: a thread per lane, the caller guards pgsOwnRow by the lane and ballots pgsSweepBad)
template <int I, bool NOFRIC>
DEV void pgsOwnRow(double& x, double r, double xf, double cL, double cH, double& d, unsigned long long) {
  const double xi = NOFRIC ? fmin(fmax(x + r, cL), cH) : fmin(fmax(x + r, cL * xf), cH * xf);
  d = xi - x;
  x = xi;
}
template <int I>
DEV void pgsFollowRow(double& xf, double& r, int fi, double ds, double narowI, unsigned long long) {
  if (fi == I) xf += ds;
  r = fma(narowI, ds, r);
}
template <bool FIRST>
DEV bool pgsSweepBad(double x, double d, double thrFirst, double relTol, double epsDiv) {
  return FIRST ? fabs(d) > thrFirst : (fabs(x) > epsDiv && fabs(d) > relTol * fabs(x));
}
#endif

// rowStep(IntTag<I0>{}, tag); ... rowStep(IntTag<I1 - 1>{}, tag);   (a compile-time unrolled row loop: the row index is an instruction operand)
template <int I0, int I1, class F, class T>
DEV void pgsRowRange(F& rowStep, T tag) {
  if constexpr (I0 < I1) {
    rowStep(IntTag<I0>{}, tag);
    pgsRowRange<I0 + 1, I1>(rowStep, tag);
  }
}

// NOFRIC: the caller knows that no row has a friction index (stage 3)
template <bool NOFRIC = false, class W, class LDS>
DEV bool coopPgs(const W& w, LDS& C, int n, CoopLcpRow& row) {
  const int maxIteration = 30;
  const double dxTh = 1e-6, relTol = 1e-3, epsDiv = 1e-9;
  const int ln = w.lane();
  const bool on = ln < n;
  const int me = on ? ln : 0;
  double narow[MAXR];       // MINUS this lane's row of A scaled by 1 / a_ii (the sign is exact, and the residual update wants it)
#pragma unroll
  for (int j = 0; j < MAXR; j++) { const double av = C.A[me * CLD + j]; narow[j] = (on && j < n) ? av : 0.0; }
  const double aii = on ? C.A[me * CLD + me] : 1.0;
  const bool inOrder = on && !(aii < epsDiv);          // rows with a_ii ~ 0 are set to 0 once and then left alone
  const double sc = inOrder ? 1.0 / aii : 1.0;
  const double nsc = -sc;
#pragma unroll
  for (int j = 0; j < MAXR; j++) narow[j] *= nsc;
  double x = on ? row.x : 0.0;
  double r0 = row.b * sc, r1 = 0.0;
#pragma unroll
  for (int j = 0; j < MAXR; j += 2) {
    r0 = fma(narow[j], j < n ? w.bcast(x, j) : 0.0, r0);
    r1 = fma(narow[j + 1], j + 1 < n ? w.bcast(x, j + 1) : 0.0, r1);
  }
  double r = r0 + r1;
  const int fi = on ? row.findex : -1;
  const bool fric = fi >= 0;
  const double cH = !inOrder ? 0.0 : row.hi, cL = !inOrder ? 0.0 : (fric ? -row.hi : row.lo);
  double xf = 1.0;
  if constexpr (!NOFRIC) {
    const double xNormal = w.shfl(x, fric ? fi : 0);
    xf = fric ? xNormal : 1.0;
  }
  const double thrFirst = inOrder ? dxTh : INFINITY;
  double d = 0.0;           // lane i: the change of row i in the sweep that is running (rows the sweep does not step keep 0 and x = 0)
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long all = __builtin_amdgcn_read_exec();
#else
  const unsigned long long all = 0ull;
#endif
  auto rowStep = [&](auto iTag, auto) {
    constexpr int i = decltype(iTag)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    pgsOwnRow<i, NOFRIC>(x, r, xf, cL, cH, d, all);
#else
    if (ln == i) pgsOwnRow<i, NOFRIC>(x, r, xf, cL, cH, d, all);
#endif
    const double ds = w.bcast(d, i);
    if constexpr (NOFRIC) r = fma(narow[i], ds, r);
    else pgsFollowRow<i>(xf, r, fi, ds, narow[i], all);
  };
  auto sweep = [&](auto nTag) {
    constexpr int NR = decltype(nTag)::value;
    pgsRowRange<0, NR>(rowStep, IntTag<0>{});       // rowStep(IntTag<0>{}, .); ... rowStep(IntTag<NR - 1>{}, .);
  };
  // "no row moved by more than the tolerance": the device's test is already a lane mask, the host emulation ballots its lanes' flags
  auto noneBad = [&](auto firstTag) -> bool {
    constexpr bool first = decltype(firstTag)::value != 0;
#if defined(__HIP_DEVICE_COMPILE__)
    return !pgsSweepBad<first>(x, d, thrFirst, relTol, epsDiv);
#else
    return w.ballot(pgsSweepBad<first>(x, d, thrFirst, relTol, epsDiv)) == 0ull;
#endif
  };
  auto solve = [&](auto nTag) -> bool {
    sweep(nTag);
    if (noneBad(IntTag<1>{})) return true;
    bool done = false;
#pragma unroll 1
    for (int iter = 1; iter < maxIteration; ++iter) {
      sweep(nTag);
      if (noneBad(IntTag<0>{})) { done = true; break; }
    }
    return done;
  };
  // (the sweep is instantiated for a third, two thirds and all of the rows; without friction rows there is at most a third)
  bool done;
  if constexpr (NOFRIC) done = solve(IntTag<MAXR / 3>{});
  else done = n <= MAXR / 3 ? solve(IntTag<MAXR / 3>{}) : (n <= 2 * MAXR / 3 ? solve(IntTag<2 * MAXR / 3>{}) : solve(IntTag<MAXR>{}));
  row.x = x;
  return done;
}

// ---- stages 1-3 of BoxedLcpConstraintSolver::solveLcp (:461-677) + registration / standardisation (:718-736) ----
// The three fallback stages read the same inputs (A, b, the pre-solve x) and none reads another's result - only WHICH result
// is kept depends on the earlier stages' success flags.  They are therefore written as three independent functions that the
// kernel runs on two wavefronts of a workgroup AT THE SAME TIME (stage 1 | stages 2 and 3, k_contact_cascade_stages); coopCascadeSelect then applies
// the reference's order of preference.  A world that falls through to the frictionless stage (two thirds of the worlds that
// reach the cascade on the metric distribution) used to pay reduce + Dantzig + reduce + 30 PGS sweeps + 30 PGS sweeps one after
// the other (5.2e5 cycles); now it pays the longer of stage 1 and stages 2 + 3.
struct CoopStageResult {
  double X;        // this lane's row of the stage's solution, mapped back to the world's rows
  int flags;       // uniform, CS_*
};
constexpr int CS_SOLVED = 1;   // the solver reported success (Dantzig: no early termination; PGS: converged)
constexpr int CS_VALID = 2;    // ... and isLCPSolutionValid accepted it
constexpr int CS_NAN = 4;      // Dantzig: NaN step length

template <class W, class LDS>
DEV void coopLoadProblem(const W& w, LDS& C, const CoopRow& R, double cfmDiag, double x0, CoopLcpRow& row, int& mapTo, int& nOut) {
  const int ln = w.lane();
  const int m = R.m;
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
  // The empty tangent rows of frictionless contacts (mu <= 1e-3, k_contact_rows_coop) do not exist in the reference's problem
  // (ContactConstraint dimension 1): take them out, from the last one down, so that the solvers see the reference's rows in the
  // reference's order - the initial permutation of dSolveLCP and with it the whole pivot sequence depend on it.
  const RowMask dead = (RowMask)w.ballot(ln < m && (!R.on || (R.fric && R.mu == 0.0)));   // ... and the rows of the world's other constrained groups
  nOut = m;
  if (dead != 0) nOut = coopKeepRows(w, C, m, (uint64_t)w.ballot(ln < m) & ~(uint64_t)dead, row, mapTo);
}
// X[o] = x_reduced[mapTo[o]]
template <class W, class LDS>
DEV double coopMapOut(const W& w, LDS& C, int m, int mapTo, double xred, int nred) {
  const int ln = w.lane();
  w.sync();
  if (ln < nred) C.v[3][ln] = xred;
  w.sync();
  return (ln < m && mapTo >= 0) ? C.v[3][mapTo] : 0.0;
}

// stage 1: reduce + Dantzig with early termination (:461-522)
template <class W>
DEV void coopCascadeStage1(const W& w, CascadeLds& C, const CoopRow& R, double X0, CoopStageResult& out) {
  CoopLcpRow row;
  int mapTo;
  int n0;
#if defined(NBL_CASCADE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  const int ln = w.lane();
  long long s1T = clock64();
#define S1_ADD(k) do { const long long n_ = clock64(); if (ln == 0) atomicAdd(&g_dzStat[k], (unsigned long long)(n_ - s1T)); s1T = n_; } while (0)
#else
#define S1_ADD(k) do { } while (0)
#endif
  coopLoadProblem(w, C, R, 0.0, X0, row, mapTo, n0);
  S1_ADD(12);
  const int nr = coopLcpReduce(w, C, n0, row, mapTo);
  S1_ADD(13);
  const int rc = coopDantzig(w, C, nr, row);
  S1_ADD(14);
  out.X = 0.0; out.flags = 0;
  if (rc == 1) {
    out.X = coopMapOut(w, C, R.m, mapTo, row.x, nr);
    out.flags = CS_SOLVED | (coopValid(w, C.v[0], R, out.X, false, 0.0) ? CS_VALID : 0);
  } else if (rc < 0) out.flags = CS_NAN;
  S1_ADD(15);
#undef S1_ADD
}
// stage 2: CFM + PGS from the pre-solve x (:539-597)
template <class W, class LDS>
DEV void coopCascadeStage2(const W& w, LDS& C, const CoopRow& R, double X0, double cfm, CoopStageResult& out) {
  CoopLcpRow row;
  int mapTo;
  int n0;
  coopLoadProblem(w, C, R, cfm, X0, row, mapTo, n0);
  const int nr = coopLcpReduce(w, C, n0, row, mapTo);
  out.X = 0.0; out.flags = 0;
  if (coopPgs(w, C, nr, row)) {
    out.X = coopMapOut(w, C, R.m, mapTo, row.x, nr);
    out.flags = CS_SOLVED | (coopValid(w, C.v[0], R, out.X, false, cfm) ? CS_VALID : 0);
  }
}
// stage 3: drop friction, PGS from zero (:606-677); its result is used whatever the solver says
template <class W, class LDS>
DEV void coopCascadeStage3(const W& w, LDS& C, const CoopRow& R, double X0, double cfm, CoopStageResult& out) {
  CoopLcpRow row;
  int mapTo;
  int n0;
  coopLoadProblem(w, C, R, cfm, X0, row, mapTo, n0);
  const int nr = coopLcpRemoveFriction(w, C, n0, row, mapTo);
  row.x = 0.0;
  const bool ok3 = coopPgs<true>(w, C, nr, row);    // (nr <= MAX_CONTACTS rows, none with a friction index)
  out.X = coopMapOut(w, C, R.m, mapTo, row.x, nr);
  out.flags = ok3 ? CS_SOLVED : 0;
}

struct CoopCascadeOut {
  double X;          // impulses of this lane's row
  CoopClasses K;
  double cfm;
  uint32_t st;       // NBL_ST_* bits to OR into the world's status
  bool pinvValid;
};

// The reference's order of preference over the three stage results (BoxedLcpConstraintSolver.cpp:461-687).  X0: the pre-solve x.
template <class W>
DEV void coopCascadeChoose(const W& w, int m, double X0, double fallbackCfm, const CoopStageResult& r1, const CoopStageResult& r2,
                           const CoopStageResult& r3, double& X, double& cfm, bool& ignoreFriction, uint32_t& st) {
  const int ln = w.lane();
  auto hasNan = [&](double x) -> bool { return w.ballot(ln < m && x != x) != 0ull; };
  bool success = false;
  st = 0; ignoreFriction = false; cfm = 0.0; X = X0;
  if (r1.flags & CS_SOLVED) {
    X = r1.X;
    success = (r1.flags & CS_VALID) != 0;
    if (success) st |= 0x4u;
  }
  if ((r1.flags & CS_NAN) || hasNan(X)) { success = false; X = 0.0; st |= 0x40u; }
  if (!success) {
    cfm = fallbackCfm;
    if (r2.flags & CS_SOLVED) {
      X = r2.X;
      success = (r2.flags & CS_VALID) != 0;
      if (success) st |= 0x8u;
    }
  }
  if (!success) {
    ignoreFriction = true;
    X = r3.X;
    st |= 0x10u;
    if (!(r3.flags & CS_SOLVED)) st |= 0x20u;
  }
  if (hasNan(X)) { X = 0.0; st |= 0x40u; }
}

// ... then registration, classification and standardisation of the chosen solution (:718-736)
template <class W>
DEV void coopCascadeSelect(const W& w, CoopLds& S, const CoopRow& R, double X0, double fallbackCfm, const CoopStageResult& r1,
                           const CoopStageResult& r2, const CoopStageResult& r3, CoopCascadeOut& out) {
  double X, cfm;
  bool ignoreFriction;
  uint32_t st;
  coopCascadeChoose(w, R.m, X0, fallbackCfm, r1, r2, r3, X, cfm, ignoreFriction, st);
  bool pinvValid = false;
  const bool std = coopStandardizeLoop(w, S, R, X, cfm, ignoreFriction, 0u, pinvValid, out.K);
  if (std) st |= 0x100u;
  // (pinvValid also when the loop ended on an invalid standardised solution: S.P is then still Q^+ of the classification handed back,
  //  factorised in that very iteration - the record wants exactly that matrix, and factorising it again cost the slowest worlds of
  //  k_contact_cascade_final 65-100 k of their 200 k cycles)
  out.X = X; out.cfm = cfm; out.st = st; out.pinvValid = pinvValid;
}

}  // namespace NBL_NS
