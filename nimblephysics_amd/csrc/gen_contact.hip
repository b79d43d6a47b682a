// gen_contact.hip — the dense part of the contact stage for worlds with ANY number of contacts (the general instantiation of the library,
// NBL_MAXC = 64: up to 64 contacts = 192 LCP rows, 64 colliders, 512 collider pairs per world), one world per wavefront:
//   k_contact_rows_gen      contact Jacobians, b, the unit-impulse tests and A                          (k_contact_rows_coop with row tiles)
//   k_contact_solve_gen     the whole solver cascade of a world - stage 0, then only if needed stages 1-3 in the reference's order - group by
//                           group, the record's Q^+, v' = v_pre + M^-1 J^T x                            (gen_lcp_dev.hpp / gen_dantzig_dev.hpp)
//   k_bwd_contact_a_gen     the dense (c x c) part of the contact adjoint;  k_bwd_contact_b_gen  its tree part;  k_bwd_bounce_gen
// Same record, same scratch rows and the same mathematics as coop_kernels.hip, whose kernels are written around lane = LCP row with at most
// 64 rows and register-resident columns: here a lane strides through the rows, matrices live in the world's slice of HBM, vectors - and,
// for a world of few rows, the matrices a factorisation works on - in LDS (round 6: 0.58 -> 2 M world-steps/s on eight-contact worlds).
// Nothing of the metric path runs through this file (nimble_amd_dispatch.cpp hands a model to this instantiation only when it asks for more
// than 16 contact slots / 32 colliders / 64 collider pairs); it exists so that no legal world gets a truncated answer.
#include "gen_lcp_dev.hpp"
#include "gen_dantzig_dev.hpp"
#include "coop_wave_dev.hpp"

namespace NBL_NS {

struct GenWaveDev {
  DEV int lane() const { return (int)(threadIdx.x & 63u); }
  DEV int lanes() const { return 64; }
  // one wavefront per workgroup: the barrier also makes the wave's stores to its world's HBM scratch visible to its other lanes
  DEV void sync() const { __syncthreads(); }
  DEV double maxAll(double v) const { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }
  DEV int minAllI(int v) const { for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
  DEV double sumAll(double v) const { for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o); return v; }   // (a + b on both partners: every lane ends with the same bits)
  DEV bool anyAll(bool b) const { return __ballot(b ? 1 : 0) != 0ull; }
  // the value of lane `src` (wave-uniform): v_readlane
  DEV double bcast(double v, int src) const {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
  }
  DEV int bcastI(int v, int src) const { return __builtin_amdgcn_readlane(v, src); }
  // orders the wave's own LDS traffic (a wave's LDS instructions execute in order; this only stops the compiler from moving them)
  DEV void fence() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
};

// the world's scratch: genScratchDoubles(ld) per world after the contact-backward rows of the workspace; mat[3] (the pseudo-inverse) is the
// record's own Q^+ block (same leading dimension), so the last factorisation of the cascade lands where the backward pass reads it
DEV GenScratch genScratchOf(double* gws, int64_t world, double* recordPinv, int ld) {
  GenScratch S;
  const size_t mat = (size_t)ld * ld;
  double* base = gws + (size_t)world * genScratchDoubles(ld);
  S.ld = ld;
  S.mat[0] = base; S.mat[1] = base + mat; S.mat[2] = base + 2 * mat; S.mat[3] = recordPinv;
  S.mat[4] = base + 3 * mat;
  S.vec = base + 4 * mat;
  return S;
}

// ======================================================================================================================================
// rows: see k_contact_rows_coop (coop_kernels.hip) for the mathematics - world-aligned frame with its origin at the root of each tree, one
// wrench per row and side, impulses up the two ancestor chains, velocity changes down the tree, A[r][c] = F_c . (dV_A - dV_B).  Here the rows
// are processed in tiles of `ts` (a lane = a row of the tile; the velocity-change field [body][6][ts] is what limits a tile: LDS).
//   lds doubles: F[the model's rows][6] x 2, Sw / AISw / Vw [nb][6], psi[nb], origin[nb][3], free[nFree][54], acc[nb][6][ts], contact bodies
// ======================================================================================================================================
__global__ __launch_bounds__(64) void k_contact_rows_gen(DevModel mdl, const DevBody* __restrict__ bodies, const DevContactModel* __restrict__ cm, int64_t B,
                                                         double* __restrict__ saved, SavedLayout lay, const double* __restrict__ ws, int ts) {
  extern __shared__ __attribute__((aligned(16))) double ldsG[];
  const int nb = mdl.nb;
  const GenWaveDev w;
  const int ln = w.lane();
  const int ldr = lay.ldr;              // leading dimension of the record's dense blocks and of the world's scratch matrices
  double* Fs = ldsG;
  double* FsB = Fs + 6 * ldr;           // (the wrench tables sized by the MODEL's rows, like the record: genLeadingDim)
  double* Sw = FsB + 6 * ldr;
  double* AISw = Sw + 6 * nb;
  double* Vw = AISw + 6 * nb;
  double* psiL = Vw + 6 * nb;
  double* orig = psiL + nb;
  double* freeL = orig + 3 * nb;
  double* acc = freeL + 54 * mdl.nFree;               // [body][6][ts]
  int* cbody = reinterpret_cast<int*>(acc + (size_t)6 * nb * ts);   // [2][MAX_CONTACTS]
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x;
  if (b >= mdl.b1) return;
  const int nC = (int)svAt(saved, lay.nc, B, b);
  const int m = 3 * nC;
  if (m == 0) return;
  Ctx c = makeCtx(mdl, bodies, nullptr, const_cast<double*>(ws), B, b, saved, &lay);
  double* dn = denseOf(saved, lay, B, b);
  auto ld6 = [](const double* base) -> V6 { double a[6]; for (int e = 0; e < 6; e++) a[e] = base[e]; return fromArr(a); };
  auto st6 = [](double* base, V6 x) { double a[6]; toArr(x, a); for (int e = 0; e < 6; e++) base[e] = a[e]; };
  // ---- lane = body: frame origins, world-frame joint axis, AI*S, twist at v_pre; the blocks of the free-joint bodies ----
  for (int i = ln; i < nb; i += 64) {
    const DevBody& bd = bodies[i];
    const V3 o = ldTAt(c, bd.root, WS_TW).p;
    orig[3 * i] = o.x; orig[3 * i + 1] = o.y; orig[3 * i + 2] = o.z;
    T12 TW = ldTAt(c, i, WS_TW);
    TW.p = TW.p - o;
    st6(Vw + 6 * i, AdT(TW, ldV6(c, i, WS_VTW)));
    if (bd.jtype != JT_FREE) {
      st6(Sw + 6 * i, AdT(TW, cV6(bd.S)));
      st6(AISw + 6 * i, dAdInvT(TW, ldV6(c, i, WS_AIS)));
      psiL[i] = wsAt(c, i, WS_PSI);
    } else {
      double* fl = freeL + 54 * bd.freeIdx;
      for (int e = 0; e < 54; e++) {
        const int slot = e < 21 ? WS_PSI + e : (e < 42 ? WS_AI + (e - 21) : WS_TW + (e - 42));
        fl[e] = e >= 51 ? 0.0 : wsAt(c, i, slot);      // a free joint is the root of its tree: no translation in the frame of its own origin
      }
    }
  }
  w.sync();
  struct RowGeom { V6 F, FB; int bA, bB; uint64_t mA, mB; bool on; };
  auto rowGeom = [&](int row) -> RowGeom {
    RowGeom g;
    g.on = row < m;
    const int rr = g.on ? row : 0;
    const int ci = rr / 3, kk = rr % 3;
    const int r0 = lay.contacts + ci * CR_SIZE;
    const V3 p = mk3(svAt(saved, r0 + CR_POINT, B, b), svAt(saved, r0 + CR_POINT + 1, B, b), svAt(saved, r0 + CR_POINT + 2, B, b));
    const V3 nrm = mk3(svAt(saved, r0 + CR_NORMAL, B, b), svAt(saved, r0 + CR_NORMAL + 1, B, b), svAt(saved, r0 + CR_NORMAL + 2, B, b));
    const int bxA = (int)svAt(saved, r0 + CR_BOXA, B, b), bxB = (int)svAt(saved, r0 + CR_BOXB, B, b);
    const bool isLim = (int)svAt(saved, r0 + CR_TYPE, B, b) == CT_LIMIT;
    V3 t1, t2;
    tangentBasis(nrm, t1, t2);
    // a frictionless contact (mu <= 1e-3) keeps its three row slots, the two tangent rows EMPTY (k_contact_rows_coop)
    const double muRow = isLim ? 0.0 : fmin(crMuOf(cm, bxA), crMuOf(cm, bxB));
    const V3 dirOn = kk == 0 ? nrm : (kk == 1 ? t1 : t2);
    const V3 dir = (kk != 0 && !(muRow > 1e-3)) ? mk3(0.0, 0.0, 0.0) : dirOn;
    g.bA = crBodyOf(cm, bxA); g.bB = crBodyOf(cm, bxB);
    const int iA = g.bA < 0 ? 0 : g.bA, iB = g.bB < 0 ? 0 : g.bB;
    const V3 oA = mk3(orig[3 * iA], orig[3 * iA + 1], orig[3 * iA + 2]), oB = mk3(orig[3 * iB], orig[3 * iB + 1], orig[3 * iB + 2]);
    g.F = mk6(cross(p - oA, dir), dir); g.FB = mk6(cross(p - oB, dir), dir);
    if (isLim) {
      // joint-limit row: the generalized unit impulse sigma e_d as the wrench pair (+F on the joint's child body, -F on its parent),
      // F = sigma S_d / |S_d|^2 (JointLimitConstraint::applyUnitImpulse / getVelocityChange, JointLimitConstraint.cpp:293-349)
      const double limSigma = svAt(saved, r0 + CR_EA_FIXED + 1, B, b);
      const V6 Sd = ld6(Sw + 6 * iA);
      const double s2 = dot(Sd, Sd);
      g.F = kk == 0 ? (limSigma / s2) * Sd : zero6();
      g.FB = g.F;
    }
    g.mA = g.bA >= 0 ? cm->ancestors[g.bA] : 0ull;
    g.mB = g.bB >= 0 ? cm->ancestors[g.bB] : 0ull;
    return g;
  };
  // ---- pass 1, every row: wrenches, the two bodies of every contact, b = -J^T V (+ bouncing) ----
  for (int row = ln; row < m; row += 64) {
    const RowGeom g = rowGeom(row);
    const int ci = row / 3, kk = row % 3;
    const int r0 = lay.contacts + ci * CR_SIZE;
    st6(Fs + 6 * row, g.F); st6(FsB + 6 * row, g.FB);
    if (kk == 0) { cbody[ci] = g.bA; cbody[MAX_CONTACTS + ci] = g.bB; }
    double rel = 0;
    if (g.bA >= 0) rel -= dot(g.F, ld6(Vw + 6 * g.bA));
    if (g.bB >= 0) rel += dot(g.FB, ld6(Vw + 6 * g.bB));
    if (kk == 0) {
      // "bouncing" (ContactConstraint.cpp:393-441 / 470-512): penetration correction and restitution, see k_contact_rows_coop
      const int bxA = (int)svAt(saved, r0 + CR_BOXA, B, b), bxB = (int)svAt(saved, r0 + CR_BOXB, B, b);
      const bool isLim = (int)svAt(saved, r0 + CR_TYPE, B, b) == CT_LIMIT;
      double bouncing = 0.0;
      if (cm->penetrationCorrection && !isLim) {
        double bv = svAt(saved, r0 + CR_DEPTH, B, b) - 0.0;
        if (bv < 0.0) bv = 0.0;
        else { bv *= 0.01 * (1.0 / mdl.dt); if (bv > 1e-3) bv = 1e-3; }
        bouncing = bv;
      }
      const double eR = isLim ? 0.0 : cm->boxes[(unsigned)bxA < (unsigned)MAX_BOXES ? bxA : 0].restitution * cm->boxes[(unsigned)bxB < (unsigned)MAX_BOXES ? bxB : 0].restitution;
      double coeff = 0.0;
      if (eR > 1e-3) {
        const double rv = rel * eR;
        if (rv > 1e-1) { coeff = eR; if (rv > bouncing) bouncing = rv > 1e+2 ? 1e+2 : rv; }
      }
      rel += bouncing;
      svAt(saved, lay.rest + ci, B, b) = coeff;
    }
    svAt(saved, lay.b + row, B, b) = rel;
  }
  w.sync();
  // ---- pass 2, tile by tile: A_c column, the unit-impulse test, M^-1 J^T column, row of A ----
  for (int t0 = 0; t0 < m; t0 += ts) {
    const int row = t0 + ln;
    const bool mine = ln < ts;
    const RowGeom g = rowGeom(mine ? row : m);
    const bool on = mine && g.on;
    const V6 F = g.F, FB = g.FB;
    const int bA = g.bA, bB = g.bB;
    const uint64_t mA = g.mA, mB = g.mB;
    auto accAt = [&](int body, int e) -> double& { return acc[((size_t)body * 6 + e) * ts + ln]; };
    auto ldAcc = [&](int body) -> V6 { double a[6]; for (int e = 0; e < 6; e++) a[e] = accAt(body, e); return fromArr(a); };
    auto stAcc = [&](int body, V6 x) { double a[6]; toArr(x, a); for (int e = 0; e < 6; e++) accAt(body, e) = a[e]; };
    if (on) {
      // constraint forces in joint space (DCC::getConstraintForces): A_c[i] = sigma_i s_i . F
      for (int i = 0; i < nb; i++) {
        const DevBody& bd = bodies[i];
        const bool pa = (mA >> i) & 1ull, pb = (mB >> i) & 1ull;
        const double mult = (pa && pb) ? 0.0 : (pa ? 1.0 : (pb ? -1.0 : 0.0));
        const V6 Fi = pb ? FB : F;
        if (bd.jtype != JT_FREE) dn[lay.aall + bd.dofOff * ldr + row] = mult * dot(ld6(Sw + 6 * i), Fi);
        else {
          double v6[6];
          toArr(dAdT(cT(bd.Tcj), dAdT(cT(freeL + 54 * bd.freeIdx + 42), Fi)), v6);
          for (int e = 0; e < 6; e++) dn[lay.aall + (bd.dofOff + e) * ldr + row] = mult * v6[e];
        }
        for (int e = 0; e < 6; e++) accAt(i, e) = 0.0;
      }
      // leaf -> root: bias impulses along the two ancestor chains (world wrenches)
      const uint64_t chain = mA | mB;
      for (int i = nb - 1; i >= 0; i--) {
        if (!((chain >> i) & 1ull)) continue;
        const DevBody& bd = bodies[i];
        V6 Bi = ldAcc(i);
        if (i == bA) Bi = Bi - F;
        if (i == bB) Bi = Bi + FB;
        stAcc(i, Bi);
        if (bd.jtype != JT_FREE && bd.parent >= 0) {
          const double uimp = -dot(ld6(Sw + 6 * i), Bi);
          stAcc(bd.parent, ldAcc(bd.parent) + Bi + (psiL[i] * uimp) * ld6(AISw + 6 * i));
        }
      }
      // root -> leaf: velocity changes of every body (world twists), joint-space response
      for (int i = 0; i < nb; i++) {
        const DevBody& bd = bodies[i];
        const V6 X = bd.parent >= 0 ? ldAcc(bd.parent) : zero6();
        const V6 Bi = ((chain >> i) & 1ull) ? ldAcc(i) : zero6();
        if (bd.jtype != JT_FREE) {
          const V6 S = ld6(Sw + 6 * i);
          const double dq = psiL[i] * (-dot(S, Bi) - dot(ld6(AISw + 6 * i), X));
          stAcc(i, X + dq * S);
          dn[lay.massed + bd.dofOff * ldr + row] = dq;
        } else {
          const double* fl = freeL + 54 * bd.freeIdx;
          const T12 Tcj = cT(bd.Tcj), TW = cT(fl + 42);
          LDL6 f;
          for (int e = 0; e < 15; e++) f.l[e] = fl[e];
          for (int e = 0; e < 6; e++) f.d[e] = fl[15 + e];
          S6 AIb;
          for (int e = 0; e < 21; e++) AIb.a[e] = fl[21 + e];
          const V6 Xb = AdInvT(TW, X);
          double r[6], u[6], pj[6];
          toArr(dAdT(Tcj, dAdT(TW, Bi)), u);
          toArr(dAdT(Tcj, mul(AIb, Xb)), pj);
          for (int e = 0; e < 6; e++) r[e] = -u[e] - pj[e];
          ldl6Solve(f, r);
          stAcc(i, AdT(TW, Xb + AdT(Tcj, fromArr(r))));
          for (int e = 0; e < 6; e++) dn[lay.massed + (bd.dofOff + e) * ldr + row] = r[e];
        }
      }
      // row of A: relative-velocity response at every row of the contacts c2 >= ci, mirrored into the earlier rows
      const int ci = row / 3;
      for (int c2 = ci; c2 < nC; c2++) {
        const int b2A = cbody[c2], b2B = cbody[MAX_CONTACTS + c2];
        const V6 dVA = b2A >= 0 ? ldAcc(b2A) : zero6(), dVB = b2B >= 0 ? ldAcc(b2B) : zero6();
        for (int k2 = 0; k2 < 3; k2++) {
          const int col = 3 * c2 + k2;
          const double val = dot(ld6(Fs + 6 * col), dVA) - dot(ld6(FsB + 6 * col), dVB);
          dn[lay.A + row * ldr + col] = val;
          if (c2 > ci) dn[lay.A + col * ldr + row] = val;
        }
      }
    }
    w.sync();
  }
}

// ======================================================================================================================================
// the solver cascade of one world
// ======================================================================================================================================
struct GenFinal {           // the result row by row while the groups are solved one after the other (carved out of the pool behind the rows)
  double *X, *E, *cfm, *xcache;
  int* cls;
};
#ifndef GEN_SOLVE_FAST_MATS
#define GEN_SOLVE_FAST_MATS 0     // fast (LDS) matrices of the step kernel: measured - 8 kB of LDS more per world cost more occupancy than the LDS-resident Gauss-Seidel matrix gains
#endif
__host__ __device__ inline size_t genFinalDoubles(int cap) { return (size_t)4 * cap + ((size_t)cap + 1) / 2; }
// dynamic LDS of k_contact_solve_gen for a model of `rows` rows: the rows' pool + the final-result arrays
// ... + ONE fast matrix of GEN_FAST_N x GEN_FAST_N (8 kB: the scaled matrix of the Gauss-Seidel sweeps of a problem of up to 32 rows; with
// 190 registers per lane eight worlds share a CU, so up to 20 kB of LDS per world are free)
// ... + the 16 scratch VECTORS of the cascade (the Dantzig driver's x, w, dx, dw, ell, ..., the compacted problem's x, b, lo, hi, the
// candidate) when the model has at most GEN_VEC_LDS_ROWS rows: in HBM scratch every barrier after a store to one of them waited for the
// store to land (hundreds of barriers per world); the matrices stay in HBM scratch.
constexpr int GEN_VEC_LDS_ROWS = 96;
// ... and never fewer than GEN_VEC_LDS_MIN doubles: the pool also lends itself, packed by the WORLD's rows, to the working pair of the
// pseudo-inverse (2 m^2) and to the matrix of the Gauss-Seidel sweeps (n^2) - 1152 doubles hold the pair of a world of eight contacts
// whatever the model's slot count is (a 16-slot model would otherwise send its eight-contact worlds to HBM scratch: 1.46 against 1.74 M/s)
constexpr int GEN_VEC_LDS_MIN = 1152;      // (1536 - room for the Dantzig matrix as well - was measured: the solve kernel unchanged, its neighbours 2.5 % slower)
__host__ __device__ inline size_t genSolveVecDoubles(int rows) { return rows <= GEN_VEC_LDS_ROWS ? ((size_t)16 * rows > GEN_VEC_LDS_MIN ? (size_t)16 * rows : (size_t)GEN_VEC_LDS_MIN) : 0; }
__host__ __device__ inline size_t genSolveLdsBytes(int rows) {
  const int cap = genRowsCap(rows);
  return (genRowsDoubles(cap) + genFinalDoubles(cap) + GEN_SOLVE_FAST_MATS * GEN_FAST_N * GEN_FAST_N + genSolveVecDoubles(rows)) * sizeof(double);
}

__global__ __launch_bounds__(64) NBL_WAVES(NBL_W_SOLVE_GEN) void k_contact_solve_gen(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B, double* __restrict__ saved,
                                                          SavedLayout lay, const double* __restrict__ cacheIn, double* __restrict__ cacheOut,
                                                          double* __restrict__ next, uint32_t* __restrict__ status, double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double ldsRows[];
  const GenWaveDev w;
  const int ln = w.lane();
  const int ldr = lay.ldr;              // leading dimension of the record's dense blocks and of the world's scratch matrices
  GenRows R;
  GenFinal Fn;
  {
    const int cap = genRowsCap(ldr);
    genRowsCarve(R, ldsRows, cap);
    double* f = ldsRows + genRowsDoubles(cap);
    Fn.X = f; Fn.E = f + cap; Fn.cfm = f + 2 * cap; Fn.xcache = f + 3 * cap; Fn.cls = reinterpret_cast<int*>(f + 4 * cap);
  }
  double* fastMat = ldsRows + genRowsDoubles(genRowsCap(ldr)) + genFinalDoubles(genRowsCap(ldr));
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x;
  if (b >= mdl.b1) return;
  const int n = mdl.n;
  const double ncD = svAt(saved, lay.nc, B, b);
  const int nC = (int)ncD;
  const int m = 3 * nC;
  double* nv = next + (int64_t)n * B;
  double* dn = denseOf(saved, lay, B, b);
  const double* A = dn + lay.A;
  if (ln == 0 && status) status[b] |= (ncD - (double)nC > 0.25 ? 0x80u : 0u);
  if (m == 0) {
    for (int r = ln; r <= MAX_ROWS; r += 64) if (cacheOut) cacheOut[(int64_t)r * B + b] = 0.0;
    for (int r = ln; r < MAX_ROWS; r += 64) { svAt(saved, lay.x + r, B, b) = 0.0; svAt(saved, lay.cls + r, B, b) = 0.0; svAt(saved, lay.cfm + r, B, b) = 0.0; }
    if (ln == 0) svAt(saved, lay.pflag, B, b) = 0.0;
    for (int d = ln; d < n; d += 64) svAt(saved, lay.w + d, B, b) = 0.0;
    return;
  }
  GenScratch S = genScratchOf(gws, b, dn + lay.pinv, ldr);
  if (GEN_SOLVE_FAST_MATS > 0) { S.fast = fastMat; S.fastN = GEN_FAST_N; S.fastMats = GEN_SOLVE_FAST_MATS; }
  if (genSolveVecDoubles(ldr) > 0) { S.vec = fastMat + GEN_SOLVE_FAST_MATS * GEN_FAST_N * GEN_FAST_N; S.vecFast = true; S.vecDoubles = (int)genSolveVecDoubles(ldr); }      // the cascade's vectors in LDS
  GEN_T0();
  GEN_CNT(10);
  // ---- the rows ----
  const bool haveCache = cacheIn && ((int)cacheIn[(int64_t)MAX_ROWS * B + b] == m);
  int nLimMine = 0;
  for (int r = ln; r < m; r += 64) {
    const int r0 = lay.contacts + (r / 3) * CR_SIZE;
    const int cA = (int)saved[(int64_t)(r0 + CR_BOXA) * B + b], cB = (int)saved[(int64_t)(r0 + CR_BOXB) * B + b];
    const double muA = crMuOf(cm, cA), muB = crMuOf(cm, cB);
    const bool lim = cA >= CR_BODY_CODE && (r % 3) == 0;
    R.lim[r] = lim ? 1 : 0;
    R.neg[r] = (lim && saved[(int64_t)(r0 + CR_EA_FIXED + 1) * B + b] < 0.0) ? 1 : 0;
    nLimMine += lim;
    double mu = muA < muB ? muA : muB;
    if (!(mu > 1e-3)) mu = 0.0;
    R.mu[r] = mu;
    R.Bv[r] = saved[(int64_t)(lay.b + r) * B + b];
    R.fric[r] = (r % 3) != 0; R.fp[r] = r - (r % 3);
    R.rowOn[r] = 1; R.on[r] = 1;
    double cn = 0.0;
    for (int i = 0; i < m; i++) { const double a = A[(size_t)i * ldr + r]; cn = fma(a, a, cn); }
    R.colNorm[r] = cn;
    Fn.xcache[r] = haveCache ? (R.neg[r] ? -1.0 : 1.0) * cacheIn[(int64_t)r * B + b] : 0.0;
    Fn.X[r] = 0.0; Fn.E[r] = 0.0; Fn.cfm[r] = 0.0; Fn.cls[r] = RC_NOT_CLAMPING;
  }
  R.m = m; R.ld = ldr;
  const int nLim = (int)w.sumAll((double)nLimMine);
  R.anyLim = nLim > 0;
  if (ln == 0 && status) status[b] |= (nC - nLim > 0 ? 0x1u : 0u) | (nLim > 0 ? 0x400u : 0u);
  // ---- constrained groups (ConstraintSolver::buildConstrainedGroups :724-780, ContactConstraint::uniteSkeletons :879-907): skeletons
  //      connected by a contact between two reactive bodies are one group; groups are numbered by their first contact (coopGroups) ----
  if (cm->oneSkeleton) {               // (one skeleton: one constrained group - nothing to label; the labelling below is lane 0 alone)
    for (int r = ln; r < m; r += 64) R.gid[r] = 0;
    if (ln == 0) R.iscal[2] = 1;
  } else if (ln == 0) {
    int* lab = R.perm;                 // label of skeleton s (< 64)
    int* cu = reinterpret_cast<int*>(R.t0);
    int* cv = cu + MAX_CONTACTS;
    for (int s = 0; s < 64; s++) lab[s] = s;
    for (int c = 0; c < nC; c++) {
      const int r0 = lay.contacts + c * CR_SIZE;
      const int bA = crBodyOf(cm, (int)saved[(int64_t)(r0 + CR_BOXA) * B + b]), bB = crBodyOf(cm, (int)saved[(int64_t)(r0 + CR_BOXB) * B + b]);
      const int sA = bA >= 0 ? cm->skelOf[bA] : -1, sB = bB >= 0 ? cm->skelOf[bB] : -1;
      int u = sA >= 0 ? sA : sB;
      int v = (sA >= 0 && sB >= 0) ? sB : u;
      if (u < 0) { u = 0; v = 0; }
      cu[c] = u; cv[c] = v;
      const int lu = lab[u], lv = lab[v];
      const int mn = lu < lv ? lu : lv;
      for (int s = 0; s < 64; s++) if (lab[s] == lu || lab[s] == lv) lab[s] = mn;
    }
    int nGroups = 0;
    for (int c = 0; c < nC; c++) {
      const int comp = lab[cu[c]];
      int g = -1;
      for (int c2 = 0; c2 < c; c2++) if (lab[cu[c2]] == comp) { g = R.gid[3 * c2]; break; }
      if (g < 0) g = nGroups++;
      R.gid[3 * c] = g; R.gid[3 * c + 1] = g; R.gid[3 * c + 2] = g;
    }
    R.iscal[2] = nGroups;
  }
  w.sync();
  const int nGroups = R.iscal[2];
  uint32_t stAll = 0x100u;
  bool anyFail = false, pinvValidWorld = false;
  for (int g = 0; g < nGroups; g++) {
    for (int r = ln; r < m; r += 64) { R.on[r] = R.gid[r] == g; R.X[r] = (haveCache && R.on[r]) ? Fn.xcache[r] : 0.0; }
    w.sync();
    bool pinvValid = false;
    GenClasses K;
    double cfmG = 0.0;
    GEN_T(0);
    const bool ok = genStage0(w, A, ldr, R, S, haveCache, pinvValid, K);
    GEN_T(1);
    if (!ok) {
      anyFail = true;
      uint32_t st = 0;
      genCascade(w, A, ldr, R, S, cm->fallbackCfm, cfmG, st, pinvValid, K);
      GEN_T(12);
      stAll = (stAll & ~0x100u) | (st & ~0x100u) | (stAll & st & 0x100u);
    }
    for (int r = ln; r < m; r += 64)
      if (R.on[r]) { Fn.X[r] = R.X[r]; Fn.cls[r] = R.cls[r]; Fn.E[r] = R.E[r]; Fn.cfm[r] = cfmG; }
    w.sync();
    pinvValidWorld = nGroups == 1 && pinvValid;
  }
  // ---- the world's final classification and the record's Q^+ (of the whole clamping set: block diagonal over the groups, each block with
  //      its group's CFM; joint-limit rows out: the reference's backward pass gives them no constraint-force column) ----
  int ncMine = 0, nuMine = 0;
  bool limClamp = false;
  for (int r = ln; r < m; r += 64) {
    R.on[r] = 1;
    R.cls[r] = Fn.cls[r]; R.E[r] = Fn.E[r]; R.X[r] = Fn.X[r];
    if (R.lim[r] && R.cls[r] == RC_CLAMPING) limClamp = true;
  }
  w.sync();
  const bool anyLimClamp = w.anyAll(limClamp);
  for (int r = ln; r < m; r += 64) {
    if (anyLimClamp && R.lim[r]) R.cls[r] = RC_NOT_CLAMPING;
    ncMine += R.cls[r] == RC_CLAMPING; nuMine += R.cls[r] == RC_UPPER_BOUND;
  }
  w.sync();
  GenClasses K;
  K.nc = (int)w.sumAll((double)ncMine); K.nu = (int)w.sumAll((double)nuMine);
  bool pinvValid = pinvValidWorld && !anyLimClamp;
  GEN_T(0);
  if (!pinvValid) {
    if (K.nc > 0) {
      const GenPinvPair pp = genPinvPair(S, m);
      genBuildQ(w, A, ldr, R, K, 0.0, pp.M, Fn.cfm, pp.ld);
      genPinv(w, R, pp.M, pp.G, S.mat[2], S.mat[3], m, K.nc, K.nu == 0, pp.ld, R.t0, 3 * R.cap);
    } else {
      for (int j = ln; j < m; j += 64) for (int i = 0; i < m; i++) S.mat[3][(size_t)i * ldr + j] = 0.0;
      w.sync();
    }
    pinvValid = K.nc > 0 || anyLimClamp;
  }
  GEN_T(2);
  // ---- outputs (coopContactOutputs) ----
  for (int r = ln; r < MAX_ROWS; r += 64) {
    const bool in = r < m;
    const double X = in ? Fn.X[r] : 0.0;
    double code = 0.0;
    if (in) code = R.lim[r] ? (Fn.cls[r] == RC_CLAMPING ? 3.0 : 0.0) : (Fn.cls[r] == RC_UPPER_BOUND ? (Fn.E[r] > 0 ? 2.0 : -2.0) : (double)Fn.cls[r]);
    svAt(saved, lay.x + r, B, b) = X;
    svAt(saved, lay.cls + r, B, b) = code;
    svAt(saved, lay.cfm + r, B, b) = in ? Fn.cfm[r] : 0.0;
    if (cacheOut) cacheOut[(int64_t)r * B + b] = (in && R.neg[r]) ? -X : X;
    // the velocity change the BACKWARD pass works with: the impulses of the clamping rows and, for a friction row on its bound, E times its
    // normal's (BackpropSnapshot.cpp:980-1066); joint-limit rows: never
    if (in) R.t1[r] = R.lim[r] ? 0.0 : (Fn.cls[r] == RC_CLAMPING ? X : (Fn.cls[r] == RC_UPPER_BOUND ? Fn.E[r] * Fn.X[R.fp[r]] : 0.0));
  }
  if (ln == 0 && cacheOut) cacheOut[(int64_t)MAX_ROWS * B + b] = (double)m;
  if (ln == 0) svAt(saved, lay.pflag, B, b) = pinvValid ? 1.0 : 0.0;
  w.sync();
  bool bad = false;
  for (int d = ln; d < n; d += 64) {
    double wd = 0.0, wb = 0.0;
    for (int r = 0; r < m; r++) {
      const double ms = dn[lay.massed + d * ldr + r];
      wd = fma(ms, Fn.X[r], wd); wb = fma(ms, R.t1[r], wb);
    }
    svAt(saved, lay.w + d, B, b) = wb;
    const double vNext = svAt(saved, lay.vpre + d, B, b) + wd;
    nv[(int64_t)d * B + b] = vNext;
    if (!__builtin_isfinite(vNext)) bad = true;
  }
  const bool nan = w.anyAll(bad);
  if (ln == 0 && status) status[b] |= (anyFail ? stAll : (0x2u | 0x100u)) | (nan ? 0x40u : 0u);
  GEN_T(3);
}

// ======================================================================================================================================
// backward, dense part: k_bwd_contact_a_coop with loops (the header of contact_backward.hip derives the quantities)
//   (Q x)_r   = (A xE)_r + cfm_r x_r      xE = x on clamping rows, E_u x_normal(u) on upper-bound rows  ("spread")
//   (Q^T y)_s = t_s + sum_{u in ub(s)} E_u t_u + cfm_s y_s,  t = A y                                     ("fold")
// ======================================================================================================================================
struct GenBwdA {
  double cfm[GR], x[GR], b[GR], mu[GR], E[GR], rest[GR];
  double t[GR], fbar[GR], muv[GR], fls[GR], tmp[GR], tmp2[GR];
  double al[3][GR], be[3][GR], beE[3][GR], muB[GR], fcE[GR];
  double g[MAX_DOF_CONTACT];
  int fp[GR];
  unsigned char clamp[GR], ub[GR], fric[GR], pad_[GR];
};

__global__ __launch_bounds__(64) void k_bwd_contact_a_gen(DevModel mdl, const DevContactModel* __restrict__ cm, int64_t B, double* __restrict__ saved,
                                                          SavedLayout lay, const double* __restrict__ gnext, double* __restrict__ lws,
                                                          double* __restrict__ gws) {
  __shared__ GenBwdA L;
  const GenWaveDev w;
  const int ln = w.lane();
  const int ldr = lay.ldr;              // leading dimension of the record's dense blocks and of the world's scratch matrices
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x;
  if (b >= mdl.b1) return;
  const int n = mdl.n;
  const double* gvn = gnext + (int64_t)n * B;
  double* dn = denseOf(saved, lay, B, b);
  const double* A = dn + lay.A;
  const double* P = dn + lay.pinv;
  const int m = 3 * (int)svAt(saved, lay.nc, B, b);
  const bool pflag = svAt(saved, lay.pflag, B, b) != 0.0;
  bool anyClamp = false, limNoCfm = false;
  int nuMine = 0;
  for (int r = ln; r < m; r += 64) {
    const double cv = svAt(saved, lay.cls + r, B, b);
    const int r0c = lay.contacts + (r / 3) * CR_SIZE;
    const int bxA = (int)svAt(saved, r0c + CR_BOXA, B, b), bxB = (int)svAt(saved, r0c + CR_BOXB, B, b);
    const double muA = crMuOf(cm, bxA), muB = crMuOf(cm, bxB);
    L.mu[r] = muA < muB ? muA : muB;
    L.fric[r] = (r % 3) != 0; L.fp[r] = r - (r % 3);
    L.clamp[r] = cv == 1.0; L.ub[r] = (cv == 2.0 || cv == -2.0);
    L.E[r] = L.ub[r] ? (cv > 0 ? L.mu[r] : -L.mu[r]) : 0.0;
    L.cfm[r] = svAt(saved, lay.cfm + r, B, b);
    L.x[r] = svAt(saved, lay.x + r, B, b);
    L.b[r] = svAt(saved, lay.b + r, B, b);
    L.rest[r] = svAt(saved, lay.rest + r / 3, B, b);
    anyClamp = anyClamp || L.clamp[r];
    nuMine += L.ub[r];
    if (cv == 3.0 && L.cfm[r] == 0.0) limNoCfm = true;
  }
  // contact adjoint active <=> some row is clamping (the same test k_bwd_recompute_coop makes for LB_FLAG)
  if (!w.anyAll(anyClamp)) return;
  const bool anyLimNoCfm = w.anyAll(limNoCfm);
  const int nu = (int)w.sumAll((double)nuMine);
  for (int d = ln; d < n; d += 64) L.g[d] = gvn[(int64_t)d * B + b];
  w.sync();
  auto fold = [&](const double* t, double* out) {     // normal rows collect E_u t_u of their contact's upper-bound rows
    for (int r = ln; r < m; r += 64) {
      double f = 0.0;
      if (nu > 0 && !L.fric[r] && r + 2 < m) f = (L.ub[r + 1] ? L.E[r + 1] * t[r + 1] : 0.0) + (L.ub[r + 2] ? L.E[r + 2] * t[r + 2] : 0.0);
      out[r] = f;
    }
    w.sync();
  };
  auto spread = [&](const double* x, double* out) {   // upper-bound rows ride on their normal row: E_u x_normal
    for (int r = ln; r < m; r += 64) out[r] = L.clamp[r] ? x[r] : (L.ub[r] ? L.E[r] * x[L.fp[r]] : 0.0);
    w.sync();
  };
  auto ax = [&](const double* x, double* out) {       // A x, A symmetric
    for (int r = ln; r < m; r += 64) out[r] = genFmaSeq(0, m, 0.0, [&](int j) { return A[(size_t)j * ldr + r]; }, [&](int j) { return x[j]; });      // (operands four steps ahead: gen_lcp_dev.hpp)
    w.sync();
  };
  auto pinvApply = [&](const double* x, double* out, bool trans) {
    for (int i = ln; i < m; i += 64) out[i] = genFmaSeq(0, m, 0.0, [&](int k) { return trans ? P[(size_t)k * ldr + i] : P[(size_t)i * ldr + k]; }, [&](int k) { return x[k]; });
    w.sync();
  };
  // fbar = Abar^T lambda1 = (M^-1 A_c)^T g: the saved impulse tests applied to g
  for (int r = ln; r < m; r += 64) L.t[r] = genFmaSeq(0, n, 0.0, [&](int d) { return dn[lay.massed + d * ldr + r]; }, [&](int d) { return L.g[d]; });
  w.sync();
  fold(L.t, L.tmp);
  for (int r = ln; r < m; r += 64) L.fbar[r] = L.clamp[r] ? L.t[r] + L.tmp[r] : 0.0;
  w.sync();
  if (!pflag) {   // cannot happen: the forward kernel always leaves Q^+ of the final classification in the record; make it loud
    for (int d = ln; d < n; d += 64) lws[(int64_t)(LB_GVP + d) * B + b] = __builtin_nan("");
    return;
  }
  // ---- was Q inverted precisely?  ||I - Q Q^+||_F^2 < 1e-18 on the clamping block (BackpropSnapshot.cpp:2964-2984, see k_bwd_contact_a_coop) ----
  bool precise;
  {
    double acc = 0.0;
    for (int e = ln; e < m * m; e += 64) {
      const int r = e / m, j = e - r * m;
      if (!L.clamp[r] || !L.clamp[j]) continue;
      // spread(Q^+)[k][j] = sc_k P[src_k][j]; the four terms of a trip fetch their operands side by side (genFmaSeq)
      double y = genFmaSeq(0, m, 0.0, [&](int k) { return A[(size_t)k * ldr + r]; },
                           [&](int k) { return L.clamp[k] ? P[(size_t)k * ldr + j] : (L.ub[k] ? L.E[k] * P[(size_t)L.fp[k] * ldr + j] : 0.0); });
      y += L.cfm[r] * P[(size_t)r * ldr + j];
      const double dlt = ((r == j) ? 1.0 : 0.0) - y;
      acc = fma(dlt, dlt, acc);
    }
    precise = w.sumAll(acc) < 1e-18;
    if (anyLimNoCfm) precise = false;
  }
  for (int r = ln; r < m; r += 64) L.tmp[r] = L.clamp[r] ? L.b[r] : 0.0;      // bcl
  w.sync();
  pinvApply(L.fbar, L.muv, true);           // mu = (Q^+)^T fbar
  pinvApply(L.tmp, L.fls, false);           // Q^+ b, the reference's least-squares f_c
  for (int r = ln; r < m; r += 64) { L.al[0][r] = -L.muv[r]; L.be[0][r] = L.fls[r]; }
  w.sync();
  spread(L.fls, L.tmp2);
  ax(L.tmp2, L.t);
  for (int r = ln; r < m; r += 64) L.al[1][r] = L.clamp[r] ? L.tmp[r] - (L.t[r] + L.cfm[r] * L.fls[r]) : 0.0;
  w.sync();
  pinvApply(L.muv, L.be[1], false);
  pinvApply(L.fls, L.al[2], true);
  ax(L.muv, L.t);
  fold(L.t, L.tmp2);
  for (int r = ln; r < m; r += 64) L.be[2][r] = L.clamp[r] ? L.fbar[r] - (L.t[r] + L.tmp2[r] + L.cfm[r] * L.muv[r]) : 0.0;
  w.sync();
  if (precise) { for (int r = ln; r < m; r += 64) { L.al[1][r] = 0.0; L.be[1][r] = 0.0; L.al[2][r] = 0.0; L.be[2][r] = 0.0; } w.sync(); }
  // bounce diagonals (CGGM.cpp:770, BackpropSnapshot.cpp:3099-3146)
  for (int r = ln; r < m; r += 64) L.muB[r] = ((r % 3) == 0) ? L.muv[r] * (1.0 + L.rest[r]) : L.muv[r];
  w.sync();
  for (int k = 0; k < 3; k++) spread(L.be[k], L.beE[k]);
  spread(L.x, L.fcE);
  // coefficient vectors for the DOF lanes
  for (int d = ln; d < n; d += 64) {
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    auto term = [&](int r, double ms, double aa) {
      for (int k = 0; k < 3; k++) { acc[k] = fma(L.clamp[r] ? L.al[k][r] : 0.0, ms, acc[k]); acc[3 + k] = fma(L.beE[k][r], ms, acc[3 + k]); }
      acc[6] = fma(L.clamp[r] ? L.muB[r] : 0.0, aa, acc[6]);
    };
    int r = 0;
    for (; r + 3 < m; r += 4) {      // (the record's entries of four rows first - every one an L2 round trip -, then the rows in order)
      const double m0 = dn[lay.massed + d * ldr + r], m1 = dn[lay.massed + d * ldr + r + 1], m2 = dn[lay.massed + d * ldr + r + 2], m3 = dn[lay.massed + d * ldr + r + 3];
      const double a0 = dn[lay.aall + d * ldr + r], a1 = dn[lay.aall + d * ldr + r + 1], a2 = dn[lay.aall + d * ldr + r + 2], a3 = dn[lay.aall + d * ldr + r + 3];
      term(r, m0, a0); term(r + 1, m1, a1); term(r + 2, m2, a2); term(r + 3, m3, a3);
    }
    for (; r < m; r++) term(r, dn[lay.massed + d * ldr + r], dn[lay.aall + d * ldr + r]);
    for (int k = 0; k < 3; k++) {
      lws[(int64_t)(LB_S + k * MAX_DOF_CONTACT + d) * B + b] = acc[k];
      lws[(int64_t)(LB_P + k * MAX_DOF_CONTACT + d) * B + b] = acc[3 + k];
    }
    lws[(int64_t)(LB_GVP + d) * B + b] = L.g[d] - acc[6];
  }
  // coefficients of z_row on the bases [lambda1, v_pre, p1, p2, p3, s1, s2, s3]
  for (int r = ln; r < MAX_ROWS; r += 64) {
    double cf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < m) {
      cf[0] = L.fcE[r]; cf[1] = L.clamp[r] ? -L.muB[r] : 0.0;
      for (int k = 0; k < 3; k++) { cf[2 + k] = L.clamp[r] ? L.al[k][r] : 0.0; cf[5 + k] = L.beE[k][r]; }
    }
    for (int k = 0; k < 8; k++) lws[(int64_t)(LB_COEF + r * 8 + k) * B + b] = cf[k];
  }
}

// ======================================================================================================================================
// backward, tree part: k_bwd_contact_b_coop with the LCP rows in tiles of 64 (the phases are described there)
//   lds doubles: FW[nb][9][6] D[nb][54] { tmp[54][64] | TF[nb][9][6] } TW[nb][12] contact bodies
// ======================================================================================================================================
template <bool CAPS>
__global__ __launch_bounds__(64) void k_bwd_contact_b_gen(DevModel mdl, const DevBody* __restrict__ bodies, const DevContactModel* __restrict__ cm, int64_t B,
                                                          double* __restrict__ saved, SavedLayout lay, const double* __restrict__ ws, double* __restrict__ lws) {
  extern __shared__ __attribute__((aligned(16))) double ldsB[];
  const GenWaveDev w;
  const int ln = w.lane();
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x;
  if (b >= mdl.b1) return;
  if (lws[(int64_t)LB_FLAG * B + b] == 0.0) return;
  const int nb = mdl.nb;
  double* FW = ldsB;
  double* D = FW + nb * 54;
  double* TF = D + nb * 54;      // written after the row phase: shares its storage with tmp
  double* tmp = TF;
  double* TWs = TF + (nb * 54 > 54 * 64 ? nb * 54 : 54 * 64);   // [nb][12] world transforms
  int* cbody = reinterpret_cast<int*>(TWs + nb * 12);           // [2][MAX_CONTACTS]
  Ctx c = makeCtx(mdl, bodies, nullptr, const_cast<double*>(ws), B, b, saved, &lay);
  LaneMem SV; SV.base = saved; SV.B = B; SV.b = b;
  const double* q = saved;
  auto ld6 = [](const double* base, int stride) -> V6 { double a[6]; for (int e = 0; e < 6; e++) a[e] = base[e * stride]; return fromArr(a); };
  auto st6 = [](double* base, int stride, V6 x) { double a[6]; toArr(x, a); for (int e = 0; e < 6; e++) base[e * stride] = a[e]; };
  const int m = 3 * (int)svAt(saved, lay.nc, B, b);
  const int nC = m / 3;
  const V3 worldOrigin = mk3(wsAt(c, 0, WS_TW + 9), wsAt(c, 0, WS_TW + 10), wsAt(c, 0, WS_TW + 11));
  for (int idx = ln; idx < nb * 12; idx += 64) {
    const int e = idx % 12;
    TWs[idx] = wsAt(c, idx / 12, WS_TW + e) - (e == 9 ? worldOrigin.x : (e == 10 ? worldOrigin.y : (e == 11 ? worldOrigin.z : 0.0)));
  }
  for (int ci = ln; ci < nC; ci += 64) {
    const int q0 = lay.contacts + ci * CR_SIZE;
    cbody[ci] = crBodyOf(cm, (int)svAt(saved, q0 + CR_BOXA, B, b));
    cbody[MAX_CONTACTS + ci] = crBodyOf(cm, (int)svAt(saved, q0 + CR_BOXB, B, b));
  }
  for (int idx = ln; idx < nb * 54; idx += 64) D[idx] = 0.0;
  w.sync();
  // ---- phase 1a: world twists of the nine joint-rate fields, then prefix sums down the tree (bodies are listed parents first) ----
  for (int item = ln; item < nb * 9; item += 64) {
    const int i = item / 9, f = item - 9 * i;
    const double* src; const int64_t stride = B;
    if (f == 0) src = lws + (int64_t)LB_LAM1 * B + b;
    else if (f == 1) src = saved + (int64_t)lay.vpre * B + b;
    else if (f <= 4) src = lws + (int64_t)(LB_P + (f - 2) * MAX_DOF_CONTACT) * B + b;
    else if (f <= 7) src = lws + (int64_t)(LB_S + (f - 5) * MAX_DOF_CONTACT) * B + b;
    else src = saved + (int64_t)lay.w * B + b;
    const DevBody& bd = bodies[i];
    V6 tw;
    if (bd.jtype == JT_FREE) {
      const int o = bd.dofOff;
      tw = AdT(cT(bd.Tcj), mk6(mk3(src[o * stride], src[(o + 1) * stride], src[(o + 2) * stride]),
                               mk3(src[(o + 3) * stride], src[(o + 4) * stride], src[(o + 5) * stride])));
    } else tw = src[bd.dofOff * stride] * cV6(bd.S);
    st6(FW + item * 6, 1, AdT(cT(TWs + 12 * i), tw));
  }
  w.sync();
  if (ln < 54) {
    for (int i = 1; i < nb; i++) {
      const int par = bodies[i].parent;
      if (par >= 0) FW[i * 54 + ln] += FW[par * 54 + ln];
    }
  }
  w.sync();
  // ---- phase 2: per-row constants, tile by tile (63 rows = 21 contacts: a contact's three rows never straddle two tiles), side A then side B ----
  constexpr int TSB = 63;
  for (int t0 = 0; t0 < m; t0 += TSB) {
    const int row = (ln < TSB) ? t0 + ln : m;
    double cf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool any = false;
    ContactRec CR;
    CR.bA = -1; CR.bB = -1; CR.type = 0;
    V6 Fw = zero6(), TA = zero6(), TB = zero6();
    RowTerms RT;
    RT.vertexTerm = RT.faceTerm = RT.edgeTermA = RT.edgeTermB = zero6();
    RT.commonAngular = mk3(0, 0, 0);
    if (row < m) {
      const int ci = row / 3, k = row % 3;
      for (int e = 0; e < 8; e++) { cf[e] = lws[(int64_t)(LB_COEF + row * 8 + e) * B + b]; any = any || cf[e] != 0.0; }
      if (any) {
        CR = loadContactRec<CAPS>(SV, lay, cm, ci);
        CR.p = CR.p - worldOrigin;
        if (CR.type >= CT_EDGE_EDGE) CR.eAP = CR.eAP - worldOrigin;
        if (CR.type == CT_EDGE_EDGE || CR.type == CT_SPHERE_SPHERE || CR.type >= CT_PIPE_SPHERE) CR.eBP = CR.eBP - worldOrigin;
        const TangentFrame TF_ = tangentFrameOf(CR.nrm);
        const V3 d = k == 0 ? CR.nrm : (k == 1 ? TF_.t1 : TF_.t2);
        Fw = mk6(cross(CR.p, d), d);
        auto twistOf = [&](int body) -> V6 {   // world twist of `body` under the joint rates z_row
          V6 z = zero6();
          if (body < 0) return z;
          for (int e = 0; e < 8; e++) z = z + cf[e] * ld6(FW + (body * 9 + e) * 6, 1);
          return z;
        };
        TA = twistOf(CR.bA); TB = twistOf(CR.bB);
        RT = contactRowTerms<CAPS>(CR, TF_, k, d, TA - TB);
      }
    }
    const bool aIsVertex = (CR.type == CT_VERTEX_FACE);
    const int cFirst = t0 / 3, cLast = (t0 + TSB < m ? t0 + TSB : m) / 3;
    for (int side = 0; side < 2; side++) {
      {
        const double sgn = side == 0 ? 1.0 : -1.0;
        const int start = side == 0 ? CR.bA : CR.bB;
        const bool vertexSide = (side == 0) == aIsVertex;
        V6 Cc = zero6();
        double sc = 0.0;
        if (any && start >= 0) {
          Cc = -sgn * dad(side == 0 ? TA : TB, Fw);
          if (CR.type == CT_VERTEX_FACE || CR.type == CT_FACE_VERTEX) Cc = Cc + (vertexSide ? RT.vertexTerm : RT.faceTerm);
          else if (CR.type >= CT_EDGE_EDGE) Cc = Cc + (side == 0 ? RT.edgeTermA : RT.edgeTermB);
          sc = sgn;
        }
        double c6[6], f6[6];
        toArr(Cc, c6); toArr(Fw, f6);
        for (int e = 0; e < 6; e++) tmp[e * 64 + ln] = c6[e];
        for (int e = 0; e < 8; e++) for (int x = 0; x < 6; x++) tmp[(6 + e * 6 + x) * 64 + ln] = sc * cf[e] * f6[x];
      }
      w.sync();
      if (ln < 54) {
        for (int ci = cFirst; ci < cLast; ci++) {
          const int st = cbody[side * MAX_CONTACTS + ci];
          const int l0 = 3 * ci - t0;
          if (st >= 0) D[st * 54 + ln] += (tmp[ln * 64 + l0] + tmp[ln * 64 + l0 + 1]) + tmp[ln * 64 + l0 + 2];
        }
      }
      w.sync();
    }
    if (cm->selfCollision) {
      {
        const bool both = any && CR.type == CT_EDGE_EDGE && CR.bA >= 0 && CR.bB >= 0;
        const V3 ca = both ? RT.commonAngular : mk3(0, 0, 0);
        tmp[0 * 64 + ln] = ca.x; tmp[1 * 64 + ln] = ca.y; tmp[2 * 64 + ln] = ca.z;
      }
      w.sync();
      if (ln < 3) {
        for (int ci = cFirst; ci < cLast; ci++) {
          const int bA = cbody[ci], bB = cbody[MAX_CONTACTS + ci];
          if (bA < 0 || bB < 0) continue;
          const uint64_t common = cm->ancestors[bA] & cm->ancestors[bB];
          if (common == 0ull) continue;
          const int lca = 63 - __builtin_clzll(common);
          const int l0 = 3 * ci - t0;
          D[lca * 54 + ln] += (tmp[ln * 64 + l0] + tmp[ln * 64 + l0 + 1]) + tmp[ln * 64 + l0 + 2];
        }
      }
      w.sync();
    }
  }
  // ---- phase 1b (after the rows: TF takes over tmp's storage): local wrenches of the nine fields, world frame ----
  for (int item = ln; item < nb * 9; item += 64) {
    const int i = item / 9;
    const T12 TW = cT(TWs + 12 * i);
    const V6 twB = AdInvT(TW, ld6(FW + item * 6, 1));
    st6(TF + item * 6, 1, dAdInvT(TW, mul(cS6(bodies[i].G), twB)));
  }
  w.sync();
  // ---- phase 3: subtree sums, leaf -> root (D and the transmitted wrenches) ----
  if (ln < 54) {
    for (int i = nb - 1; i >= 1; i--) {
      const int par = bodies[i].parent;
      if (par >= 0) { D[par * 54 + ln] += D[i * 54 + ln]; TF[par * 54 + ln] += TF[i * 54 + ln]; }
    }
  }
  w.sync();
  // ---- phase 4 ----
  for (int bl = ln; bl < nb; bl += 64) {
    const DevBody& bd = bodies[bl];
    const int i = (bd.jtype == JT_BALL || bd.jtype == JT_FREEC) ? bl - bd.ballComp : bl;
    const int par = bodies[i].parent;
    V6 xiW = ld6(D + i * 54, 1);
    if (par >= 0) {
      const double* FWp = FW + par * 54;
      for (int e = 0; e < 8; e++) xiW = xiW + dad(ld6(FWp + e * 6, 1), ld6(D + i * 54 + 6 + e * 6, 1));
      for (int k = 0; k < 4; k++) {
        const int ADJ = k == 0 ? 0 : 4 + k, ACC = k == 0 ? 8 : 1 + k;   // (lambda1, w), (s_k, p_k)
        xiW = xiW - dad(ld6(FWp + ADJ * 6, 1), ld6(TF + i * 54 + ACC * 6, 1)) - dad(ld6(FWp + ACC * 6, 1), ld6(TF + i * 54 + ADJ * 6, 1));
      }
    }
    double qb[6];
    applyHt(bd, q, B, b, dAdT(cT(TWs + 12 * i), xiW), qb);
    for (int k = 0; k < bd.ndof; k++) lws[(int64_t)(LB_QX + bd.dofOff + k) * B + b] = qb[k];
  }
}

// ======================================================================================================================================
// the reference's bounce approximation of the position Jacobians (k_bwd_bounce of coop_kernels.hip): X = I - sum_i c_i a_i a_i^T,
// c = G^+ (e + |a|^2), G_ij = (a_i . a_j)^2 over the bouncing constraints (clamping normal rows whose contact bounced)
// ======================================================================================================================================
__global__ __launch_bounds__(64) void k_bwd_bounce_gen(DevModel mdl, const DevBody* __restrict__ bodies, int64_t B, const double* __restrict__ savedC,
                                                       SavedLayout lay, const double* __restrict__ gnext, double* __restrict__ lws, double* __restrict__ gws) {
  double* saved = const_cast<double*>(savedC);
  extern __shared__ __attribute__((aligned(16))) double ldsRows[];
  GenRows R;
  genRowsCarve(R, ldsRows, genRowsCap(MAX_CONTACTS));
  __shared__ double yq[MAX_DOF_CONTACT], yv[MAX_DOF_CONTACT], tq[MAX_CONTACTS], tv[MAX_CONTACTS], rhs[MAX_CONTACTS], cRow[MAX_CONTACTS];
  __shared__ int brow[MAX_CONTACTS];
  const GenWaveDev w;
  const int ln = w.lane();
  const int ldr = lay.ldr;              // leading dimension of the record's dense blocks and of the world's scratch matrices
  const int64_t b = mdl.b0 + (int64_t)blockIdx.x;
  if (b >= mdl.b1) return;
  const int n = mdl.n;
  const int m = 3 * (int)svAt(saved, lay.nc, B, b);
  if (ln == 0) {
    int nbn = 0;
    for (int r = 0; r < m; r += 3)
      if (svAt(saved, lay.cls + r, B, b) == 1.0 && svAt(saved, lay.rest + r / 3, B, b) > 0.0) brow[nbn++] = r;
    R.iscal[0] = nbn;
  }
  R.ld = ldr;                                           // (genPinv below works in the world's scratch matrices)
  w.sync();
  const int nbn = R.iscal[0];
  if (nbn == 0) {                                       // nothing bounced in this world: X = I
    for (int d = ln; d < n; d += 64) lws[(int64_t)(LB_VX + d) * B + b] = 0.0;
    return;
  }
  const double* dn = denseOf(saved, lay, B, b);
  const double* q = saved;
  const double* v = saved + (int64_t)n * B;
  // ---- y_q = posPos^T gq', y_v = velPos^T gq' (lane = body; the exp / log VJPs like in the reverse sweep) ----
  for (int bi = ln; bi < mdl.nb; bi += 64) {
    const DevBody& bd = bodies[bi];
    const int o = bd.dofOff;
    if (bd.jtype == JT_FREEC) {
      const int d0 = o - bd.ballComp, cmp = bd.ballComp;
      auto at3 = [&](const double* x, int k0) { return mk3(x[(int64_t)(d0 + k0) * B + b], x[(int64_t)(d0 + k0 + 1) * B + b], x[(int64_t)(d0 + k0 + 2) * B + b]); };
      double posT[6], velT[6];
      se3IntegrationVjp(at3(q, 0), at3(v, 0), at3(v, 3), mdl.dt, at3(gnext, 0), at3(gnext, 3), posT, velT);
      for (int k = 0; k < 6; k++) if (k == cmp) { yq[o] = posT[k]; yv[o] = velT[k]; }
    } else if (bd.jtype == JT_BALL) {
      const int d0 = o - bd.ballComp, cmp = bd.ballComp;
      V3 posr, velw;
      so3IntegrationVjp(mk3(q[(int64_t)(d0 + 0) * B + b], q[(int64_t)(d0 + 1) * B + b], q[(int64_t)(d0 + 2) * B + b]),
                        mk3(v[(int64_t)(d0 + 0) * B + b], v[(int64_t)(d0 + 1) * B + b], v[(int64_t)(d0 + 2) * B + b]), mdl.dt,
                        mk3(gnext[(int64_t)(d0 + 0) * B + b], gnext[(int64_t)(d0 + 1) * B + b], gnext[(int64_t)(d0 + 2) * B + b]), posr, velw);
      yq[o] = cmp == 0 ? posr.x : (cmp == 1 ? posr.y : posr.z);
      yv[o] = cmp == 0 ? velw.x : (cmp == 1 ? velw.y : velw.z);
    } else if (bd.jtype != JT_FREE) {
      const double g = gnext[(int64_t)o * B + b];
      yq[o] = g; yv[o] = mdl.dt * g;
    } else {
      auto at3 = [&](const double* x, int k0) { return mk3(x[(int64_t)(o + k0) * B + b], x[(int64_t)(o + k0 + 1) * B + b], x[(int64_t)(o + k0 + 2) * B + b]); };
      double posT[6], velT[6];
      se3IntegrationVjp(at3(q, 0), at3(v, 0), at3(v, 3), mdl.dt, at3(gnext, 0), at3(gnext, 3), posT, velT);
      for (int k = 0; k < 6; k++) { yq[o + k] = posT[k]; yv[o + k] = velT[k]; }
    }
  }
  w.sync();
  // ---- G (nbn x nbn, compact) -> scratch, t = a_i . y, rhs = e_i + |a_i|^2 ----
  const GenScratch S = genScratchOf(gws, b, gws + (size_t)b * genScratchDoubles(ldr) + (size_t)3 * ldr * ldr, ldr);   // (P in the last scratch matrix: the record's Q^+ stays)
  for (int e = ln; e < nbn * nbn; e += 64) {
    const int i = e / nbn, k = e - i * nbn;
    double dotik = 0.0;
    for (int d = 0; d < n; d++) dotik = fma(dn[lay.aall + d * ldr + brow[i]], dn[lay.aall + d * ldr + brow[k]], dotik);
    S.mat[0][(size_t)i * ldr + k] = dotik * dotik;
  }
  for (int i = ln; i < nbn; i += 64) {
    double a2 = 0.0, sq = 0.0, sv = 0.0;
    for (int d = 0; d < n; d++) { const double a = dn[lay.aall + d * ldr + brow[i]]; sq = fma(a, yq[d], sq); sv = fma(a, yv[d], sv); a2 = fma(a, a, a2); }
    tq[i] = sq; tv[i] = sv; rhs[i] = svAt(saved, lay.rest + brow[i] / 3, B, b) + a2;
  }
  w.sync();
  genPinv(w, R, S.mat[0], S.mat[1], S.mat[2], S.mat[3], nbn, nbn);
  for (int i = ln; i < nbn; i += 64) { double s = 0.0; for (int k = 0; k < nbn; k++) s = fma(S.mat[3][(size_t)i * ldr + k], rhs[k], s); cRow[i] = s; }
  w.sync();
  // ---- (X - I) y = -A_b (c o t)   (lane = DOF) ----
  for (int d = ln; d < n; d += 64) {
    double dq = 0.0, dv = 0.0;
    for (int i = 0; i < nbn; i++) {
      const double a = dn[lay.aall + d * ldr + brow[i]];
      dq = fma(-a, cRow[i] * tq[i], dq); dv = fma(-a, cRow[i] * tv[i], dv);
    }
    lws[(int64_t)(LB_QX + d) * B + b] += dq;
    lws[(int64_t)(LB_VX + d) * B + b] = dv;
  }
}

// Self-test of the general Dantzig driver (nbl_selftest_lcp_dantzig with n > 48): one wavefront per problem of a batch of n-row boxed LCPs
// with explicit bounds, exactly the code k_contact_solve_gen runs in its stage 1.  Problems are dense [count][n * n] / [count][n].
__global__ __launch_bounds__(64) void k_selftest_dantzig_gen(int count, int n, const double* __restrict__ A, const double* __restrict__ b,
                                                            const double* __restrict__ lo, const double* __restrict__ hi, const int32_t* __restrict__ findex,
                                                            double* __restrict__ x, int32_t* __restrict__ rc, double* __restrict__ gws) {
  const GenWaveDev w;
  const int ln = w.lane();
  const int64_t pb = blockIdx.x;
  if (pb >= count) return;
  const GenScratch S = genScratchOf(gws, pb, gws + (size_t)pb * genScratchDoubles(GR) + (size_t)3 * GR * GR, GR);   // (self-test: leading dimension = the cap)
  GenProblem P; GenDantzigMem D;
  genCarve(S, P, D, n);
  for (int j = ln; j < n; j += 64) {
    for (int i = 0; i < n; i++) P.A[(size_t)i * GR + j] = A[(pb * n + i) * n + j];
    P.b[j] = b[pb * n + j]; P.lo[j] = lo[pb * n + j]; P.hi[j] = hi[pb * n + j]; P.findex[j] = findex[pb * n + j]; P.x[j] = 0.0;
  }
  w.sync();
  __shared__ double sb[4 * GR];      // the driver's four shared vectors (the step kernel lends it GenRows::t0 .. t3)
  const int r = genDantzigPar(w, D, n, P.x, sb, sb + GR, sb + 2 * GR, sb + 3 * GR, S.mat[0]);
  if (ln == 0) rc[pb] = r;
  for (int i = ln; i < n; i += 64) x[pb * n + i] = r == 1 ? P.x[i] : 0.0;
}

// Self-test of the general solver cascade (nbl_selftest_lcp_cascade): one wavefront per problem, the rows set up like k_contact_solve_gen
// sets them up (three per contact, mu per contact, optionally a mask of the rows of one constrained group), then genStage0 / genCascade.
__global__ __launch_bounds__(64) void k_selftest_cascade_gen(int count, int m, const double* __restrict__ A, const double* __restrict__ b,
                                                            const double* __restrict__ mu, int haveCache, const double* __restrict__ xcache,
                                                            const uint8_t* __restrict__ on, double fallbackCfm, double* __restrict__ x,
                                                            int32_t* __restrict__ cls, uint32_t* __restrict__ st, double* __restrict__ cfmOut,
                                                            double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double ldsRows[];
  GenRows R;
  genRowsCarve(R, ldsRows, GR);
  const GenWaveDev w;
  const int ln = w.lane();
  const int64_t pb = blockIdx.x;
  if (pb >= count) return;
  // per problem: the scratch of a world with the cap as leading dimension (five matrices + 16 vectors): M, G, T, then the block that serves
  // as Q^+ AND - never at the same time - as the cascade's problem matrix, the vectors, and in the fifth block the problem handed in
  double* base = gws + (size_t)pb * genScratchDoubles(GR);
  const GenScratch S = genScratchOf(gws, pb, base + (size_t)3 * GR * GR, GR);
  double* Ap = base + (size_t)4 * GR * GR + 16 * GR;
  for (int j = ln; j < m; j += 64) for (int i = 0; i < m; i++) Ap[(size_t)i * GR + j] = A[((size_t)pb * m + i) * m + j];
  R.m = m; R.ld = GR;
  w.sync();
  for (int r = ln; r < m; r += 64) {
    double mr = mu[(size_t)pb * (m / 3) + r / 3];
    if (!(mr > 1e-3)) mr = 0.0;
    R.mu[r] = mr; R.Bv[r] = b[(size_t)pb * m + r];
    R.fric[r] = (r % 3) != 0; R.fp[r] = r - (r % 3);
    R.lim[r] = 0; R.neg[r] = 0; R.rowOn[r] = 1;
    R.on[r] = on ? on[(size_t)pb * m + r] : 1;
    double cn = 0.0;
    if (R.on[r]) for (int i = 0; i < m; i++) { const double a = Ap[(size_t)i * GR + r]; cn = fma(a, a, cn); }
    R.colNorm[r] = cn;
    R.X[r] = (haveCache && R.on[r]) ? xcache[(size_t)pb * m + r] : 0.0;
  }
  w.sync();
  bool pinvValid = false;
  GenClasses K;
  double cfmG = 0.0;
  uint32_t stG = 0x2u | 0x100u;
  if (!genStage0(w, Ap, GR, R, S, haveCache != 0, pinvValid, K)) genCascade(w, Ap, GR, R, S, fallbackCfm, cfmG, stG, pinvValid, K);
  for (int r = ln; r < m; r += 64) { x[(size_t)pb * m + r] = R.X[r]; cls[(size_t)pb * m + r] = R.on[r] ? R.cls[r] : 0; }
  if (ln == 0) { st[pb] = stG; cfmOut[pb] = cfmG; }
}

}  // namespace NBL_NS
