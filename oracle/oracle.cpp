// oracle/oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement ("oracle") of the reference hot path
//   neural::forwardPass -> World::step            dart/neural/NeuralUtils.cpp:26-66, dart/simulation/World.cpp:221-333
//   BackpropSnapshot::backpropState               dart/neural/BackpropSnapshot.cpp:121-194, 382-479
// one world at a time, scalar fp64, dense n x n Jacobians exactly like the reference forms them.
//
// PARITY PINNING: the upstream library cannot be built in this image (Eigen, libccd, assimp, ...
// are absent), and its tests hold no stored expected outputs for the step or its gradients.  The
// oracle is therefore pinned by the reference's own *property* tests restated in
// tests/test_oracle_props.py (finite-difference agreement of every Jacobian, VJP == J^T g,
// multi-step backprop vs brute force, M*Minv = I) and, for the LCP stage, by the literal fixtures of
// unittests/unit/test_LCPUtils.cpp and by the vendored ODE Dantzig solver compiled from the reference
// sources (oracle/_ref), by the reference's own PgsBoxedLcpSolver::solve compiled the same way (bit for bit); for the narrow phase by the reference's own dBoxBox / box-sphere / sphere-sphere functions compiled from
// DARTCollide.cpp (oracle/ref_build.py, tests/test_oracle_contact.py: bit for bit on random pairs).  Where neither exists
// DESIGN.md says "parity unpinned".
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <thread>

// ---- per-thread bump arena for the timed batch path ---------------------------------------------------------------------
// The restatement allocates dozens of small std::vector / MatX temporaries per world-step; with one cloned world per host
// thread (the reference's own concurrency model, MultiShot.cpp:66-70) on a 256-thread box the allocator, not the arithmetic,
// bounded the contact path (11x speed-up on 256 threads against 49x without contacts).  Inside nbo_step_batch every worker
// thread serves `operator new` from a private arena with per-size free lists (no lock, no system call, blocks reused while
// they are still in cache); everything else (and every other thread) falls through to malloc.  The library is linked with -Bsymbolic so that only ITS allocations come here.
namespace {
// size-class free lists on top of a bump region: a freed block is the next one handed out for its class, so the temporaries of
// a world-step keep reusing the same few hundred kB of (cache-resident) memory instead of streaming through megabytes
struct Arena {
  static constexpr int GRAIN = 64, NCLS = 1024;   // classes of 64 B up to 64 kB; larger requests go to malloc
  char* base = nullptr;
  size_t cap = 0, off = 0;
  void* freeList[NCLS + 1];
  bool active = false;
};
thread_local Arena* tlArena = nullptr;
inline void* arenaAlloc(size_t n) {
  Arena* a = tlArena;
  if (a && a->active) {
    const size_t need = n + 16;                    // 16-byte header: the block's class
    const size_t cls = (need + Arena::GRAIN - 1) / Arena::GRAIN;
    if (cls <= (size_t)Arena::NCLS) {
      char* blk = (char*)a->freeList[cls];
      if (blk) a->freeList[cls] = *(void**)(blk + 16);
      else if (a->off + cls * Arena::GRAIN <= a->cap) { blk = a->base + a->off; a->off += cls * Arena::GRAIN; }
      if (blk) { *(size_t*)blk = cls; return blk + 16; }
    }
  }
  void* p = std::malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
inline void arenaFree(void* p) noexcept {
  if (!p) return;
  Arena* a = tlArena;
  if (a && p >= (void*)a->base && p < (void*)(a->base + a->cap)) {
    char* blk = (char*)p - 16;
    const size_t cls = *(size_t*)blk;
    *(void**)(blk + 16) = a->freeList[cls];
    a->freeList[cls] = blk;
    return;
  }
  std::free(p);
}
}  // namespace
void* operator new(size_t n) { return arenaAlloc(n); }
void* operator new[](size_t n) { return arenaAlloc(n); }
void operator delete(void* p) noexcept { arenaFree(p); }
void operator delete[](void* p) noexcept { arenaFree(p); }
void operator delete(void* p, size_t) noexcept { arenaFree(p); }
void operator delete[](void* p, size_t) noexcept { arenaFree(p); }

#include "contact.hpp"
#include "dynamics.hpp"

using namespace nbo;

namespace {

struct Snapshot {  // what BackpropSnapshot captures (BackpropSnapshot.cpp:33-118)
  VecX q, v, tau, vPre;
  ContactResult contact;
  bool valid = false;
};

struct Oracle {
  Model model;
  Snapshot snap;
  VecX lcpCache;  // BoxedLcpConstraintSolver::mX, persists across steps (:176-187)
};

// World::step with gradients enabled.
void stepWorld(Oracle& o, const s_t* q, const s_t* v, const s_t* tau, s_t* qNext, s_t* vNext, uint32_t* status) {
  const Model& m = o.model;
  std::vector<Kin> kin;
  std::vector<Art> art;
  kinematics(m, q, v, kin);
  articulatedInertias(m, kin, art);
  VecX qdd(m.n, 0.0), vPre(m.n, 0.0);
  // World.cpp:226-233: computeForwardDynamics(); integrateVelocities(dt)
  forwardDynamics(m, kin, art, q, v, tau, qdd.data());
  for (int i = 0; i < m.n; i++) vPre[i] = v[i] + m.dt * qdd[i];  // GenericJoint.hpp:1410-1414

  Snapshot& s = o.snap;
  s.q.assign(q, q + m.n);
  s.v.assign(v, v + m.n);
  s.tau.assign(tau, tau + m.n);
  s.vPre = vPre;
  s.valid = true;

  // World.cpp:247 -> ConstraintSolver::solve() -> integrateVelocitiesFromImpulses()
  VecX vOut = vPre;
  uint32_t st = 0;
  solveContacts(m, kin, art, q, vPre.data(), o.lcpCache, s.contact, vOut.data(), &st);
  for (int i = 0; i < m.n; i++) vNext[i] = vOut[i];
  // World.cpp:250,307-333: positions integrate with the *initial* velocity (mParallelVelocityAndPositionUpdates)
  integratePositions(m, q, v, m.dt, qNext);
  if (status) *status = st;
}

// The five dense step Jacobians of the last step (BackpropSnapshot::getPosPosJacobian :1263-1335, getVelPosJacobian :1338-1400,
// getPosVelJacobian :762-821, getVelVelJacobian :643-759, getControlForceVelJacobian :482-574; §3.3 / Appendix A.6-A.7).
// Naming as in the reference: "XY" = d(Y at t+1) / d(X at t).
void stepJacobians(Oracle& o, MatX& posPos, MatX& velPos, MatX& posVel, MatX& velVel, MatX& forceVel) {
  const Model& m = o.model;
  const Snapshot& s = o.snap;
  const int n = m.n;
  const s_t dt = m.dt;
  std::vector<Kin> kin;
  std::vector<Art> art;
  kinematics(m, s.q.data(), s.v.data(), kin);
  articulatedInertias(m, kin, art);
  MatX Minv = invMassMatrix(m, kin, art);
  VecX zero(n, 0.0), C(n, 0.0);
  // World::getCoriolisAndGravityAndExternalForces (World.cpp:1958-1971)
  inverseDynamics(m, kin, s.v.data(), zero.data(), true, true, C.data());
  MatX dCdq = jacobianOfC(m, kin, s.q.data(), s.v.data(), false);
  MatX dCdv = jacobianOfC(m, kin, s.q.data(), s.v.data(), true);

  posJacobians(m, s.q.data(), s.v.data(), dt, posPos, velPos);
  if (s.contact.m > 0 && !s.contact.restCoeff.empty()) {   // BackpropSnapshot.cpp:1304-1305, 1372-1373
    MatX X = bounceApproximationJacobian(m, s.contact);
    posPos = matmul(posPos, X);
    velPos = matmul(velPos, X);
  }

  forceVel = MatX(n, n); velVel = MatX(n, n); posVel = MatX(n, n);
  const ContactResult& cr = s.contact;
  if (cr.numClamping == 0) {
    // no clamping constraints: BackpropSnapshot.cpp:521-524, 686-689, and getVelJacobianWrt with empty A_c
    VecX r(n);
    for (int i = 0; i < n; i++)
      r[i] = dt * (s.tau[i] - C[i] - m.damping[i] * s.v[i] - m.spring[i] * (s.q[i] - m.rest[i] + dt * s.v[i]));
    VecX w = matvec(Minv, r);
    MatX dMw = jacobianOfMx(m, kin, s.q.data(), w.data());
    // getJacobianOfMinv = -Minv * d(M w)/dq  (Skeleton.cpp:2072-2077)
    MatX dM = matmul(Minv, dMw);
    MatX MinvdCdq = matmul(Minv, dCdq), MinvdCdv = matmul(Minv, dCdv);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        forceVel(i, j) = dt * Minv(i, j);
        velVel(i, j) = (i == j ? 1.0 : 0.0) - dt * Minv(i, j) * m.damping[j] - dt * dt * Minv(i, j) * m.spring[j] -
                       dt * MinvdCdv(i, j);
        posVel(i, j) = -dM(i, j) - dt * MinvdCdq(i, j) - dt * Minv(i, j) * m.spring[j];
      }
  } else {
    contactJacobians(m, kin, art, s.q.data(), s.v.data(), s.tau.data(), s.vPre, cr, Minv, C, dCdq, dCdv, forceVel,
                     velVel, posVel);
  }

}

// BackpropSnapshot::backprop (:121-194): the loss gradient through the dense Jacobians, then clipLossGradientsToBounds.
void backpropWorld(Oracle& o, const s_t* gqNext, const s_t* gvNext, s_t* gq, s_t* gv, s_t* gtau) {
  const Model& m = o.model;
  const Snapshot& s = o.snap;
  const int n = m.n;
  MatX posPos, velPos, posVel, velVel, forceVel;
  stepJacobians(o, posPos, velPos, posVel, velVel, forceVel);
  VecX gqn(gqNext, gqNext + n), gvn(gvNext, gvNext + n);
  VecX a = matTvec(posPos, gqn), b = matTvec(posVel, gvn), c = matTvec(velPos, gqn), d = matTvec(velVel, gvn),
       e = matTvec(forceVel, gvn);
  for (int i = 0; i < n; i++) {
    gq[i] = a[i] + b[i];
    gv[i] = c[i] + d[i];
    gtau[i] = e[i];
  }
  // clipLossGradientsToBounds (BackpropSnapshot.cpp:425-479)
  for (int i = 0; i < n; i++) {
    if (s.q[i] == m.posLo[i] && gq[i] > 0) gq[i] = 0;
    if (s.q[i] == m.posHi[i] && gq[i] < 0) gq[i] = 0;
    if (s.v[i] == m.velLo[i] && gv[i] > 0) gv[i] = 0;
    if (s.v[i] == m.velHi[i] && gv[i] < 0) gv[i] = 0;
    if (s.tau[i] == m.forceLo[i] && gtau[i] > 0) gtau[i] = 0;
    if (s.tau[i] == m.forceHi[i] && gtau[i] < 0) gtau[i] = 0;
  }
}

}  // namespace

extern "C" {

void* nbo_create(const nbl_model_desc* d) {
  Oracle* o = new Oracle();
  o->model = buildModel(d);
  return o;
}
void nbo_destroy(void* h) { delete (Oracle*)h; }
int nbo_num_dofs(void* h) { return ((Oracle*)h)->model.n; }

// state = [q; v] (World::setState), action through the action map (World::setAction)
static void splitAction(const Model& m, const double* action, VecX& tau) {
  tau.assign(m.n, 0.0);
  for (size_t i = 0; i < m.actionMap.size(); i++) tau[m.actionMap[i]] = action[i];
}

// lcp cache handling: cacheIn may be NULL => keep the handle's persistent cache (reference behaviour);
// cacheLen < 0 => reset to "wrong size" (forces guessSolution).
void nbo_set_lcp_cache(void* h, const double* x, int len) {
  Oracle* o = (Oracle*)h;
  if (len <= 0) o->lcpCache.clear();
  else o->lcpCache.assign(x, x + len);
}
// test instrument (Model::lcpNoiseUlps): ulps = 0 switches it off
// absolute: 0 relative noise, 1 absolute noise, 2 no noise but A recomputed as J M^-1 J^T (Model::lcpAlternateA), 3 noise in units of
// the rounding-error bound of every entry (Model::lcpNoiseBound)
void nbo_set_lcp_noise(void* h, int ulps, uint64_t seed, int absolute) {
  Oracle* o = (Oracle*)h;
  o->model.lcpAlternateA = absolute == 2;
  o->model.lcpNoiseBound = absolute == 3 ? ulps : 0;
  if (absolute >= 2) ulps = 0;
  o->model.lcpNoiseUlps = ulps; o->model.lcpNoiseSeed = seed; o->model.lcpNoiseSample = 0; o->model.lcpNoiseAbsolute = absolute != 0;
}
// LCP cache in the device's interchange format (Model::lcpCacheSlots)
// test instrument (Model::pinvNoiseUlps): ulps = 0 switches it off
void nbo_set_pinv_noise(void* h, int ulps, uint64_t seed) {
  Oracle* o = (Oracle*)h;
  o->model.pinvNoiseUlps = ulps; o->model.pinvNoiseSample = 0; if (ulps > 0) o->model.lcpNoiseSeed = seed;
}
void nbo_set_lcp_cache_slots(void* h, int on) { ((Oracle*)h)->model.lcpCacheSlots = on != 0; }
// test instrument, not the reference's behaviour (dynamics.hpp::posJacobiansExact): exact position-integration Jacobians of free / ball joints
// (0 off, 1 in extended precision, 2 the same formulas in doubles)
void nbo_set_exact_position_jacobians(void* h, int on) { ((Oracle*)h)->model.exactPosJacobians = on; }
// test instrument (Model::lcpForced): len <= 0 switches it off
void nbo_set_lcp_forced(void* h, const double* x, int len, int cfmStage) {
  Oracle* o = (Oracle*)h;
  o->model.lcpForcedCfm = cfmStage != 0;
  if (len <= 0) o->model.lcpForced.clear();
  else o->model.lcpForced.assign(x, x + len);
}
int nbo_get_lcp_cache(void* h, double* x, int cap) {
  Oracle* o = (Oracle*)h;
  int len = (int)o->lcpCache.size();
  for (int i = 0; i < len && i < cap; i++) x[i] = o->lcpCache[i];
  return len;
}

int nbo_step(void* h, const double* state, const double* action, double* nextState, uint32_t* status) {
  Oracle* o = (Oracle*)h;
  const Model& m = o->model;
  VecX tau;
  splitAction(m, action, tau);
  stepWorld(*o, state, state + m.n, tau.data(), nextState, nextState + m.n, status);
  return 0;
}

int nbo_backprop(void* h, const double* gradNext, double* gradState, double* gradAction) {
  Oracle* o = (Oracle*)h;
  const Model& m = o->model;
  if (!o->snap.valid) return -1;
  VecX gtau(m.n, 0.0);
  backpropWorld(*o, gradNext, gradNext + m.n, gradState, gradState + m.n, gtau.data());
  for (size_t i = 0; i < m.actionMap.size(); i++) gradAction[i] = gtau[m.actionMap[i]];
  return 0;
}

// Batched convenience used by the parity tests and by bench.py's cpu_baseline leg: B independent
// worlds, world-major arrays [B][2n], [B][k]; each world starts from `cold` LCP cache unless
// lcpIn/lcpLen given ([B][m] + per-world lengths).  `threads` host threads, one cloned world per thread
// (the reference's own model: MultiShot.cpp:66-70).
// World::getStateJacobian / getActionJacobian of the LAST step (World.cpp:2210-2243): out row-major [2n][2n] / [2n][k]
// (the action Jacobian takes the columns of forceVel of the DOFs in the action space, actionMap[k]).
int nbo_state_jacobian(void* h, double* out) {
  Oracle& o = *(Oracle*)h;
  if (!o.snap.valid) return -1;
  const int n = o.model.n;
  MatX posPos, velPos, posVel, velVel, forceVel;
  stepJacobians(o, posPos, velPos, posVel, velVel, forceVel);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      out[(size_t)i * 2 * n + j] = posPos(i, j);
      out[(size_t)i * 2 * n + n + j] = velPos(i, j);
      out[(size_t)(n + i) * 2 * n + j] = posVel(i, j);
      out[(size_t)(n + i) * 2 * n + n + j] = velVel(i, j);
    }
  return 0;
}
int nbo_action_jacobian(void* h, const int32_t* actionMap, int k, double* out) {
  Oracle& o = *(Oracle*)h;
  if (!o.snap.valid) return -1;
  const int n = o.model.n;
  MatX posPos, velPos, posVel, velVel, forceVel;
  stepJacobians(o, posPos, velPos, posVel, velVel, forceVel);
  for (int i = 0; i < 2 * n * k; i++) out[i] = 0.0;
  for (int a = 0; a < k; a++)
    for (int i = 0; i < n; i++) out[(size_t)(n + i) * k + a] = forceVel(i, actionMap[a]);
  return 0;
}

int nbo_step_batch(void* h, int64_t B, const double* state, const double* action, const double* gradNext,
                   double* nextState, double* gradState, double* gradAction, uint32_t* status, int threads,
                   const double* lcpIn, const int32_t* lcpLenIn, double* lcpOut, int32_t* lcpLenOut, int lcpStride) {
  Oracle* base = (Oracle*)h;
  const Model& m = base->model;
  const int n = m.n, k = (int)m.actionMap.size();
  if (threads < 1) threads = 1;
  const char* arenaEnv = std::getenv("NBO_ARENA");
  const bool useArena = !(arenaEnv && arenaEnv[0] == '0');
  auto work = [&](int t) {
    Arena arena;
    if (useArena) {
      arena.cap = (size_t)64 << 20;
      arena.base = (char*)std::malloc(arena.cap);
      if (!arena.base) arena.cap = 0;
      for (auto& f : arena.freeList) f = nullptr;
      arena.active = arena.cap > 0;
      tlArena = &arena;
    }
    {
      Oracle o;                                   // everything it allocates lives (and dies) inside the arena's lifetime
      o.model = m;
      // contiguous chunk of worlds per thread (one cloned world per thread stepping its share)
      const int64_t per = (B + threads - 1) / threads, b0 = t * per, b1 = std::min<int64_t>(B, b0 + per);
      o.model.lcpNoiseSample = (uint64_t)b0; o.model.pinvNoiseSample = (uint64_t)b0;
      for (int64_t b = b0; b < b1; b++) {
        if (lcpIn && lcpLenIn && lcpLenIn[b] > 0) o.lcpCache.assign(lcpIn + b * lcpStride, lcpIn + b * lcpStride + lcpLenIn[b]);
        else o.lcpCache.clear();
        VecX tau;
        splitAction(m, action + b * k, tau);
        uint32_t st = 0;
        stepWorld(o, state + b * 2 * n, state + b * 2 * n + n, tau.data(), nextState + b * 2 * n,
                  nextState + b * 2 * n + n, &st);
        if (status) status[b] = st;
        if (lcpOut && lcpLenOut) {
          lcpLenOut[b] = (int32_t)o.lcpCache.size();
          for (size_t i = 0; i < o.lcpCache.size() && (int)i < lcpStride; i++) lcpOut[b * lcpStride + i] = o.lcpCache[i];
        }
        if (gradNext && gradState) {
          VecX gtau(n, 0.0);
          backpropWorld(o, gradNext + b * 2 * n, gradNext + b * 2 * n + n, gradState + b * 2 * n,
                        gradState + b * 2 * n + n, gtau.data());
          if (gradAction)
            for (int i = 0; i < k; i++) gradAction[b * k + i] = gtau[m.actionMap[i]];
        }
      }
    }
    if (std::getenv("NBO_ARENA_STATS") && t == 0) std::fprintf(stderr, "[oracle] arena high-water mark: %zu bytes\n", arena.off);
    arena.active = false;
    tlArena = nullptr;
    if (arena.base) std::free(arena.base);
  };
  if (threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  return 0;
}

// ---- introspection for the property tests (each returns row-major n x n or n) ----
static void ctx(Oracle* o, const double* q, const double* v, std::vector<Kin>& kin, std::vector<Art>& art) {
  kinematics(o->model, q, v, kin);
  articulatedInertias(o->model, kin, art);
}
void nbo_mass_matrix(void* h, const double* q, double* out) {
  Oracle* o = (Oracle*)h;
  std::vector<Kin> kin; std::vector<Art> art;
  VecX z(o->model.n, 0.0);
  ctx(o, q, z.data(), kin, art);
  MatX M = massMatrix(o->model, kin);
  std::copy(M.d.begin(), M.d.end(), out);
}
void nbo_coriolis_gravity(void* h, const double* q, const double* v, double* out) {
  Oracle* o = (Oracle*)h;
  std::vector<Kin> kin; std::vector<Art> art;
  ctx(o, q, v, kin, art);
  VecX z(o->model.n, 0.0);
  inverseDynamics(o->model, kin, v, z.data(), true, true, out);
}
void nbo_forward_dynamics(void* h, const double* q, const double* v, const double* tau, double* qdd) {
  Oracle* o = (Oracle*)h;
  std::vector<Kin> kin; std::vector<Art> art;
  ctx(o, q, v, kin, art);
  forwardDynamics(o->model, kin, art, q, v, tau, qdd);
}
void nbo_jac_C(void* h, const double* q, const double* v, int wrtVel, double* out) {
  Oracle* o = (Oracle*)h;
  std::vector<Kin> kin; std::vector<Art> art;
  ctx(o, q, v, kin, art);
  MatX J = jacobianOfC(o->model, kin, q, v, wrtVel != 0);
  std::copy(J.d.begin(), J.d.end(), out);
}
void nbo_jac_Mx(void* h, const double* q, const double* x, double* out) {
  Oracle* o = (Oracle*)h;
  std::vector<Kin> kin; std::vector<Art> art;
  VecX z(o->model.n, 0.0);
  ctx(o, q, z.data(), kin, art);
  MatX J = jacobianOfMx(o->model, kin, q, x);
  std::copy(J.d.begin(), J.d.end(), out);
}
// impulse test: delta-velocity for a unit impulse `imp` (6) applied on body `body` in its own frame
void nbo_impulse_response(void* h, const double* q, int body, const double* imp, double* delV) {
  Oracle* o = (Oracle*)h;
  std::vector<Kin> kin; std::vector<Art> art;
  VecX z(o->model.n, 0.0);
  ctx(o, q, z.data(), kin, art);
  std::vector<Vec6> imps(o->model.nb, zero6());
  for (int i = 0; i < 6; i++) imps[body][i] = imp[i];
  impulseDynamics(o->model, kin, art, imps, delV);
}
void nbo_body_world_transform(void* h, const double* q, int body, double* T12) {
  Oracle* o = (Oracle*)h;
  std::vector<Kin> kin;
  VecX z(o->model.n, 0.0);
  kinematics(o->model, q, z.data(), kin);
  for (int i = 0; i < 9; i++) T12[i] = kin[body].Tworld.R.m[i];
  for (int i = 0; i < 3; i++) T12[9 + i] = kin[body].Tworld.p[i];
}
void nbo_integrate_positions(void* h, const double* q, const double* v, double* qn) {
  Oracle* o = (Oracle*)h;
  integratePositions(o->model, q, v, o->model.dt, qn);
}

// contact introspection (defined in contact.hpp)
int nbo_last_contacts(void* h, double* out /* [C][12]: p(3) n(3) depth type bodyA bodyB boxA boxB */, int cap) {
  Oracle* o = (Oracle*)h;
  const ContactResult& cr = o->snap.contact;
  int C = (int)cr.contacts.size();
  for (int i = 0; i < C && i < cap; i++) {
    const Contact& c = cr.contacts[i];
    double* r = out + 12 * i;
    for (int k = 0; k < 3; k++) { r[k] = c.point[k]; r[3 + k] = c.normal[k]; }
    r[6] = c.depth; r[7] = c.type; r[8] = c.bodyA; r[9] = c.bodyB; r[10] = c.boxA; r[11] = c.boxB;
  }
  return C;
}
int nbo_last_lcp(void* h, double* A, double* b, double* x, double* lo, double* hi, int32_t* findex, int32_t* rowClass,
                 int cap) {
  Oracle* o = (Oracle*)h;
  const ContactResult& cr = o->snap.contact;
  int mrows = cr.m;
  if (mrows > cap) return -mrows;
  for (int i = 0; i < mrows; i++) {
    for (int j = 0; j < mrows; j++) A[i * mrows + j] = cr.A(i, j);
    b[i] = cr.b[i]; x[i] = cr.x[i]; lo[i] = cr.lo[i]; hi[i] = cr.hi[i]; findex[i] = cr.findex[i];
    rowClass[i] = cr.rowClass[i];
  }
  return mrows;
}

// ---- direct access to the collision / LCP building blocks for the fixture tests ----
int nbo_box_box(const double* T1, const double* size1, const double* T2, const double* size2, double clip,
                double* out /* [8][22]: point normal depth type edgeAFixed edgeADir edgeBFixed edgeBDir */) {
  std::vector<Contact> cs;
  Vec3 h1 = mk3(0.5 * size1[0], 0.5 * size1[1], 0.5 * size1[2]), h2 = mk3(0.5 * size2[0], 0.5 * size2[1], 0.5 * size2[2]);
  int n = boxBox(loadIso(T1), h1, loadIso(T2), h2, clip, cs);
  for (int i = 0; i < n && i < 8; i++) {
    double* r = out + 22 * i;
    const Contact& c = cs[i];
    for (int k = 0; k < 3; k++) {
      r[k] = c.point[k]; r[3 + k] = c.normal[k]; r[8 + k] = c.edgeAFixedPoint[k]; r[11 + k] = c.edgeADir[k];
      r[14 + k] = c.edgeBFixedPoint[k]; r[17 + k] = c.edgeBDir[k];
    }
    r[6] = c.depth; r[7] = c.type;
  }
  return n;
}
// which: 0 = box (T1, size1) vs sphere (T2, radius size2[0]), 1 = sphere vs box, 2 = sphere vs sphere.  out: up to 4 contacts of 32
// doubles: point normal depth type sphereCenter face1..3Normal locked(3) centerA centerB radiusA radiusB
int nbo_sphere_pair(int which, const double* T1, const double* size1, const double* T2, const double* size2, double clip, double* out) {
  std::vector<Contact> cs;
  int n;
  if (which == 0) n = sphereBoxPair(false, size2[0], loadIso(T2), mk3(0.5 * size1[0], 0.5 * size1[1], 0.5 * size1[2]), loadIso(T1), clip, cs);
  else if (which == 1) n = sphereBoxPair(true, size1[0], loadIso(T1), mk3(0.5 * size2[0], 0.5 * size2[1], 0.5 * size2[2]), loadIso(T2), clip, cs);
  else n = sphereSphere(size1[0], loadIso(T1), size2[0], loadIso(T2), clip, cs);
  for (int i = 0; i < n && i < 4; i++) {
    double* r = out + 32 * i;
    const Contact& c = cs[i];
    for (int k = 0; k < 3; k++) {
      r[k] = c.point[k]; r[3 + k] = c.normal[k]; r[8 + k] = c.sphereCenter[k];
      r[11 + k] = c.faceNormal[0][k]; r[14 + k] = c.faceNormal[1][k]; r[17 + k] = c.faceNormal[2][k];
      r[23 + k] = c.centerA[k]; r[26 + k] = c.centerB[k];
    }
    r[6] = c.depth; r[7] = c.type; r[20] = c.faceLocked[0]; r[21] = c.faceLocked[1]; r[22] = c.faceLocked[2]; r[29] = c.radiusA; r[30] = c.radiusB; r[31] = 0;
  }
  return n;
}
// which: 0 = capsule vs capsule, 1 = sphere (size1[0]) vs capsule, 2 = capsule vs sphere; a capsule's size = (radius, height).
// out: 48 doubles (the layout of ref_collide_capsule, oracle/ref_boxbox_epilogue.hpp)
int nbo_capsule_pair(int which, const double* T1, const double* size1, const double* T2, const double* size2, double clip, double* out) {
  std::vector<Contact> cs;
  int n;
  if (which == 0) n = capsuleCapsule(size1[1], size1[0], loadIso(T1), size2[1], size2[0], loadIso(T2), clip, cs);
  else if (which == 1) n = sphereCapsulePair(true, size1[0], loadIso(T1), size2[1], size2[0], loadIso(T2), clip, cs);
  else n = sphereCapsulePair(false, size2[0], loadIso(T2), size1[1], size1[0], loadIso(T1), clip, cs);
  if (n) {
    const Contact& c = cs[0];
    double* o = out;
    for (int k = 0; k < 3; k++) {
      o[k] = c.point[k]; o[3 + k] = c.normal[k]; o[8 + k] = c.centerA[k]; o[11 + k] = c.centerB[k];
      o[16 + k] = c.sphereCenter[k]; o[19 + k] = c.pipeDir[k]; o[22 + k] = c.pipeFixedPoint[k]; o[25 + k] = c.pipeClosestPoint[k];
      o[30 + k] = c.edgeAFixedPoint[k]; o[33 + k] = c.edgeADir[k]; o[36 + k] = c.edgeBFixedPoint[k]; o[39 + k] = c.edgeBDir[k];
      o[42 + k] = c.edgeAClosestPoint[k]; o[45 + k] = c.edgeBClosestPoint[k];
    }
    o[6] = c.depth; o[7] = c.type; o[14] = c.radiusA; o[15] = c.radiusB; o[28] = c.sphereRadius; o[29] = c.pipeRadius;
  }
  return n;
}
static LcpProblem mkProblem(int n, const double* A, const double* x, const double* b, const double* lo, const double* hi,
                            const int32_t* findex) {
  LcpProblem p;
  p.A = MatX(n, n);
  for (int i = 0; i < n * n; i++) p.A.d[i] = A[i];
  p.x.assign(x, x + n); p.b.assign(b, b + n); p.lo.assign(lo, lo + n); p.hi.assign(hi, hi + n);
  p.findex.assign(findex, findex + n);
  return p;
}
int nbo_lcp_valid(int n, const double* A, const double* x, const double* b, const double* lo, const double* hi,
                  const int32_t* findex, int ignoreFriction) {
  LcpProblem p = mkProblem(n, A, x, b, lo, hi, findex);
  return isLCPSolutionValid(p.A, p.x, p.b, p.hi, p.lo, p.findex, ignoreFriction != 0) ? 1 : 0;
}
void nbo_lcp_guess(int n, const double* A, const double* b, const int32_t* findex, double* x) {
  MatX Am(n, n);
  for (int i = 0; i < n * n; i++) Am.d[i] = A[i];
  VecX g = guessSolution(Am, VecX(b, b + n), std::vector<int>(findex, findex + n));
  for (int i = 0; i < n; i++) x[i] = g[i];
}
int nbo_lcp_pgs(int n, const double* A, double* x, const double* b, const double* lo, const double* hi, const int32_t* findex,
                int maxIter, double dx, double rel, double eps) {
  LcpProblem p = mkProblem(n, A, x, b, lo, hi, findex);
  bool ok = pgsSolve(p, maxIter, dx, rel, eps);
  for (int i = 0; i < n; i++) x[i] = p.x[i];
  return ok ? 1 : 0;
}
int nbo_lcp_dantzig(int n, const double* A, double* x, const double* b, const double* lo, const double* hi,
                    const int32_t* findex, int early) {
  LcpProblem p = mkProblem(n, A, x, b, lo, hi, findex);
  int ok = dantzigSolve(p, early != 0);
  for (int i = 0; i < n; i++) x[i] = p.x[i];
  return ok;
}
// stages 1-3 of the solver cascade (contact.hpp lcpCascade) on plain arrays; returns isLCPSolutionValid of the result on the
// matrix the solver ended with (CFM on the diagonal when a fallback stage was used), friction ignored when it was dropped
int nbo_lcp_cascade(int n, const double* A, const double* x0, const double* b, const double* lo, const double* hi,
                    const int32_t* findex, double fallbackCfm, double* xOut, uint32_t* statusOut, double* cfmOut) {
  LcpProblem p = mkProblem(n, A, x0, b, lo, hi, findex);
  VecX X;
  s_t cfm = 0;
  bool noFric = false;
  uint32_t st = 0;
  lcpCascade(p.A, p.b, p.lo, p.hi, p.findex, p.x, fallbackCfm, X, cfm, noFric, st);
  for (int i = 0; i < n; i++) xOut[i] = X[i];
  if (statusOut) *statusOut = st;
  if (cfmOut) *cfmOut = cfm;
  MatX Ac = p.A;
  for (int i = 0; i < n; i++) Ac(i, i) += cfm;
  return isLCPSolutionValid(Ac, X, p.b, p.hi, p.lo, p.findex, noFric) ? 1 : 0;
}
// reduce / removeFriction: returns reduced size; outputs reduced problem and mapOut [n][nr]
int nbo_lcp_reduce(int n, const double* A, const double* x, const double* b, const double* lo, const double* hi,
                   const int32_t* findex, int removeFriction, double* Ar, double* xr, double* br, double* lor, double* hir,
                   int32_t* fr, double* mapOut) {
  LcpProblem p = mkProblem(n, A, x, b, lo, hi, findex);
  MatX mo = removeFriction ? removeFrictionLcp(p) : reduceLcp(p);
  int nr = (int)p.x.size();
  for (int i = 0; i < nr * nr; i++) Ar[i] = p.A.d[i];
  for (int i = 0; i < nr; i++) { xr[i] = p.x[i]; br[i] = p.b[i]; lor[i] = p.lo[i]; hir[i] = p.hi[i]; fr[i] = p.findex[i]; }
  for (int i = 0; i < n * nr; i++) mapOut[i] = mo.d[i];
  return nr;
}
int nbo_cod_solve(int rows, int cols, const double* A, const double* b, double* x) {
  MatX Am(rows, cols);
  for (int i = 0; i < rows * cols; i++) Am.d[i] = A[i];
  int rank = 0;
  VecX r = codSolve(Am, VecX(b, b + rows), &rank);
  for (int i = 0; i < cols; i++) x[i] = r[i];
  return rank;
}
// The spatial-algebra primitives of oracle/spatial.hpp by name, for the comparison with the reference's own functions compiled into
// oracle/_ref/libgeometry_ref.so (tests/test_oracle_ref_geometry.py).  in0 / in1: the arguments in the layouts of that library's entry
// points (3 x 3 / 6 x 6 row-major; transforms 12 doubles R row-major then p; 6-vectors [omega; v]).  Returns the number of outputs.
int nbo_prim(const char* name, const double* in0, const double* in1, double* out) {
  const std::string f(name);
  auto iso = [](const double* t) { Iso T; for (int i = 0; i < 9; i++) T.R.m[i] = t[i]; T.p = mk3(t[9], t[10], t[11]); return T; };
  auto v3 = [](const double* x) { return mk3(x[0], x[1], x[2]); };
  auto v6 = [](const double* x) { Vec6 r; for (int i = 0; i < 6; i++) r.v[i] = x[i]; return r; };
  auto o3 = [&](const Vec3& v) { for (int i = 0; i < 3; i++) out[i] = v[i]; return 3; };
  auto o6 = [&](const Vec6& v) { for (int i = 0; i < 6; i++) out[i] = v[i]; return 6; };
  auto o33 = [&](const Mat3& v) { for (int i = 0; i < 9; i++) out[i] = v.m[i]; return 9; };
  if (f == "expMapRot") return o33(expMapRot(v3(in0)));
  if (f == "expMapJac") return o33(expMapJac(v3(in0)));
  if (f == "expAngular") return o33(expAngular(v3(in0)));
  if (f == "makeSkewSymmetric") return o33(skew(v3(in0)));
  if (f == "logMap") { Mat3 R; for (int i = 0; i < 9; i++) R.m[i] = in0[i]; return o3(logMap(R)); }
  if (f == "AdT") return o6(AdT(iso(in0), v6(in1)));
  if (f == "AdInvT") return o6(AdInvT(iso(in0), v6(in1)));
  if (f == "AdInvRLinear") return o6(AdInvRLinear(iso(in0), v3(in1)));
  if (f == "ad") return o6(ad(v6(in0), v6(in1)));
  if (f == "dad") return o6(dad(v6(in0), v6(in1)));
  if (f == "dAdT") return o6(dAdT(iso(in0), v6(in1)));
  if (f == "dAdInvT") return o6(dAdInvT(iso(in0), v6(in1)));
  if (f == "tangentBasis") { Vec3 t1, t2; tangentBasis(v3(in0), t1, t2); for (int i = 0; i < 3; i++) { out[i] = t1[i]; out[3 + i] = t2[i]; } return 6; }
  if (f == "tangentBasisGradient") { Vec3 t1, t2; tangentBasisGradient(v3(in0), v3(in1), t1, t2); for (int i = 0; i < 3; i++) { out[i] = t1[i]; out[3 + i] = t2[i]; } return 6; }
  if (f == "contactPointGradient")
    return o3(contactPointGradient(v3(in0), v3(in0 + 3), v3(in0 + 6), v3(in0 + 9), v3(in0 + 12), v3(in0 + 15), v3(in0 + 18), v3(in0 + 21)));
  if (f == "contactPointGradientRadii")   // in1 = (radiusA, radiusB)
    return o3(contactPointGradient(v3(in0), v3(in0 + 3), v3(in0 + 6), v3(in0 + 9), v3(in0 + 12), v3(in0 + 15), v3(in0 + 18), v3(in0 + 21), in1[0], in1[1]));
  if (f == "closestPointOnLineGradient")
    return o3(closestPointOnLineGradient(v3(in0), v3(in0 + 3), v3(in0 + 6), v3(in0 + 9), v3(in0 + 12), v3(in0 + 15)));
  if (f == "transformInertia") {
    Mat6 I; for (int i = 0; i < 36; i++) I.m[i] = in1[i];
    const Mat6 r = transformInertia(iso(in0), I);
    for (int i = 0; i < 36; i++) out[i] = r.m[i];
    return 36;
  }
  return -1;
}
}
