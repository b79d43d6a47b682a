// nimble_amd.hip — host side of the C ABI declared in include/nimble_amd.h.
//
// Thin: validates the model description, uploads the constants once, sizes the workspace and
// launches the kernels of kernels.hip on the caller's stream.  No torch types, no allocation on
// the hot path, no host<->device synchronisation inside step_forward/step_backward.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "abi_variants.h"   // (before the ABI header: the two instantiations of the library rename its entry points, see there)
#include "../../include/nimble_amd.h"
#include "kernels.hip"
#include "contact_kernels.hip"
#include "contact_backward.hip"
#if NBL_GENERAL
#include "gen_contact.hip"      // the general instantiation: any number of contact rows (up to 192), rows looped over instead of mapped to lanes
#else
#include "coop_kernels.hip"
#endif
#include "coop_tree.hip"
#include "inertia_backward.hip"

using namespace NBL_NS;

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(NBL_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int NBL_MAX_SLICES = 8;
enum KernelId { K_FWD = 0, K_DETECT, K_BWD, K_RECOMPUTE, K_BWD_FINAL, K_SOLVE_COOP, K_BWD_A_COOP, K_ROWS_COOP, K_BWD_B_COOP, K_FWD_COOP, K_RECOMPUTE_COOP, K_BWD_FINAL_COOP, K_TREE_TO_LANES, K_CASCADE_COOP, K_CASCADE_FINAL, K_BWD_BOUNCE, K_CASCADE_FUSED, K_FWD_DETECT, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"k_step_forward", "k_contact_detect",
                                           "k_step_backward", "k_bwd_recompute",
                                           "k_bwd_final", "k_contact_solve_coop", "k_bwd_contact_a_coop", "k_contact_rows_coop", "k_bwd_contact_b_coop", "k_step_forward_coop", "k_bwd_recompute_coop",
                                           "k_bwd_final_coop", "k_tree_to_lanes", "k_contact_cascade_stages", "k_contact_cascade_final", "k_bwd_bounce", "k_contact_cascade_fused", "k_forward_detect_coop"};
struct TimedLaunch {
  hipEvent_t start, stop;
  int kernel;
};
}  // namespace

struct nbl_model {
  int device = 0;
  DevModel mdl;
  DevBody* dBodies = nullptr;
  DevDof* dDofs = nullptr;
  int nb = 0, n = 0, k = 0, maxContacts = 0;
  bool hasContact = false;
  DevContactModel* dContact = nullptr;
  SavedLayout lay;
  bool timing = false;      // kernel timing requested
  bool timingNow = false;   // ... and this call is one of the sampled ones
  int timingPeriod = 1;     // every timingPeriod-th forward / backward call carries HIP events
  int64_t fwdCalls = 0, bwdCalls = 0;
  int slices = 0;                    // batch slices over HIP streams (0 = auto), nbl_set_slices / NBL_SLICES
  bool deferJoin = false;            // nbl_set_deferred_join: every slice on an internal stream, calls return without joining (nbl_join_slices joins)
  std::vector<hipStream_t> side;     // internal streams of slices 1..
  std::vector<hipEvent_t> sideDone;
  hipEvent_t fork = nullptr;
  // one auxiliary stream per slice: k_bwd_recompute_coop (lambda1 for the tree part) runs there while k_bwd_contact_a_coop runs
  // on the slice's own stream - neither needs the other's result
  std::vector<hipStream_t> aux;
  std::vector<hipEvent_t> auxFork, auxJoin;
  int wpbFwd = 4, wpbBwd = 4;        // worlds per workgroup of the lane = body tree kernels (chosen for LDS occupancy)
  size_t ldsFwd = 0, ldsBwd = 0;     // dynamic LDS of those workgroups
  bool auxOverlap = false;           // NBL_AUX_OVERLAP=1: k_bwd_recompute_coop on an auxiliary stream next to k_bwd_contact_a_coop.
                                     // Measured: no gain (1 slice 7.65 vs 7.68 M/s, 2 slices 8.52 vs 8.57) and a loss once the
                                     // streams exceed four (4 slices 7.4 vs 9.0 M/s): the chip is already shared by the slices.
  bool fusedCascade = false;         // NBL_FUSED_CASCADE=1: stages + final part in one launch (k_contact_cascade_fused).  Measured on MI355X,
                                     // metric distribution: one 1024-world slice alone 453 us per step against 503 (the final part of a world no
                                     // longer waits for the slowest Dantzig run of the launch), four slices in flight 5.93 against 6.11 M/s: the
                                     // fused kernel holds the final part's 256 registers through the stages (2 waves per SIMD instead of 3) and
                                     // the other slices' tree kernels wait for the CUs (k_step_forward_coop 70 -> 111 us).  Off by default.
  bool fusedDetect = true;           // NBL_FUSED_DETECT=0: the narrow phase as a launch of its own after the forward tree kernel
  bool detectSplit = true;           // NBL_DETECT_SPLIT=0: one lane per world in k_contact_detect (collider pairs one after the other)
  int detectWl = 0;                  // NBL_DETECT_WL: cap of the worlds per narrow-phase workgroup of k_forward_detect_coop (0: a full wavefront)
  int fkBodies = 0;                  // bodies on the ancestor chains of the colliders (forward kinematics of the fused narrow phase)
  int rowsPack = 1;                  // worlds per wavefront of k_contact_rows_coop (2: the 24-row build, <= 32 device bodies; NBL_ROWS_PACK=1 forces 1)
  int nPairs = 0;                    // candidate collider pairs of the model
  bool multiGroup = false;           // colliders on more than one skeleton: a world can hold several constrained groups
  bool coopFinal = true;             // the backward sweeps too, in the world frame (NBL_COOP_FINAL=0: one world per lane, fed by k_tree_to_lanes)
  bool coopTree = false;             // tree sweeps one world per wavefront (needs the saved tree block, nb and n <= 64)
  int treeLanes = 0;                 // worlds per workgroup of the one-world-per-lane tree kernels (0 = pick from B); nbl_set_launch_lanes
  std::vector<DevBody> hBodies;      // host copy of the body constants (nbl_set_body_inertia patches one entry)
  int userBodies = 0;                // bodies of the caller's description; bodyMap: caller's body index -> device body (empty: identity;
  std::vector<int32_t> bodyMap;      // models with ball joints carry two extra massless bodies per ball joint, expandBallJoints)
  int deviceBody(int body) const { return bodyMap.empty() ? body : bodyMap[body]; }
  DevInertiaParam* dParams = nullptr; // registered inertia parameters (nbl_set_inertia_params)
  int nParams = 0;
  void* staging = nullptr;           // pinned host staging area of the stream-ordered uploads (nbl_set_body_inertias)
  size_t stagingBytes = 0;
  hipEvent_t staged = nullptr;       // recorded after the last upload that reads the staging area
  std::vector<TimedLaunch> pending;
  double fwdMs = 0, bwdMs = 0;
  int64_t fwdCount = 0, bwdCount = 0;
  double kMs[K_COUNT] = {0};
  int64_t kCount[K_COUNT] = {0};
};

// spatial inertia (Inertia.cpp:1368-1383), packed symmetric (upper triangle, row by row)
static void packSpatialInertia(double m, const double* c, const double* I, double* out21) {
  double Ic[3][3] = {{I[0], I[3], I[4]}, {I[3], I[1], I[5]}, {I[4], I[5], I[2]}};
  double C[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
  double G[6][6];
  std::memset(G, 0, sizeof(G));
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) {
      double cct = 0;
      for (int k = 0; k < 3; k++) cct += C[r][k] * C[cc][k];
      G[r][cc] = Ic[r][cc] + m * cct;
      G[r][3 + cc] = m * C[r][cc];
      G[3 + r][cc] = m * C[cc][r];
    }
  G[3][3] = G[4][4] = G[5][5] = m;
  int idx = 0;
  for (int r = 0; r < 6; r++)
    for (int cc = r; cc < 6; cc++) out21[idx++] = G[r][cc];
}

// ---- batch slicing over HIP streams ----------------------------------------------------------------------------------
// A call can process its worlds as several contiguous slices whose kernels overlap on internal HIP streams (slice 0 on the
// caller's stream, the others fork from / join into it with events, so the call keeps stream semantics and stays capturable
// in a hipGraph).  The join at the end of every forward / backward call prevents the most useful overlap, which is between the
// FORWARD of one slice and the BACKWARD of another: callers that own the whole fwd+bwd loop get the most by running one World per
// slice on its own stream (bench.py --streams: 6.86 M/s at B = 4096).  Inside one call, measured on MI355X at the end of round 4
// (metric distribution, one World): B = 4096: 1 slice 5.26, 2 slices 5.76, 3 slices 5.48, 4 slices 5.20 M/s; B = 8192: 6.36 / 6.85 -
// two slices let the tail of one slice's launch (its slowest world) overlap with the other's.  Default (auto): 2 slices from 4096
// worlds for models with colliders, else 1; NBL_SLICES / nbl_set_slices.
// slices of a rollout: every slice runs ALL its steps on its stream and the slices join only at the end, so the steps of
// different slices overlap (the per-call join is what makes in-call slicing useless for a single step)
static int rolloutSlicesFor(const nbl_model* m, int64_t B) {
  int sl = m->slices > 0 ? m->slices : (B >= 4096 ? 4 : (B >= 2048 ? 2 : 1));
  if (sl > NBL_MAX_SLICES) sl = NBL_MAX_SLICES;
  while (sl > 1 && B / sl < 256) sl--;
  return sl;
}
static int slicesFor(const nbl_model* m, int64_t B) {
  if (m->deferJoin) return rolloutSlicesFor(m, B);      // nothing joins inside the call: as many slices as a rollout takes (4 from 4096 worlds)
  int sl = m->slices > 0 ? m->slices : ((m->hasContact && B >= 4096) ? 2 : 1);
  if (sl > NBL_MAX_SLICES) sl = NBL_MAX_SLICES;
  while (sl > 1 && B / sl < 256) sl--;
  return sl;
}
static int32_t ensureSideStreams(nbl_model* m, int need) {
  while ((int)m->side.size() < need) {
    hipStream_t st; hipEvent_t ev;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    m->side.push_back(st); m->sideDone.push_back(ev);
  }
  if (!m->fork) HIP_TRY(hipEventCreateWithFlags(&m->fork, hipEventDisableTiming));
  return NBL_OK;
}
static int32_t ensureAux(nbl_model* m, int need) {
  while ((int)m->aux.size() < need) {
    hipStream_t st; hipEvent_t e1, e2;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    m->aux.push_back(st); m->auxFork.push_back(e1); m->auxJoin.push_back(e2);
  }
  return NBL_OK;
}
// worlds per slice and the slices of a call
static int64_t slicePer(int64_t B, int sl) { return (((B + sl - 1) / sl) + 15) & ~(int64_t)15; }
// run fn(slice index, first world, one-past-last world, stream) for every slice, fork/join around the caller's stream.
// DEFERRED JOIN (nbl_set_deferred_join): slice 0 runs on the caller's stream, the others on internal streams of the handle
// (ordered after the caller's stream only where the caller says so: nbl_fork_slices), and the call returns WITHOUT making the caller's
// stream wait for them: the slices of
// consecutive calls - the forward pass of one, the backward pass of another - overlap like the four-handle pattern's, with ONE handle.
// Whoever consumes a result does so on the slice's stream (nbl_slice_stream) or after nbl_join_slices.
template <class Fn>
static int32_t forSlices(nbl_model* m, int64_t B, int sl, hipStream_t s, Fn fn) {
  const bool defer = m->deferJoin;
  if (sl > 1) {
    const int32_t rc = ensureSideStreams(m, sl - 1);
    if (rc != NBL_OK) return rc;
    // (deferred: the call does not touch the caller's stream at all - an event recorded there per call queues behind whatever shares its
    //  hardware queue, a slice's kernels included, and the slices of the NEXT call then wait for it: measured, the steps ran in lock-step)
    if (!defer) HIP_TRY(hipEventRecord(m->fork, s));
  }
  const int64_t per = slicePer(B, sl);
  for (int i = 0; i < sl; i++) {
    const int64_t b0 = (int64_t)i * per, b1 = std::min(B, b0 + per);
    if (b0 >= b1) break;
    hipStream_t st = i == 0 ? s : m->side[i - 1];          // (slice 0 on the caller's stream in both modes: it is one of the workers)
    if (i > 0 && !defer) HIP_TRY(hipStreamWaitEvent(st, m->fork, 0));
    const int32_t rc = fn(i, b0, b1, st);
    if (rc != NBL_OK) return rc;
    if (i > 0 && !defer) { HIP_TRY(hipEventRecord(m->sideDone[i - 1], st)); HIP_TRY(hipStreamWaitEvent(s, m->sideDone[i - 1], 0)); }
  }
  return NBL_OK;
}

namespace {
// Ball joints (NBL_JOINT_BALL, BallJoint.cpp) run on the device as three coincident single-axis joints x, y, z at zero angle, the first
// one carrying exp(q): two massless bodies are inserted before every ball-jointed body.  Mass matrix, impulse tests and contact
// Jacobians of that chain are the ball joint's; the joint accelerations differ by the closed-form term (wy wz, -wx wz, wx wy) (the
// chain's axes turn with its own rates, the ball's do not; tests/test_ball_joint.py), positions integrate on SO(3) and position
// derivatives go through H(q) = [expMapJac(q)^T; 0] of the first of the three.  DOF numbering is unchanged; body indices of the
// caller's description are mapped to the last body of each triple (the one that carries mass, colliders and children).
struct ExpandedDesc {
  nbl_model_desc desc;
  std::vector<int32_t> parent, jointType, dofOffset, boxBody, bodySkeleton, bodyMap, ballComp, bodySelf;
  std::vector<double> Tpj, Tcj, axis, mass, com, inertia, pitch;
};
constexpr int NBL_JOINT_FREE_CHAIN = 6;   // internal (= JT_FREEC): one of the six coincident axes of a free joint below the root
bool hasBallJoints(const nbl_model_desc* d) {   // ... or free joints below the root, which are expanded the same way
  for (int i = 0; i < d->n_bodies; i++)
    if (d->joint_type[i] == NBL_JOINT_BALL || (d->joint_type[i] == NBL_JOINT_FREE && d->parent[i] >= 0)) return true;
  return false;
}
void expandBallJoints(const nbl_model_desc* d, ExpandedDesc& e) {
  static const double I12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  e.bodyMap.assign(d->n_bodies, -1);
  int srcBody = 0;
  auto push = [&](int parent, int jt, int dofOff, const double* Tpj, const double* Tcj, const double* ax, double mass, const double* com,
                  const double* inertia, int skel, int comp) {
    static const double z6[6] = {0, 0, 0, 0, 0, 0};
    e.parent.push_back(parent); e.jointType.push_back(jt); e.dofOffset.push_back(dofOff);
    e.Tpj.insert(e.Tpj.end(), Tpj, Tpj + 12); e.Tcj.insert(e.Tcj.end(), Tcj, Tcj + 12); e.axis.insert(e.axis.end(), ax, ax + 3);
    e.mass.push_back(mass); e.com.insert(e.com.end(), com ? com : z6, (com ? com : z6) + 3);
    e.inertia.insert(e.inertia.end(), inertia ? inertia : z6, (inertia ? inertia : z6) + 6);
    e.bodySkeleton.push_back(skel); e.ballComp.push_back(comp);
    e.bodySelf.push_back(d->body_self_collision ? d->body_self_collision[srcBody] : 0);
    e.pitch.push_back(jt == NBL_JOINT_SCREW && d->pitch ? d->pitch[srcBody] : 0.1);
  };
  // skeleton ids: the caller's, or (default: one skeleton per tree) the root of the tree in the CALLER's numbering - any id shared by
  // exactly the bodies of one tree will do
  auto rootOf = [&](int body) { while (d->parent[body] >= 0) body = d->parent[body]; return body; };
  for (int i = 0; i < d->n_bodies; i++) {
    srcBody = i;
    const int par = d->parent[i] < 0 ? -1 : e.bodyMap[d->parent[i]];
    const int skel = d->body_skeleton ? d->body_skeleton[i] : rootOf(i);
    const bool freeBelowRoot = d->joint_type[i] == NBL_JOINT_FREE && d->parent[i] >= 0;
    if (freeBelowRoot) {
      // six coincident axes at zero displacement behind T_pj [exp(q_r), q_p]: rotations x, y, z, then translations x, y, z (the order of the
      // free joint's DOFs; S = Ad(T_cj) is constant in the child frame, FreeJoint.cpp:1049-1056), the first five bodies massless.
      // qdd_free = qdd_chain + [(wy wz, -wx wz, wx wy); w x u]
      for (int k = 0; k < 6; k++) {
        const double ax[3] = {k % 3 == 0 ? 1.0 : 0.0, k % 3 == 1 ? 1.0 : 0.0, k % 3 == 2 ? 1.0 : 0.0};
        const bool last = k == 5;
        push(k == 0 ? par : (int)e.parent.size() - 1, NBL_JOINT_FREE_CHAIN, d->dof_offset[i] + k, k == 0 ? d->T_pj + 12 * i : I12,
             last ? d->T_cj + 12 * i : I12, ax, last ? d->mass[i] : 0.0, last ? d->com + 3 * i : nullptr, last ? d->inertia + 6 * i : nullptr,
             skel, k);
      }
    } else if (d->joint_type[i] != NBL_JOINT_BALL) {
      push(par, d->joint_type[i], d->dof_offset[i], d->T_pj + 12 * i, d->T_cj + 12 * i, d->axis + 3 * i, d->mass[i], d->com + 3 * i,
           d->inertia + 6 * i, skel, 0);
    } else {
      for (int k = 0; k < 3; k++) {
        const double ax[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
        const bool last = k == 2;
        push(k == 0 ? par : (int)e.parent.size() - 1, NBL_JOINT_BALL, d->dof_offset[i] + k, k == 0 ? d->T_pj + 12 * i : I12,
             last ? d->T_cj + 12 * i : I12, ax, last ? d->mass[i] : 0.0, last ? d->com + 3 * i : nullptr, last ? d->inertia + 6 * i : nullptr,
             skel, k);
      }
    }
    e.bodyMap[i] = (int)e.parent.size() - 1;
  }
  e.boxBody.resize(d->n_boxes > 0 ? d->n_boxes : 0);
  for (int i = 0; i < d->n_boxes; i++) e.boxBody[i] = d->box_body[i] < 0 ? d->box_body[i] : (d->box_body[i] < d->n_bodies ? e.bodyMap[d->box_body[i]] : 1 << 20);
  e.desc = *d;
  e.desc.n_bodies = (int32_t)e.parent.size();
  e.desc.parent = e.parent.data(); e.desc.joint_type = e.jointType.data(); e.desc.dof_offset = e.dofOffset.data();
  e.desc.T_pj = e.Tpj.data(); e.desc.T_cj = e.Tcj.data(); e.desc.axis = e.axis.data();
  e.desc.mass = e.mass.data(); e.desc.com = e.com.data(); e.desc.inertia = e.inertia.data();
  e.desc.box_body = e.boxBody.data(); e.desc.body_skeleton = e.bodySkeleton.data(); e.desc.pitch = e.pitch.data();
  e.desc.body_self_collision = e.bodySelf.data();
}
}  // namespace

extern "C" {

const char* nbl_last_error(void) { return g_err.c_str(); }
int32_t nbl_version(void) { return (0 << 16) | NBL_ABI_MINOR; }

int32_t nbl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t nbl_model_create(const nbl_model_desc* d, int32_t device, nbl_model** out) {
  if (!d || !out) return fail(NBL_E_BADARG, "null argument");
  *out = nullptr;
  if (d->n_bodies <= 0 || d->n_dofs <= 0) return fail(NBL_E_BADARG, "empty model");
  int ndev = nbl_device_count();
  if (ndev <= 0) return fail(NBL_E_NOGPU, "no HIP device visible: the batched timestep has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(NBL_E_BADARG, "device index out of range");
  for (int i = 0; i < d->n_bodies; i++)
    if (d->parent[i] < -1 || d->parent[i] >= i) return fail(NBL_E_BADARG, "bodies must be listed parents-before-children");
  ExpandedDesc expanded;
  const int userBodies = d->n_bodies;
  const bool ballModel = hasBallJoints(d);
  if (ballModel) {
    if (d->body_skeleton)
      for (int i = 0; i < d->n_bodies; i++)
        if (d->body_skeleton[i] < 0 || d->body_skeleton[i] >= 64) return fail(NBL_E_BADARG, "body_skeleton must lie in [0, 64)");
    expandBallJoints(d, expanded);
    d = &expanded.desc;
  }

  std::vector<DevBody> hb(d->n_bodies);
  std::vector<DevDof> hd(d->n_dofs);
  int off = 0;
  for (int i = 0; i < d->n_bodies; i++) {
    DevBody& b = hb[i];
    std::memset(&b, 0, sizeof(b));
    b.parent = d->parent[i];
    b.jtype = d->joint_type[i];
    if (b.parent < -1 || b.parent >= i) return fail(NBL_E_BADARG, "bodies must be listed parents-before-children");
    if (b.jtype == NBL_JOINT_WELD)
      return fail(NBL_E_UNSUPPORTED, "weld joints must be merged into their parent before upload (ModelDescription.merge_welds)");
    if (b.jtype != NBL_JOINT_REVOLUTE && b.jtype != NBL_JOINT_PRISMATIC && b.jtype != NBL_JOINT_FREE && b.jtype != NBL_JOINT_BALL &&
        b.jtype != NBL_JOINT_SCREW && !(ballModel && b.jtype == NBL_JOINT_FREE_CHAIN))
      return fail(NBL_E_UNSUPPORTED, "joint type outside the hot-path scope (revolute, prismatic, screw, free, ball)");
    if (b.jtype == NBL_JOINT_FREE && b.parent != -1) return fail(NBL_E_BADARG, "internal: a free joint below the root was not expanded");
    b.freeIdx = -1; b.ballComp = ballModel ? expanded.ballComp[i] : 0;
    b.root = b.parent < 0 ? i : hb[b.parent].root; b.padr = 0;
    b.level = b.parent < 0 ? 0 : hb[b.parent].level + 1;
    b.rank = 0;
    for (int j = 0; j < i; j++) if (hb[j].parent == b.parent) b.rank++;
    b.ndof = (b.jtype == NBL_JOINT_FREE) ? 6 : 1;
    b.dofOff = d->dof_offset[i];
    if (b.dofOff != off) return fail(NBL_E_BADARG, "dof_offset must be the running sum of joint DOFs");
    off += b.ndof;
    for (int k = 0; k < 12; k++) { b.Tpj[k] = d->T_pj[12 * i + k]; b.Tcj[k] = d->T_cj[12 * i + k]; }
    // inverse of T_cj
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) b.TcjInv[3 * r + c] = b.Tcj[3 * c + r];
    for (int r = 0; r < 3; r++)
      b.TcjInv[9 + r] = -(b.Tcj[0 + r] * b.Tcj[9] + b.Tcj[3 + r] * b.Tcj[10] + b.Tcj[6 + r] * b.Tcj[11]);
    for (int k = 0; k < 3; k++) b.axis[k] = d->axis[3 * i + k];
    // constant relative Jacobian: AdTAngular / AdTLinear (RevoluteJoint.cpp:141-152, PrismaticJoint.cpp)
    const double* R = b.Tcj;
    const double* p = b.Tcj + 9;
    double Ra[3];
    for (int r = 0; r < 3; r++) Ra[r] = R[3 * r] * b.axis[0] + R[3 * r + 1] * b.axis[1] + R[3 * r + 2] * b.axis[2];
    const bool chainRot = b.jtype == NBL_JOINT_FREE_CHAIN && b.ballComp < 3, chainLin = b.jtype == NBL_JOINT_FREE_CHAIN && b.ballComp >= 3;
    if (b.jtype == NBL_JOINT_REVOLUTE || b.jtype == NBL_JOINT_BALL || chainRot) {   // ball / free below the root: one of the coincident axes (expandBallJoints)
      b.S[0] = Ra[0]; b.S[1] = Ra[1]; b.S[2] = Ra[2];
      b.S[3] = p[1] * Ra[2] - p[2] * Ra[1];
      b.S[4] = p[2] * Ra[0] - p[0] * Ra[2];
      b.S[5] = p[0] * Ra[1] - p[1] * Ra[0];
    } else if (b.jtype == NBL_JOINT_PRISMATIC || chainLin) {
      b.S[3] = Ra[0]; b.S[4] = Ra[1]; b.S[5] = Ra[2];
    } else if (b.jtype == NBL_JOINT_SCREW) {            // Ad(T_cj) [axis; h axis], ScrewJoint.cpp:160-179
      b.screwRate = (d->pitch ? d->pitch[i] : 0.1) / (2.0 * M_PI);
      b.S[0] = Ra[0]; b.S[1] = Ra[1]; b.S[2] = Ra[2];
      b.S[3] = p[1] * Ra[2] - p[2] * Ra[1] + b.screwRate * Ra[0];
      b.S[4] = p[2] * Ra[0] - p[0] * Ra[2] + b.screwRate * Ra[1];
      b.S[5] = p[0] * Ra[1] - p[1] * Ra[0] + b.screwRate * Ra[2];
    }
    packSpatialInertia(d->mass[i], d->com + 3 * i, d->inertia + 6 * i, b.G);
  }
  if (off != d->n_dofs) return fail(NBL_E_BADARG, "n_dofs does not match the joints");
  const double inf = INFINITY;
  for (int j = 0; j < d->n_dofs; j++) {
    DevDof& f = hd[j];
    f.damping = d->damping ? d->damping[j] : 0.0;
    f.spring = d->spring ? d->spring[j] : 0.0;
    f.rest = d->rest ? d->rest[j] : 0.0;
    f.posLo = d->pos_lo ? d->pos_lo[j] : -inf; f.posHi = d->pos_hi ? d->pos_hi[j] : inf;
    f.velLo = d->vel_lo ? d->vel_lo[j] : -inf; f.velHi = d->vel_hi ? d->vel_hi[j] : inf;
    f.forceLo = d->force_lo ? d->force_lo[j] : -inf; f.forceHi = d->force_hi ? d->force_hi[j] : inf;
    f.actionIndex = -1;
    f.pad = 0;
  }
  for (int a = 0; a < d->n_action; a++) {
    int j = d->action_map[a];
    if (j < 0 || j >= d->n_dofs) return fail(NBL_E_BADARG, "action mapping out of bounds");  // World.cpp:2118-2135
    if (hd[j].actionIndex != -1) return fail(NBL_E_UNSUPPORTED, "two action entries map to the same DOF");
    hd[j].actionIndex = a;
  }

  // ---- contact model: box colliders, candidate pairs (CollisionFilter.cpp:105-154), ancestor masks ----
  DevContactModel hc;
  std::memset(&hc, 0, sizeof(hc));
  // joint-limit constraint rows (dof_limit_enforced): the joints with a finite limit to enforce, one entry per DOF (= per device body)
  std::vector<int> limitDofs;
  if (d->dof_limit_enforced)
    for (int i = 0; i < d->n_bodies; i++) {
      const int jt = d->joint_type[i];
      if (jt == NBL_JOINT_WELD) continue;
      if (jt == NBL_JOINT_FREE) {
        // the free-joint root is solved in its body frame by the impulse tests (no world-frame axis per DOF): no rows for its coordinates
        for (int k = 0; k < 6; k++) {
          const int j = d->dof_offset[i] + k;
          if (d->dof_limit_enforced[j] && (std::isfinite(hd[j].posLo) || std::isfinite(hd[j].posHi)))
            return fail(NBL_E_UNSUPPORTED, "dof_limit_enforced with a finite limit on the coordinates of a free-joint root");
        }
        continue;
      }
      // (the three / six coincident single-axis bodies of a ball joint / a free joint below the root carry the joint's own generalized
      //  velocities, so a unit impulse on one of their DOFs IS the joint's JointLimitConstraint::applyUnitImpulse)
      const int j = d->dof_offset[i];
      if (d->dof_limit_enforced[j] && (std::isfinite(hd[j].posLo) || std::isfinite(hd[j].posHi))) limitDofs.push_back(i);
    }
  if (!limitDofs.empty() && d->max_contacts <= 0)
    return fail(NBL_E_BADARG, "dof_limit_enforced needs max_contacts > 0: a joint-limit row takes one contact slot of the LCP");
  bool hasContact = (d->n_boxes > 0 || !limitDofs.empty()) && d->max_contacts > 0;
  bool multiGroupModel = !limitDofs.empty();   // the general instantiation of the contact kernels carries the joint-limit rows
  if (hasContact) {
    // (NBL_E_CAPACITY: the 24-row build hands such a model on to the 48-row build, nimble_amd_dispatch.cpp; from that one it is final)
    if (d->n_boxes > MAX_BOXES) return fail(NBL_E_CAPACITY, "more than " + std::to_string(MAX_BOXES) + " colliders: outside the device path");
    if (d->max_contacts > MAX_CONTACTS) return fail(NBL_E_CAPACITY, "max_contacts above " + std::to_string(MAX_CONTACTS) + " is outside the device path");
    if (d->n_bodies > 64) return fail(NBL_E_UNSUPPORTED, "contact path supports at most 64 bodies");
    if (d->n_dofs > MAX_DOF_CONTACT) return fail(NBL_E_UNSUPPORTED, "contact path supports at most 64 DOFs");
    hc.nBoxes = d->n_boxes;
    hc.nLimitDofs = (int)limitDofs.size();
    for (int k = 0; k < hc.nLimitDofs; k++) {
      const int body = limitDofs[k], j = d->dof_offset[body];
      hc.limitDof[k] = j; hc.limitBody[k] = body; hc.limitLo[k] = hd[j].posLo; hc.limitHi[k] = hd[j].posHi;
    }
    hc.maxContacts = d->max_contacts;
    hc.clippingDepth = d->contact_clipping_depth;
    hc.fallbackCfm = d->fallback_cfm;
    hc.penetrationCorrection = d->penetration_correction != 0;
    auto rootOf = [&](int body) { while (body >= 0 && d->parent[body] >= 0) body = d->parent[body]; return body; };
    auto skelOf = [&](int body) { return d->body_skeleton ? (int)d->body_skeleton[body] : rootOf(body); };   // default: one skeleton per tree
    for (int i = 0; i < d->n_boxes; i++) {
      DevBox& bx = hc.boxes[i];
      bx.body = d->box_body[i];
      if (bx.body >= d->n_bodies) return fail(NBL_E_BADARG, "box collider attached to an unknown body");
      for (int k = 0; k < 12; k++) bx.T[k] = d->box_T[12 * i + k];
      bx.shape = d->box_shape ? d->box_shape[i] : NBL_SHAPE_BOX;
      if (bx.shape != NBL_SHAPE_BOX && bx.shape != NBL_SHAPE_SPHERE && bx.shape != NBL_SHAPE_CAPSULE)
        return fail(NBL_E_UNSUPPORTED, "collider shape outside the device path (box, sphere, capsule)");
      for (int k = 0; k < 3; k++) bx.half[k] = bx.shape == NBL_SHAPE_SPHERE ? d->box_size[3 * i] : 0.5 * d->box_size[3 * i + k];   // sphere: radius
      if (bx.shape == NBL_SHAPE_CAPSULE) {   // (radius, height, -) -> (radius, height / 2, 0)
        bx.half[0] = d->box_size[3 * i]; bx.half[1] = 0.5 * d->box_size[3 * i + 1]; bx.half[2] = 0.0;
        if (!(bx.half[0] > 0.0) || !(bx.half[1] >= 0.0)) return fail(NBL_E_BADARG, "capsule collider needs radius > 0 and height >= 0");
      }
      bx.mu = d->box_mu[i];
      bx.restitution = d->box_restitution ? d->box_restitution[i] : 0.0;
      if (!(bx.restitution >= 0.0)) return fail(NBL_E_BADARG, "negative restitution coefficient");
      if (!(bx.mu >= 0.0)) return fail(NBL_E_BADARG, "negative friction coefficient");   // mu <= 1e-3: frictionless contacts (one live row)
    }
    for (int i = 0; i + 1 < d->n_boxes; i++)
      for (int j = i + 1; j < d->n_boxes; j++) {
        int bi = d->box_body[i], bj = d->box_body[j];
        // same body (or both fixed to the world) - unless the caller merged welded bodies and the two colliders belong to different
        // BodyNodes of the reference (box_node): those are tested like any two bodies of a skeleton (their rows are empty: no relative motion)
        if (bi == bj && !(bi >= 0 && d->box_node && d->box_node[i] != d->box_node[j])) continue;
        if (bi >= 0 && bj >= 0 && skelOf(bi) == skelOf(bj)) {
          // same skeleton (BodyNodeCollisionFilter::ignoresCollision, CollisionFilter.cpp:138-148): only with the self-collision check on, and
          // without the adjacent-body check not between a body and its parent (the description's parent: a ball joint's / a free joint's
          // chain of coincident bodies is one joint)
          const int fi = d->body_self_collision ? d->body_self_collision[bi] : 0, fj = d->body_self_collision ? d->body_self_collision[bj] : 0;
          if (!((fi & 1) && (fj & 1))) continue;
          auto realParent = [&](int b) { if (ballModel) for (int k = expanded.ballComp[b]; k > 0; k--) b = d->parent[b]; return d->parent[b]; };
          const bool adjacent = (d->box_node && d->box_node_parent)
                                    ? (d->box_node_parent[i] == d->box_node[j] || d->box_node_parent[j] == d->box_node[i])
                                    : (realParent(bi) == bj || realParent(bj) == bi);
          if (!((fi & 2) && (fj & 2)) && adjacent) continue;
          hc.selfCollision = 1;
        }
        if ((hc.boxes[i].shape == NBL_SHAPE_CAPSULE) != (hc.boxes[j].shape == NBL_SHAPE_CAPSULE)
            && (hc.boxes[i].shape == NBL_SHAPE_BOX || hc.boxes[j].shape == NBL_SHAPE_BOX))
          return fail(NBL_E_UNSUPPORTED, "a capsule collider can meet a box collider: that pair runs libccd's MPR in the reference (DARTCollide.cpp:4422-4645), outside this path");
        if (hc.nPairs >= MAX_PAIRS) return fail(NBL_E_CAPACITY, "more than " + std::to_string(MAX_PAIRS) + " collider pairs: outside the device path");
        hc.pairA[hc.nPairs] = i;
        hc.pairB[hc.nPairs] = j;
        hc.nPairs++;
      }
    for (int i = 0; i < d->n_boxes; i++)
      for (int j = i + 1; j < d->n_boxes; j++)
        if (d->box_body[i] >= 0 && d->box_body[j] >= 0 && skelOf(d->box_body[i]) != skelOf(d->box_body[j])) multiGroupModel = true;
    hc.oneSkeleton = 1; hc.pad0_ = 0;
    for (int bdy = 1; bdy < d->n_bodies; bdy++) if (skelOf(bdy) != skelOf(0)) hc.oneSkeleton = 0;
    for (int bdy = 0; bdy < d->n_bodies; bdy++) {
      hc.skelOf[bdy] = skelOf(bdy);
      if (hc.skelOf[bdy] < 0 || hc.skelOf[bdy] >= 64) return fail(NBL_E_BADARG, "body_skeleton must lie in [0, 64)");
      uint64_t mask = 0;
      for (int a = bdy; a >= 0; a = d->parent[a]) mask |= (1ull << a);
      hc.ancestors[bdy] = mask;
    }
  }

  if (hasContact) {
    // the per-world LDS images of k_contact_rows_coop ([body][6][row] velocity changes, launchForward) and k_bwd_contact_b_coop
    int nFreeRoots = 0;
    for (int i = 0; i < d->n_bodies; i++) if (d->joint_type[i] == NBL_JOINT_FREE) nFreeRoots++;
#if NBL_GENERAL
    // (the general kernels tile their rows: gen_contact.hip; their LDS images fit for every model of at most 64 bodies)
    const size_t rowsLds = 0, bLds = ((size_t)d->n_bodies * 120 + std::max((size_t)d->n_bodies * 54, (size_t)54 * 64) + MAX_CONTACTS) * sizeof(double);
    (void)nFreeRoots;
#else
    const size_t rowsLds = ((size_t)d->n_bodies * 6 * MAX_ROWS + 12 * MAX_ROWS + 19 * (size_t)d->n_bodies + 54 * (size_t)nFreeRoots + MAX_CONTACTS) * sizeof(double);
    const size_t bLds = ((size_t)d->n_bodies * 120 + std::max((size_t)d->n_bodies * 54, (size_t)54 * MAX_ROWS) + MAX_CONTACTS) * sizeof(double);
#endif
    if (std::max(rowsLds, bLds) > 160u * 1024u)
      return fail(NBL_E_UNSUPPORTED, "bodies x LCP rows exceed the 160 kB of LDS of a compute unit (" + std::to_string(d->n_bodies) + " device bodies, " +
                                         std::to_string(MAX_ROWS) + " rows): " + (MAX_CONTACTS > 8 ? "use max_contacts <= 8 or fewer bodies" : "fewer bodies"));
  }

  nbl_model* m = new nbl_model();
  m->hasContact = hasContact;
  m->multiGroup = multiGroupModel;
  {
    SavedLayout& L = m->lay;
    const int n = d->n_dofs;
    L.n = n; L.q = 0; L.v = n; L.tau = 2 * n;
    if (!hasContact) { L.vpre = L.w = L.nc = L.contacts = L.x = L.b = L.cls = L.cfm = L.pflag = L.rest = -1; L.total = 3 * n;
                       L.A = L.massed = L.aall = L.pinv = -1; L.dense = 0; L.ldr = MAX_ROWS; }
    else {
      L.vpre = 3 * n; L.w = 4 * n; L.nc = 5 * n; L.contacts = L.nc + 1; L.x = L.contacts + MAX_CONTACTS * CR_SIZE;
      L.b = L.x + MAX_ROWS; L.cls = L.b + MAX_ROWS; L.cfm = L.cls + MAX_ROWS; L.pflag = L.cfm + MAX_ROWS; L.rest = L.pflag + 1; L.total = L.rest + MAX_CONTACTS;   // cfm: one constant per row (its group's)
#if NBL_GENERAL
      const int ldr = genLeadingDim(d->max_contacts);      // the general builds size record and scratch by the MODEL's rows, not by their cap
#else
      const int ldr = MAX_ROWS;
#endif
      L.ldr = ldr;
      L.A = 0; L.massed = L.A + ldr * ldr; L.aall = L.massed + n * ldr; L.pinv = L.aall + n * ldr;
      L.dense = L.pinv + ldr * ldr;
    }
    bool saveTree = true;   // trade 8 * WS_KEEP * n_bodies bytes per world and step for the ABA re-run of the backward pass
    if (const char* e0 = getenv("NBL_SAVE_TREE")) saveTree = atoi(e0) != 0;
    bool coopTree = true;
    const int nbp = (d->n_bodies + 3) & ~3;
    // the lane = body tree kernels keep the sweep state of `wpb` worlds + one copy of the model constants in LDS; pick the
    // worlds per workgroup that maximise the resident wavefronts per CU (160 kB)
    int nFree = 0;
    for (int i = 0; i < d->n_bodies; i++) if (d->joint_type[i] == NBL_JOINT_FREE) hb[i].freeIdx = nFree++;
    m->mdl.nFree = nFree;
    const size_t modelLds = (size_t)d->n_bodies * sizeof(DevBody) + (size_t)d->n_dofs * sizeof(DevDof);
    // worlds packed into one wavefront of the tree kernels (lane = body of one of them): 64 / nbp, e.g. 4 for the 16-body metric
    // model.  NBL_TREE_PACK caps it (1 = one world per wavefront).  The packed worlds of a wave need their LDS images together.
    int wpw = std::max(1, 64 / nbp);
    if (const char* e11 = getenv("NBL_TREE_PACK")) wpw = std::max(1, std::min(wpw, atoi(e11)));
    {
      const size_t worst = (size_t)std::max(coopWorldDoubles<PROF_FWD>(nbp, nFree), coopWorldDoubles<PROF_BWD>(nbp, nFree)) * sizeof(double);
      while (wpw > 1 && modelLds + (size_t)wpw * worst > 160u * 1024u) wpw--;
    }
    m->mdl.pad = wpw;
    auto pickWpb = [&](size_t perWorld, int& wpb, size_t& bytes) {   // wavefronts per workgroup
      int best = 0, bestWaves = 0;
      int wCap = TREE_WPB_MAX;
      if (const char* e17 = getenv("NBL_TREE_WPB")) wCap = std::max(1, std::min(TREE_WPB_MAX, atoi(e17)));
      for (int w = 1; w <= wCap; w++) {
        const size_t need = modelLds + (size_t)w * wpw * perWorld;
        if (need > 160u * 1024u) break;
        int waves = (int)((160u * 1024u) / need) * w;
        if (waves > 8) waves = 8;   // these kernels use 256 VGPRs: two wavefronts per SIMD is all a CU can hold
        if (waves > bestWaves) { bestWaves = waves; best = w; }
      }
      wpb = best; bytes = modelLds + (size_t)best * wpw * perWorld;
    };
    pickWpb((size_t)coopWorldDoubles<PROF_FWD>(nbp, nFree) * sizeof(double), m->wpbFwd, m->ldsFwd);
    pickWpb((size_t)coopWorldDoubles<PROF_BWD>(nbp, nFree) * sizeof(double), m->wpbBwd, m->ldsBwd);
    const size_t coopTreeLds = (m->wpbFwd > 0 && m->wpbBwd > 0) ? 0 : (size_t)1 << 30;
    if (const char* e5 = getenv("NBL_COOP_TREE")) coopTree = atoi(e5) != 0;
    if (coopTreeLds > 160u * 1024u) coopTree = false;
    // (k_bwd_contact_b_coop's per-world LDS image, (174 n_bodies + 54 * 24 + 8) doubles, fits 160 kB for every model the contact path accepts: <= 64 bodies)
    if (const char* e6 = getenv("NBL_COOP_FINAL")) m->coopFinal = atoi(e6) != 0;
    if (const char* e10 = getenv("NBL_AUX_OVERLAP")) m->auxOverlap = atoi(e10) != 0;
    if (const char* e13 = getenv("NBL_DETECT_SPLIT")) m->detectSplit = atoi(e13) != 0;
    if (const char* e14 = getenv("NBL_FUSED_CASCADE")) m->fusedCascade = atoi(e14) != 0;
    if (const char* e15 = getenv("NBL_FUSED_DETECT")) m->fusedDetect = atoi(e15) != 0;
    if (const char* e19 = getenv("NBL_DETECT_WL")) m->detectWl = atoi(e19);
    m->nPairs = hc.nPairs;
    if (hasContact) {
      uint64_t need = 0ull;
      for (int i = 0; i < hc.nBoxes; i++) if (hc.boxes[i].body >= 0) need |= hc.ancestors[hc.boxes[i].body];
      m->fkBodies = __builtin_popcountll(need);
    }
    {
      const size_t oneWorld = ((size_t)d->n_bodies * 6 * MAX_ROWS + 12 * MAX_ROWS + 19 * (size_t)d->n_bodies + 54 * (size_t)nFree + MAX_CONTACTS) * sizeof(double);
      m->rowsPack = (MAX_ROWS <= 32 && d->n_bodies <= 32 && 2 * oneWorld <= 160u * 1024u) ? 2 : 1;
      if (const char* e16 = getenv("NBL_ROWS_PACK")) m->rowsPack = std::max(1, std::min(m->rowsPack, atoi(e16)));
    }
    // measured (MI355X, B = 4096, world-frame sweeps): 5.5 vs 3.8 M/s with colliders, 16.4 vs 11.0 M/s without
    m->coopTree = coopTree && saveTree && d->n_bodies <= 64 && d->n_dofs <= 64;
    if (const char* e7 = getenv("NBL_COOP_TREE_FORCE")) m->coopTree = atoi(e7) != 0 && saveTree && d->n_bodies <= 64 && d->n_dofs <= 64 && coopTreeLds <= 160u * 1024u;
    L.treeNbp = m->coopTree ? nbp : 0;
    L.treeRows = !saveTree ? 0 : (m->coopTree ? TREE_ROWS * nbp + TREE_FREE * nFree : d->n_bodies * WS_KEEP);
    m->mdl.nbp = nbp;
  }
  m->device = device;
  if (const char* e1 = getenv("NBL_TREE_LANES")) m->treeLanes = atoi(e1);
  m->nb = d->n_bodies; m->n = d->n_dofs; m->k = d->n_action; m->maxContacts = d->max_contacts;
  m->mdl.hasBounce = 0; m->mdl.hasCapsule = 0;
  if (hasContact)
    for (int pi = 0; pi < hc.nPairs; pi++)
      if (hc.boxes[hc.pairA[pi]].restitution * hc.boxes[hc.pairB[pi]].restitution > 1e-3) m->mdl.hasBounce = 1;
  if (hasContact)
    for (int i = 0; i < hc.nBoxes; i++)
      if (hc.boxes[i].shape == NBL_SHAPE_CAPSULE) m->mdl.hasCapsule = 1;
  if (m->mdl.hasBounce && !(m->coopTree && m->coopFinal)) {
    nbl_model_destroy(m);
    return fail(NBL_E_UNSUPPORTED, "restitution needs the wavefront-per-world kernels (NBL_COOP_TREE / NBL_COOP_FINAL are off or the model does not fit them)");
  }
  if (ballModel && !(m->coopTree && m->coopFinal)) {
    nbl_model_destroy(m);
    return fail(NBL_E_UNSUPPORTED, "ball joints need the wavefront-per-world kernels (NBL_COOP_TREE / NBL_COOP_FINAL / NBL_SAVE_TREE are off or the model, three device bodies per ball joint, does not fit them)");
  }
  m->userBodies = userBodies;
  if (ballModel) m->bodyMap = expanded.bodyMap;
  m->mdl.nb = m->nb; m->mdl.n = m->n; m->mdl.nAction = m->k; m->mdl.b0 = 0; m->mdl.b1 = 0;   // mdl.pad: worlds per wavefront, set above
  if (const char* e9 = getenv("NBL_SLICES")) m->slices = atoi(e9);
  m->mdl.maxLevel = 0; m->mdl.maxRank = 0;
  for (const DevBody& hbI : hb) { if (hbI.level > m->mdl.maxLevel) m->mdl.maxLevel = hbI.level; if (hbI.parent >= 0 && hbI.rank > m->mdl.maxRank) m->mdl.maxRank = hbI.rank; }
  for (int k = 0; k < 3; k++) m->mdl.gravity[k] = d->gravity[k];
  m->mdl.dt = d->dt;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc((void**)&m->dBodies, sizeof(DevBody) * hb.size());
  if (e == hipSuccess) e = hipMalloc((void**)&m->dDofs, sizeof(DevDof) * hd.size());
  m->hBodies = hb;
  if (e == hipSuccess) e = hipMemcpy(m->dBodies, hb.data(), sizeof(DevBody) * hb.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->dDofs, hd.data(), sizeof(DevDof) * hd.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && hasContact) e = hipMalloc((void**)&m->dContact, sizeof(DevContactModel));
  if (e == hipSuccess && hasContact) e = hipMemcpy(m->dContact, &hc, sizeof(DevContactModel), hipMemcpyHostToDevice);
  // (k_contact_detect: 160 kB less its static arrays - the remembered points and the clip polygons)
  if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_contact_detect, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             std::min(104 * 1024, 160 * 1024 - (SEEN_POINTS * 3 * DETECT_LS + 48 * DETECT_LS) * (int)sizeof(double)));
#if NBL_GENERAL
  if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_contact_rows_gen, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_bwd_contact_b_gen<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_bwd_contact_b_gen<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#else
  if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_contact_rows_coop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if constexpr (MAX_ROWS <= 32)
    if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_contact_rows_coop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_bwd_contact_b_coop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess && hasContact) e = hipFuncSetAttribute((const void*)k_bwd_contact_b_coop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_step_forward_coop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_forward_detect_coop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bwd_recompute_coop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bwd_final_coop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    std::string msg = std::string("model upload failed: ") + hipGetErrorString(e);
    nbl_model_destroy(m);
    return fail(NBL_E_HIP, msg);
  }
  *out = m;
  return NBL_OK;
}

void nbl_model_destroy(nbl_model* m) {
  if (!m) return;
  for (auto& t : m->pending) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
  for (auto st : m->side) hipStreamDestroy(st);
  for (auto st : m->aux) hipStreamDestroy(st);
  for (auto ev : m->auxFork) hipEventDestroy(ev);
  for (auto ev : m->auxJoin) hipEventDestroy(ev);
  for (auto ev : m->sideDone) hipEventDestroy(ev);
  if (m->fork) hipEventDestroy(m->fork);
  if (m->dBodies) hipFree(m->dBodies);
  if (m->dDofs) hipFree(m->dDofs);
  if (m->dContact) hipFree(m->dContact);
  if (m->dParams) hipFree(m->dParams);
  if (m->staging) hipHostFree(m->staging);
  if (m->staged) hipEventDestroy(m->staged);
  delete m;
}

int32_t nbl_model_num_dofs(const nbl_model* m) { return m ? m->n : 0; }
int32_t nbl_model_num_action(const nbl_model* m) { return m ? m->k : 0; }
int32_t nbl_model_lcp_rows(const nbl_model* m) { return (m && m->hasContact) ? MAX_ROWS + 1 : 0; }
int32_t nbl_model_max_contacts(const nbl_model* m) { return (m && m->hasContact) ? MAX_CONTACTS : 0; }

// bytes of the workspace before the scratch matrices of the general instantiation (tree slots, contact-backward rows, the slice lists)
static size_t workspaceHeadBytes(const nbl_model* m, int64_t B) {
  const size_t head = ((size_t)m->nb * WS_PER_BODY + (m->hasContact ? LB_TOTAL : 0)) * sizeof(double) * (size_t)B +
                      (m->hasContact ? ((size_t)B + 16) * sizeof(int32_t) : 0);
  return (head + 255) & ~(size_t)255;
}
size_t nbl_workspace_bytes(const nbl_model* m, int64_t B) {
  if (!m || B <= 0) return 0;
#if NBL_GENERAL
  // + the per-world scratch matrices of the general contact kernels (gen_lcp_dev.hpp: 5 matrices of ld x ld + 16 vectors, ld = the model's rows:
  // 1.5 MB per world at 64 slots, 62 kB x 5 at 28)
  if (m->hasContact) return workspaceHeadBytes(m, B) + genScratchDoubles(m->lay.ldr) * sizeof(double) * (size_t)B;
#endif
  return ((size_t)m->nb * WS_PER_BODY + (m->hasContact ? LB_TOTAL : 0)) * sizeof(double) * (size_t)B +
         (m->hasContact ? ((size_t)B + 16) * sizeof(int32_t) : 0);
}
size_t nbl_saved_bytes(const nbl_model* m, int64_t B) {
  if (!m || B <= 0) return 0;
  return ((size_t)m->lay.total + (size_t)m->lay.dense + (size_t)m->lay.treeRows) * sizeof(double) * (size_t)B;
}


// Worlds per workgroup of the one-world-per-lane kernels.  Measured on MI355X at B = 4096 (a sweep over nbl_set_launch_lanes):
// 16-lane workgroups are marginally faster than 64 for the tree kernels (more CUs busy), smaller ones lose (the
// kernels are bound by memory transactions per wave-instruction, which do not shrink with the lane count).
static int pickLanes(int64_t B, int requested, int maxLanes) {
  int l = requested > 0 ? requested : (maxLanes > 16 && B < 65536 ? 16 : maxLanes);
  if (l > maxLanes) l = maxLanes;
  if (l < 1) l = 1;
  return l;
}
static void beginTiming(nbl_model* m, hipStream_t s, int kernel) {
  if (!m->timingNow) return;
  TimedLaunch t;
  hipEventCreate(&t.start);
  hipEventCreate(&t.stop);
  t.kernel = kernel;
  hipEventRecord(t.start, s);
  m->pending.push_back(t);
}
static void endTiming(nbl_model* m, hipStream_t s) {
  if (!m->timingNow) return;
  hipEventRecord(m->pending.back().stop, s);
}
// NBL_DEBUG_SYNC=1 (developer switch): wait for every kernel and name it on stderr - a device fault then points at its launch
static const bool g_dbgSync = getenv("NBL_DEBUG_SYNC") && atoi(getenv("NBL_DEBUG_SYNC")) != 0;
#define TIMED(kid, launch) do { beginTiming(m, s, kid); launch; endTiming(m, s); \
    if (g_dbgSync) { const hipError_t de_ = hipStreamSynchronize(s); fprintf(stderr, "[nbl] %s: %s\n", kKernelNames[kid], hipGetErrorString(de_)); } } while (0)

// the kernels of one forward step for the worlds [b0, b1) on stream s (slice si owns fail list / counter si)
static int32_t launchForward(nbl_model* m, int64_t B, int si, int64_t b0, int64_t b1, hipStream_t s, const double* state,
                             const double* action, const double* lcp_cache_in, double* next_state, double* lcp_cache_out,
                             void* saved, uint32_t* status, void* workspace) {
  const int tl = pickLanes(B, m->treeLanes, 64);
  double* lws = (double*)workspace + (size_t)m->nb * WS_PER_BODY * (size_t)B;
  int32_t* failListAll = (int32_t*)(lws + (size_t)LB_TOTAL * (size_t)B);
  uint32_t* failCountAll = (uint32_t*)(failListAll + B);

    const int64_t cnt = b1 - b0;
    DevModel mdl = m->mdl;
    mdl.b0 = b0; mdl.b1 = b1;
    dim3 grid((unsigned)((cnt + tl - 1) / tl)), block(tl);
    const size_t treeLds = m->ldsFwd;
    const int64_t perBlockF = (int64_t)std::max(1, m->wpbFwd) * std::max(1, (int)m->mdl.pad);   // worlds per workgroup
    const dim3 treeGrid((unsigned)((cnt + perBlockF - 1) / perBlockF)), treeBlock(64 * std::max(1, m->wpbFwd));
    // lanes per world of the narrow phase: the collider pairs of a world side by side (k_contact_detect / contactDetectBody)
    const int ppwD = !m->detectSplit ? 1 : (m->nPairs >= 4 ? 4 : (m->nPairs >= 2 ? 2 : 1));
    int wlD = std::min(tl, 64 / ppwD);                             // worlds per narrow-phase workgroup
    if (m->detectWl > 0) wlD = std::max(1, std::min(wlD, m->detectWl));
    const int seenPts = NBL_GENERAL ? std::min(2 * m->maxContacts, (int)SEEN_POINTS) : (int)SEEN_POINTS;   // (contactDetectBody)
    auto detectLdsFor = [&](int wl) -> size_t {
      return ((size_t)seenPts * 3 + 48) * (size_t)(wl * ppwD) * sizeof(double) +
             (ppwD > 1 ? (size_t)wl * (ppwD - 1) * (8 * CR_SIZE) * sizeof(double) + (size_t)wl * (ppwD - 1) * sizeof(int) : 0) +
             (size_t)m->nb * sizeof(DevBody) + 32 +   // + the body constants of the narrow phase's own forward kinematics
             sizeof(DevContactModel) +                // + the collider model
             (size_t)wl * m->fkBodies * 12 * sizeof(double);   // + the joint transforms of the bodies on the collider chains
    };
    // (many colliders on long chains - a humanoid on the 48-row build: fewer worlds per workgroup rather than the narrow phase as a launch of its own)
    while (wlD > 1 && detectLdsFor(wlD) > 160u * 1024u) wlD /= 2;
    const size_t detectLds = detectLdsFor(wlD);
    const bool fusedDetect = m->hasContact && m->coopTree && saved && m->fusedDetect && std::max(treeLds, detectLds) <= 160u * 1024u;
    if (fusedDetect) {
      const int nDetect = (int)((cnt + wlD - 1) / wlD);
      TIMED(K_FWD_DETECT, hipLaunchKernelGGL(k_forward_detect_coop, dim3(treeGrid.x + (unsigned)nDetect), treeBlock, std::max(treeLds, detectLds), s, mdl,
                                             m->dBodies, m->dDofs, m->dContact, B, state, action, next_state, (double*)saved, status, m->lay,
                                             (double*)workspace, failCountAll + si, ppwD, wlD, nDetect));
    } else if (m->coopTree && (saved || !m->hasContact))
      TIMED(K_FWD_COOP, hipLaunchKernelGGL(k_step_forward_coop, treeGrid, treeBlock, treeLds, s, mdl, m->dBodies, m->dDofs, B,
                                           state, action, next_state, (double*)saved, status, m->lay, m->hasContact ? 1 : 0));
    else
      TIMED(K_FWD, hipLaunchKernelGGL(k_step_forward, grid, block, 0, s, mdl, m->dBodies, m->dDofs, B, state, action, next_state,
                                      (double*)saved, status, (double*)workspace, m->lay));
    if (m->hasContact) {
      if (!fusedDetect) {
        // lanes per world of the narrow phase: the collider pairs of a world side by side (k_contact_detect)
        const int ppw = !m->detectSplit ? 1 : (m->nPairs >= 4 ? 4 : (m->nPairs >= 2 ? 2 : 1));
        // worlds per workgroup: its threads (wl * ppw) index the kernel's static LDS slices, whose lane stride is DETECT_LS (16 in the
        // general builds: their 128 remembered points per world do not fit 64 lanes' worth of LDS)
        const int wl = std::max(1, std::min(tl, DETECT_LS / ppw));
        const size_t stageBytes = ppw > 1 ? (size_t)wl * (ppw - 1) * (8 * CR_SIZE) * sizeof(double) + (size_t)wl * (ppw - 1) * sizeof(int) : 0;
        TIMED(K_DETECT, hipLaunchKernelGGL(k_contact_detect, dim3((unsigned)((cnt + wl - 1) / wl)), dim3(wl * ppw), stageBytes, s, mdl, m->dBodies,
                                           m->dContact, B, (double*)saved, m->lay, status, (double*)workspace, m->coopTree ? 0 : 1,
                                           failCountAll + si, ppw));
      }
#if NBL_GENERAL
      {
        // the general kernels (gen_contact.hip), one wavefront per world: rows in tiles of `ts` (what the [body][6][ts] field of the
        // impulse tests leaves of the LDS), then the whole solver cascade of a world in one launch
        double* gws = (double*)((char*)workspace + workspaceHeadBytes(m, B));
        // ... a tile no wider than the model's rows need, and small enough for three worlds per CU (48 kB) while it still holds 32 rows: a
        // 24-slot model runs its eight-contact worlds in one pass of 32 rows at 35 kB instead of one of 64 rows at 71 kB (two worlds per CU)
        int ts = 64;
        const int ldrRows = m->lay.ldr;
        auto rowsLdsFor = [&](int t) -> size_t {
          return ((size_t)12 * ldrRows + 22 * (size_t)m->nb + 54 * (size_t)m->mdl.nFree + (size_t)6 * m->nb * t) * sizeof(double) + 2 * MAX_CONTACTS * sizeof(int);
        };
        while (ts > 8 && rowsLdsFor(ts) > 150u * 1024u) ts /= 2;
        while (ts > 32 && (ts / 2 >= ldrRows || rowsLdsFor(ts) > 48u * 1024u)) ts /= 2;
        if (const char* e = getenv("NBL_ROWS_TS")) { const int t = atoi(e); if (t >= 8 && t <= 64 && rowsLdsFor(t) <= 150u * 1024u) ts = t; }
        TIMED(K_ROWS_COOP, hipLaunchKernelGGL(k_contact_rows_gen, dim3((unsigned)cnt), dim3(64), rowsLdsFor(ts), s, mdl, m->dBodies, m->dContact, B,
                                              (double*)saved, m->lay, (const double*)workspace, ts));
        TIMED(K_SOLVE_COOP, hipLaunchKernelGGL(k_contact_solve_gen, dim3((unsigned)cnt), dim3(64), genSolveLdsBytes(m->lay.ldr), s, mdl, m->dContact, B, (double*)saved, m->lay,
                                               lcp_cache_in, lcp_cache_out, next_state, status, gws));
      }
      (void)lws; (void)failListAll;
#else
      {
        // worlds per wavefront of the row kernel: two (lanes 0..31 | 32..63) in the 24-row build when the model's bodies fit a half
        const int wpw = m->rowsPack;
        const size_t rowsLds = (size_t)wpw * ((size_t)m->nb * 6 * MAX_ROWS + 12 * MAX_ROWS + 19 * (size_t)m->nb + 54 * (size_t)m->mdl.nFree + MAX_CONTACTS) *
                               sizeof(double);   // acc, Fw, Sw/AISw/Vw/psi, free-joint blocks, contact bodies
        const dim3 rowsGrid((unsigned)((cnt + wpw - 1) / wpw));
        if constexpr (MAX_ROWS <= 32) {
          if (wpw == 2)
            TIMED(K_ROWS_COOP, hipLaunchKernelGGL(k_contact_rows_coop<2>, rowsGrid, dim3(64), rowsLds, s, mdl, m->dBodies, m->dContact, B,
                                                  (double*)saved, m->lay, (const double*)workspace));
        }
        if (wpw == 1)
          TIMED(K_ROWS_COOP, hipLaunchKernelGGL(k_contact_rows_coop<1>, rowsGrid, dim3(64), rowsLds, s, mdl, m->dBodies, m->dContact, B,
                                                (double*)saved, m->lay, (const double*)workspace));
      }
      int32_t* failList = failListAll + b0;          // the slice's own compacted list and counter
      uint32_t* failCount = failCountAll + si;   // zeroed by k_contact_detect
      const bool mg = m->multiGroup;
      TIMED(K_SOLVE_COOP, hipLaunchKernelGGL(mg ? k_contact_solve_coop<true> : k_contact_solve_coop<false>, dim3((unsigned)cnt), dim3(64), 0, s, mdl, m->dContact, B,
                                             (double*)saved, m->lay, lcp_cache_in, lcp_cache_out, next_state, status, lws,
                                             failList, failCount));
      if (m->fusedCascade) {
        TIMED(K_CASCADE_FUSED, hipLaunchKernelGGL(mg ? k_contact_cascade_fused<true> : k_contact_cascade_fused<false>, dim3((unsigned)cnt), dim3(128), 0, s, mdl, m->dContact, B,
                                                  (double*)saved, m->lay, lcp_cache_out, next_state, status, lws, failList, failCount));
      } else {
        TIMED(K_CASCADE_COOP, hipLaunchKernelGGL(mg ? k_contact_cascade_stages<true> : k_contact_cascade_stages<false>, dim3((unsigned)cnt), dim3(128), 0, s, mdl, m->dContact, B,
                                                 (double*)saved, m->lay, lws, failList, failCount));
        TIMED(K_CASCADE_FINAL, hipLaunchKernelGGL(mg ? k_contact_cascade_final<true> : k_contact_cascade_final<false>, dim3((unsigned)cnt), dim3(64), 0, s, mdl, m->dContact, B,
                                                  (double*)saved, m->lay, lcp_cache_out, next_state, status, lws, failList, failCount));
      }
#endif
    }
  return NBL_OK;
}

int32_t nbl_step_forward(nbl_model* m, int64_t B, const double* state, const double* action, const double* lcp_cache_in,
                         double* next_state, double* lcp_cache_out, void* saved, uint32_t* status, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!m || !state || !action || !next_state || !workspace) return fail(NBL_E_BADARG, "null argument");
  if (m->hasContact && !saved) return fail(NBL_E_BADARG, "models with colliders need the saved record (it doubles as the contact scratch)");
  if (B <= 0) return fail(NBL_E_BADARG, "B must be positive");
  if (workspace_bytes < nbl_workspace_bytes(m, B)) return fail(NBL_E_WORKSPACE, "workspace too small");
  m->timingNow = m->timing && (m->fwdCalls++ % m->timingPeriod == 0);
  const int32_t rc = forSlices(m, B, slicesFor(m, B), (hipStream_t)stream, [&](int si, int64_t b0, int64_t b1, hipStream_t s) -> int32_t {
    return launchForward(m, B, si, b0, b1, s, state, action, lcp_cache_in, next_state, lcp_cache_out, saved, status, workspace);
  });
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipGetLastError());
  return NBL_OK;
}

// the kernels of one backward step for the worlds [b0, b1) on stream s
static int32_t launchBackward(nbl_model* m, int64_t B, int si, int64_t b0, int64_t b1, hipStream_t s, const void* saved,
                              const double* grad_next_state, double* grad_state, double* grad_action, void* workspace) {
  const int tl = pickLanes(B, m->treeLanes, 64);
  SavedLayout layLanes = m->lay;   // for the one-world-per-lane sweep fed by k_tree_to_lanes: kept slots in the workspace
  layLanes.treeRows = 0; layLanes.treeNbp = 0;
  double* lws = (double*)workspace + (size_t)m->nb * WS_PER_BODY * (size_t)B;
  double* sv = (double*)const_cast<void*>(saved);

    const int64_t cnt = b1 - b0;
    DevModel mdl = m->mdl;
    mdl.b0 = b0; mdl.b1 = b1;
    dim3 grid((unsigned)((cnt + tl - 1) / tl)), block(tl);
    const size_t treeLds = m->ldsBwd;
    const int64_t perBlockB = (int64_t)std::max(1, m->wpbBwd) * std::max(1, (int)m->mdl.pad);   // worlds per workgroup
    const dim3 treeGrid((unsigned)((cnt + perBlockB - 1) / perBlockB)), treeBlock(64 * std::max(1, m->wpbBwd));
    const dim3 t2lGrid((unsigned)((m->lay.treeRows + 31) / 32), (unsigned)((cnt + 31) / 32));
    if (!m->hasContact && m->coopTree && !m->coopFinal) {
      TIMED(K_TREE_TO_LANES, hipLaunchKernelGGL(k_tree_to_lanes, t2lGrid, dim3(256), 0, s, (const double*)saved, m->lay, m->nb, B, b0, b1, (double*)workspace, m->dBodies));
      TIMED(K_BWD, hipLaunchKernelGGL(k_step_backward, grid, block, 0, s, mdl, m->dBodies, m->dDofs, B, (const double*)saved, layLanes,
                                      grad_next_state, grad_state, grad_action, (double*)workspace, 1));
    } else if (!m->hasContact && m->coopTree) {
      TIMED(K_BWD_FINAL_COOP, hipLaunchKernelGGL(k_bwd_final_coop, treeGrid, treeBlock, treeLds, s, mdl, m->dBodies, m->dDofs, B,
                                                 (const double*)saved, m->lay, grad_next_state, grad_state, grad_action,
                                                 (const double*)nullptr, m->nParams > 0 ? (double*)workspace : (double*)nullptr));
    } else if (!m->hasContact) {
      TIMED(K_BWD, hipLaunchKernelGGL(k_step_backward, grid, block, 0, s, mdl, m->dBodies, m->dDofs, B, (const double*)saved, m->lay,
                                      grad_next_state, grad_state, grad_action, (double*)workspace, 0));
    } else {
      // lambda1 = M^-1 g (tree kernel) and the dense contact adjoint do not depend on each other: with the wavefront-per-world
      // kernels the first runs on the slice's auxiliary stream, the second on its own stream, and they join before
      // k_bwd_contact_b_coop, which needs both.
      const bool forkRecompute = m->coopTree && m->auxOverlap;
      if (forkRecompute) {
        const int32_t rc = ensureAux(m, si + 1);
        if (rc != NBL_OK) return rc;
        HIP_TRY(hipEventRecord(m->auxFork[si], s));
        HIP_TRY(hipStreamWaitEvent(m->aux[si], m->auxFork[si], 0));
        {
          hipStream_t s = m->aux[si];   // shadows the slice's stream for the launch and its timing events
          TIMED(K_RECOMPUTE_COOP, hipLaunchKernelGGL(k_bwd_recompute_coop, treeGrid, treeBlock, treeLds, s, mdl, m->dBodies,
                                                     m->dDofs, B, (const double*)saved, m->lay, grad_next_state, lws));
        }
        HIP_TRY(hipEventRecord(m->auxJoin[si], m->aux[si]));
      } else if (m->coopTree)
        TIMED(K_RECOMPUTE_COOP, hipLaunchKernelGGL(k_bwd_recompute_coop, treeGrid, treeBlock, treeLds, s, mdl, m->dBodies,
                                                   m->dDofs, B, (const double*)saved, m->lay, grad_next_state, lws));
      else
        TIMED(K_RECOMPUTE, hipLaunchKernelGGL(k_bwd_recompute, grid, block, 0, s, mdl, m->dBodies, m->dDofs, B,
                                              (const double*)saved, m->lay, grad_next_state, (double*)workspace, lws));
#if NBL_GENERAL
      double* gws = (double*)((char*)workspace + workspaceHeadBytes(m, B));
      TIMED(K_BWD_A_COOP, hipLaunchKernelGGL(k_bwd_contact_a_gen, dim3((unsigned)cnt), dim3(64), 0, s, mdl, m->dContact, B, sv,
                                             m->lay, grad_next_state, lws, gws));
      if (forkRecompute) HIP_TRY(hipStreamWaitEvent(s, m->auxJoin[si], 0));
      {
        const size_t bLds = ((size_t)m->nb * 120 + std::max((size_t)m->nb * 54, (size_t)54 * 64) + MAX_CONTACTS) * sizeof(double);   // FW D {tmp | TF} TW contact bodies
        if (mdl.hasCapsule)
          TIMED(K_BWD_B_COOP, hipLaunchKernelGGL(k_bwd_contact_b_gen<true>, dim3((unsigned)cnt), dim3(64), bLds, s, mdl, m->dBodies, m->dContact, B,
                                                 sv, m->lay, (const double*)workspace, lws));
        else
          TIMED(K_BWD_B_COOP, hipLaunchKernelGGL(k_bwd_contact_b_gen<false>, dim3((unsigned)cnt), dim3(64), bLds, s, mdl, m->dBodies, m->dContact, B,
                                                 sv, m->lay, (const double*)workspace, lws));
      }
      if (mdl.hasBounce)
        TIMED(K_BWD_BOUNCE, hipLaunchKernelGGL(k_bwd_bounce_gen, dim3((unsigned)cnt), dim3(64), genRowsDoubles(genRowsCap(MAX_CONTACTS)) * sizeof(double), s, mdl, m->dBodies, B, (const double*)saved, m->lay,
                                               grad_next_state, lws, gws));
#else
      TIMED(K_BWD_A_COOP, hipLaunchKernelGGL(k_bwd_contact_a_coop, dim3((unsigned)cnt), dim3(64), 0, s, mdl, m->dContact, B, sv,
                                             m->lay, grad_next_state, lws));
      if (forkRecompute) HIP_TRY(hipStreamWaitEvent(s, m->auxJoin[si], 0));
      {
        const size_t bLds = ((size_t)m->nb * 120 + std::max((size_t)m->nb * 54, (size_t)54 * MAX_ROWS) + MAX_CONTACTS) * sizeof(double);   // FW D {tmp | TF} TW contact bodies
        if (mdl.hasCapsule)
          TIMED(K_BWD_B_COOP, hipLaunchKernelGGL(k_bwd_contact_b_coop<true>, dim3((unsigned)cnt), dim3(64), bLds, s, mdl, m->dBodies, m->dContact, B,
                                                 sv, m->lay, (const double*)workspace, lws));
        else
          TIMED(K_BWD_B_COOP, hipLaunchKernelGGL(k_bwd_contact_b_coop<false>, dim3((unsigned)cnt), dim3(64), bLds, s, mdl, m->dBodies, m->dContact, B,
                                                 sv, m->lay, (const double*)workspace, lws));
      }
      if (mdl.hasBounce)
        TIMED(K_BWD_BOUNCE, hipLaunchKernelGGL(k_bwd_bounce, dim3((unsigned)cnt), dim3(64), 0, s, mdl, m->dBodies, B, (const double*)saved, m->lay,
                                               grad_next_state, lws));
#endif
      if (m->coopTree && !m->coopFinal) {
        TIMED(K_TREE_TO_LANES, hipLaunchKernelGGL(k_tree_to_lanes, t2lGrid, dim3(256), 0, s, (const double*)saved, m->lay, m->nb, B, b0, b1, (double*)workspace, m->dBodies));
        TIMED(K_BWD_FINAL, hipLaunchKernelGGL(k_bwd_final, grid, block, 0, s, mdl, m->dBodies, m->dDofs, B, (const double*)saved, layLanes,
                                              grad_next_state, grad_state, grad_action, (double*)workspace, (const double*)lws, 1));
      } else if (m->coopTree)
        TIMED(K_BWD_FINAL_COOP, hipLaunchKernelGGL(k_bwd_final_coop, treeGrid, treeBlock, treeLds, s, mdl, m->dBodies, m->dDofs,
                                                   B, (const double*)saved, m->lay, grad_next_state, grad_state, grad_action,
                                                   (const double*)lws, m->nParams > 0 ? (double*)workspace : (double*)nullptr));
      else
        TIMED(K_BWD_FINAL, hipLaunchKernelGGL(k_bwd_final, grid, block, 0, s, mdl, m->dBodies, m->dDofs, B, (const double*)saved, m->lay,
                                              grad_next_state, grad_state, grad_action, (double*)workspace, (const double*)lws, 0));
    }
  return NBL_OK;
}

int32_t nbl_step_backward(nbl_model* m, int64_t B, const void* saved, const double* grad_next_state, double* grad_state,
                          double* grad_action, void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !saved || !grad_next_state || !grad_state || !grad_action || !workspace) return fail(NBL_E_BADARG, "null argument");
  if (B <= 0) return fail(NBL_E_BADARG, "B must be positive");
  if (workspace_bytes < nbl_workspace_bytes(m, B)) return fail(NBL_E_WORKSPACE, "workspace too small");
  m->timingNow = m->timing && (m->bwdCalls++ % m->timingPeriod == 0);
  const int32_t rc = forSlices(m, B, slicesFor(m, B), (hipStream_t)stream, [&](int si, int64_t b0, int64_t b1, hipStream_t s) -> int32_t {
    return launchBackward(m, B, si, b0, b1, s, saved, grad_next_state, grad_state, grad_action, workspace);
  });
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipGetLastError());
  return NBL_OK;
}

// ---- inertia ("mass") parameters: World::setMasses / lossWrtMass (World.cpp:1821-1824, BackpropSnapshot.cpp:167-179) ----
extern "C++" {   // (internal helpers: C++ linkage keeps them local to this instantiation of the library)
namespace {
// keeps the calling thread's current device across an entry point that has to work on the model's device
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// pinned staging area of a model (stream-ordered uploads read from it; `staged` marks the last upload still reading it)
int32_t ensureStaging(nbl_model* m, size_t bytes) {
  if (m->stagingBytes >= bytes) return NBL_OK;
  if (m->staging) { HIP_TRY(hipEventSynchronize(m->staged)); HIP_TRY(hipHostFree(m->staging)); m->staging = nullptr; m->stagingBytes = 0; }
  HIP_TRY(hipHostMalloc(&m->staging, bytes, hipHostMallocDefault));
  m->stagingBytes = bytes;
  if (!m->staged) HIP_TRY(hipEventCreateWithFlags(&m->staged, hipEventDisableTiming));
  return NBL_OK;
}
}  // namespace
}  // extern "C++"

// Replaces the inertial constants of `count` bodies with ONE stream-ordered copy on `stream`: launches issued on that stream
// afterwards see the new values, launches issued before it the old ones; no device synchronisation.
int32_t nbl_set_body_inertias(nbl_model* m, int32_t count, const int32_t* bodies, const double* mass, const double* com,
                              const double* inertia, void* stream) {
  if (!m || count < 0 || (count > 0 && (!bodies || !mass || !com || !inertia))) return fail(NBL_E_BADARG, "bad argument");
  if (count == 0) return NBL_OK;
  for (int i = 0; i < count; i++) {
    if (bodies[i] < 0 || bodies[i] >= m->userBodies) return fail(NBL_E_BADARG, "body index out of range");
    if (!(mass[i] > 0)) return fail(NBL_E_BADARG, "mass must be positive");
  }
  DeviceGuard guard(m->device);
  if (!guard.ok) return fail(NBL_E_HIP, "hipSetDevice failed");
  const size_t bytes = sizeof(DevBody) * (size_t)m->nb;
  const int32_t rc = ensureStaging(m, bytes + sizeof(DevInertiaParam) * (size_t)std::max(64, m->nParams));
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipEventSynchronize(m->staged));   // the previous upload has finished reading the staging area (normally long ago)
  for (int i = 0; i < count; i++) packSpatialInertia(mass[i], com + 3 * i, inertia + 6 * i, m->hBodies[m->deviceBody(bodies[i])].G);
  std::memcpy(m->staging, m->hBodies.data(), bytes);
  HIP_TRY(hipMemcpyAsync(m->dBodies, m->staging, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipEventRecord(m->staged, (hipStream_t)stream));
  return NBL_OK;
}

int32_t nbl_set_body_inertia(nbl_model* m, int32_t body, double mass, const double* com, const double* inertia) {
  if (!m || !com || !inertia) return fail(NBL_E_BADARG, "null argument");
  DeviceGuard guard(m->device);
  HIP_TRY(hipDeviceSynchronize());   // this entry point keeps its documented semantics: no launch in flight sees a half-written body
  const int32_t rc = nbl_set_body_inertias(m, 1, &body, &mass, com, inertia, nullptr);
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  return NBL_OK;
}

// The registered inertia parameters.  A table of another size is registration-time work (device synchronisation, reallocation); a
// table of the SAME size - every World.setMasses with new values - is one stream-ordered copy on `stream` through the pinned staging
// area, like nbl_set_body_inertias: launches issued on that stream before the call read the old table, later ones the new one.
int32_t nbl_set_inertia_params_on(nbl_model* m, int32_t count, const int32_t* bodies, const double* dG, void* stream) {
  if (!m || count < 0 || (count > 0 && (!bodies || !dG))) return fail(NBL_E_BADARG, "bad argument");
  for (int p = 0; p < count; p++)
    if (bodies[p] < 0 || bodies[p] >= m->userBodies) return fail(NBL_E_BADARG, "inertia parameter on an unknown body");
  DeviceGuard guard(m->device);
  if (!guard.ok) return fail(NBL_E_HIP, "hipSetDevice failed");
  std::vector<DevInertiaParam> hp(count);
  for (int p = 0; p < count; p++) {
    hp[p].body = m->deviceBody(bodies[p]); hp[p].pad = 0;
    const double* D = dG + 36 * (size_t)p;
    int idx = 0;
    for (int r = 0; r < 6; r++)
      for (int c = r; c < 6; c++) hp[p].dG[idx++] = 0.5 * (D[6 * r + c] + D[6 * c + r]);
  }
  if (count != m->nParams) {
    // the table changes size (registration time, not the per-step path): wait for the launches that read the old one
    HIP_TRY(hipDeviceSynchronize());
    if (m->dParams) { HIP_TRY(hipFree(m->dParams)); m->dParams = nullptr; }
    m->nParams = 0;
    if (count == 0) return NBL_OK;
    HIP_TRY(hipMalloc((void**)&m->dParams, sizeof(DevInertiaParam) * (size_t)count));
    HIP_TRY(hipMemcpy(m->dParams, hp.data(), sizeof(DevInertiaParam) * (size_t)count, hipMemcpyHostToDevice));
    m->nParams = count;
    return NBL_OK;
  }
  if (count == 0) return NBL_OK;
  const size_t bodyBytes = sizeof(DevBody) * (size_t)m->nb, tableBytes = sizeof(DevInertiaParam) * (size_t)count;
  const int32_t rc = ensureStaging(m, bodyBytes + std::max(tableBytes, sizeof(DevInertiaParam) * 64));
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipEventSynchronize(m->staged));   // the previous upload has finished reading the staging area
  std::memcpy((char*)m->staging + bodyBytes, hp.data(), tableBytes);
  HIP_TRY(hipMemcpyAsync(m->dParams, (char*)m->staging + bodyBytes, tableBytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipEventRecord(m->staged, (hipStream_t)stream));
  return NBL_OK;
}

// The same without a stream: ordered against EVERYTHING in flight on the device (synchronises it before and after the copy).
int32_t nbl_set_inertia_params(nbl_model* m, int32_t count, const int32_t* bodies, const double* dG) {
  if (!m) return fail(NBL_E_BADARG, "bad argument");
  DeviceGuard guard(m->device);
  if (!guard.ok) return fail(NBL_E_HIP, "hipSetDevice failed");
  HIP_TRY(hipDeviceSynchronize());
  const int32_t rc = nbl_set_inertia_params_on(m, count, bodies, dG, nullptr);
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  return NBL_OK;
}

int32_t nbl_num_inertia_params(const nbl_model* m) { return m ? m->nParams : 0; }

static void launchInertia(nbl_model* m, int64_t B, int64_t b0, int64_t b1, hipStream_t s, const void* saved, double* grad_params,
                          int accumulate, void* workspace) {
  DevModel mdl = m->mdl;
  mdl.b0 = b0; mdl.b1 = b1;
  SavedLayout lay = m->lay;
  if (m->coopTree && !m->coopFinal) { lay.treeRows = 0; lay.treeNbp = 0; }   // k_tree_to_lanes left the kept slots in the workspace
  hipLaunchKernelGGL(k_bwd_inertia, dim3((unsigned)((b1 - b0 + 63) / 64)), dim3(64), 0, s, mdl, m->dBodies, m->dDofs, B,
                     (const double*)saved, lay, m->dParams, m->nParams, grad_params, accumulate, (double*)workspace);
}

int32_t nbl_backward_inertia(nbl_model* m, int64_t B, const void* saved, double* grad_params, int32_t accumulate, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!m || !saved || !grad_params || !workspace) return fail(NBL_E_BADARG, "null argument");
  if (B <= 0) return fail(NBL_E_BADARG, "B must be positive");
  if (m->nParams <= 0) return fail(NBL_E_BADARG, "no inertia parameters registered (nbl_set_inertia_params)");
  if (workspace_bytes < nbl_workspace_bytes(m, B)) return fail(NBL_E_WORKSPACE, "workspace too small");
  launchInertia(m, B, 0, B, (hipStream_t)stream, saved, grad_params, accumulate, workspace);
  HIP_TRY(hipGetLastError());
  return NBL_OK;
}

// ---- self-test: the device Dantzig driver on caller-supplied problems (host pointers) ----------------------------------
int32_t nbl_selftest_lcp_dantzig(int32_t count, int32_t n, const double* A, const double* b, const double* lo, const double* hi,
                                 const int32_t* findex, double* x, int32_t* rc) {
  return nbl_selftest_lcp_dantzig_timed(count, n, A, b, lo, hi, findex, x, rc, 1, nullptr);
}

#if NBL_GENERAL
// The self-tests of the general instantiations work in HBM scratch like the step: genScratchDoubles(GR) doubles per problem (1.5 MB at 192
// rows, 5.9 MB at 384) whatever n is.  A batch that does not fit the device's free memory is refused with the number that would fit, instead
// of a bare HIP out-of-memory error.
static int32_t selftestScratchCheck(const char* what, int32_t count) {
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return NBL_OK;
  const size_t per = genScratchDoubles(GR) * sizeof(double);
  if ((size_t)count * per > freeB / 2)
    return fail(NBL_E_BADARG, std::string(what) + ": this instantiation (" + std::to_string(GR) + " rows) needs " + std::to_string(per >> 10) + " kB of device scratch per problem; at most " +
                                  std::to_string(freeB / 2 / per) + " problems per call fit half of the free device memory, " + std::to_string(count) + " were given");
  return NBL_OK;
}
#endif

int32_t nbl_selftest_lcp_dantzig_timed(int32_t count, int32_t n, const double* A, const double* b, const double* lo, const double* hi,
                                       const int32_t* findex, double* x, int32_t* rc, int32_t reps, double* ms_per_launch) {
  if (!A || !b || !lo || !hi || !findex || !x || !rc) return fail(NBL_E_BADARG, "null argument");
  if (reps < 1) return fail(NBL_E_BADARG, "reps must be positive");
  if (count <= 0 || n <= 0 || n > MAX_ROWS) return fail(NBL_E_BADARG, "count must be positive and 1 <= n <= " + std::to_string(MAX_ROWS));
  if (nbl_device_count() <= 0) return fail(NBL_E_NOGPU, "no HIP device visible");
#if NBL_GENERAL
  if (const int32_t rcS = selftestScratchCheck("nbl_selftest_lcp_dantzig", count)) return rcS;
#endif
  const size_t nv = (size_t)count * n, nm = nv * n;
  double *dA = nullptr, *dv = nullptr;
  int32_t* di = nullptr;
  hipError_t e = hipMalloc((void**)&dA, nm * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dv, 4 * nv * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&di, (nv + count) * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(dA, A, nm * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv, b, nv * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv + nv, lo, nv * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv + 2 * nv, hi, nv * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(di, findex, nv * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (ms_per_launch) { e = hipEventCreate(&t0); if (e == hipSuccess) e = hipEventCreate(&t1); }
#if NBL_GENERAL
    double* dScratch = nullptr;      // the general driver works in HBM: one world's worth of scratch per problem
    e = hipMalloc((void**)&dScratch, (size_t)count * genScratchDoubles(GR) * sizeof(double));
#define NBL_SELFTEST_DANTZIG(...) hipLaunchKernelGGL(k_selftest_dantzig_gen, dim3((unsigned)count), dim3(64), 0, 0, __VA_ARGS__, dScratch)
#else
#define NBL_SELFTEST_DANTZIG(...) hipLaunchKernelGGL(k_selftest_dantzig, dim3((unsigned)count), dim3(64), 0, 0, __VA_ARGS__)
#endif
    if (e == hipSuccess && ms_per_launch) {     // one untimed launch first (code load)
      NBL_SELFTEST_DANTZIG(count, n, dA, dv, dv + nv, dv + 2 * nv, di, dv + 3 * nv, di + nv);
      e = hipEventRecord(t0, 0);
    }
    for (int r = 0; r < reps && e == hipSuccess; r++)
      NBL_SELFTEST_DANTZIG(count, n, dA, dv, dv + nv, dv + 2 * nv, di, dv + 3 * nv, di + nv);
    if (e == hipSuccess && ms_per_launch) e = hipEventRecord(t1, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && ms_per_launch) {
      float ms = 0.f;
      e = hipEventElapsedTime(&ms, t0, t1);
      *ms_per_launch = (double)ms / reps;
    }
    if (t0) hipEventDestroy(t0);
    if (t1) hipEventDestroy(t1);
#if NBL_GENERAL
    if (dScratch) hipFree(dScratch);
#endif
  }
  if (e == hipSuccess) e = hipMemcpy(x, dv + 3 * nv, nv * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(rc, di + nv, count * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (dA) hipFree(dA);
  if (dv) hipFree(dv);
  if (di) hipFree(di);
  if (e != hipSuccess) return fail(NBL_E_HIP, std::string("nbl_selftest_lcp_dantzig: ") + hipGetErrorString(e));
  return NBL_OK;
}

// ---- self-test: the solver cascade of the general instantiation on caller-supplied contact LCPs (host pointers) ---------------------
int32_t nbl_selftest_lcp_cascade(int32_t count, int32_t mRows, const double* A, const double* b, const double* mu, int32_t have_cache,
                                 const double* x_cache, const uint8_t* on, double fallback_cfm, double* x, int32_t* cls, uint32_t* st, double* cfm) {
#if NBL_GENERAL
  if (!A || !b || !mu || !x || !cls || !st || !cfm || (have_cache && !x_cache)) return fail(NBL_E_BADARG, "null argument");
  if (count <= 0 || mRows <= 0 || mRows > MAX_ROWS || mRows % 3) return fail(NBL_E_BADARG, "count must be positive and m a multiple of 3 up to " + std::to_string(MAX_ROWS));
  if (nbl_device_count() <= 0) return fail(NBL_E_NOGPU, "no HIP device visible");
  if (const int32_t rcS = selftestScratchCheck("nbl_selftest_lcp_cascade", count)) return rcS;
  const size_t nv = (size_t)count * mRows, nm = nv * mRows, nc = (size_t)count * (mRows / 3);
  double *dA = nullptr, *dv = nullptr, *dS = nullptr;
  int32_t* di = nullptr;
  uint8_t* dOn = nullptr;
  hipError_t e = hipMalloc((void**)&dA, nm * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dv, (3 * nv + nc + count) * sizeof(double));       // b, xcache, x, mu, cfm
  if (e == hipSuccess) e = hipMalloc((void**)&di, (nv + count) * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&dS, (size_t)count * genScratchDoubles(GR) * sizeof(double));
  if (e == hipSuccess && on) e = hipMalloc((void**)&dOn, nv);
  if (e == hipSuccess) e = hipMemcpy(dA, A, nm * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv, b, nv * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && have_cache) e = hipMemcpy(dv + nv, x_cache, nv * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv + 3 * nv, mu, nc * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && on) e = hipMemcpy(dOn, on, nv, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_selftest_cascade_gen, dim3((unsigned)count), dim3(64), genRowsDoubles(GR) * sizeof(double), 0, count, mRows, dA, dv, dv + 3 * nv, have_cache, dv + nv, dOn, fallback_cfm,
                       dv + 2 * nv, di, (uint32_t*)(di + nv), dv + 3 * nv + nc, dS);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(x, dv + 2 * nv, nv * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(cls, di, nv * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(st, di + nv, count * sizeof(uint32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(cfm, dv + 3 * nv + nc, count * sizeof(double), hipMemcpyDeviceToHost);
  if (dA) hipFree(dA);
  if (dv) hipFree(dv);
  if (di) hipFree(di);
  if (dS) hipFree(dS);
  if (dOn) hipFree(dOn);
  if (e != hipSuccess) return fail(NBL_E_HIP, std::string("nbl_selftest_lcp_cascade: ") + hipGetErrorString(e));
  return NBL_OK;
#else
  (void)count; (void)mRows; (void)A; (void)b; (void)mu; (void)have_cache; (void)x_cache; (void)on; (void)fallback_cfm; (void)x; (void)cls; (void)st; (void)cfm;
  return fail(NBL_E_UNSUPPORTED, "nbl_selftest_lcp_cascade addresses the general instantiation");
#endif
}

// ---- self-test: the device pseudo-inverses on caller-supplied matrices (host pointers) ----------------------------------------
int32_t nbl_selftest_pinv_rows(int32_t count, int32_t rows, const double* Q, const int32_t* cTrue, int32_t route, double* P, int32_t* rank,
                               int32_t reps, double* ms_per_launch) {
  if (rows != MAX_ROWS) return fail(NBL_E_BADARG, "rows must be " + std::to_string(MAX_ROWS) + " in this instantiation of the library");
  return nbl_selftest_pinv(count, Q, cTrue, route, P, rank, reps, ms_per_launch);
}
int32_t nbl_selftest_pinv(int32_t count, const double* Q, const int32_t* cTrue, int32_t route, double* P, int32_t* rank, int32_t reps,
                          double* ms_per_launch) {
#if NBL_GENERAL
  (void)count; (void)Q; (void)cTrue; (void)route; (void)P; (void)rank; (void)reps; (void)ms_per_launch;
  return fail(NBL_E_UNSUPPORTED, "nbl_selftest_pinv addresses the 24- and 48-row instantiations (the general one is tested through its steps)");
#else
  if (!Q || !cTrue || !P || !rank) return fail(NBL_E_BADARG, "null argument");
  if (count <= 0 || reps < 1 || (route != 0 && route != 1)) return fail(NBL_E_BADARG, "bad count / reps / route");
  if (nbl_device_count() <= 0) return fail(NBL_E_NOGPU, "no HIP device visible");
  const size_t nm = (size_t)count * MAX_ROWS * MAX_ROWS;
  double *dQ = nullptr, *dP = nullptr;
  int32_t* di = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  hipError_t e = hipMalloc((void**)&dQ, nm * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dP, nm * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&di, 2 * (size_t)count * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(dQ, Q, nm * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(di, cTrue, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipEventCreate(&t0);
  if (e == hipSuccess) e = hipEventCreate(&t1);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_selftest_pinv, dim3((unsigned)count), dim3(64), 0, 0, count, dQ, di, route, dP, di + count);   // untimed (code load)
    e = hipEventRecord(t0, 0);
    for (int r = 0; r < reps && e == hipSuccess; r++)
      hipLaunchKernelGGL(k_selftest_pinv, dim3((unsigned)count), dim3(64), 0, 0, count, dQ, di, route, dP, di + count);
    if (e == hipSuccess) e = hipEventRecord(t1, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t0, t1);
    if (ms_per_launch) *ms_per_launch = (double)ms / reps;
  }
  if (e == hipSuccess) e = hipMemcpy(P, dP, nm * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(rank, di + count, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (t0) hipEventDestroy(t0);
  if (t1) hipEventDestroy(t1);
  if (dQ) hipFree(dQ);
  if (dP) hipFree(dP);
  if (di) hipFree(di);
  if (e != hipSuccess) return fail(NBL_E_HIP, std::string("nbl_selftest_pinv: ") + hipGetErrorString(e));
  return NBL_OK;
#endif
}

#ifdef NBL_CASCADE_TIMING
int32_t nbl_debug_dantzig_stats(unsigned long long* out16, int32_t reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dzStat), sizeof(unsigned long long) * 16));   // (out16: 24 entries, the last 8 = g_pinvStat)
  HIP_TRY(hipMemcpyFromSymbol(out16 + 16, HIP_SYMBOL(g_pinvStat), sizeof(unsigned long long) * 8));
  if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dzStat), z, sizeof(z))); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pinvStat), z, sizeof(unsigned long long) * 8));
               HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dzStatSlow), z, sizeof(z))); }
  return NBL_OK;
}
int32_t nbl_debug_dantzig_stats_slow(unsigned long long* out16) {   // the same sums over the solves of more than NBL_DZ_SLOW cycles
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dzStatSlow), sizeof(unsigned long long) * 16));
  return NBL_OK;
}
#endif

#if defined(NBL_GEN_TIMING) && NBL_GENERAL
int32_t nbl_debug_gen_stats(unsigned long long* out32, int32_t reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_genStat), sizeof(unsigned long long) * 32));
  if (reset) { unsigned long long z[32] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_genStat), z, sizeof(z))); }
  return NBL_OK;
}
#endif

#ifdef NBL_PHASE_TIMING
int32_t nbl_debug_phase_stamps(unsigned long long* out64) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phaseStamp), sizeof(unsigned long long) * 64));
  return NBL_OK;
}
#endif

// ---- T-step rollout -------------------------------------------------------------------
extern "C++" {   // (internal helpers: C++ linkage keeps them local to this instantiation of the library)
namespace {
size_t alignUp(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace
}  // extern "C++"

size_t nbl_rollout_workspace_bytes(const nbl_model* m, int64_t B) {
  if (!m || B <= 0) return 0;
  // step workspace + two LCP warm-start buffers + the running state cotangent
  return alignUp(nbl_workspace_bytes(m, B)) + 2 * alignUp((size_t)(MAX_ROWS + 1) * B * sizeof(double)) +
         alignUp((size_t)2 * m->n * B * sizeof(double));
}

extern "C++" {   // (internal helpers: C++ linkage keeps them local to this instantiation of the library)
namespace {
__global__ __launch_bounds__(256) void k_add_rows(double* __restrict__ dst, const double* __restrict__ src, int64_t B, int64_t b0,
                                                  int64_t b1, int rows) {   // dst[r][b] += src[r][b] for b in [b0, b1)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, cnt = b1 - b0;
  if (i >= cnt * rows) return;
  const int64_t r = i / cnt, b = b0 + (i - r * cnt);
  dst[r * B + b] += src[r * B + b];
}
__global__ __launch_bounds__(256) void k_copy_rows(double* __restrict__ dst, const double* __restrict__ src, int64_t B, int64_t b0,
                                                   int64_t b1, int rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, cnt = b1 - b0;
  if (i >= cnt * rows) return;
  const int64_t r = i / cnt, b = b0 + (i - r * cnt);
  dst[r * B + b] = src[r * B + b];
}

// The trajectory buffers of one rollout call.  segment == 0: record t lives at saved + t * savedBytes (all T resident);
// segment > 0 (checkpointed): only `segment` records are resident, record t at slot t % segment, and the LCP warm start that
// enters step k * segment is kept in checkpoints[k] so that the segment can be run again, bit for bit, by the backward pass.
struct RolloutBufs {
  int32_t T, segment;
  const double* actions;
  int64_t actionStride;
  double* states;
  char* saved;
  char* checkpoints;
  uint32_t* status;
  bool warm;
  double* cache[2];
  size_t stateElems, savedBytes, cacheBytes;
  char* record(int32_t t) const { return saved ? saved + (size_t)(segment > 0 ? t % segment : t) * savedBytes : nullptr; }
};

// steps [t0, t1) of one batch slice on its stream
int32_t rolloutStepsOfSlice(nbl_model* m, int64_t B, int si, int64_t b0, int64_t b1, hipStream_t s, const RolloutBufs& r, int32_t t0,
                            int32_t t1, bool recompute, void* workspace) {
  const unsigned cblocks = (unsigned)(((b1 - b0) * (MAX_ROWS + 1) + 255) / 256);
  for (int32_t t = t0; t < t1; t++) {
    const bool warm = m->hasContact && r.warm && t > 0;
    const int32_t rc = launchForward(m, B, si, b0, b1, s, r.states + (size_t)t * r.stateElems, r.actions + (size_t)t * r.actionStride,
                                     warm ? r.cache[(t + 1) & 1] : nullptr, r.states + (size_t)(t + 1) * r.stateElems,
                                     m->hasContact ? r.cache[t & 1] : nullptr, r.record(t),
                                     (r.status && !recompute) ? r.status + (size_t)t * B : nullptr, workspace);
    if (rc != NBL_OK) return rc;
    if (!recompute && r.segment > 0 && m->hasContact && r.warm && (t + 1) % r.segment == 0 && t + 1 < r.T)   // the warm start step t+1 reads
      hipLaunchKernelGGL(k_copy_rows, dim3(cblocks), dim3(256), 0, s, (double*)(r.checkpoints + (size_t)((t + 1) / r.segment) * r.cacheBytes),
                         (const double*)r.cache[t & 1], B, b0, b1, MAX_ROWS + 1);
  }
  return NBL_OK;
}

RolloutBufs rolloutBufs(const nbl_model* m, int64_t B, int32_t T, int32_t segment, const double* actions, int64_t action_stride,
                        double* states, void* saved, void* checkpoints, uint32_t* status, int32_t warm_start, void* workspace) {
  RolloutBufs r;
  r.T = T; r.segment = segment; r.actions = actions; r.actionStride = action_stride; r.states = states;
  r.saved = (char*)saved; r.checkpoints = (char*)checkpoints; r.status = status; r.warm = warm_start != 0;
  r.cacheBytes = alignUp((size_t)(MAX_ROWS + 1) * B * sizeof(double));
  char* base = (char*)workspace + alignUp(nbl_workspace_bytes(m, B));
  r.cache[0] = (double*)base; r.cache[1] = (double*)(base + r.cacheBytes);
  r.stateElems = (size_t)2 * m->n * B; r.savedBytes = nbl_saved_bytes(m, B);
  return r;
}
}  // namespace
}  // extern "C++"

size_t nbl_rollout_checkpoint_bytes(const nbl_model* m, int64_t B, int32_t T, int32_t segment) {
  if (!m || B <= 0 || T <= 0 || segment <= 0) return 0;
  return (size_t)((T + segment - 1) / segment) * alignUp((size_t)(MAX_ROWS + 1) * B * sizeof(double));
}

int32_t nbl_rollout_forward_checkpointed(nbl_model* m, int64_t B, int32_t T, int32_t segment, const double* state0, const double* actions,
                                         int64_t action_stride, double* states, void* saved, void* checkpoints, uint32_t* status,
                                         int32_t warm_start, void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !state0 || !actions || !states || !workspace) return fail(NBL_E_BADARG, "null argument");
  if (B <= 0 || T <= 0 || segment < 0) return fail(NBL_E_BADARG, "B and T must be positive, segment >= 0");
  if (workspace_bytes < nbl_rollout_workspace_bytes(m, B)) return fail(NBL_E_WORKSPACE, "rollout workspace too small");
  if (m->hasContact && !saved) return fail(NBL_E_BADARG, "models with colliders need the saved records");
  if (segment > 0 && m->hasContact && warm_start && !checkpoints) return fail(NBL_E_BADARG, "a checkpointed warm-started rollout needs the checkpoint buffer");
  hipStream_t s0 = (hipStream_t)stream;
  const RolloutBufs r = rolloutBufs(m, B, T, segment, actions, action_stride, states, saved, checkpoints, status, warm_start, workspace);
  HIP_TRY(hipMemcpyAsync(states, state0, r.stateElems * sizeof(double), hipMemcpyDeviceToDevice, s0));
  m->timingNow = false;
  // slice-major: every slice runs its T steps on its own stream; the slices only join at the end
  const int32_t rc = forSlices(m, B, rolloutSlicesFor(m, B), s0, [&](int si, int64_t b0, int64_t b1, hipStream_t s) -> int32_t {
    return rolloutStepsOfSlice(m, B, si, b0, b1, s, r, 0, T, false, workspace);
  });
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipGetLastError());
  return NBL_OK;
}

int32_t nbl_rollout_forward(nbl_model* m, int64_t B, int32_t T, const double* state0, const double* actions,
                            int64_t action_stride, double* states, void* saved, uint32_t* status, int32_t warm_start,
                            void* workspace, size_t workspace_bytes, void* stream) {
  return nbl_rollout_forward_checkpointed(m, B, T, 0, state0, actions, action_stride, states, saved, nullptr, status, warm_start, workspace,
                                          workspace_bytes, stream);
}

int32_t nbl_rollout_backward(nbl_model* m, int64_t B, int32_t T, const void* saved, const double* grad_states,
                             double* grad_state0, double* grad_actions, void* workspace, size_t workspace_bytes,
                             void* stream) {
  return nbl_rollout_backward_inertia(m, B, T, saved, grad_states, grad_state0, grad_actions, nullptr, workspace, workspace_bytes, stream);
}

int32_t nbl_rollout_backward_inertia(nbl_model* m, int64_t B, int32_t T, const void* saved, const double* grad_states,
                                     double* grad_state0, double* grad_actions, double* grad_params, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return nbl_rollout_backward_checkpointed(m, B, T, 0, nullptr, nullptr, 0, (void*)saved, nullptr, 0, grad_states, grad_state0, grad_actions,
                                           grad_params, workspace, workspace_bytes, stream);
}

int32_t nbl_rollout_backward_checkpointed(nbl_model* m, int64_t B, int32_t T, int32_t segment, double* states, const double* actions,
                                          int64_t action_stride, void* saved, const void* checkpoints, int32_t warm_start,
                                          const double* grad_states, double* grad_state0, double* grad_actions, double* grad_params,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !saved || !grad_states || !grad_state0 || !grad_actions || !workspace) return fail(NBL_E_BADARG, "null argument");
  if (grad_params && m->nParams <= 0) return fail(NBL_E_BADARG, "no inertia parameters registered (nbl_set_inertia_params)");
  if (B <= 0 || T <= 0 || segment < 0) return fail(NBL_E_BADARG, "B and T must be positive, segment >= 0");
  if (segment > 0 && segment < T && (!states || !actions)) return fail(NBL_E_BADARG, "a checkpointed backward pass needs the states and actions of the forward call");
  if (segment > 0 && segment < T && m->hasContact && warm_start && !checkpoints) return fail(NBL_E_BADARG, "a checkpointed warm-started rollout needs the checkpoint buffer");
  if (workspace_bytes < nbl_rollout_workspace_bytes(m, B)) return fail(NBL_E_WORKSPACE, "rollout workspace too small");
  hipStream_t s0 = (hipStream_t)stream;
  const RolloutBufs r = rolloutBufs(m, B, T, segment, actions, action_stride, states, saved, (void*)checkpoints, nullptr, warm_start, workspace);
  double* g = (double*)((char*)workspace + alignUp(nbl_workspace_bytes(m, B)) + 2 * r.cacheBytes);   // running cotangent of states[t+1]
  const size_t actElems = (size_t)m->k * B;
  const int rows = 2 * m->n;
  const int32_t seg = segment > 0 ? segment : T, nSeg = (T + seg - 1) / seg;
  m->timingNow = false;
  const int32_t rc = forSlices(m, B, rolloutSlicesFor(m, B), s0, [&](int si, int64_t b0, int64_t b1, hipStream_t s) -> int32_t {
    const unsigned blocks = (unsigned)(((b1 - b0) * rows + 255) / 256), cblocks = (unsigned)(((b1 - b0) * (MAX_ROWS + 1) + 255) / 256);
    hipLaunchKernelGGL(k_copy_rows, dim3(blocks), dim3(256), 0, s, g, grad_states + (size_t)T * r.stateElems, B, b0, b1, rows);
    for (int32_t k = nSeg - 1; k >= 0; k--) {
      const int32_t t0 = k * seg, t1 = std::min(T, t0 + seg);
      if (k != nSeg - 1) {
        // the records of this segment were overwritten by later ones: run its steps again from states[t0] and the checkpointed
        // warm start (the forward kernels are bit-reproducible, so the records - and states[t0+1 .. t1], rewritten in passing -
        // are the ones the forward call produced).  The last segment's records are still resident.
        if (m->hasContact && r.warm && t0 > 0)
          hipLaunchKernelGGL(k_copy_rows, dim3(cblocks), dim3(256), 0, s, r.cache[(t0 + 1) & 1], (const double*)(r.checkpoints + (size_t)k * r.cacheBytes),
                             B, b0, b1, MAX_ROWS + 1);
        const int32_t rf = rolloutStepsOfSlice(m, B, si, b0, b1, s, r, t0, t1, true, workspace);
        if (rf != NBL_OK) return rf;
      }
      for (int32_t t = t1 - 1; t >= t0; t--) {
        // the kernels of one backward step re-read the incoming cotangent after the first outputs are written, so the
        // output must not alias it: every step writes into grad_state0 and the running cotangent is copied back
        const int32_t rb = launchBackward(m, B, si, b0, b1, s, r.record(t), g, grad_state0, grad_actions + (size_t)t * actElems, workspace);
        if (rb != NBL_OK) return rb;
        if (grad_params) launchInertia(m, B, b0, b1, s, r.record(t), grad_params, t != T - 1, workspace);
        hipLaunchKernelGGL(k_add_rows, dim3(blocks), dim3(256), 0, s, grad_state0, grad_states + (size_t)t * r.stateElems, B, b0, b1, rows);
        if (t > 0) hipLaunchKernelGGL(k_copy_rows, dim3(blocks), dim3(256), 0, s, g, (const double*)grad_state0, B, b0, b1, rows);
      }
    }
    return NBL_OK;
  });
  if (rc != NBL_OK) return rc;
  HIP_TRY(hipGetLastError());
  return NBL_OK;
}

int32_t nbl_transpose_to_soa(const double* src_bd, double* dst_db, int64_t B, int32_t d, void* stream) {
  if (!src_bd || !dst_db || B <= 0 || d <= 0) return fail(NBL_E_BADARG, "bad transpose argument");
  dim3 grid((unsigned)((d + 31) / 32), (unsigned)((B + 31) / 32)), block(256);
  hipLaunchKernelGGL(k_transpose, grid, block, 0, (hipStream_t)stream, src_bd, dst_db, B, (int64_t)d);
  HIP_TRY(hipGetLastError());
  return NBL_OK;
}
int32_t nbl_transpose_from_soa(const double* src_db, double* dst_bd, int64_t B, int32_t d, void* stream) {
  if (!src_db || !dst_bd || B <= 0 || d <= 0) return fail(NBL_E_BADARG, "bad transpose argument");
  dim3 grid((unsigned)((B + 31) / 32), (unsigned)((d + 31) / 32)), block(256);
  hipLaunchKernelGGL(k_transpose, grid, block, 0, (hipStream_t)stream, src_db, dst_bd, (int64_t)d, B);
  HIP_TRY(hipGetLastError());
  return NBL_OK;
}

int32_t nbl_set_slices(nbl_model* m, int32_t slices) {
  if (!m) return fail(NBL_E_BADARG, "null model");
  if (slices < 0 || slices > NBL_MAX_SLICES) return fail(NBL_E_BADARG, "slices must be 0 (auto) .. 8");
  m->slices = slices;
  return NBL_OK;
}
int32_t nbl_slices_for(const nbl_model* m, int64_t B) { return (m && B > 0) ? slicesFor(m, B) : 0; }
int32_t nbl_set_deferred_join(nbl_model* m, int32_t enabled) {
  if (!m) return fail(NBL_E_BADARG, "null model");
  m->deferJoin = enabled != 0;
  if (m->deferJoin) { const int32_t rc = ensureSideStreams(m, NBL_MAX_SLICES - 1); if (rc != NBL_OK) return rc; }   // (nbl_fork_slices before the first call reaches every stream)
  return NBL_OK;
}
int32_t nbl_slice_stream(nbl_model* m, int64_t B, int32_t slice, void** stream, int64_t* first_world, int64_t* end_world) {
  if (!m || B <= 0) return fail(NBL_E_BADARG, "null model or B <= 0");
  if (!m->deferJoin) return fail(NBL_E_BADARG, "nbl_slice_stream: the handle is not in deferred-join mode (nbl_set_deferred_join)");
  const int sl = slicesFor(m, B);
  if (slice < 0 || slice >= sl) return fail(NBL_E_BADARG, "slice out of range");
  const int32_t rc = ensureSideStreams(m, sl - 1);
  if (rc != NBL_OK) return rc;
  const int64_t per = slicePer(B, sl);
  if (stream) *stream = slice == 0 ? nullptr : (void*)m->side[slice - 1];      // (NULL: slice 0 runs on the stream the calls are given)
  if (first_world) *first_world = std::min(B, (int64_t)slice * per);
  if (end_world) *end_world = std::min(B, (int64_t)(slice + 1) * per);
  return NBL_OK;
}
int32_t nbl_fork_slices(nbl_model* m, void* stream) {
  if (!m) return fail(NBL_E_BADARG, "null model");
  if (!m->fork) HIP_TRY(hipEventCreateWithFlags(&m->fork, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(m->fork, (hipStream_t)stream));
  for (size_t i = 0; i < m->side.size(); i++) HIP_TRY(hipStreamWaitEvent(m->side[i], m->fork, 0));
  return NBL_OK;
}
int32_t nbl_join_slices(nbl_model* m, void* stream) {
  if (!m) return fail(NBL_E_BADARG, "null model");
  for (size_t i = 0; i < m->side.size(); i++) {
    HIP_TRY(hipEventRecord(m->sideDone[i], m->side[i]));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, m->sideDone[i], 0));
  }
  return NBL_OK;
}

int32_t nbl_set_launch_lanes(nbl_model* m, int32_t tree_lanes, int32_t lcp_lanes) {
  if (!m) return fail(NBL_E_BADARG, "null model");
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if ((tree_lanes != 0 && (!pow2(tree_lanes) || tree_lanes > 64)) || (lcp_lanes != 0 && (!pow2(lcp_lanes) || lcp_lanes > LCP_LANES)))
    return fail(NBL_E_BADARG, "lanes must be 0 (auto) or a power of two (tree <= 64, lcp <= 16)");
  m->treeLanes = tree_lanes; (void)lcp_lanes;   // the dense kernels are one world per wavefront: nothing to choose
  return NBL_OK;
}

int32_t nbl_set_timing(nbl_model* m, int32_t enabled) {
  if (!m) return fail(NBL_E_BADARG, "null model");
  m->timing = enabled != 0;
  m->timingPeriod = enabled > 1 ? enabled : 1;
  m->fwdCalls = m->bwdCalls = 0;
  if (!enabled) {
    for (auto& t : m->pending) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
    m->pending.clear();
    m->fwdMs = m->bwdMs = 0;
    m->fwdCount = m->bwdCount = 0;
    for (int i = 0; i < K_COUNT; i++) { m->kMs[i] = 0; m->kCount[i] = 0; }
  }
  return NBL_OK;
}
int32_t nbl_get_timing(nbl_model* m, double* fwd_ms_sum, int64_t* fwd_count, double* bwd_ms_sum, int64_t* bwd_count) {
  if (!m) return fail(NBL_E_BADARG, "null model");
  for (auto& t : m->pending) {
    HIP_TRY(hipEventSynchronize(t.stop));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
    m->kMs[t.kernel] += ms;
    m->kCount[t.kernel]++;
    const bool isBwd = t.kernel == K_BWD || t.kernel == K_RECOMPUTE || t.kernel == K_BWD_FINAL ||
                       t.kernel == K_BWD_A_COOP || t.kernel == K_BWD_B_COOP || t.kernel == K_RECOMPUTE_COOP || t.kernel == K_BWD_FINAL_COOP || t.kernel == K_TREE_TO_LANES;
    if (isBwd) { m->bwdMs += ms; if (t.kernel == K_BWD || t.kernel == K_BWD_FINAL || t.kernel == K_BWD_FINAL_COOP) m->bwdCount++; }
    else { m->fwdMs += ms; if (t.kernel == K_FWD || t.kernel == K_FWD_COOP) m->fwdCount++; }
    hipEventDestroy(t.start);
    hipEventDestroy(t.stop);
  }
  m->pending.clear();
  if (fwd_ms_sum) *fwd_ms_sum = m->fwdMs;
  if (fwd_count) *fwd_count = m->fwdCount;
  if (bwd_ms_sum) *bwd_ms_sum = m->bwdMs;
  if (bwd_count) *bwd_count = m->bwdCount;
  return NBL_OK;
}

int32_t nbl_kernel_count(void) { return K_COUNT; }
const char* nbl_kernel_name(int32_t i) { return (i >= 0 && i < K_COUNT) ? kKernelNames[i] : ""; }
int32_t nbl_kernel_timing(nbl_model* m, int32_t i, double* ms_sum, int64_t* count) {
  if (!m || i < 0 || i >= K_COUNT) return fail(NBL_E_BADARG, "bad kernel index");
  int32_t rc = nbl_get_timing(m, nullptr, nullptr, nullptr, nullptr);  // drains pending events
  if (rc != NBL_OK) return rc;
  if (ms_sum) *ms_sum = m->kMs[i];
  if (count) *count = m->kCount[i];
  return NBL_OK;
}

}  // extern "C"
