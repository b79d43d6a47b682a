// model_dev.hpp — device-resident model constants and workspace layout.
//
// The model (topology, joint frames, spatial inertias, limits) is identical for every world of a
// batch.  It is stored once in HBM as an array of DevBody and read with wave-uniform indices, so
// hipcc emits scalar loads (s_load_dwordx*) through the scalar cache: the constants occupy SGPRs,
// not VGPRs or LDS bandwidth.  Per-world state lives in VGPRs; per-body intermediates that must
// survive between the tree sweeps go to a structure-of-arrays workspace  ws[slot * B + world]
// so that every access of a wavefront is one contiguous 512-byte line.
#pragma once
#include <stdint.h>

// Minimum waves per SIMD the register allocator must leave room for, per kernel (512 registers per lane per SIMD, VGPRs + AGPRs:
// 2 -> 256, 3 -> 168, 4 -> 128).  The defaults are the measured optimum (profiles/r02c_occupancy_sweep.json); -DNBL_W_<KERNEL>=k overrides.
#define NBL_WAVES(k) __attribute__((amdgpu_waves_per_eu(k)))
#ifndef NBL_W_SOLVE_GEN
#define NBL_W_SOLVE_GEN 2      // (round 6: capped to 128 registers - four wavefronts per SIMD - it spills 324 B and measures 6 % slower)
#endif
#ifndef NBL_W_SOLVE
#define NBL_W_SOLVE 2
#endif
#ifndef NBL_W_CFINAL
#define NBL_W_CFINAL 2
#endif
#ifndef NBL_W_BWDA
#define NBL_W_BWDA 1
#endif
#ifndef NBL_W_BWDB
#define NBL_W_BWDB 1
#endif
#ifndef NBL_W_ROWS
#define NBL_W_ROWS 1
#endif
#ifndef NBL_W_STAGES
#define NBL_W_STAGES 1
#endif
#ifndef NBL_W_FWD
#define NBL_W_FWD 1
#endif
#ifndef NBL_W_RECOMP
#define NBL_W_RECOMP 1
#endif
#ifndef NBL_W_BFINAL
#define NBL_W_BFINAL 1
#endif

// The device namespace of this instantiation of the library (the library is built several times from one set of sources, see
// abi_variants.h: every build names its own - -DNBL_NS=nbl_c16 - so that their kernels and constants never meet at link time).
#ifndef NBL_NS
#define NBL_NS nbl
#endif
namespace NBL_NS {

// Developer instrumentation (tools/phase_timing.py builds with -DNBL_PHASE_TIMING): cycle stamps of the first wavefront of a
// launch at phase boundaries of the tree kernels.  Compiled out of the shipped library.
#ifdef NBL_PHASE_TIMING
__device__ unsigned long long g_phaseStamp[64];
#define NBL_PHASE(k) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_phaseStamp[k] = clock64(); } while (0)
#define NBL_PHASE_FIRST(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phaseStamp[k] = clock64(); } while (0)   // (the narrow-phase workgroups come first)
#else
#define NBL_PHASE(k) do { } while (0)
#define NBL_PHASE_FIRST(k) do { } while (0)
#endif

constexpr int JT_REVOLUTE = 0, JT_PRISMATIC = 1, JT_FREE = 2, JT_BALL = 4, JT_SCREW = 5, JT_FREEC = 6;   // = NBL_JOINT_*; JT_BALL: one of the three coincident axes of a ball joint; JT_FREEC (internal): one of the SIX coincident axes (3 rotations, 3 translations) of a free joint below the root

struct DevBody {
  int32_t parent, jtype, dofOff, ndof;
  double Tpj[12];     // parent body -> joint   (R row-major, p)
  double Tcj[12];     // child body  -> joint
  double TcjInv[12];
  double axis[3];
  double S[6];        // 1-DOF joints: constant relative Jacobian column in the child frame
  double G[21];       // spatial inertia, packed symmetric
  double screwRate;    // JT_SCREW: translation along the axis per radian (pitch / 2 pi)
  int32_t level, rank; // depth in the tree, index among the children of the parent (coop tree kernels)
  int32_t freeIdx;     // index among the free-joint bodies (-1 otherwise): their extra LDS block in the coop tree kernels
  int32_t root, padr;  // the root body of this body's tree: origin of the translated "world" frame its spatial quantities are carried in
  int32_t ballComp;    // JT_BALL: 0, 1, 2 = the x, y, z body of a ball joint's triple (consecutive body indices and DOFs; the x body
                       // carries T_pj exp(q), the z body T_cj, mass and children).  1-DOF code treats all three as revolute joints
};

struct DevDof {
  double damping, spring, rest;
  double posLo, posHi, velLo, velHi, forceLo, forceHi;
  int32_t actionIndex;  // index into the action vector, -1 = unactuated (tau = 0)
  int32_t pad;
};

struct DevModel {
  int32_t nb, n, nAction, pad;   // pad: worlds packed into one wavefront of the lane = body tree kernels (>= 1)
  int32_t maxLevel, maxRank, nbp, nFree;  // tree depth, max sibling rank, bodies padded, free-joint bodies (coop tree kernels)
  int32_t hasBounce, hasCapsule;          // some collider pair can bounce (restitution product > 1e-3): k_bwd_bounce runs; some collider is a capsule
  double gravity[3];
  double dt;
  int64_t b0, b1;     // the worlds [b0, b1) this launch processes (the batch may be sliced over several HIP streams)
};

// Workspace slots per body (doubles per world).
constexpr int WS_T = 0;        // 12  relative transform T = T_pj Q(q) T_cj^-1
constexpr int WS_V = 12;       // 6   body twist
constexpr int WS_AI = 18;      // 21  articulated inertia (children accumulate here first)
constexpr int WS_AIS = 39;     // 6   AI*S (1-DOF)
constexpr int WS_PSI = 45;     // 1 (1-DOF)  or 21 (free joint: LDL^T of S^T AI S)
constexpr int WS_BACC = 66;    // 6   bias-force accumulator from children (dead after sweep 2 of the ABA)
constexpr int WS_VTW = WS_BACC; // 6   body twist at the pre-contact velocity, written after the ABA (contact models)
constexpr int WS_U = 72;       // 6   joint-space total force / rhs
constexpr int WS_A = 78;       // 6   body acceleration (gravity carried as base acceleration -g)
constexpr int WS_TW = 84;      // 12  world transform of the body
// Slots [0, WS_KEEP) are the forward tree state the backward pass needs: when the saved record carries a tree block
// (SavedLayout::treeRows) the forward kernels write them there and the backward pass reads them back instead of
// re-running the three ABA sweeps; the slots from WS_KEEP on are scratch in the workspace.
constexpr int WS_KEEP = 96;
// The WORLD-MAJOR tree block of the saved record (SavedLayout::treeNbp > 0, the lane = body tree kernels) is compact: rows
// [TREE_ROWS][nbp] for the kept slots every body owns (T V AIS PSI[0] BACC/VTW U[0] A TW), then [TREE_FREE] doubles per free-joint body
// for the slots only a free joint owns (AI - its one reader is the 6 x 6 solve of a free-joint root -, PSI[1..20], U[1..5]):
// 50 nbp + 46 nFree doubles per world and step instead of 96 nbp (Atlas-20: 6.8 kB instead of 12.3 kB).
constexpr int TREE_ROWS = 50, TREE_FREE = 46;
__host__ __device__ inline int treeCompactRow(int s) {   // kept slot -> row (>= 0), or -(1 + entry of the free-joint part)
  if (s < 18) return s;                         // T V                 rows 0..17      (WS_T, WS_V)
  if (s < 39) return -(1 + (s - 18));           // AI                  free 0..20      (WS_AI)
  if (s <= 45) return 18 + (s - 39);            // AIS PSI[0]          rows 18..24     (WS_AIS, WS_PSI)
  if (s < 66) return -(1 + 21 + (s - 46));      // PSI[1..20]          free 21..40
  if (s <= 72) return 25 + (s - 66);            // BACC / VTW, U[0]    rows 25..31     (WS_BACC, WS_U)
  if (s < 78) return -(1 + 41 + (s - 73));      // U[1..5]             free 41..45
  return 32 + (s - 78);                         // A TW                rows 32..49     (WS_A, WS_TW)
}
__host__ __device__ inline int treeRowSlot(int r) { return r < 18 ? r : (r < 25 ? 39 + (r - 18) : (r < 32 ? 66 + (r - 25) : 78 + (r - 32))); }
__host__ __device__ inline int treeFreeSlot(int k) { return k < 21 ? 18 + k : (k < 41 ? 46 + (k - 21) : 73 + (k - 41)); }
static_assert(WS_V == 12 && WS_AI == 18 && WS_AIS == 39 && WS_PSI == 45 && WS_BACC == 66 && WS_U == 72 && WS_A == 78 && WS_TW == 84, "treeCompactRow follows the slot map above");
constexpr int WS_BIMP = 96;    // 6   impulse-bias accumulator (M^-1 solves)
constexpr int WS_UIMP = 102;   // 6
constexpr int WS_W = 108;      // 6   twist generated by lambda = M^-1 g   (adjoint of transmitted force)
constexpr int WS_FACC = 114;   // 6   transmitted-force accumulator
constexpr int WS_ABAR = 120;   // 6   acceleration adjoint accumulator
constexpr int WS_VBAR = 126;   // 6   velocity adjoint accumulator
// contact backward: twist fields generated by joint-rate vectors (body frame / world frame) and accumulators
constexpr int NFIELD = 9;       // 0 lambda1, 1 v_pre, 2-4 p_k, 5-7 s_k, 8 w
constexpr int WS_FB = 132;      // NFIELD x 6 body-frame twists
constexpr int WS_FW = 186;      // 8 x 6 world-frame twists of fields 0..7
constexpr int WS_PAIRF = 234;   // 4 x 6 transmitted-force accumulators of the mass-matrix pairs
constexpr int WS_PAIRA = 258;   // 4 x 6 acceleration-adjoint accumulators
constexpr int WS_XI = 282;      // 6   world-frame adjoint of the joint's position twist (contact geometry)
constexpr int WS_PER_BODY = 288;

// ---- contact stage -------------------------------------------------------------------------
// The library is built TWICE from these sources (__graft_entry__.build / nimble_amd_dispatch.cpp): NBL_MAXC = 8 (24 LCP rows: the fast
// instantiation every BASELINE config runs) and NBL_MAXC = 16 (48 LCP rows, 32 colliders, 64 collider pairs: namespace nbl_c16); a model
// is given to one or the other when it is created, by max_contacts / its collider and pair counts.
#ifndef NBL_MAXC
#define NBL_MAXC 8
#endif
// A THIRD instantiation, NBL_MAXC = 64 (192 LCP rows, 64 colliders, 512 collider pairs: namespace nbl_c64), is the GENERAL one: its dense
// contact kernels (gen_contact.hip) loop over the rows instead of mapping them to lanes - slow, and without a compile-time row budget
// that a legal world could exceed in practice (the reference itself has none, ConstraintSolver.cpp:563-606).
constexpr int MAX_CONTACTS = NBL_MAXC;       // per world (8 frictional contacts = 24 LCP rows; 16 = 48 rows; 64 = 192 rows; 128 = 384 rows: the general code again)
constexpr int MAX_ROWS = 3 * MAX_CONTACTS;
constexpr int MAX_BOXES = NBL_MAXC > 16 ? 64 : (NBL_MAXC > 8 ? 32 : 16);      // (<= 64: collider codes below CR_BODY_CODE)
constexpr int MAX_PAIRS = NBL_MAXC > 16 ? 512 : (NBL_MAXC > 8 ? 64 : 32);
#define NBL_GENERAL (NBL_MAXC > 16)          // the general instantiation of the dense contact stage
constexpr int SEEN_POINTS = 2 * MAX_CONTACTS;   // narrow phase: points the duplicate filter compares with (the kept contacts + unique points the depth filter dropped)
constexpr int MAX_DOF_CONTACT = 64;   // lane = DOF in the wavefront kernels (round 2: 40)

constexpr int SHAPE_BOX = 0, SHAPE_SPHERE = 1, SHAPE_CAPSULE = 2;   // NBL_SHAPE_*
struct DevBox {       // a collider: box (half extents), sphere (radius in half[0]) or capsule (radius in half[0], half the cylinder height in half[1])
  int32_t body, shape;
  double T[12];     // shape frame in the body frame
  double half[3];
  double mu;
  double restitution;   // BodyNode restitution coefficient of the owning body (ContactConstraint.cpp:95-97: e = e_A e_B)
};

struct DevContactModel {
  int32_t nBoxes, nPairs, maxContacts, penetrationCorrection;
  double clippingDepth, fallbackCfm;
  int32_t pairA[MAX_PAIRS], pairB[MAX_PAIRS];
  DevBox boxes[MAX_BOXES];
  uint64_t ancestors[64];   // bit i of ancestors[b]: body i is b or an ancestor of b
  int32_t skelOf[64];       // per body: its skeleton (constrained groups unite skeletons, ConstraintSolver.cpp:724-780); < 64
  // joint-limit constraint rows (JointLimitConstraint.cpp; nbl_model_desc.dof_limit_enforced): the single-DOF joints that enforce a finite
  // position limit.  An active one becomes a pseudo-contact of the record (CT_LIMIT) after the world's contacts
  int32_t nLimitDofs, selfCollision;   // selfCollision: some pair of colliders sits on one skeleton (body_self_collision)
  int32_t oneSkeleton, pad0_;          // oneSkeleton: every body of the model is on one skeleton - a world has at most ONE constrained group
  int32_t limitDof[MAX_DOF_CONTACT], limitBody[MAX_DOF_CONTACT];
  double limitLo[MAX_DOF_CONTACT], limitHi[MAX_DOF_CONTACT];
};
// A joint-limit row in the contact record: type CT_LIMIT, the two "colliders" are the codes CR_BODY_CODE + 1 + body of the joint's child
// body (A) and of its parent body (B; world = -1), CR_EA_FIXED = (DOF, sigma, 0).  sigma = +1 at the lower limit; at the upper limit the
// row is carried NEGATED (sigma = -1: x' = -x >= 0, row and column of A and b negated - exact in IEEE arithmetic) so that every stage sees
// a frictionless normal row with bounds [0, inf); the warm-start cache and LCPUtils::guessSolution's `b > 0` use the reference's sign.
constexpr int CT_LIMIT = 30, CR_BODY_CODE = 64;
__device__ __forceinline__ int crBodyOf(const DevContactModel* __restrict__ cm, int code) { return code >= CR_BODY_CODE ? code - CR_BODY_CODE - 1 : cm->boxes[code].body; }
__device__ __forceinline__ double crMuOf(const DevContactModel* __restrict__ cm, int code) { return code >= CR_BODY_CODE ? 0.0 : cm->boxes[code].mu; }

// Per-contact record kept in the saved-for-backward buffer (doubles per world)
constexpr int CR_POINT = 0, CR_NORMAL = 3, CR_DEPTH = 6, CR_TYPE = 7, CR_BOXA = 8, CR_BOXB = 9;
constexpr int CR_EA_FIXED = 10, CR_EA_DIR = 13, CR_EB_FIXED = 16, CR_EB_DIR = 19, CR_SIZE = 22;
// sphere contacts reuse the four edge slots: SPHERE_BOX / BOX_SPHERE: EA_FIXED = sphere centre, EA_DIR / EB_FIXED / EB_DIR = the
// three box face normals (zero when the face is not "locked"); SPHERE_SPHERE: EA_FIXED = centre A, EB_FIXED = centre B,
// EA_DIR = (radius A, radius B, 0).  Capsule contacts: see collision_dev.hpp (capsuleCapsule).

// Layout of the saved record.  Rows q .. pflag are lane-interleaved (row index; every row holds B doubles, `total` rows);
// the dense per-world block follows: world b owns `dense` contiguous doubles at saved[total * B + b * dense] and
// A / massed / aall / pinv are offsets inside it (world-major, so that one wavefront per world reads them coalesced):
//   A      MAX_ROWS x MAX_ROWS   Delassus matrix           massed  n x MAX_ROWS   M^-1 J^T (impulse tests)
//   aall   n x MAX_ROWS          J^T (constraint forces)   pinv    MAX_ROWS x MAX_ROWS   Q^+ of the final standardisation
struct SavedLayout {
  int32_t n, q, v, tau, vpre, w, nc, contacts, x, b, cls, cfm, pflag, rest, total;   // rest: MAX_CONTACTS rows, restitution coefficient of the contacts that bounced
  int32_t A, massed, aall, pinv, dense;
  int32_t ldr;        // leading dimension of A / massed / aall / pinv: MAX_ROWS in the 24- / 48-row builds; in the general builds the model's own rows
                      // (3 x max_contacts rounded up to 8: genLeadingDim), so that record and scratch are sized by the model, not by the cap
  int32_t treeRows;   // doubles per world of the tree block after the dense block (0: tree state not saved, backward recomputes)
  int32_t treeNbp;    // 0: lane-interleaved rows [body * WS_KEEP + slot][B];  > 0: compact world-major blocks [b]{[TREE_ROWS][treeNbp], [nFree][TREE_FREE]} (coop tree kernels)
};
// Extra lane-interleaved scratch rows after the per-body workspace (contact stage)
constexpr int LW_JA = 0;                               // MAX_ROWS x 6 body-frame wrench on body A per row
constexpr int LW_JB = LW_JA + MAX_ROWS * 6;            // same for body B
constexpr int LW_TOTAL = LW_JB + MAX_ROWS * 6;         // (the per-world factorisations live in LDS, not here)
// contact backward scratch
constexpr int LCP_LANES = 16;                             // worlds per workgroup in the LDS-staged dense kernels
constexpr int LCP_LDS_BYTES = 2 * MAX_ROWS * MAX_ROWS * 8 * LCP_LANES;
constexpr int LB_LAM1 = LW_TOTAL;                          // lambda1 = M^-1 g
constexpr int LB_GVP = LB_LAM1 + MAX_DOF_CONTACT;          // cotangent of the pre-contact velocity
constexpr int LB_QX = LB_GVP + MAX_DOF_CONTACT;            // extra position cotangent from the contact stage
constexpr int LB_S = LB_QX + MAX_DOF_CONTACT;              // 3 x n   s_k = M^-1 A_c alpha_k
constexpr int LB_P = LB_S + 3 * MAX_DOF_CONTACT;           // 3 x n   p_k = M^-1 Abar beta_k
constexpr int LB_COEF = LB_P + 3 * MAX_DOF_CONTACT;        // MAX_ROWS x 8 coefficients of z_row on the bases
constexpr int LB_FLAG = LB_COEF + MAX_ROWS * 8;            // 1: contact adjoint active
constexpr int LB_VX = LB_FLAG + 1;                         // extra velocity cotangent: the bounce approximation applied to velPos^T gq' (k_bwd_bounce)
constexpr int LB_TOTAL = LB_VX + MAX_DOF_CONTACT;

}  // namespace NBL_NS
