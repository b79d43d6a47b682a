/*
 * nimble_amd.h — C ABI of the MI355X-native batched differentiable timestep.
 *
 * This is the drop-in boundary for ONE hot path of nimblephysics: the call
 *   nimble.timestep(world, state, action)                      python/nimblephysics/timestep.py:63-69
 * which in the reference goes through pybind11 into
 *   neural::forwardPass(world)                                  dart/neural/NeuralUtils.cpp:26-66
 *     -> World::step(true)                                      dart/simulation/World.cpp:221-254
 *   BackpropSnapshot::backpropState(world, grad)                dart/neural/BackpropSnapshot.cpp:382-420
 *
 * Everything here is plain C: pointers, sizes, ints.  No torch types.
 * All batched arrays are DEVICE pointers to fp64 in structure-of-arrays layout
 *   x[d * B + b]      d = DOF (or row) index, b = world index   ("[dof][B]")
 * so that one wavefront (64 consecutive worlds) reads one coalesced 512-byte line
 * per DOF.  The caller owns every buffer; the library owns only the model handle
 * and a workspace sized at nbl_workspace_bytes().
 *
 * Threading: one handle may be used from one host thread / one stream at a time
 * (the reference's World is not thread-safe either, World.cpp:114-172).
 * Multi-GPU: one handle per device, the batch is sharded by the caller.
 *
 * Return value of every int function: 0 = ok, <0 = error (see NBL_E_*).  This
 * replaces the reference's "print to std::cerr and ignore the call"
 * (World.cpp:2027-2033, 2063-2070) with a status the caller must check.
 */
#ifndef NIMBLE_AMD_H
#define NIMBLE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- joint types (dart/dynamics/{Revolute,Prismatic,Free,Weld}Joint.cpp) ---- */
#define NBL_JOINT_REVOLUTE 0
#define NBL_JOINT_PRISMATIC 1
#define NBL_JOINT_FREE 2 /* DART_USE_IDENTITY_JACOBIAN build: S = Ad(T_cj), FreeJoint.cpp:1049-1056.  Anywhere in the tree (as a tree root: the
                            fast path; below other bodies: six coincident single-axis joints internally, like NBL_JOINT_BALL) */
#define NBL_JOINT_WELD 3 /* 0 DOF. The GPU library requires welds to be merged into the parent
                            (host model builder does this); the CPU oracle accepts them. */
#define NBL_JOINT_SCREW 5 /* 1 DOF, ScrewJoint.cpp:160-232: rotation about `axis` coupled with a translation of `pitch` per turn along it,
                             S = Ad(T_cj) [axis; axis pitch / 2 pi], T = T_pj expMap(S_local q) T_cj^-1 */
#define NBL_JOINT_BALL 4 /* 3 DOF, BallJoint.cpp (DART_USE_IDENTITY_JACOBIAN build): positions = exponential-map vector of the joint
                            rotation, velocities = angular velocity in the child joint frame, S = Ad(T_cj)[:, 0:3] (:441-452),
                            q' = log(exp(q) exp(v dt)) (:333-349).  Anywhere in the tree.  The library runs it as three coincident
                            single-axis joints (x, y, z at zero angle, the first carrying exp(q)) - identical velocity-level dynamics -
                            and takes position derivatives through H(q) = [expMapJac(q)^T; 0] (:282-289); body indices of the
                            description stay valid in every entry point. */

/* ---- error codes ---- */
#define NBL_OK 0
#define NBL_E_BADARG -1
#define NBL_E_UNSUPPORTED -2 /* model feature outside the hot-path scope */
#define NBL_E_HIP -3         /* a HIP runtime call failed (nbl_last_error() has text) */
#define NBL_E_WORKSPACE -4   /* workspace too small for this B */
#define NBL_E_NOGPU -5

/* ---- per-lane status bits written to status[b] by nbl_step_forward ----
 * A world with several constrained groups (body_skeleton below) runs the solver once per group: STAGE0 and STANDARDIZED are set
 * when they hold for EVERY group, the other bits when they hold for ANY group. */
#define NBL_ST_CONTACT 0x1u       /* >=1 contact constraint was active */
#define NBL_ST_LCP_STAGE0 0x2u    /* warm-start / guess classification was a valid LCP solution (BoxedLcpConstraintSolver.cpp:434-457) */
#define NBL_ST_LCP_PIVOT 0x4u     /* pivoting (Dantzig-equivalent) stage used */
#define NBL_ST_LCP_PGS 0x8u       /* CFM + PGS fallback used */
#define NBL_ST_LCP_NOFRIC 0x10u   /* friction dropped fallback used */
#define NBL_ST_LCP_FAILED 0x20u   /* every stage failed its validity check: like the reference, the last stage's (frictionless PGS) iterate is applied as is
                                     (BoxedLcpConstraintSolver.cpp:590-676); the impulses are zeroed only if that iterate is non-finite (:678-687, NBL_ST_NAN) */
#define NBL_ST_NAN 0x40u          /* non-finite value seen: in the LCP stages, or in the world's next state (NaN / Inf inputs); other worlds are unaffected */
#define NBL_ST_CONTACT_OVERFLOW 0x80u /* more contacts (+ active joint-limit rows) than max_contacts: extra ones dropped; or a contact kept after
                                        2 x nbl_model_max_contacts() distinct points from the narrow phases (kept or dropped by the depth filter):
                                        the duplicate filter's memory */
#define NBL_ST_STANDARDIZED 0x100u /* least-squares standardized x replaced solver x (CGGM.cpp:321-332) */
#define NBL_ST_JOINT_LIMIT 0x400u  /* >=1 joint-limit constraint row was active (dof_limit_enforced) */
#define NBL_ST_GRAD_PARTIAL 0x200u /* reserved (never set: the EDGE_EDGE contact-geometry gradient terms, DCC.cpp:397-424,
                                      700-735, are evaluated by the device backward) */

/*
 * Model description.  One model is shared by all B worlds of a batch; worlds differ only in
 * (q, v, tau) and the LCP warm start.  Bodies are listed parents-before-children.
 * Every body has exactly one parent joint.  Transforms are 12 doubles: R row-major (9) then p (3).
 */
typedef struct nbl_model_desc {
  int32_t n_bodies;
  int32_t n_dofs;
  const int32_t* parent;     /* [n_bodies] parent body index, -1 = world */
  const int32_t* joint_type; /* [n_bodies] NBL_JOINT_* */
  const int32_t* dof_offset; /* [n_bodies] first DOF of the parent joint */
  const double* T_pj;        /* [n_bodies][12] parent body -> joint  (Joint::mT_ParentBodyToJoint) */
  const double* T_cj;        /* [n_bodies][12] child body  -> joint  (Joint::mT_ChildBodyToJoint) */
  const double* axis;        /* [n_bodies][3] unit axis for revolute/prismatic */
  const double* mass;        /* [n_bodies] */
  const double* com;         /* [n_bodies][3] local COM (Inertia::mCenterOfMass) */
  const double* inertia;     /* [n_bodies][6] Ixx Iyy Izz Ixy Ixz Iyz about the COM (Inertia.cpp:1368-1383) */
  const double* damping;     /* [n_dofs] GenericJoint mDampingCoefficients */
  const double* spring;      /* [n_dofs] mSpringStiffnesses */
  const double* rest;        /* [n_dofs] mRestPositions */
  const double* pos_lo;      /* [n_dofs] limits, used only by clipLossGradientsToBounds (BackpropSnapshot.cpp:425-479) */
  const double* pos_hi;
  const double* vel_lo;
  const double* vel_hi;
  const double* force_lo;
  const double* force_hi;
  double gravity[3]; /* World::mGravity, default (0,-9.81,0) in the configs */
  double dt;         /* World::mTimeStep, default 1e-3 (World.cpp:76) */

  /* action space: tau[action_map[i]] = action[i], unmapped tau = 0 (World.cpp:2061-2086) */
  int32_t n_action;
  const int32_t* action_map; /* [n_action] */

  /* ---- contact: box colliders (dBoxBox, DARTCollide.cpp:764-1450) and spheres (box_shape below) ---- */
  int32_t n_boxes;
  const int32_t* box_body;  /* [n_boxes] body index, -1 = fixed to the world (immobile skeleton) */
  const double* box_T;      /* [n_boxes][12] shape transform in the body frame */
  const double* box_size;   /* [n_boxes][3] full side lengths */
  const double* box_mu;     /* [n_boxes] friction coefficient of the owning body (default 1, BodyNodeAspect.hpp:47) */
  int32_t max_contacts;     /* per world, <= 128; rows m = 3 * max_contacts.  The reference keeps every contact of every pair
                               (ConstraintSolver.cpp:563-606); a world with more than its model's slots is truncated and flagged
                               NBL_ST_CONTACT_OVERFLOW.  The library holds three instantiations of its contact stage: models with max_contacts
                               <= 8, <= 16 colliders and <= 32 collider pairs run the 24-row one (the fast one), up to 16 contacts / 32
                               colliders / 64 pairs the 48-row one, everything up to 64 contacts (192 rows) / 64 colliders / 512 pairs the
                               GENERAL one, whose dense kernels loop over the rows (since round 6 at 2 M world-steps/s on eight-contact worlds, 29 % of the 24-row
                               build's rate; on worlds that fill 12 - 16 slots the 48-row one is 2 - 3.7 x faster), roomy enough that a model can always be given
                               the slots its colliders can fill (a tower of ten cubes: 40 contacts in one constrained group).  A model that
                               asks for 65 .. 128 slots gets the same general code with 384 rows (four times the scratch and record per world:
                               ten cubes each turned against the next touch in clipped octagons, 80 contacts) */

  /* ---- options mirrored from the reference defaults ---- */
  double contact_clipping_depth; /* 0.03  World.cpp:86 */
  double fallback_cfm;           /* 1e-4  World.cpp:85 */

  /* ---- collider shapes (appended; NULL = every collider is a box) ----
   * [n_boxes] NBL_SHAPE_BOX | NBL_SHAPE_SPHERE | NBL_SHAPE_CAPSULE.  A sphere's radius is box_size[3*i]; sphere-box, box-sphere and
   * sphere-sphere pairs follow collideSphereBox / collideBoxSphere / collideSphereSphere (DARTCollide.cpp:1482-1880).
   * A capsule (CapsuleShape: axis = z of the shape frame) has radius box_size[3*i] and cylinder height box_size[3*i+1];
   * capsule-capsule, sphere-capsule and capsule-sphere pairs follow collideCapsuleCapsule / collideSphereCapsule /
   * collideCapsuleSphere (DARTCollide.cpp:4183-4420).  A model in which a capsule can meet a BOX is refused: that pair runs
   * libccd's MPR in the reference (DARTCollide.cpp:4422-4645), a third-party iterative algorithm outside this path. */
  const int32_t* box_shape;

  /* ---- restitution (appended; NULL = 0 everywhere, the reference's default BodyNodeAspect.hpp:48) ----
   * [n_boxes] restitution coefficient of the owning body.  A contact bounces when e = e_A * e_B > 1e-3 and e times its approach
   * speed exceeds 0.1 m/s: b_normal += min(e * b_normal, 100) (ContactConstraint.cpp:95-110, 395-442); the backward pass carries
   * the bounce diagonals 1 + e and the reference's bounce approximation of the position Jacobians
   * (BackpropSnapshot.cpp:1131-1226). */
  const double* box_restitution;

  /* ---- penetration correction (appended; 0 = off, the reference's default: ConstraintSolver.cpp:69-71, "it breaks our gradients") ----
   * World::setPenetrationCorrectionEnabled(true): b_normal += min(max(depth - 0, 0) * 0.01 / dt, 1e-3), unless the contact bounces
   * harder than that (ContactConstraint.cpp:393-441 with DART_ERROR_ALLOWANCE / DART_ERP / DART_MAX_ERV, :45-47).  Like the
   * reference's analytical Jacobians the backward pass treats the correction velocity as a constant. */
  int32_t penetration_correction;

  /* ---- skeletons (appended; NULL = every tree of the model is its own skeleton) ----
   * [n_bodies] index of the dart::dynamics::Skeleton a body belongs to.  The reference solves one LCP per CONSTRAINED GROUP: the
   * skeletons connected by contacts between two reactive bodies (ConstraintSolver.cpp:724-780, ContactConstraint.cpp:879-907;
   * contacts with world-fixed colliders do not connect anything).  Each group runs the solver cascade on its own, so one
   * object that needs the fallback stages does not change the solution of the others. */
  const int32_t* body_skeleton;

  /* ---- screw joints (appended; NULL = 0.1 for every screw joint, ScrewJointAspect's default) ----
   * [n_bodies] ScrewJoint::mPitch: translation along the axis per full turn; read for NBL_JOINT_SCREW bodies only. */
  const double* pitch;

  /* ---- joint-limit constraint rows (appended; NULL = none, the reference's default: Joint::isPositionLimitEnforced is false,
   * JointAspect.hpp:165) ----
   * [n_dofs] non-zero: the DOF's joint enforces its position limits (Joint::setPositionLimitEnforced).  A DOF at or beyond pos_lo /
   * pos_hi then adds one row to the LCP of its skeleton's constrained group, after the contact rows (JointLimitConstraint.cpp:182-290,
   * ConstraintSolver.cpp:641-696): unit impulse on the DOF, b = -qdot (error allowance 0), bounds [0, inf) at the lower and
   * (-inf, 0] at the upper limit.  Every joint type except the free-joint root (refused when such a coordinate has a finite limit);
   * a limit row takes one of the max_contacts contact slots.  The backward pass follows the reference's: its
   * DifferentiableContactConstraint gives a non-contact constraint a zero constraint-force column (DCC.cpp:51-99), so the row drops out
   * of every Jacobian. */
  const int32_t* dof_limit_enforced;

  /* ---- self-collision (appended; NULL = off, the reference's default: Skeleton::mEnabledSelfCollisionCheck is false) ----
   * [n_bodies] bit 0: the body's skeleton checks self-collisions (Skeleton::enableSelfCollisionCheck), bit 1: also between adjacent
   * bodies (enableAdjacentBodyCheck).  Two colliders of one skeleton are tested when bit 0 is set on both bodies and, unless bit 1
   * is set, neither body is the other's parent (BodyNodeCollisionFilter::ignoresCollision, CollisionFilter.cpp:105-154).  A DOF
   * above both bodies of such a contact moves it rigidly (DofContactType::SELF_COLLISION, DCC.cpp:116-130) and gets no
   * constraint force from it (getControlForceMultiple 0). */
  const int32_t* body_self_collision;
  /* [n_boxes] (appended; NULL = the collider's body and that body's parent) the BodyNode a collider belongs to and that node's parent, in
   * any numbering: "adjacent" above means box_node_parent[i] == box_node[j] or the reverse.  A caller that merges welded bodies before
   * nbl_model_create passes the identities of the unmerged BodyNodes here (a body welded to its neighbour's child is not adjacent to it). */
  const int32_t* box_node;
  const int32_t* box_node_parent;
} nbl_model_desc;

#define NBL_SHAPE_BOX 0
#define NBL_SHAPE_SPHERE 1
#define NBL_SHAPE_CAPSULE 2

typedef struct nbl_model nbl_model; /* opaque */

/* Human-readable text for the last error on this thread. */
const char* nbl_last_error(void);

/* Library/ABI version (major<<16 | minor).  The minor number counts the revisions that APPENDED fields to the model description struct: a NULL
 * pointer / zero in an appended field always means "the behaviour before that field existed"; a caller that zero-initialises
 * the struct and is compiled against this header keeps working, a caller compiled against minor k needs a library of minor >= k:
 *   minor 1: the struct up to and including pitch;
 *   minor 2: + dof_limit_enforced, body_self_collision, box_node, box_node_parent; NBL_SHAPE_CAPSULE; NBL_ST_JOINT_LIMIT;
 *   minor 3: max_contacts up to 16 (32 colliders, 64 pairs); + nbl_model_max_contacts, nbl_selftest_pinv_rows; the Dantzig self-test takes n <= 48.
 *   minor 4: max_contacts up to 64 (64 colliders, 512 pairs: the general instantiation); the Dantzig self-test takes n <= 192;
 *            nbl_workspace_bytes of such a model includes 1.5 MB of scratch matrices per world; max_contacts 65 .. 128: a second general
 *            instantiation of 384 rows (5.9 MB of scratch per world), the Dantzig self-test then takes n <= 384.
 *   minor 5: + nbl_set_deferred_join, nbl_slice_stream, nbl_fork_slices, nbl_join_slices (one handle, slices that are not joined per call). */
#define NBL_ABI_MINOR 5
int32_t nbl_version(void);

/* Number of visible HIP devices (0 if none). */
int32_t nbl_device_count(void);

/*
 * Create a model on `device`.  Copies everything out of desc.
 * Replaces: constructing a nimble.simulation.World + skeletons (World.cpp:93-172).
 */
int32_t nbl_model_create(const nbl_model_desc* desc, int32_t device, nbl_model** out);
void nbl_model_destroy(nbl_model* m);

int32_t nbl_model_num_dofs(const nbl_model* m);
int32_t nbl_model_num_action(const nbl_model* m);
int32_t nbl_model_lcp_rows(const nbl_model* m); /* rows of the LCP warm-start buffer: 3 * nbl_model_max_contacts() impulses + 1 row holding the row count they belong to; 0 without colliders */
int32_t nbl_model_max_contacts(const nbl_model* m); /* contact slots per world of the instantiation the model runs on: 8, 16, 64 or 128 (>= desc.max_contacts); 0 without colliders */

/* Bytes of scratch the library needs for a batch of B worlds (forward or backward). */
size_t nbl_workspace_bytes(const nbl_model* m, int64_t B);

/*
 * Bytes of the per-step "saved for backward" record for B worlds: what the reference keeps in a
 * BackpropSnapshot (dart/neural/BackpropSnapshot.cpp:33-118): q_t, v_t, tau_t, the pre-step LCP
 * cache and the constraint-group results.  Caller-owned device memory.
 */
size_t nbl_saved_bytes(const nbl_model* m, int64_t B);

/*
 * One differentiable timestep for B worlds.
 *   state   [2n][B]  = [q; v]                              World::setState   World.cpp:2024-2036
 *   action  [k][B]                                          World::setAction  World.cpp:2061-2086
 *   lcp_cache_in  [m][B] or NULL (cold: guessSolution)      BoxedLcpConstraintSolver.cpp:202-208
 *   next_state [2n][B]                                      World::getState   World.cpp:2040-2047
 *   lcp_cache_out [m][B] or NULL
 *   saved      nbl_saved_bytes() bytes or NULL (no backward wanted)
 *   status     [B] uint32 or NULL
 * `stream` is a hipStream_t (void* so this header needs no HIP include); NULL = default stream.
 * Replaces: neural::forwardPass (NeuralUtils.cpp:26-66).
 */
int32_t nbl_step_forward(nbl_model* m, int64_t B, const double* state, const double* action,
                         const double* lcp_cache_in, double* next_state, double* lcp_cache_out,
                         void* saved, uint32_t* status, void* workspace, size_t workspace_bytes,
                         void* stream);

/*
 * Vector-Jacobian product of the step.
 *   grad_next_state [2n][B]  dL/d[q';v']
 *   grad_state      [2n][B]  dL/d[q;v]        (lossWrtState)
 *   grad_action     [k][B]   dL/daction       (lossWrtAction)
 * Replaces: BackpropSnapshot::backpropState (BackpropSnapshot.cpp:382-420), including
 * clipLossGradientsToBounds (:425-479).
 */
int32_t nbl_step_backward(nbl_model* m, int64_t B, const void* saved, const double* grad_next_state,
                          double* grad_state, double* grad_action, void* workspace,
                          size_t workspace_bytes, void* stream);

/*
 * Inertia ("mass") parameters: World::tuneMass / setMasses / lossWrtMass
 * (dart/simulation/World.cpp:1027-1035, 1821-1824; dart/neural/WithRespectToMass.cpp:50-140;
 *  dart/neural/BackpropSnapshot.cpp:153, 167-179, 580-640).
 *
 * nbl_set_body_inertia   replaces the inertial constants of one body (host pointers; com[3]; inertia[6] = Ixx Iyy Izz Ixy
 *                        Ixz Iyz about the COM, as in nbl_model_desc).  Synchronises the device: call it between steps.
 * nbl_set_inertia_params registers `count` scalar parameters theta_p: parameter p moves the spatial inertia of body
 *                        bodies[p] along dG[p] (6x6 row-major, symmetrised) - e.g. G/m for WrtMassBodyNodeEntryType::
 *                        INERTIA_MASS, whose setter scales the whole tensor (Inertia.cpp:157-179).  count = 0 clears them.
 * nbl_backward_inertia   grad_params [count][B]: dL/dtheta_p of every world for the step recorded in `saved`.  Call it
 *                        right after nbl_step_backward of the same record on the same stream and workspace (it uses the
 *                        adjoint joint rates that call leaves in the workspace).  accumulate != 0 adds to grad_params.
 *                        The reference finite-differences M^-1 and C for these parameters (Skeleton.cpp:1826-1829,
 *                        2078-2081); this is the closed form (csrc/inertia_backward.hip).
 */
int32_t nbl_set_body_inertia(nbl_model* m, int32_t body, double mass, const double* com, const double* inertia);
/* The same for `count` bodies at once (host pointers: bodies[count], mass[count], com[count][3], inertia[count][6]) as ONE
 * stream-ordered copy on `stream`: launches issued on that stream afterwards see the new constants; no device synchronisation
 * and the calling thread's current device is left as it was.  A backward pass reads the model's CURRENT constants: run the
 * backward of a step before changing the masses for the next one. */
int32_t nbl_set_body_inertias(nbl_model* m, int32_t count, const int32_t* bodies, const double* mass, const double* com,
                              const double* inertia, void* stream);
int32_t nbl_set_inertia_params(nbl_model* m, int32_t count, const int32_t* bodies, const double* dG);
/* The same, stream-ordered: a table of the size already registered (World::setMasses with new values) is ONE asynchronous copy on
 * `stream` - launches issued on that stream before the call read the old table, later ones the new one, no device synchronisation;
 * a table of another size is registration-time work and synchronises the device.  (nbl_set_inertia_params synchronises the device
 * around the copy in every case.) */
int32_t nbl_set_inertia_params_on(nbl_model* m, int32_t count, const int32_t* bodies, const double* dG, void* stream);
int32_t nbl_num_inertia_params(const nbl_model* m);
int32_t nbl_backward_inertia(nbl_model* m, int64_t B, const void* saved, double* grad_params, int32_t accumulate,
                             void* workspace, size_t workspace_bytes, void* stream);

/*
 * T-step trajectory rollout on the device: the loop of SingleShot::getStates /
 * SingleShot::backpropGradientWrt (dart/trajectory/SingleShot.cpp:539-598, 598-700) over forwardPass / backprop, without
 * a host round trip per step.
 *   forward:   states[0] = state0;  states[t+1] = step(states[t], actions[t])                       t = 0 .. T-1
 *     state0   [2n][B]
 *     actions  [T][k][B], or one [k][B] block used for every step when action_stride == 0 (else action_stride = k*B)
 *     states   [T+1][2n][B] out
 *     saved    T * nbl_saved_bytes(m, B) bytes (record t at byte offset t * nbl_saved_bytes), or NULL (no backward wanted)
 *     status   [T][B] uint32 or NULL
 *     warm_start != 0: step t starts the LCP from step t-1's solution (the reference's solver carries mX between steps,
 *                BoxedLcpConstraintSolver.cpp:176-187); 0: LCPUtils::guessSolution every step
 *   backward:  g_T = grad_states[T];  (g_t', grad_actions[t]) = step_backward(saved[t], g_{t+1});  g_t = g_t' + grad_states[t]
 *     grad_states  [T+1][2n][B]  dL/dstates[t] as it enters the loss directly (zeros where the loss does not look)
 *     grad_state0  [2n][B] out,  grad_actions [T][k][B] out
 * workspace: nbl_rollout_workspace_bytes(m, B) bytes.
 */
size_t nbl_rollout_workspace_bytes(const nbl_model* m, int64_t B);
int32_t nbl_rollout_forward(nbl_model* m, int64_t B, int32_t T, const double* state0, const double* actions,
                            int64_t action_stride, double* states, void* saved, uint32_t* status, int32_t warm_start,
                            void* workspace, size_t workspace_bytes, void* stream);
int32_t nbl_rollout_backward(nbl_model* m, int64_t B, int32_t T, const void* saved, const double* grad_states,
                             double* grad_state0, double* grad_actions, void* workspace, size_t workspace_bytes,
                             void* stream);
/* The same with the inertia ("mass") parameters of nbl_set_inertia_params: grad_params [count][B] receives the sum over the T
 * steps of dL/dtheta_p (the masses are constant along the trajectory); NULL = nbl_rollout_backward. */
int32_t nbl_rollout_backward_inertia(nbl_model* m, int64_t B, int32_t T, const void* saved, const double* grad_states,
                                     double* grad_state0, double* grad_actions, double* grad_params, void* workspace,
                                     size_t workspace_bytes, void* stream);

/*
 * Checkpointed rollout: the saved records of `segment` steps are resident instead of T (a record is 26.7 kB per world-step on the
 * metric model - nbl_saved_bytes / B -, the states 16 n bytes).  The forward pass writes record t into slot t % segment of `saved`
 * (segment * nbl_saved_bytes(m, B) bytes) and keeps the LCP warm start entering every segment in `checkpoints`
 * (nbl_rollout_checkpoint_bytes(m, B, T, segment) bytes; may be NULL when warm_start == 0 or the model has no colliders).  The
 * backward pass walks the segments from the last to the first: it runs the steps of a segment again from states[k * segment] -
 * the forward kernels are bit-reproducible, so the records are the ones the forward call produced and the gradients are bit for
 * bit those of the unsegmented rollout - and then backpropagates through them.  The last segment is still resident and is not
 * recomputed; cost: one extra forward pass over the other T - segment steps.  (The role of the reference's re-simulation from
 * a stored state, RestorableSnapshot + forwardPass, for trajectories that do not fit.)
 *   segment == 0 (or >= T): exactly nbl_rollout_forward / nbl_rollout_backward_inertia.
 *   backward: `states`, `actions`, `action_stride`, `warm_start` as passed to / returned by the forward call (states[t0+1 .. t1] of
 *   a recomputed segment are rewritten with identical values); grad_params may be NULL.
 */
size_t nbl_rollout_checkpoint_bytes(const nbl_model* m, int64_t B, int32_t T, int32_t segment);
int32_t nbl_rollout_forward_checkpointed(nbl_model* m, int64_t B, int32_t T, int32_t segment, const double* state0, const double* actions,
                                         int64_t action_stride, double* states, void* saved, void* checkpoints, uint32_t* status,
                                         int32_t warm_start, void* workspace, size_t workspace_bytes, void* stream);
int32_t nbl_rollout_backward_checkpointed(nbl_model* m, int64_t B, int32_t T, int32_t segment, double* states, const double* actions,
                                          int64_t action_stride, void* saved, const void* checkpoints, int32_t warm_start,
                                          const double* grad_states, double* grad_state0, double* grad_actions, double* grad_params,
                                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- self-test ----
 * Runs the library's device restatement of the reference's Dantzig driver (dSolveLCP, dart/external/odelcpsolver/lcp.cpp:780-1113,
 * nub = 0, earlyTermination = true; the stage-1 code of the LCP cascade) on `count` caller-supplied n-row problems, one wavefront
 * per problem.  HOST pointers: A [count][n*n] row-major (only the lower triangles are read), b / lo / hi / findex [count][n] with
 * the bounds as DantzigBoxedLcpSolver::solve hands them over (friction rows: lo = -mu, hi = mu, findex = their normal row);
 * outputs x [count][n] and rc [count] (1 solved, 0 early termination, -1 NaN step).  n <= 48 (n <= 24 runs the 24-row instantiation's
 * code, larger n the 48-row one's).  Synchronous; for tests: on identical
 * inputs x and rc are bit-identical to the reference solver's. */
int32_t nbl_selftest_lcp_dantzig(int32_t count, int32_t n, const double* A, const double* b, const double* lo, const double* hi,
                                 const int32_t* findex, double* x, int32_t* rc);
/* The same, launched `reps` times back to back between two HIP events (after one untimed launch): *ms_per_launch = average duration of
 * one launch over the `count` problems (NULL: not timed).  The micro-benchmark of the stage-1 solver (tools/dantzig_bench.py). */
int32_t nbl_selftest_lcp_dantzig_timed(int32_t count, int32_t n, const double* A, const double* b, const double* lo, const double* hi,
                                       const int32_t* findex, double* x, int32_t* rc, int32_t reps, double* ms_per_launch);
/* The solver cascade of the GENERAL instantiation (csrc/gen_lcp_dev.hpp, gen_dantzig_dev.hpp: what a model with max_contacts > 16 runs) on
 * caller-supplied contact LCPs (HOST pointers): `count` problems of m = 3 * contacts rows each (m <= 192): A [count][m][m], b [count][m],
 * mu [count][m / 3], x_cache [count][m] (the warm start; have_cache = 0: LCPUtils::guessSolution instead), on [count][m] bytes (NULL: every
 * row; else the rows of the constrained group at hand).  Per problem: stage 0 and, if that fails, stages 1-3 in the reference's order, then
 * the classification / standardisation: x [count][m], row classes cls [count][m] (0 / 1 / 2), status bits st [count] (NBL_ST_* of the LCP:
 * 0x2 stage 0, 0x4 Dantzig, 0x8 CFM + PGS, 0x10 friction dropped, 0x20 PGS not converged, 0x40 NaN, 0x100 standardised), cfm [count].
 * Exactly the device code of k_contact_solve_gen; tests/test_gpu_general.py compares it with the same code compiled for the host. */
int32_t nbl_selftest_lcp_cascade(int32_t count, int32_t m, const double* A, const double* b, const double* mu, int32_t have_cache,
                                 const double* x_cache, const uint8_t* on, double fallback_cfm, double* x, int32_t* cls, uint32_t* st, double* cfm);

/* Runs the library's device pseudo-inverses on `count` caller-supplied 24 x 24 matrices (HOST pointers: Q [count][24*24] row-major,
 * rows / columns outside the block of interest zero; cTrue [count] = size of that block, Eigen's `size` in the rank threshold), one
 * wavefront per matrix: route 0 = column-pivoted Householder QR + complete orthogonal decomposition (any matrix; what
 * CGGM.cpp:280 / LCPUtils.cpp:113 get from Eigen), route 1 = two Cholesky factorisations (symmetric positive semi-definite input:
 * A restricted to the guess rows, Q without upper-bound rows).  Outputs P [count][24*24] and rank [count]; launched `reps` times
 * between two HIP events, *ms_per_launch (may be NULL) = average launch duration.  For tests and tools/pinv_bench.py. */
int32_t nbl_selftest_pinv(int32_t count, const double* Q, const int32_t* cTrue, int32_t route, double* P, int32_t* rank, int32_t reps,
                          double* ms_per_launch);
/* The same on rows x rows matrices, rows = 24 or 48: the pseudo-inverses of the 24-row and of the 48-row instantiation of the contact stage. */
int32_t nbl_selftest_pinv_rows(int32_t count, int32_t rows, const double* Q, const int32_t* cTrue, int32_t route, double* P, int32_t* rank,
                               int32_t reps, double* ms_per_launch);

/*
 * Layout helpers: the Python surface takes world-major tensors [B][d] like a stack of the
 * reference's 1-D state vectors; these transpose to/from the library's [d][B] layout on device.
 */
int32_t nbl_transpose_to_soa(const double* src_bd, double* dst_db, int64_t B, int32_t d, void* stream);
int32_t nbl_transpose_from_soa(const double* src_db, double* dst_bd, int64_t B, int32_t d, void* stream);

/* Average duration (ms) of the last timed launches, measured with HIP events on the launch stream. */
/* Launch shape of the one-world-per-lane tree kernels (models over 64 bodies / DOFs, NBL_COOP_TREE=0, the narrow phase):
 * worlds per workgroup, a power of two <= 64; 0 = default (16 below 65536 worlds, else 64).  Results do not depend on it;
 * environment NBL_TREE_LANES sets the initial value.  lcp_lanes is accepted for source compatibility and ignored: the
 * dense contact kernels are one world per wavefront. */
int32_t nbl_set_launch_lanes(nbl_model* m, int32_t tree_lanes, int32_t lcp_lanes);
/* Batch slicing: the worlds of a call are processed as `slices` contiguous ranges whose kernels overlap on internal HIP
 * streams forked from / joined into the caller's stream (0 = default: 2 from 4096 worlds on for a model with colliders, else 1;
 * max 8; environment NBL_SLICES).  Results do not depend on it; every call joins before it returns, so a caller who owns the whole
 * forward + backward loop gets more from one model handle per slice on its own stream.  nbl_slices_for: the number a
 * call with B worlds will use. */
int32_t nbl_set_slices(nbl_model* m, int32_t slices);
int32_t nbl_slices_for(const nbl_model* m, int64_t B);
/* Deferred join (ABI minor 5): with it enabled nbl_step_forward / nbl_step_backward return with their slices in flight: slice 0 on the
 * `stream` of the call (always hand in the same one), the others on internal streams of the handle, none waiting for another and
 * nothing recorded or awaited on `stream` per call (four busy streams is what the hardware runs side by side: the caller's is one
 * of them).  The forward pass of one slice overlaps the
 * backward pass of another across consecutive calls on ONE handle (what a caller otherwise gets from one handle per slice on its
 * own stream; the reference has no counterpart: its World::step is synchronous on one CPU thread).  Ordering against the caller's
 * streams is explicit: nbl_fork_slices(m, stream) makes every slice stream wait for what `stream` holds (after uploading inputs
 * there), nbl_join_slices(m, stream) makes `stream` wait for everything the slices hold (before consuming results there).  A result
 * can also be consumed on its slice's own stream: nbl_slice_stream gives the stream (a hipStream_t; NULL for slice 0: the calls' own) and the world range
 * [first_world, end_world) of slice `slice` of a call with B worlds - enqueue the per-slice loss there.  The buffers of a call must
 * stay valid until its slices have run.  nbl_slices_for(m, B) is the slice count (4 from 4096 worlds on).  Results are bit for bit
 * those of the joined calls. */
int32_t nbl_set_deferred_join(nbl_model* m, int32_t enabled);
int32_t nbl_slice_stream(nbl_model* m, int64_t B, int32_t slice, void** stream, int64_t* first_world, int64_t* end_world);
int32_t nbl_fork_slices(nbl_model* m, void* stream);
int32_t nbl_join_slices(nbl_model* m, void* stream);
/* enabled = 0: off (and reset); 1: HIP events around every kernel launch; N > 1: around the launches of every N-th forward /
 * backward call only (sampling keeps the perturbation of a timed region below 1 %). */
int32_t nbl_set_timing(nbl_model* m, int32_t enabled);
int32_t nbl_get_timing(nbl_model* m, double* fwd_ms_sum, int64_t* fwd_count, double* bwd_ms_sum,
                       int64_t* bwd_count);
/* Per-kernel breakdown of the same measurements (names match the rocprofv3 kernel trace). */
int32_t nbl_kernel_count(void);
const char* nbl_kernel_name(int32_t i);
int32_t nbl_kernel_timing(nbl_model* m, int32_t i, double* ms_sum, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* NIMBLE_AMD_H */
