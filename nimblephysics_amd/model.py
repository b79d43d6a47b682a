"""Model descriptions for the batched timestep.

One *model* (tree topology, joint types/axes, fixed transforms, spatial inertias, box colliders,
friction) is shared by all worlds of a batch.  This module is the host-side replacement for the
small part of `nimble.simulation.World` / `dart::dynamics::Skeleton` construction that the hot path
needs (reference: dart/simulation/World.cpp:93-172, dart/utils/urdf/DartLoader.cpp:388-560,
dart/utils/SkelParser.cpp).  It carries no dynamics; it only produces the flat arrays of
`struct nbl_model_desc` (include/nimble_amd.h).

Multi-DOF joints whose transform is a product of one-parameter motions - Euler (any axis order), universal, translational,
translational-2D and planar joints - are expanded at construction time into chains of revolute / prismatic joints through
MASSLESS virtual links (`compound_chain`): T_pj * M_1(q_0) ... M_k(q_{k-1}) * T_cj^-1 is literally the reference's
updateRelativeTransform for these joints (EulerJoint.cpp:1333-1340 with Geometry.cpp:1767-1797, UniversalJoint.cpp:193-201,
TranslationalJoint.cpp:127-135, TranslationalJoint2D.cpp:232-242, PlanarJoint.cpp:296-307), the generalized coordinates and
their order are the same, positions integrate the same way (q += dt v), so the dynamics and every derivative are the
reference's - the kernels only ever see 1-DOF joints.  Ball joints (exponential coordinates) cannot be written this way.

Weld joints are merged into their parents at build time (`merge_welds`): a welded body and its
parent are one rigid body, so the merged model is mathematically identical to the reference's
0-DOF WeldJoint treatment (dart/dynamics/WeldJoint.cpp, Skeleton.cpp:12588-12608) while the GPU
kernels only ever see 1-DOF and free joints.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _abi

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def rpy_to_matrix(rpy: Sequence[float]) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix (R = Rz(y) Ry(p) Rx(r))."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ],
        dtype=np.float64,
    )


def make_transform(xyz=(0.0, 0.0, 0.0), rpy=(0.0, 0.0, 0.0), R: Optional[np.ndarray] = None) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = rpy_to_matrix(rpy) if R is None else np.asarray(R, dtype=np.float64)
    T[:3, 3] = np.asarray(xyz, dtype=np.float64)
    return T


def _inv(T: np.ndarray) -> np.ndarray:
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


def _t12(T: np.ndarray) -> np.ndarray:
    return np.concatenate([T[:3, :3].reshape(9), T[:3, 3]])


@dataclass
class BodySpec:
    name: str
    parent: int  # index into the body list, -1 = world
    joint_type: str  # revolute | prismatic | free | weld, or a compound type (COMPOUND_JOINTS) expanded by ModelDescription
    joint_name: str = ""
    axis: Sequence[float] = (0.0, 0.0, 1.0)   # 1-DOF joints; compound joints: `axes`
    T_pj: np.ndarray = field(default_factory=lambda: np.eye(4))
    T_cj: np.ndarray = field(default_factory=lambda: np.eye(4))
    mass: float = 1.0
    com: Sequence[float] = (0.0, 0.0, 0.0)
    inertia: Sequence[float] = (1.0, 1.0, 1.0, 0.0, 0.0, 0.0)  # ixx iyy izz ixy ixz iyz about COM
    damping: Sequence[float] = ()
    spring: Sequence[float] = ()
    rest: Sequence[float] = ()
    pos_lo: Sequence[float] = ()
    pos_hi: Sequence[float] = ()
    vel_lo: Sequence[float] = ()
    vel_hi: Sequence[float] = ()
    force_lo: Sequence[float] = ()
    force_hi: Sequence[float] = ()
    friction: float = 1.0  # BodyNodeAspect.hpp:47
    axes: Sequence[Sequence[float]] = ()   # compound joints with free axes (universal: 2, translational2d: 2, planar: 2 in-plane axes)
    beta: Sequence[float] = (1.0, 1.0, 1.0)   # BodyNode::mBeta (BodyNode.cpp:1301 ctor default ones): the COM moves along beta under an INERTIA_COM_MU mass entry
    limit_enforced: bool = False   # Joint::isPositionLimitEnforced (JointAspect.hpp:165: off by default): the joint's position limits become LCP rows
    self_collision: bool = False        # Skeleton::isEnabledSelfCollisionCheck of the body's skeleton (off by default): colliders of one skeleton meet
    adjacent_body_check: bool = False   # Skeleton::isEnabledAdjacentBodyCheck: ... also those of a body and its parent
    skeleton: int = -1   # index of the dart Skeleton the body belongs to; -1 (every body of the model) = one skeleton per tree
    pitch: float = 0.1   # screw joints: translation along the axis per turn (ScrewJoint::mPitch, default 0.1)


# ---- compound joints: (kind of each one-parameter motion, default axes) ------------------------------------------------
_E = {"x": (1.0, 0.0, 0.0), "y": (0.0, 1.0, 0.0), "z": (0.0, 0.0, 1.0)}
COMPOUND_JOINTS = {"universal": 2, "translational": 3, "translational2d": 2, "planar": 3,
                   **{"euler_" + o: 3 for o in ("xyz", "zyx", "zxy", "xzy", "yxz", "yzx")}}


def compound_chain(b: "BodySpec"):
    """[(joint type, axis)] of the one-parameter motions of compound joint b, in DOF order."""
    jt = b.joint_type
    ax = [tuple(float(x) for x in a) for a in b.axes]
    if jt.startswith("euler_"):
        # R = R_a(q0) R_b(q1) R_c(q2) for order "abc" (eulerXYZToMatrix = Rx Ry Rz, Geometry.cpp:1767-1797); `axes` may carry the
        # flipped unit axes of EulerJoint's flipAxisMap (R_a(f q) = R_{f a}(q))
        return [("revolute", tuple(ax[k]) if ax else _E[c]) for k, c in enumerate(jt[6:])]
    if jt == "universal":
        if len(ax) != 2:
            raise ValueError(f"{b.name}: a universal joint needs two axes")
        return [("revolute", ax[0]), ("revolute", ax[1])]
    if jt == "translational":
        return [("prismatic", _E["x"]), ("prismatic", _E["y"]), ("prismatic", _E["z"])]
    if jt in ("translational2d", "planar"):
        if not ax:
            ax = [_E["x"], _E["y"]]                       # the XY plane (both joints' default)
        if len(ax) != 2:
            raise ValueError(f"{b.name}: {jt} needs two in-plane axes")
        a1 = np.asarray(ax[0], dtype=np.float64); a1 = a1 / np.linalg.norm(a1)
        a2 = np.asarray(ax[1], dtype=np.float64); a2 = a2 / np.linalg.norm(a2)
        if jt == "translational2d":
            return [("prismatic", tuple(float(x) for x in a1)), ("prismatic", tuple(float(x) for x in a2))]
        d = float(a1 @ a2)                                 # setArbitraryPlane (PlanarJointAspect.cpp:115-136)
        if abs(d) > 1e-6:
            a2 = a2 - d * a1; a2 = a2 / np.linalg.norm(a2)
        rot = np.cross(a1, a2); rot = rot / np.linalg.norm(rot)
        return [("prismatic", tuple(float(x) for x in a1)), ("prismatic", tuple(float(x) for x in a2)), ("revolute", tuple(float(x) for x in rot))]
    raise ValueError(f"{b.name}: unknown compound joint {jt}")


def expand_compound_joints(bodies, boxes):
    """Replace every compound joint by its chain of 1-DOF joints through massless virtual links.  Returns (bodies, boxes, index
    of each input body in the output list)."""
    if not any(b.joint_type in COMPOUND_JOINTS for b in bodies):
        return list(bodies), list(boxes), list(range(len(bodies)))
    out, where = [], []
    for b in bodies:
        parent = -1 if b.parent < 0 else where[b.parent]
        if b.joint_type not in COMPOUND_JOINTS:
            nb = BodySpec(**{**b.__dict__}); nb.parent = parent
            where.append(len(out)); out.append(nb)
            continue
        chain = compound_chain(b)
        k = len(chain)

        def dof(vals, i):
            vals = tuple(vals)
            if len(vals) not in (0, k):
                raise ValueError(f"{b.name}: per-DOF property with {len(vals)} entries on a {k}-DOF joint")
            return (vals[i],) if vals else ()
        for i, (jt, axis) in enumerate(chain):
            last = i == k - 1
            nb = BodySpec(
                b.name if last else f"{b.name}#v{i}", parent, jt, f"{b.joint_name or b.name}_{i}", axis=axis,
                T_pj=np.array(b.T_pj, dtype=np.float64) if i == 0 else np.eye(4), T_cj=np.array(b.T_cj, dtype=np.float64) if last else np.eye(4),
                mass=b.mass if last else 0.0, com=tuple(b.com) if last else (0.0, 0.0, 0.0),
                inertia=tuple(b.inertia) if last else (0.0,) * 6,
                **{key: dof(getattr(b, key), i) for key in ("damping", "spring", "rest", "pos_lo", "pos_hi", "vel_lo", "vel_hi", "force_lo", "force_hi")},
                friction=b.friction, beta=tuple(b.beta) if last else (1.0, 1.0, 1.0), limit_enforced=b.limit_enforced, self_collision=b.self_collision, adjacent_body_check=b.adjacent_body_check, skeleton=b.skeleton)
            parent = len(out)
            out.append(nb)
        where.append(len(out) - 1)
    # colliders follow their bodies; their BodyNode identity for the adjacent-body rule of self-collision (BodyNodeCollisionFilter::
    # areAdjacentBodies compares getParentBodyNode, CollisionFilter.cpp:150-154) is the REAL body and its REAL parent - not the massless
    # virtual link '#v*' the chain put between them
    def node_of(bx):
        if bx.body < 0:
            return -1, -2
        if bx.node >= 0:                      # identities from an earlier stage (merge_welds): keep them
            return bx.node, bx.node_parent
        p = bodies[bx.body].parent
        return where[bx.body], (where[p] if p >= 0 else -1)
    nboxes = [BoxSpec(bx.body if bx.body < 0 else where[bx.body], bx.T, tuple(bx.size), bx.mu, bx.shape, bx.restitution, *node_of(bx)) for bx in boxes]
    return out, nboxes, where

@dataclass
class BoxSpec:
    """A collider: a box (full side lengths `size`), with shape = "sphere" a sphere of radius size[0], or with shape = "capsule" a
    capsule of radius size[0] and cylinder height size[1] along the z axis of its frame (dynamics::CapsuleShape)."""
    body: int  # -1 = fixed to the world
    T: np.ndarray
    size: Sequence[float]
    mu: float = 1.0
    shape: str = "box"
    restitution: float = 0.0   # BodyNode restitution coefficient of the owning body (default 0: no bounce)
    node: int = -1             # after merge_welds: the body of the ORIGINAL description that carried the collider (BodyNode identity for
    node_parent: int = -2      # the adjacent-body rule of self-collision) and that body's parent there; -1 / -2: `body` and its parent


def SphereSpec(body: int, T: np.ndarray, radius: float, mu: float = 1.0) -> "BoxSpec":
    return BoxSpec(body, T, (float(radius),) * 3, mu, "sphere")


def CapsuleSpec(body: int, T: np.ndarray, radius: float, height: float, mu: float = 1.0) -> "BoxSpec":
    return BoxSpec(body, T, (float(radius), float(height), 0.0), mu, "capsule")


SHAPE_CODES = {"box": 0, "sphere": 1, "capsule": 2}


def _inertia_matrix(i6) -> np.ndarray:
    ixx, iyy, izz, ixy, ixz, iyz = i6
    return np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], dtype=np.float64)


def _inertia6(I: np.ndarray):
    return (I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2])


class ModelDescription:
    """Flat, topologically sorted description of one articulated model + its colliders."""

    def __init__(self, name: str, bodies: List[BodySpec], boxes: Optional[List[BoxSpec]] = None,
                 gravity=(0.0, -9.81, 0.0), dt: float = 1e-3, action_map: Optional[Sequence[int]] = None,
                 max_contacts: int = 0, contact_clipping_depth: float = 0.03, fallback_cfm: float = 1e-4,
                 penetration_correction: bool = False):
        self.name = name
        # compound joints -> chains of 1-DOF joints (module docstring); body_index[i] = where input body i ended up
        self.bodies, self.boxes, self.body_index = expand_compound_joints(list(bodies), list(boxes or []))
        self.gravity = tuple(float(g) for g in gravity)
        self.dt = float(dt)
        self._action_map = None if action_map is None else [int(a) for a in action_map]
        self.max_contacts = int(max_contacts)
        self.contact_clipping_depth = float(contact_clipping_depth)
        self.fallback_cfm = float(fallback_cfm)
        self.penetration_correction = bool(penetration_correction)   # World::setPenetrationCorrectionEnabled, off by default
        for i, b in enumerate(self.bodies):
            if not (-1 <= b.parent < i):
                raise ValueError(f"body {i} ({b.name}): parent {b.parent} must precede it")
            if b.joint_type not in _abi.JOINT_NAMES:
                raise ValueError(f"unsupported joint type {b.joint_type!r} (hot-path scope: revolute, prismatic, free, ball, weld and the compound joints)")

    # ---- sizes ------------------------------------------------------------------------------
    def joint_ndof(self, i: int) -> int:
        return _abi.JOINT_NDOF[_abi.JOINT_NAMES[self.bodies[i].joint_type]]

    @property
    def num_dofs(self) -> int:
        return sum(self.joint_ndof(i) for i in range(len(self.bodies)))

    @property
    def action_map(self) -> List[int]:
        # default action space = all DOFs (World.cpp:2053-2058)
        return list(range(self.num_dofs)) if self._action_map is None else list(self._action_map)

    def set_action_space(self, mapping: Sequence[int]):
        n = self.num_dofs
        mapping = [int(a) for a in mapping]
        for a in mapping:
            if a < 0 or a >= n:
                raise ValueError(f"action mapping {a} out of bounds [0,{n})")  # World.cpp:2118-2135 prints+ignores
        self._action_map = mapping

    def dof_names(self) -> List[str]:
        out = []
        for b in self.bodies:
            nd = _abi.JOINT_NDOF[_abi.JOINT_NAMES[b.joint_type]]
            if nd == 1:
                out.append(b.joint_name or b.name)
            elif nd == 6:
                out += [f"{b.joint_name or b.name}_{s}" for s in ("rot_x", "rot_y", "rot_z", "pos_x", "pos_y", "pos_z")]
        return out

    # ---- weld merging -----------------------------------------------------------------------
    def merge_welds(self) -> "ModelDescription":
        """Return an equivalent model without weld joints (see module docstring)."""
        bodies = self.bodies
        nb = len(bodies)
        # T_fix[i]: frame of body i expressed in the frame of the body it is merged into
        target = list(range(nb))
        T_in_target = [np.eye(4) for _ in range(nb)]
        acc = {}  # target index -> (mass, com, I_about_com)
        keep = []
        for i, b in enumerate(bodies):
            if b.joint_type == "weld":
                T = b.T_pj @ _inv(b.T_cj)  # child frame in parent frame (constant)
                if b.parent < 0:
                    target[i] = -1
                    T_in_target[i] = T
                else:
                    target[i] = target[b.parent]
                    T_in_target[i] = T_in_target[b.parent] @ T
            else:
                keep.append(i)
            t = target[i]
            if t >= 0:
                Tt = T_in_target[i]
                R, p = Tt[:3, :3], Tt[:3, 3]
                m_i = float(b.mass)
                c_i = R @ np.asarray(b.com, dtype=np.float64) + p
                I_i = R @ _inertia_matrix(b.inertia) @ R.T
                if t not in acc:
                    acc[t] = []
                acc[t].append((m_i, c_i, I_i))
        new_index = {old: k for k, old in enumerate(keep)}
        out = []
        for old in keep:
            b = bodies[old]
            parts = acc[old]
            M = sum(p[0] for p in parts)
            if M > 0:
                com = sum(p[0] * p[1] for p in parts) / M
            else:
                com = np.zeros(3)
            I = np.zeros((3, 3))
            for m_i, c_i, I_i in parts:
                d = c_i - com
                I += I_i + m_i * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            if b.parent < 0:
                parent_new, T_pj = -1, b.T_pj
            else:
                tp = target[b.parent]
                if tp < 0:  # parent chain welded to the world
                    parent_new, T_pj = -1, T_in_target[b.parent] @ b.T_pj
                else:
                    parent_new, T_pj = new_index[tp], T_in_target[b.parent] @ b.T_pj
            nbdy = BodySpec(**{**b.__dict__})
            nbdy.parent = parent_new
            nbdy.T_pj = T_pj
            nbdy.mass = M
            nbdy.com = tuple(com)
            nbdy.inertia = _inertia6(I)
            out.append(nbdy)
        boxes = []
        for bx in self.boxes:
            if bx.body < 0:
                boxes.append(BoxSpec(-1, bx.T.copy(), tuple(bx.size), bx.mu, bx.shape, bx.restitution))
            else:
                t = target[bx.body]
                Tb = T_in_target[bx.body] @ bx.T
                node = bx.node if bx.node >= 0 else bx.body
                boxes.append(BoxSpec(-1 if t < 0 else new_index[t], Tb, tuple(bx.size), bx.mu, bx.shape, bx.restitution, node,
                                     bx.node_parent if bx.node >= 0 else bodies[bx.body].parent))
        m = ModelDescription(self.name, out, boxes, self.gravity, self.dt, self._action_map, self.max_contacts,
                             self.contact_clipping_depth, self.fallback_cfm, self.penetration_correction)
        return m

    def weld_targets(self):
        """For every body: (index of the body of merge_welds() that carries it, or -1 for the world; its frame in that body)."""
        target, T_in, keep = list(range(len(self.bodies))), [np.eye(4) for _ in self.bodies], []
        for i, b in enumerate(self.bodies):
            if b.joint_type == "weld":
                T = b.T_pj @ _inv(b.T_cj)
                target[i] = -1 if b.parent < 0 else target[b.parent]
                T_in[i] = T if b.parent < 0 else T_in[b.parent] @ T
            else:
                keep.append(i)
        new_index = {old: k for k, old in enumerate(keep)}
        return [(-1 if t < 0 else new_index[t]) for t in target], T_in

    def body_skeletons(self) -> List[int]:
        """Per body: the dart Skeleton it belongs to - the ids the loaders recorded, or (none recorded) the index of its tree's root.
        The reference solves one LCP per constrained group of skeletons (ConstraintSolver.cpp:724-780) and never collides two
        bodies of one skeleton."""
        if all(b.skeleton >= 0 for b in self.bodies):
            return [int(b.skeleton) for b in self.bodies]
        # untagged bodies (-1): one skeleton per tree, numbered after the recorded ids so that a tagged and an untagged model
        # merged into one world (loaders.with_ground) neither share a skeleton nor split one
        base = max([int(b.skeleton) for b in self.bodies if b.skeleton >= 0], default=-1) + 1
        out, tree_id = [], {}
        for i, b in enumerate(self.bodies):
            if b.skeleton >= 0:
                out.append(int(b.skeleton))
            elif b.parent < 0 or self.bodies[b.parent].skeleton >= 0:
                tree_id[i] = base + len(tree_id)        # the root of an untagged (sub)tree
                out.append(tree_id[i])
            else:
                out.append(out[b.parent])
        return out

    def has_welds(self) -> bool:
        return any(b.joint_type == "weld" for b in self.bodies)

    # ---- flat arrays / C struct -------------------------------------------------------------
    def flat(self) -> dict:
        nb = len(self.bodies)
        n = self.num_dofs
        inf = float("inf")
        a = {
            "parent": np.zeros(nb, np.int32), "joint_type": np.zeros(nb, np.int32), "dof_offset": np.zeros(nb, np.int32),
            "T_pj": np.zeros((nb, 12)), "T_cj": np.zeros((nb, 12)), "axis": np.zeros((nb, 3)), "mass": np.zeros(nb),
            "com": np.zeros((nb, 3)), "inertia": np.zeros((nb, 6)),
            "damping": np.zeros(n), "spring": np.zeros(n), "rest": np.zeros(n), "dof_limit_enforced": np.zeros(n, np.int32),
            "pos_lo": np.full(n, -inf), "pos_hi": np.full(n, inf), "vel_lo": np.full(n, -inf), "vel_hi": np.full(n, inf),
            "force_lo": np.full(n, -inf), "force_hi": np.full(n, inf),
        }
        off = 0
        for i, b in enumerate(self.bodies):
            jt = _abi.JOINT_NAMES[b.joint_type]
            nd = _abi.JOINT_NDOF[jt]
            a["parent"][i] = b.parent
            a["joint_type"][i] = jt
            a["dof_offset"][i] = off
            a["T_pj"][i] = _t12(np.asarray(b.T_pj, dtype=np.float64))
            a["T_cj"][i] = _t12(np.asarray(b.T_cj, dtype=np.float64))
            ax = np.asarray(b.axis, dtype=np.float64)
            nrm = np.linalg.norm(ax)
            a["axis"][i] = ax / nrm if nrm > 0 else ax  # RevoluteJoint::setAxis normalizes
            a["mass"][i] = b.mass
            a["com"][i] = b.com
            a["inertia"][i] = b.inertia
            for key in ("damping", "spring", "rest", "pos_lo", "pos_hi", "vel_lo", "vel_hi", "force_lo", "force_hi"):
                vals = getattr(b, key)
                if len(vals):
                    if len(vals) != nd:
                        raise ValueError(f"body {b.name}: {key} has {len(vals)} entries, joint has {nd} dofs")
                    a[key][off:off + nd] = vals
            a["dof_limit_enforced"][off:off + nd] = 1 if b.limit_enforced else 0
            off += nd
        nbx = len(self.boxes)
        # capsule-box pairs run libccd's MPR in the reference (DARTCollide.cpp:4422-4645), an iterative third-party algorithm outside the
        # closed-form narrow phases of this path: refuse a model in which such a pair is tested (same rule as CollisionFilter.cpp:105-154)
        if self.capsule_meets_box():
            raise ValueError("a capsule collider can meet a box collider: the reference runs libccd's MPR on that pair, outside the "
                             "closed-form narrow phases (capsule-capsule, capsule-sphere) of this path")
        a["box_body"] = np.array([bx.body for bx in self.boxes], np.int32).reshape(nbx)
        a["box_T"] = np.array([_t12(bx.T) for bx in self.boxes], np.float64).reshape(nbx, 12)
        a["box_size"] = np.array([bx.size for bx in self.boxes], np.float64).reshape(nbx, 3)
        a["box_mu"] = np.array([bx.mu for bx in self.boxes], np.float64).reshape(nbx)
        a["box_shape"] = np.array([SHAPE_CODES[bx.shape] for bx in self.boxes], np.int32).reshape(nbx)
        a["box_restitution"] = np.array([bx.restitution for bx in self.boxes], np.float64).reshape(nbx)
        a["box_node"] = np.array([bx.node if bx.node >= 0 else bx.body for bx in self.boxes], np.int32).reshape(nbx)
        a["box_node_parent"] = np.array([bx.node_parent if bx.node >= 0 else (self.bodies[bx.body].parent if bx.body >= 0 else -1) for bx in self.boxes], np.int32).reshape(nbx)
        a["action_map"] = np.array(self.action_map, np.int32)
        a["body_skeleton"] = np.array(self.body_skeletons(), np.int32).reshape(nb)
        a["pitch"] = np.array([b.pitch for b in self.bodies], np.float64).reshape(nb)
        a["body_self_collision"] = np.array([(1 if b.self_collision else 0) | (2 if b.adjacent_body_check else 0) for b in self.bodies], np.int32).reshape(nb)
        return a

    def colliders_are_tested(self, bi: "BoxSpec", bj: "BoxSpec", skel=None) -> bool:
        """BodyNodeCollisionFilter::ignoresCollision (CollisionFilter.cpp:105-154) for two colliders of this model: not on one body, not both
        fixed to the world; on one skeleton only with self-collision enabled and, unless the adjacent-body check is on, not on a body
        and its parent."""
        ni, nj = (bi.node if bi.node >= 0 else bi.body), (bj.node if bj.node >= 0 else bj.body)
        if bi.body == bj.body and (bi.body < 0 or ni == nj):
            return False                      # (two BodyNodes welded into one body stay two nodes for this rule)
        if bi.body < 0 or bj.body < 0:
            return True
        skel = self.body_skeletons() if skel is None else skel
        if skel[bi.body] != skel[bj.body]:
            return True
        a, b = self.bodies[bi.body], self.bodies[bj.body]
        if not (a.self_collision and b.self_collision):
            return False
        pi, pj = (bi.node_parent if bi.node >= 0 else a.parent), (bj.node_parent if bj.node >= 0 else b.parent)
        if not (a.adjacent_body_check and b.adjacent_body_check) and (pi == nj or pj == ni):
            return False
        return True

    def suggest_max_contacts(self) -> int:
        """Contact slots per world for a description that does not say (the loaders' default): the smallest of the library's three budgets that
        the collider pairs of the model cannot exceed in their usual configurations - 8 (the 24-row build, the fast one), 16 (the 48-row
        build) or, beyond that, what the pairs can hold rounded up to a multiple of 8, at most 64 (the GENERAL build: rows looped over, 29 % of the
        24-row build's rate on eight-contact worlds, no truncated answers: a tower of ten cubes holds 40 contacts).  Counted per pair that is tested at all (CollisionFilter.cpp:105-154):
        a box on a world-fixed box 4 points (the ground's face contains the other one), two moving boxes 8 (dBoxBox's clipped octagon: the
        reference keeps every point, DARTCollide.cpp:1384-1448), capsule pairs 2, every other pair 1; of the pairs of one moving collider with
        several others only as many as can touch it at once are counted (a body has 6 faces; in a pile at most 2 face contacts of 8 + 4 of 4).
        The reference itself keeps every contact (ConstraintSolver.cpp:563-606); a world that exceeds its model's slots is truncated and
        flagged NBL_ST_CONTACT_OVERFLOW."""
        if not self.boxes:
            return 0
        m = self.merge_welds() if self.has_welds() else self
        skel = m.body_skeletons()
        est = 0
        for i, bi in enumerate(m.boxes):
            for bj in m.boxes[i + 1:]:
                if m.colliders_are_tested(bi, bj, skel):
                    both_move = bi.body >= 0 and bj.body >= 0
                    est += (8 if both_move else 4) if (bi.shape == "box" and bj.shape == "box") else (2 if (bi.shape == "capsule" and bj.shape == "capsule") else 1)
        if est <= 8:
            return 8
        if est <= 16:
            return 16
        # many pairs: not all of them can be in contact at once - per moving collider at most 2 x 8 + 4 x 4 points, shared between the two sides
        movers = sum(1 for bx in m.boxes if bx.body >= 0)
        est = min(est, 16 * max(1, movers))
        slots = min(64, (est + 7) // 8 * 8)     # (a caller who expects more - boxes turned against each other touch in octagons - asks for up to 128 itself)
        # this default lands on the GENERAL instantiation of the contact stage - correct for every world of up to `slots` contacts,
        # several times slower than the 24- / 48-row builds and with a per-world scratch that grows with the square of the rows: say so once
        import warnings
        ld = (3 * slots + 7) // 8 * 8
        warnings.warn(f"{self.name}: the collider pairs of this model can hold about {est} contacts at once: defaulting to max_contacts = {slots}, which runs "
                      f"on the general build of the contact stage (rows looped over: several times slower than the 8- / 16-slot builds; "
                      f"{(5 * ld * ld + 16 * ld) * 8 / 1e6:.2f} MB of workspace per world, {(5 * ld * ld + 16 * ld) * 8 * 4096 / 1e9:.1f} GB at 4096 worlds).  "
                      f"Pass max_contacts=16 (or 8) to the loader / set ModelDescription.max_contacts to stay on the fast builds: worlds with more "
                      f"contacts are then truncated and flagged NBL_ST_CONTACT_OVERFLOW.", stacklevel=3)
        return slots

    def capsule_meets_box(self) -> bool:
        """Some capsule collider is tested against some box collider (different bodies, not both fixed to the world, different skeletons:
        the pairs CollisionFilter.cpp:105-154 lets through)."""
        skel = self.body_skeletons()
        for i, bi in enumerate(self.boxes):
            for bj in self.boxes[i + 1:]:
                if {bi.shape, bj.shape} == {"capsule", "box"} and self.colliders_are_tested(bi, bj, skel):
                    return True
        return False

    def to_desc(self):
        """Returns (ModelDesc, keepalive). The keepalive must outlive any use of the struct."""
        a = self.flat()
        for k in a:
            a[k] = np.ascontiguousarray(a[k])
        d = _abi.ModelDesc()
        d.n_bodies = len(self.bodies)
        d.n_dofs = self.num_dofs

        def pd(x):
            return x.ctypes.data_as(C.POINTER(C.c_double))

        def pi(x):
            return x.ctypes.data_as(C.POINTER(C.c_int32))

        for k in ("parent", "joint_type", "dof_offset", "box_body", "action_map", "box_shape", "body_skeleton", "dof_limit_enforced", "body_self_collision", "box_node", "box_node_parent"):
            setattr(d, k, pi(a[k]))
        for k in ("T_pj", "T_cj", "axis", "mass", "com", "inertia", "damping", "spring", "rest", "pos_lo", "pos_hi",
                  "vel_lo", "vel_hi", "force_lo", "force_hi", "box_T", "box_size", "box_mu", "box_restitution", "pitch"):
            setattr(d, k, pd(a[k]))
        d.gravity = (C.c_double * 3)(*self.gravity)
        d.dt = self.dt
        d.n_action = len(a["action_map"])
        d.n_boxes = len(self.boxes)
        # (a joint-limit row takes one of the contact slots of the LCP: a model without colliders that enforces limits still needs them)
        d.max_contacts = self.max_contacts or (8 if any(b.limit_enforced for b in self.bodies) else 0)
        d.contact_clipping_depth = self.contact_clipping_depth
        d.fallback_cfm = self.fallback_cfm
        d.penetration_correction = 1 if self.penetration_correction else 0
        return d, a

    # ---- (de)serialisation ------------------------------------------------------------------
    def to_json(self) -> dict:
        def body(b: BodySpec):
            d = {k: (np.asarray(v).tolist() if isinstance(v, (np.ndarray, tuple, list)) else v) for k, v in b.__dict__.items()
                 if k != "axes" and not (k == "skeleton" and v < 0) and not (k == "beta" and tuple(v) == (1.0, 1.0, 1.0)) and not (k in ("limit_enforced", "self_collision", "adjacent_body_check") and not v) and not (k == "pitch" and b.joint_type != "screw")}   # compound joints are already expanded: every stored joint has its single `axis`
            return d
        return {
            "name": self.name, "gravity": list(self.gravity), "dt": self.dt, "action_map": self._action_map,
            "max_contacts": self.max_contacts, "contact_clipping_depth": self.contact_clipping_depth,
            "fallback_cfm": self.fallback_cfm, **({"penetration_correction": True} if self.penetration_correction else {}), "bodies": [body(b) for b in self.bodies],
            "boxes": [{"body": bx.body, "T": np.asarray(bx.T).tolist(), "size": list(bx.size), "mu": bx.mu,
                       **({} if bx.shape == "box" else {"shape": bx.shape}),
                       **({} if bx.restitution == 0.0 else {"restitution": bx.restitution}),
                       **({} if bx.node < 0 else {"node": bx.node, "node_parent": bx.node_parent})} for bx in self.boxes],
        }

    @staticmethod
    def from_json(d: dict) -> "ModelDescription":
        bodies = []
        for b in d["bodies"]:
            b = dict(b)
            b["T_pj"] = np.array(b["T_pj"], dtype=np.float64)
            b["T_cj"] = np.array(b["T_cj"], dtype=np.float64)
            bodies.append(BodySpec(**b))
        boxes = [BoxSpec(bx["body"], np.array(bx["T"], dtype=np.float64), tuple(bx["size"]), bx.get("mu", 1.0), bx.get("shape", "box"), bx.get("restitution", 0.0), bx.get("node", -1), bx.get("node_parent", -2)) for bx in d.get("boxes", [])]
        return ModelDescription(d["name"], bodies, boxes, d.get("gravity", (0, -9.81, 0)), d.get("dt", 1e-3),
                                d.get("action_map"), d.get("max_contacts", 0), d.get("contact_clipping_depth", 0.03),
                                d.get("fallback_cfm", 1e-4), d.get("penetration_correction", False))

    @staticmethod
    def load(name_or_path: str) -> "ModelDescription":
        path = name_or_path if os.path.exists(name_or_path) else os.path.join(_DATA_DIR, name_or_path + ".json")
        with open(path) as f:
            return ModelDescription.from_json(json.load(f))


# ---------------------------------------------------------------------------------------------
# The BASELINE.json configs
# ---------------------------------------------------------------------------------------------
def single_pendulum() -> ModelDescription:
    """cfg1: data/skel/test/single_pendulum.skel - revolute z, m = 5, I = diag(1, 2, 3), damping 10 - as parsed by
    nimblephysics_amd.loaders.load_skel (SkelParser conventions) and committed as data/single_pendulum.json by
    tools/urdf_to_model.py (body world transform (0.1, 0, 0), joint frame (-0.1, 0, 0) in the child => T_pj = identity)."""
    return ModelDescription.load("single_pendulum")


def cartpole() -> ModelDescription:
    """cfg2: unittests/comprehensive/test_Gradients.cpp:1260-1316 — prismatic x + revolute z,
    pole joint offset (0,-0.5,0) from the child body, default unit mass / identity inertia."""
    sled = BodySpec("sled", -1, "prismatic", "sled_joint", axis=(1, 0, 0))
    arm = BodySpec("arm", 0, "revolute", "arm_joint", axis=(0, 0, 1), T_cj=make_transform((0, -0.5, 0)))
    return ModelDescription("cartpole", [sled, arm], gravity=(0, -9.81, 0), dt=1e-3)


def atlas(variant: str = "atlas33", ground: bool = False) -> ModelDescription:
    """cfg3/cfg5/metric: Atlas transcribed from data/sdf/atlas/atlas_v3_box_colliders.urdf by
    tools/urdf_to_model.py (parameters only; committed as nimblephysics_amd/data/*.json).

    variant "atlas33": free root + 27 revolutes (n = 33); "atlas20": arms and back_bkz welded (n = 20).
    ground=True adds data/sdf/atlas/ground.urdf's 25 x 0.05 x 25 box at y = -0.95 (top face y = -0.925).
    """
    m = ModelDescription.load(variant + ("_ground" if ground else ""))
    return m


def box_stack() -> ModelDescription:
    """cfg4: data/skel/test/box_stacking.skel restricted to the welded ground box (2 x 0.01 x 2 at y = -0.5) and its first two
    FreeJoint cubes (side 0.2, mass 0.1, shape inertia), mu = 1 - parsed by load_skel and committed as data/box_stack.json by
    tools/urdf_to_model.py.  Two cubes = 8 contacts = the device path's row budget."""
    return ModelDescription.load("box_stack")
