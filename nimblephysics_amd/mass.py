"""Inertia ("mass") parameters of a World: the host-side mirror of `neural::WithRespectToMass`.

Reference: dart/neural/WithRespectToMass.{hpp,cpp} (entry types :25-33, dim :31-47, set :50-140, get :143-),
World::tuneMass / getMassDims / getMasses / setMasses (dart/simulation/World.cpp:1014-1053, 1821-1824).

Each registered entry contributes `dim()` scalars to the world's mass vector.  For every scalar the device needs the
direction dG in which the 6x6 spatial inertia of the carrying body moves (nbl_set_inertia_params); the spatial tensor is
    G = [[I + m C C^T, m C], [m C^T, m 1]],  C = skew(com)                     (Inertia.cpp:1368-1383)
so the directions are closed form.  INERTIA_MASS goes through BodyNode::setMass -> Inertia::setMass(mass,
preserveDimsAndEuler = true), which rescales the moment with the mass (Inertia.cpp:157-179, 881-904): G is linear in m.
"""
from __future__ import annotations

import enum
from typing import List, Sequence

import numpy as np

from .model import ModelDescription, _inertia_matrix


class WrtMassBodyNodeEntryType(enum.IntEnum):
    INERTIA_MASS = 0
    INERTIA_COM = 1
    INERTIA_COM_MU = 2
    INERTIA_DIAGONAL = 3
    INERTIA_OFF_DIAGONAL = 4
    INERTIA_FULL = 5


_DIMS = {WrtMassBodyNodeEntryType.INERTIA_MASS: 1, WrtMassBodyNodeEntryType.INERTIA_COM: 3,
         WrtMassBodyNodeEntryType.INERTIA_COM_MU: 1,
         WrtMassBodyNodeEntryType.INERTIA_DIAGONAL: 3, WrtMassBodyNodeEntryType.INERTIA_OFF_DIAGONAL: 3,
         WrtMassBodyNodeEntryType.INERTIA_FULL: 10}


def _skew(c):
    return np.array([[0.0, -c[2], c[1]], [c[2], 0.0, -c[0]], [-c[1], c[0], 0.0]])


def spatial_inertia(mass: float, com, inertia6) -> np.ndarray:
    Cm = _skew(np.asarray(com, dtype=np.float64))
    G = np.zeros((6, 6))
    G[:3, :3] = _inertia_matrix(inertia6) + mass * Cm @ Cm.T
    G[:3, 3:] = mass * Cm
    G[3:, :3] = mass * Cm.T
    G[3:, 3:] = mass * np.eye(3)
    return G


def _adjoint_inverse(T: np.ndarray) -> np.ndarray:
    """6x6 matrix taking a twist [w; v] of the carrying body's frame to the frame at pose T inside it."""
    R, p = T[:3, :3], T[:3, 3]
    X = np.zeros((6, 6))
    X[:3, :3] = R.T
    X[3:, :3] = -R.T @ _skew(p)
    X[3:, 3:] = R.T
    return X


_OFF = ((0, 1), (0, 2), (1, 2))


class WrtMassEntry:
    def __init__(self, body: int, type: WrtMassBodyNodeEntryType, upper: np.ndarray, lower: np.ndarray):
        self.body, self.type, self.upper, self.lower = body, type, upper, lower

    def dim(self) -> int:
        return _DIMS[self.type]

    # WrtMassBodyNodyEntry::get (WithRespectToMass.cpp:143-)
    def get(self, model: ModelDescription) -> np.ndarray:
        b = model.bodies[self.body]
        t = WrtMassBodyNodeEntryType
        if self.type == t.INERTIA_MASS:
            return np.array([b.mass], dtype=np.float64)
        if self.type == t.INERTIA_COM:
            return np.asarray(b.com, dtype=np.float64).copy()
        if self.type == t.INERTIA_COM_MU:   # the first axis the scaling group moves (WithRespectToMass.cpp:157-166)
            k = 0 if b.beta[0] != 0 else (1 if b.beta[1] != 0 else 2)
            return np.array([b.com[k] / b.beta[k]], dtype=np.float64)
        if self.type == t.INERTIA_DIAGONAL:
            return np.asarray(b.inertia[:3], dtype=np.float64).copy()
        if self.type == t.INERTIA_OFF_DIAGONAL:
            return np.asarray(b.inertia[3:], dtype=np.float64).copy()
        return np.concatenate([[b.mass], b.com, b.inertia]).astype(np.float64)

    # WrtMassBodyNodyEntry::set (WithRespectToMass.cpp:50-140)
    def set(self, model: ModelDescription, value: np.ndarray):
        b = model.bodies[self.body]
        t = WrtMassBodyNodeEntryType
        if self.type == t.INERTIA_MASS:
            new = float(value[0])
            if not new > 0:
                raise ValueError("mass must be positive")
            if b.mass > 0 and any(x != 0 for x in b.inertia):
                b.inertia = tuple(float(x) * new / b.mass for x in b.inertia)   # same box dimensions, new mass
            b.mass = new
        elif self.type == t.INERTIA_COM:
            b.com = tuple(float(x) for x in value)
        elif self.type == t.INERTIA_COM_MU:   # COM = beta * mu (WithRespectToMass.cpp:76-92)
            b.com = tuple(float(be) * float(value[0]) for be in b.beta)
        elif self.type == t.INERTIA_DIAGONAL:
            b.inertia = tuple(float(x) for x in value) + tuple(b.inertia[3:])
        elif self.type == t.INERTIA_OFF_DIAGONAL:
            b.inertia = tuple(b.inertia[:3]) + tuple(float(x) for x in value)
        else:
            b.mass = float(value[0])
            b.com = tuple(float(x) for x in value[1:4])
            b.inertia = tuple(float(x) for x in value[4:10])

    def directions(self, model: ModelDescription) -> List[np.ndarray]:
        """dG/dtheta (6x6, the body's own frame) for each scalar of this entry at the current values."""
        b = model.bodies[self.body]
        m, c = float(b.mass), np.asarray(b.com, dtype=np.float64)
        Cm = _skew(c)
        t = WrtMassBodyNodeEntryType

        def d_mass_only():
            D = np.zeros((6, 6))
            D[:3, :3] = Cm @ Cm.T
            D[:3, 3:] = Cm
            D[3:, :3] = Cm.T
            D[3:, 3:] = np.eye(3)
            return D

        def d_com(k):
            Ck = _skew(np.eye(3)[k])
            D = np.zeros((6, 6))
            D[:3, :3] = m * (Ck @ Cm.T + Cm @ Ck.T)
            D[:3, 3:] = m * Ck
            D[3:, :3] = m * Ck.T
            return D

        def d_diag(k):
            D = np.zeros((6, 6))
            D[k, k] = 1.0
            return D

        def d_off(k):
            D = np.zeros((6, 6))
            i, j = _OFF[k]
            D[i, j] = D[j, i] = 1.0
            return D

        if self.type == t.INERTIA_MASS:
            if b.mass > 0 and any(x != 0 for x in b.inertia):
                return [spatial_inertia(m, c, b.inertia) / m]
            return [d_mass_only()]
        if self.type == t.INERTIA_COM:
            return [d_com(k) for k in range(3)]
        if self.type == t.INERTIA_COM_MU:
            return [sum(float(b.beta[k]) * d_com(k) for k in range(3))]
        if self.type == t.INERTIA_DIAGONAL:
            return [d_diag(k) for k in range(3)]
        if self.type == t.INERTIA_OFF_DIAGONAL:
            return [d_off(k) for k in range(3)]
        return [d_mass_only()] + [d_com(k) for k in range(3)] + [d_diag(k) for k in range(3)] + [d_off(k) for k in range(3)]


class WithRespectToMass:
    """The registered entries of one world, in registration order.  INERTIA_COM_MU reads BodySpec.beta (BodyNode::getBeta, set
    by the reference's scaling groups; ones by default)."""

    def __init__(self, description: ModelDescription):
        self.description = description
        self.entries: List[WrtMassEntry] = []

    def registerNode(self, body, type=WrtMassBodyNodeEntryType.INERTIA_MASS, upperBound=None, lowerBound=None) -> WrtMassEntry:
        if isinstance(body, str):
            names = [b.name for b in self.description.bodies]
            if body not in names:
                raise KeyError(f"no body named {body!r}")
            body = names.index(body)
        type = WrtMassBodyNodeEntryType(type)
        if any(e.body == body for e in self.entries):
            raise ValueError("body already registered")   # the reference keeps one entry per node (WithRespectToMass.cpp registerNode)
        d = _DIMS[type]
        up = np.full(d, np.inf) if upperBound is None else np.asarray(upperBound, dtype=np.float64).reshape(d)
        lo = np.full(d, -np.inf if type != WrtMassBodyNodeEntryType.INERTIA_MASS else 0.0) if lowerBound is None \
            else np.asarray(lowerBound, dtype=np.float64).reshape(d)
        e = WrtMassEntry(int(body), type, up, lo)
        self.entries.append(e)
        return e

    def dim(self) -> int:
        return sum(e.dim() for e in self.entries)

    def get(self) -> np.ndarray:
        return np.concatenate([e.get(self.description) for e in self.entries]) if self.entries else np.zeros(0)

    def set(self, values: Sequence[float]):
        values = np.asarray(values, dtype=np.float64).reshape(-1)
        if values.shape[0] != self.dim():
            raise ValueError(f"mass vector has {values.shape[0]} entries, world has {self.dim()} mass dims")
        cur = 0
        for e in self.entries:
            e.set(self.description, values[cur:cur + e.dim()])
            cur += e.dim()

    def upperBound(self) -> np.ndarray:
        return np.concatenate([e.upper for e in self.entries]) if self.entries else np.zeros(0)

    def lowerBound(self) -> np.ndarray:
        return np.concatenate([e.lower for e in self.entries]) if self.entries else np.zeros(0)

    def device_table(self):
        """(carrying body of the weld-merged model, dG 6x6 in that body's frame) for every scalar parameter."""
        targets, T_in = self.description.weld_targets()
        bodies, dGs = [], []
        for e in self.entries:
            t = targets[e.body]
            if t < 0:
                raise ValueError(f"body {self.description.bodies[e.body].name!r} is welded to the world: its inertia has no effect")
            X = _adjoint_inverse(T_in[e.body])
            for D in e.directions(self.description):
                bodies.append(t)
                dGs.append(X.T @ D @ X)
        return np.asarray(bodies, dtype=np.int32), np.asarray(dGs, dtype=np.float64).reshape(-1, 36)
