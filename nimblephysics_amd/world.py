"""`World`: the host-side mirror of the slice of `nimble.simulation.World` that `timestep` touches.

Reference API mirrored (python/_nimblephysics/simulation_and_neural/World.cpp:401, 502-523;
dart/simulation/World.cpp:2016-2135): getNumDofs, getStateSize, getActionSize, getActionSpace,
setActionSpace, setState, getState, setAction, getAction, getTimeStep, getGravity, step.

Differences, all forced by batching:
  * one World object holds B worlds that share a model; state/action carry a leading batch
    dimension [B, 2n] / [B, k] (a 1-D tensor behaves as B = 1);
  * tensors live on the GPU; the library works in DOF-major [d][B] layout internally;
  * size mismatches raise instead of "print to std::cerr and ignore the call" (World.cpp:2027-2033).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from .ref_layout import RefLayout

from . import _abi
from ._lib import NimbleAmdError, check, lib
from .mass import WithRespectToMass, WrtMassBodyNodeEntryType
from .model import ModelDescription


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class RolloutRecord:
    """What the backward pass of a checkpointed rollout needs (World.rollout_soa(checkpoint_every=K)): the K resident saved records,
    the LCP warm starts at the segment boundaries, and the states / actions of the forward call to run the other segments again."""

    def __init__(self, saved, segment, checkpoints, states, actions, stride, warm_start):
        self.saved, self.segment, self.checkpoints = saved, segment, checkpoints
        self.states, self.actions, self.stride, self.warm_start = states, actions, stride, warm_start

    def resident_bytes(self) -> int:
        return self.saved.numel() + self.checkpoints.numel()


class World:
    def __init__(self, model: ModelDescription, device: Optional[torch.device | int | str] = None):
        if not torch.cuda.is_available():
            raise NimbleAmdError("no HIP device visible: nimblephysics_amd has no CPU path (the CPU restatement under "
                                 "oracle/ is test infrastructure, not a fallback)")
        import copy
        self.description = copy.deepcopy(model)   # setMasses edits it
        model = self.description
        self.model = model.merge_welds() if model.has_welds() else model
        self._wrt_mass = WithRespectToMass(self.description)
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise NimbleAmdError("World must live on a GPU device")
        if self.device.index is None:                      # a bare "cuda": the current device, like torch does
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._L = lib()
        self._h = None
        self._ws = None
        self._ws_B = 0
        self._pinned = {}    # (role, shape) -> (pinned staging tensor, event of its last host-to-device copy)
        self._state = None   # [2n][B]
        self._action = None  # [k][B]
        self.lcp_cache = None  # [m][B] hidden warm start (BoxedLcpConstraintSolver::mX)
        self.last_status = None
        # the reference's coordinate layout when it has coordinates the device model does not (immobile skeletons: ref_layout.py)
        mob = getattr(self.description, "ref_dof_mobile", None)
        self.ref_layout = RefLayout(mob) if mob is not None and not all(mob) else None
        self._ref_action_map = list(range(self.ref_layout.n_ref)) if self.ref_layout is not None else None
        self._ref_state_like = None
        self._create_handle()
        if self.ref_layout is not None and self.ref_layout.n_dev != self.n:
            raise NimbleAmdError(f"the reference layout names {self.ref_layout.n_dev} mobile coordinates, the model has {self.n}")

    def _create_handle(self):
        """(Re-)upload the model constants.  The new handle is created FIRST: when the library refuses the description (a finite limit on a
        free-joint root, a capsule-box pair that self-collision exposes, ...) this raises and the World keeps its old handle; on success the
        old handle (device buffers, side streams, events) is released and everything sized by the handle is refreshed - the LCP row count
        changes when a collider-less model starts to enforce joint limits, and a warm start of the old row layout means nothing."""
        desc, keep = self.model.to_desc()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self._L.nbl_model_create(C.byref(desc), self.device.index, C.byref(h)), "nbl_model_create")
        self._destroy_handle()
        self._h, self._keep = h, keep
        self.n = self._L.nbl_model_num_dofs(self._h)
        self.k = self._L.nbl_model_num_action(self._h)
        self.m = self._L.nbl_model_lcp_rows(self._h)
        self.lcp_cache = None
        self._ws, self._ws_B, self._scratch_saved = None, 0, None
        self._uploaded_inertia = [(float(b.mass), tuple(float(x) for x in b.com), tuple(float(x) for x in b.inertia)) for b in self.model.bodies]

    def _destroy_handle(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)            # nothing in flight may still read the model buffers
            self._L.nbl_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    def clone(self) -> "World":
        """World::clone (World.cpp:114-172): an independent world with the same model, action space, registered mass parameters
        and current masses - the reference's unit of concurrency (one clone per thread), here one clone per HIP stream."""
        w = World(self.description, self.device)       # the description carries the action space and the current masses
        for e in self._wrt_mass.entries:
            w._wrt_mass.registerNode(e.body, e.type, e.upper, e.lower)
        if self._wrt_mass.entries:
            w._push_inertia_params()
        return w

    # ---- sizes / action space (World.cpp:2016-2135) -------------------------------------------
    def getNumDofs(self) -> int:
        """World::getNumDofs: every skeleton's coordinates, the immobile ones included (their layout: ref_layout.py)."""
        return self.ref_layout.n_ref if self.ref_layout is not None else self.n

    def getStateSize(self) -> int:
        return 2 * self.getNumDofs()

    def getActionSize(self) -> int:
        return len(self._ref_action_map) if self.ref_layout is not None else self.k

    def getActionSpace(self) -> List[int]:
        return list(self._ref_action_map) if self.ref_layout is not None else self.model.action_map

    def setActionSpace(self, mapping: Sequence[int]):
        if self.ref_layout is not None:       # the caller speaks the reference's coordinates: forces on immobile skeletons do nothing
            dev_map = self.ref_layout.device_action_map(mapping)
            old = self._ref_action_map
            self._ref_action_map = [int(d) for d in mapping]
            try:
                self._set_device_action_space(dev_map)
            except Exception:
                self._ref_action_map = old
                raise
            return
        self._set_device_action_space(mapping)

    def _set_device_action_space(self, mapping: Sequence[int]):
        snapshot = [list(md._action_map) if md._action_map is not None else None for md in self._descriptions()]
        try:
            self.model.set_action_space(mapping)
            self.description.set_action_space(mapping)
            self._create_handle()                          # re-upload the constants (the old handle is destroyed, not leaked)
        except Exception:
            # the library refused the new handle (or the mapping is out of bounds): the World keeps its old handle, so it must keep the
            # action map that handle was built with (like setPositionLimitEnforced / setSelfCollisionCheck roll their flags back)
            for md, am in zip(self._descriptions(), snapshot):
                md._action_map = am
            raise
        self._action = None
        if self._wrt_mass.entries:                         # the registered mass parameters survive the re-upload
            self._push_inertia_params()

    def setPenetrationCorrectionEnabled(self, enable: bool):
        """World::setPenetrationCorrectionEnabled (World.cpp:1227; off by default because the reference's analytical Jacobians
        ignore the correction velocity - so does the backward pass here)."""
        if bool(enable) != self.description.penetration_correction:
            self.description.penetration_correction = bool(enable)
            self.model.penetration_correction = bool(enable)
            self._create_handle()
            if self._wrt_mass.entries:
                self._push_inertia_params()

    def getPenetrationCorrectionEnabled(self) -> bool:
        return self.description.penetration_correction

    def _descriptions(self):
        return list({id(self.description): self.description, id(self.model): self.model}.values())

    def _reupload_or_roll_back(self, snapshot):
        """After the body flags of the descriptions changed: upload; if the library refuses the new model, put the flags back (the World
        stays usable on its old handle) and re-raise."""
        try:
            self._create_handle()
        except Exception:
            for md, flags in zip(self._descriptions(), snapshot):
                for b, (le, sc, ab) in zip(md.bodies, flags):
                    b.limit_enforced, b.self_collision, b.adjacent_body_check = le, sc, ab
            raise
        if self._wrt_mass.entries:
            self._push_inertia_params()

    def _flag_snapshot(self):
        return [[(b.limit_enforced, b.self_collision, b.adjacent_body_check) for b in md.bodies] for md in self._descriptions()]

    def setPositionLimitEnforced(self, enforced: bool, joints=None):
        """Joint::setPositionLimitEnforced (Joint.cpp:1366) on the named joints / bodies (default: every joint): their position limits
        become rows of the contact LCP (JointLimitConstraint.cpp).  Off by default, like in the reference (JointAspect.hpp:165)."""
        snapshot, changed = self._flag_snapshot(), False
        for md in self._descriptions():
            for b in md.bodies:
                if joints is None or b.joint_name in joints or b.name in joints:
                    if b.limit_enforced != bool(enforced):
                        b.limit_enforced = bool(enforced)
                        changed = True
        if changed:
            self._reupload_or_roll_back(snapshot)

    def setSelfCollisionCheck(self, enable: bool, adjacent_bodies: bool = False, skeletons=None):
        """Skeleton::setSelfCollisionCheck / setAdjacentBodyCheck (Skeleton.cpp; off by default) on the given skeleton ids (default: all):
        colliders of one skeleton meet, except - unless `adjacent_bodies` - those of a body and its parent."""
        snapshot, changed = self._flag_snapshot(), False
        for md in self._descriptions():
            ids = md.body_skeletons()
            for b, sk in zip(md.bodies, ids):
                if skeletons is None or sk in skeletons:
                    if (b.self_collision, b.adjacent_body_check) != (bool(enable), bool(enable and adjacent_bodies)):
                        b.self_collision, b.adjacent_body_check = bool(enable), bool(enable and adjacent_bodies)
                        changed = True
        if changed:
            self._reupload_or_roll_back(snapshot)

    def getPositionLimitEnforced(self):
        return {b.joint_name or b.name: bool(b.limit_enforced) for b in self.description.bodies}

    def removeDofFromActionSpace(self, index: int):
        self.setActionSpace([a for a in self.getActionSpace() if a != index])

    def getTimeStep(self) -> float:
        return self.model.dt

    def getGravity(self):
        return self.model.gravity

    # ---- workspace ------------------------------------------------------------------------------
    def _workspace(self, B: int) -> torch.Tensor:
        need = self._L.nbl_workspace_bytes(self._h, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- layout helpers: [B, d] (reference: stack of 1-D vectors) <-> [d][B] ----------------------
    def to_soa(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        B, d = x.shape
        out = torch.empty((d, B), dtype=torch.float64, device=self.device)
        check(self._L.nbl_transpose_to_soa(_ptr(x), _ptr(out), B, d, self._stream()), "nbl_transpose_to_soa")
        return out

    def from_soa(self, x: torch.Tensor) -> torch.Tensor:
        d, B = x.shape
        out = torch.empty((B, d), dtype=torch.float64, device=self.device)
        check(self._L.nbl_transpose_from_soa(_ptr(x), _ptr(out), B, d, self._stream()), "nbl_transpose_from_soa")
        return out

    def _prep(self, x: torch.Tensor, width: int, what: str) -> torch.Tensor:
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if x.dim() != 2 or x.shape[1] != width:
            raise ValueError(f"World.{what}() called with a tensor of incorrect size {tuple(x.shape)}; expected [B, {width}]")
        x = x.detach()
        if x.device.type == "cpu" and self.device.type == "cuda":
            # the reference's own convention - CPU float64 tensors in and out (python/nimblephysics/timestep.py:31-40): one pass into a pinned
            # staging buffer the World keeps per role and shape (a fresh pinned allocation costs 50-90 ms, so nothing is allocated per call),
            # then an asynchronous copy on the step's stream ("Host tensors")
            pin, ev = self._staging(what, x.shape)
            ev.synchronize()                           # the previous copy out of this buffer has been consumed
            pin.copy_(x)
            d = pin.to(self.device, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.device))
            return d
        return x.to(device=self.device, dtype=torch.float64)

    def _staging(self, role: str, shape):
        key = (role, tuple(shape))
        got = self._pinned.get(key)
        if got is None:
            if len(self._pinned) >= 16:                # callers that keep changing the batch size: do not let pinned memory pile up
                self._pinned.clear()
            got = (torch.empty(tuple(shape), dtype=torch.float64, pin_memory=True), torch.cuda.Event())
            self._pinned[key] = got
        return got

    def _to_host(self, *tensors):
        """Device tensors -> fresh CPU tensors: asynchronous copies into the World's pinned staging buffers on the current stream, ONE
        synchronisation for all of them, then a host copy each (the caller owns what it gets; the staging buffers are reused)."""
        stage = []
        for i, t in enumerate(tensors):
            pin, _ = self._staging(f"out{i}", t.shape)
            pin.copy_(t, non_blocking=True)
            stage.append(pin)
        torch.cuda.current_stream(self.device).synchronize()
        outs = [torch.empty(p.shape, dtype=p.dtype).copy_(p) for p in stage]      # pageable: the caller's own
        return outs[0] if len(outs) == 1 else tuple(outs)

    # ---- state / action API ---------------------------------------------------------------------
    def setState(self, state: torch.Tensor):
        self._one_d = state.dim() == 1                  # one world given as the reference's 1-D vector (neural.forwardPass answers alike)
        if self.ref_layout is not None:
            full = state.detach()
            self._ref_state_like = (full if full.dim() == 2 else full.unsqueeze(0)).to(device=self.device, dtype=torch.float64).clone()
            state = self.ref_layout.restrict_state(full, "World.setState()")
        self._state = self.to_soa(self._prep(state, 2 * self.n, "setState"))

    def getState(self) -> torch.Tensor:
        out = self.from_soa(self._state)
        if self.ref_layout is not None:
            like = self._ref_state_like
            if like is None or like.shape[0] != out.shape[0]:
                like = torch.zeros((out.shape[0], 2 * self.ref_layout.n_ref), dtype=torch.float64, device=self.device)
            out = self.ref_layout.expand_state(out, like)       # the coordinates of immobile skeletons stay where setState put them
        return out

    def setAction(self, action: torch.Tensor):
        if self.ref_layout is not None:
            k_ref = len(self._ref_action_map)
            if action.shape[-1] != k_ref:
                raise ValueError(f"World.setAction() called with a tensor of incorrect size {tuple(action.shape)}; expected [B, {k_ref}]")
            cols = torch.tensor(self.ref_layout.action_columns(self._ref_action_map), dtype=torch.long, device=action.device)
            action = action.detach().index_select(-1, cols)
        self._action = self.to_soa(self._prep(action, self.k, "setAction"))

    def getAction(self) -> torch.Tensor:
        out = self.from_soa(self._action)
        if self.ref_layout is not None:
            full = torch.zeros((out.shape[0], len(self._ref_action_map)), dtype=torch.float64, device=out.device)
            cols = torch.tensor(self.ref_layout.action_columns(self._ref_action_map), dtype=torch.long, device=out.device)
            out = full.index_copy(1, cols, out)
        return out

    def reset_lcp_cache(self):
        self.lcp_cache = None

    # ---- raw batched step on DOF-major device tensors ---------------------------------------------
    def step_soa(self, state: torch.Tensor, action: torch.Tensor, want_saved: bool = True):
        """state [2n][B], action [k][B] (float64, this device, contiguous) -> (next [2n][B], saved, status[B])."""
        B = state.shape[1]
        nxt = torch.empty_like(state)
        saved = None
        if want_saved:
            saved = torch.empty(self._L.nbl_saved_bytes(self._h, B), dtype=torch.uint8, device=self.device)
        elif self.m > 0:
            # models with colliders use the record as their contact scratch even when no backward pass is wanted (World.step):
            # one reusable buffer per World
            need = self._L.nbl_saved_bytes(self._h, B)
            if getattr(self, "_scratch_saved", None) is None or self._scratch_saved.numel() < need:
                self._scratch_saved = torch.empty(need, dtype=torch.uint8, device=self.device)
            saved = self._scratch_saved
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        ws = self._workspace(B)
        cache_in = self.lcp_cache if (self.lcp_cache is not None and self.lcp_cache.shape == (self.m, B)) else None
        cache_out = torch.empty((self.m, B), dtype=torch.float64, device=self.device) if self.m > 0 else None
        check(self._L.nbl_step_forward(self._h, B, _ptr(state), _ptr(action), _ptr(cache_in), _ptr(nxt), _ptr(cache_out),
                                       _ptr(saved), _ptr(status), _ptr(ws), ws.numel(), self._stream()), "nbl_step_forward")
        if cache_out is not None:
            self.lcp_cache = cache_out
        self.last_status = status
        self._last_saved = saved if want_saved else None
        return nxt, (saved if want_saved else None), status

    def backward_soa(self, saved: torch.Tensor, grad_next: torch.Tensor):
        """grad_next [2n][B] -> (grad_state [2n][B], grad_action [k][B])."""
        B = grad_next.shape[1]
        gs = torch.empty_like(grad_next)
        ga = torch.empty((self.k, B), dtype=torch.float64, device=self.device)
        ws = self._workspace(B)
        check(self._L.nbl_step_backward(self._h, B, _ptr(saved), _ptr(grad_next), _ptr(gs), _ptr(ga), _ptr(ws), ws.numel(),
                                        self._stream()), "nbl_step_backward")
        return gs, ga

    # ---- inertia ("mass") parameters (World.cpp:1014-1053, 1821-1824; WithRespectToMass.cpp) ------------
    def getWrtMass(self) -> WithRespectToMass:
        return self._wrt_mass

    def tuneMass(self, body, type=WrtMassBodyNodeEntryType.INERTIA_MASS, upperBound=None, lowerBound=None):
        """Register a body (name or index in the model description) whose inertial parameters become part of the mass vector."""
        self._wrt_mass.registerNode(body, type, upperBound, lowerBound)
        self._push_inertia_params()

    def getMassDims(self) -> int:
        return self._wrt_mass.dim()

    def getMasses(self) -> torch.Tensor:
        return torch.from_numpy(self._wrt_mass.get())

    def getMassUpperLimits(self) -> torch.Tensor:
        return torch.from_numpy(self._wrt_mass.upperBound())

    def getMassLowerLimits(self) -> torch.Tensor:
        return torch.from_numpy(self._wrt_mass.lowerBound())

    def setMasses(self, masses):
        """World::setMasses: writes the registered parameters into the model (host) and uploads the bodies whose inertial
        constants actually changed, in ONE stream-ordered copy on the current stream (no device synchronisation; an unchanged
        mass vector costs nothing).  Like the reference, a backward pass reads the world's CURRENT masses: run the backward of
        a step before changing the masses for the next one (BackpropSnapshot.cpp:142-146 restores the world's state, not its
        masses)."""
        import numpy as np
        if isinstance(masses, torch.Tensor):
            masses = masses.detach().cpu().numpy()
        self._wrt_mass.set(np.asarray(masses, dtype=np.float64))
        new = self.description.merge_welds() if self.description.has_welds() else self.description
        changed, values = [], {}
        for i, new_b in enumerate(new.bodies):
            cur = (float(new_b.mass), tuple(float(x) for x in new_b.com), tuple(float(x) for x in new_b.inertia))
            if cur != self._uploaded_inertia[i]:
                changed.append(i)
                values[i] = cur
        if changed:
            idx = np.asarray(changed, dtype=np.int32)
            mass = np.asarray([values[i][0] for i in changed], dtype=np.float64)
            com = np.asarray([values[i][1] for i in changed], dtype=np.float64).reshape(-1, 3)
            ine = np.asarray([values[i][2] for i in changed], dtype=np.float64).reshape(-1, 6)
            with torch.cuda.device(self.device):
                check(self._L.nbl_set_body_inertias(self._h, len(changed), idx.ctypes.data_as(C.c_void_p), mass.ctypes.data_as(C.c_void_p),
                                                    com.ctypes.data_as(C.c_void_p), ine.ctypes.data_as(C.c_void_p), self._stream()),
                      "nbl_set_body_inertias")
            for i in changed:                     # only now: a failed upload leaves host and device views in step
                self._uploaded_inertia[i] = values[i]
        if new is not self.model:
            new._action_map = self.model._action_map
            self.model = new
        if changed:
            self._push_inertia_params()

    def _push_inertia_params(self):
        bodies, dG = self._wrt_mass.device_table()
        cnt = int(bodies.shape[0])
        with torch.cuda.device(self.device):     # stream-ordered like the body constants it belongs to (same stream)
            check(self._L.nbl_set_inertia_params_on(self._h, cnt, bodies.ctypes.data_as(C.c_void_p) if cnt else None,
                                                    dG.ctypes.data_as(C.c_void_p) if cnt else None, self._stream()), "nbl_set_inertia_params_on")

    def backward_inertia_soa(self, saved: torch.Tensor, B: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dL/dmass of every world, [massDims][B], for the record whose backward_soa was the LAST call on this world
        (BackpropSnapshot::backprop, lossWrtMass = massVel^T lossWrtVelocity, BackpropSnapshot.cpp:153, 167-179).
        `out`: accumulate into an existing [massDims][B] tensor."""
        d = self.getMassDims()
        if d == 0:
            return torch.zeros((0, B), dtype=torch.float64, device=self.device)
        g = out if out is not None else torch.empty((d, B), dtype=torch.float64, device=self.device)
        ws = self._workspace(B)
        check(self._L.nbl_backward_inertia(self._h, B, _ptr(saved), _ptr(g), 1 if out is not None else 0, _ptr(ws), ws.numel(),
                                           self._stream()), "nbl_backward_inertia")
        return g

    # ---- dense Jacobians of the last step ------------------------------------
    def step_jacobians_soa(self, saved: torch.Tensor, B: int):
        """(d next_state / d state [2n][2n][B], d next_state / d action [2n][k][B]) of the step that produced `saved`:
        row i is the vector-Jacobian product with the unit cotangent e_i (2n backward passes; the path itself never
        forms a Jacobian)."""
        n2 = 2 * self.n
        js = torch.empty((n2, n2, B), dtype=torch.float64, device=self.device)
        ja = torch.empty((n2, self.k, B), dtype=torch.float64, device=self.device)
        g = torch.zeros((n2, B), dtype=torch.float64, device=self.device)
        # The backward pass ends with clipLossGradientsToBounds (BackpropSnapshot.cpp:425-479): with a coordinate exactly ON a limit the
        # entry of the gradient that points out of the box is zeroed - a property of backprop(), not of the Jacobians, which the
        # reference assembles without it (getStateJacobian, World.cpp:2210-2226).  The clipping looks at the sign, so of the two
        # products with +e_i and -e_i exactly one keeps such an entry: models with finite limits pay the second pass.  (A coordinate
        # whose lower and upper limit coincide and that sits on them is clipped in both: its column stays zero.)
        fl = self._limits_finite()
        for i in range(n2):
            g[i] = 1.0
            rs, ra = self.backward_soa(saved, g)
            if fl:
                g[i] = -1.0
                ms, ma = self.backward_soa(saved, g)
                rs = torch.where(rs != 0, rs, -ms); ra = torch.where(ra != 0, ra, -ma)
            js[i], ja[i] = rs, ra
            g[i] = 0.0
        return js, ja

    def _limits_finite(self) -> bool:
        if getattr(self, "_limits_finite_cache", None) is None:
            f = self.description.flat()
            self._limits_finite_cache = bool(any(np.isfinite(np.asarray(f[k], dtype=np.float64)).any()
                                                 for k in ("pos_lo", "pos_hi", "vel_lo", "vel_hi", "force_lo", "force_hi")))
        return self._limits_finite_cache

    def getStateJacobian(self) -> torch.Tensor:
        """World::getStateJacobian (World.cpp:2210-2226) of the last step: [B, 2n, 2n], out[b, i, j] = d next[i] / d state[j]."""
        if getattr(self, "_last_saved", None) is None:
            raise NimbleAmdError("getStateJacobian(): no step with a saved record has been taken")
        B = self.last_status.shape[0]
        self._last_jac = self.step_jacobians_soa(self._last_saved, B)
        J = self._last_jac[0].permute(2, 0, 1).contiguous()
        return self.ref_layout.state_jacobian(J) if self.ref_layout is not None else J

    def getActionJacobian(self) -> torch.Tensor:
        """World::getActionJacobian (World.cpp:2229-2243) of the last step: [B, 2n, k]."""
        if getattr(self, "_last_saved", None) is None:
            raise NimbleAmdError("getActionJacobian(): no step with a saved record has been taken")
        B = self.last_status.shape[0]
        jac = self.step_jacobians_soa(self._last_saved, B)
        J = jac[1].permute(2, 0, 1).contiguous()
        return self._ref_action_jacobian(J) if self.ref_layout is not None else J

    def _ref_action_jacobian(self, J: torch.Tensor) -> torch.Tensor:
        """[B, 2 n_dev, k_dev] -> [B, 2 n_ref, k_ref]: zero rows for the immobile coordinates, zero columns for forces on them"""
        lay = self.ref_layout
        rows = lay._idx(J.device, "state")
        cols = torch.tensor(lay.action_columns(self._ref_action_map), dtype=torch.long, device=J.device)
        out = torch.zeros((J.shape[0], 2 * lay.n_ref, len(self._ref_action_map)), dtype=J.dtype, device=J.device)
        out[:, rows[:, None], cols[None, :]] = J
        return out

    # ---- T-step rollout on the device ---------------------------------------
    def _rollout_workspace(self, B: int) -> torch.Tensor:
        need = self._L.nbl_rollout_workspace_bytes(self._h, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def rollout_soa(self, state0: torch.Tensor, actions: torch.Tensor, T: int = None, want_saved: bool = True,
                    warm_start: bool = True, checkpoint_every: int = 0):
        """state0 [2n][B]; actions [T][k][B], or one [k][B] block applied at every one of `T` steps
        -> (states [T+1][2n][B], saved records, status [T][B]).  checkpoint_every = K > 0: only K records stay resident and the
        backward pass recomputes the others segment by segment (nbl_rollout_forward_checkpointed); `saved` is then the RolloutRecord
        rollout_backward_soa wants."""
        if actions.dim() == 2:
            if T is None:
                raise ValueError("rollout_soa: a single [k][B] action block needs T")
            stride = 0
        else:
            T = actions.shape[0]
            stride = actions.shape[1] * actions.shape[2]
        B = state0.shape[1]
        K = int(checkpoint_every or 0)
        if K < 0:
            raise ValueError("checkpoint_every must be >= 0")
        if K >= T:
            K = 0
        states = torch.empty((T + 1, 2 * self.n, B), dtype=torch.float64, device=self.device)
        saved = None
        if want_saved:
            saved = torch.empty((K if K else T) * self._L.nbl_saved_bytes(self._h, B), dtype=torch.uint8, device=self.device)
        status = torch.empty((T, B), dtype=torch.int32, device=self.device)
        ws = self._rollout_workspace(B)
        if K and want_saved:
            ckpt = torch.empty(self._L.nbl_rollout_checkpoint_bytes(self._h, B, T, K), dtype=torch.uint8, device=self.device)
            check(self._L.nbl_rollout_forward_checkpointed(self._h, B, T, K, _ptr(state0), _ptr(actions), stride, _ptr(states), _ptr(saved),
                                                           _ptr(ckpt), _ptr(status), 1 if warm_start else 0, _ptr(ws), ws.numel(),
                                                           self._stream()), "nbl_rollout_forward_checkpointed")
            saved = RolloutRecord(saved, K, ckpt, states, actions, stride, bool(warm_start))
        else:
            check(self._L.nbl_rollout_forward(self._h, B, T, _ptr(state0), _ptr(actions), stride, _ptr(states), _ptr(saved),
                                              _ptr(status), 1 if warm_start else 0, _ptr(ws), ws.numel(), self._stream()),
                  "nbl_rollout_forward")
        self.last_status = status[-1]
        return states, saved, status

    def rollout_backward_soa(self, saved, grad_states: torch.Tensor, want_mass: bool = False):
        """grad_states [T+1][2n][B] -> (grad_state0 [2n][B], grad_actions [T][k][B]) and, with want_mass, the gradient with
        respect to the registered mass vector summed over the T steps, [massDims][B].  `saved`: what rollout_soa returned."""
        T = grad_states.shape[0] - 1
        B = grad_states.shape[2]
        g0 = torch.empty((2 * self.n, B), dtype=torch.float64, device=self.device)
        ga = torch.empty((T, self.k, B), dtype=torch.float64, device=self.device)
        ws = self._rollout_workspace(B)
        mass = want_mass and self.getMassDims() > 0
        gm = torch.empty((self.getMassDims(), B), dtype=torch.float64, device=self.device) if mass else None
        if isinstance(saved, RolloutRecord):
            r = saved
            check(self._L.nbl_rollout_backward_checkpointed(self._h, B, T, r.segment, _ptr(r.states), _ptr(r.actions), r.stride, _ptr(r.saved),
                                                            _ptr(r.checkpoints), 1 if r.warm_start else 0, _ptr(grad_states), _ptr(g0), _ptr(ga),
                                                            _ptr(gm), _ptr(ws), ws.numel(), self._stream()), "nbl_rollout_backward_checkpointed")
        elif mass:
            check(self._L.nbl_rollout_backward_inertia(self._h, B, T, _ptr(saved), _ptr(grad_states), _ptr(g0), _ptr(ga), _ptr(gm),
                                                       _ptr(ws), ws.numel(), self._stream()), "nbl_rollout_backward_inertia")
        else:
            check(self._L.nbl_rollout_backward(self._h, B, T, _ptr(saved), _ptr(grad_states), _ptr(g0), _ptr(ga), _ptr(ws),
                                               ws.numel(), self._stream()), "nbl_rollout_backward")
        if mass:
            return g0, ga, gm
        if want_mass:
            return g0, ga, torch.zeros((0, B), dtype=torch.float64, device=self.device)
        return g0, ga

    def step(self):
        """World::step on the stored state/action (no gradient bookkeeping kept)."""
        nxt, _, _ = self.step_soa(self._state, self._action, want_saved=False)
        self._state = nxt

    # ---- deferred join: one handle, slices that are not joined per call (nbl_set_deferred_join, include/nimble_amd.h) ----
    def set_deferred_join(self, enabled: bool = True):
        """With it on, step_into / backward_into (and step_soa / backward_soa) return while their slices still run on the handle's internal
        streams; consume a result on its slice's stream (`slices(B)`) or after `join()`.  Buffers handed to a call must outlive it: use
        the *_into entry points with buffers you keep (torch's caching allocator knows nothing of the internal streams)."""
        check(self._L.nbl_set_deferred_join(self._h, 1 if enabled else 0), "nbl_set_deferred_join")
        self._deferred = bool(enabled)

    def slices(self, B: int):
        """[(torch stream, first world, one-past-last world)] of a deferred-join call with B worlds"""
        out = []
        for i in range(self._L.nbl_slices_for(self._h, B)):
            st, b0, b1 = C.c_void_p(), C.c_int64(), C.c_int64()
            check(self._L.nbl_slice_stream(self._h, B, i, C.byref(st), C.byref(b0), C.byref(b1)), "nbl_slice_stream")
            stream = torch.cuda.current_stream(self.device) if not st.value else torch.cuda.ExternalStream(st.value, device=self.device)
            out.append((stream, int(b0.value), int(b1.value)))     # (slice 0 runs on the stream of the calls: the current one)
        return out

    def fork(self):
        """the handle's slice streams wait for everything the current stream holds (inputs produced there)"""
        check(self._L.nbl_fork_slices(self._h, self._stream()), "nbl_fork_slices")

    def join(self):
        """the current stream waits for everything the handle's slices hold"""
        check(self._L.nbl_join_slices(self._h, self._stream()), "nbl_join_slices")

    def step_into(self, state, action, nxt, saved, status, cache_in=None, cache_out=None):
        """nbl_step_forward into caller-owned buffers (state [2n][B], action [k][B], nxt [2n][B], saved nbl_saved_bytes, status int32 [B],
        the warm start in / out [m][B] or None)"""
        B = state.shape[1]
        ws = self._workspace(B)
        check(self._L.nbl_step_forward(self._h, B, _ptr(state), _ptr(action), _ptr(cache_in), _ptr(nxt), _ptr(cache_out),
                                       _ptr(saved), _ptr(status), _ptr(ws), ws.numel(), self._stream()), "nbl_step_forward")

    def backward_into(self, saved, grad_next, grad_state, grad_action):
        B = grad_next.shape[1]
        ws = self._workspace(B)
        check(self._L.nbl_step_backward(self._h, B, _ptr(saved), _ptr(grad_next), _ptr(grad_state), _ptr(grad_action), _ptr(ws), ws.numel(),
                                        self._stream()), "nbl_step_backward")

    def saved_bytes(self, B: int) -> int:
        return int(self._L.nbl_saved_bytes(self._h, B))

    def set_slices(self, slices: int = 0):
        """Batch slices over HIP streams (0 = auto); results are independent of it."""
        check(self._L.nbl_set_slices(self._h, slices), "nbl_set_slices")

    def slices_for(self, B: int) -> int:
        return self._L.nbl_slices_for(self._h, B)

    def set_launch_lanes(self, tree_lanes: int = 0, lcp_lanes: int = 0):
        """Worlds per workgroup (0 = auto); a launch-shape knob, results are independent of it."""
        check(self._L.nbl_set_launch_lanes(self._h, tree_lanes, lcp_lanes), "nbl_set_launch_lanes")

    # ---- kernel timing (HIP events on the launch stream) ------------------------------------------
    def set_timing(self, enabled, period: int = 1):
        """HIP-event timing of the kernels; `period` N > 1 times only every N-th forward / backward call."""
        check(self._L.nbl_set_timing(self._h, (max(1, int(period)) if enabled else 0)), "nbl_set_timing")

    def get_timing(self):
        f, b = C.c_double(0), C.c_double(0)
        fc, bc = C.c_int64(0), C.c_int64(0)
        check(self._L.nbl_get_timing(self._h, C.byref(f), C.byref(fc), C.byref(b), C.byref(bc)), "nbl_get_timing")
        out = {"fwd_ms_sum": f.value, "fwd_count": fc.value, "bwd_ms_sum": b.value, "bwd_count": bc.value, "kernels": {}}
        for i in range(self._L.nbl_kernel_count()):
            ms, cnt = C.c_double(0), C.c_int64(0)
            check(self._L.nbl_kernel_timing(self._h, i, C.byref(ms), C.byref(cnt)), "nbl_kernel_timing")
            if cnt.value:
                out["kernels"][self._L.nbl_kernel_name(i).decode()] = {"ms_sum": ms.value, "count": cnt.value}
        return out
