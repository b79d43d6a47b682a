"""The reference World's coordinate layout when the device model has fewer coordinates than the reference's.

`World::getState()` of the reference is `[q; v]` over EVERY skeleton of the world (dart/simulation/World.cpp:2016-2047), the immobile ones
(`<mobile>false</mobile>`, Skeleton::setMobile) included: World::step skips their dynamics and their integration, so their coordinates
stay where the caller put them and their rows of the step Jacobians are the identity.  The device model has no such coordinates - the SKEL
loader welds an immobile skeleton at its zero configuration - so a reference script's state vector would not fit.

`RefLayout` is the bookkeeping between the two: `mobile[i]` says whether the reference's i-th coordinate is one of the device's (in
order).  The drop-in surface (`World.getStateSize / setState / getState / setAction / getAction`, `timestep`, `rollout`, the dense Jacobian
getters) speaks the reference's layout; frozen coordinates pass through a step unchanged (next q = q, next v = v: identity rows of the
vector-Jacobian product, zero columns for their control forces) and MUST sit at the configuration the model was loaded in (zero): their
colliders were placed there once.  The raw SoA entry points (`step_soa`, `backward_soa`, ...) keep the device's layout.
"""
from typing import List, Sequence

import torch


class RefLayout:
    def __init__(self, mobile: Sequence[bool]):
        self.mobile = [bool(m) for m in mobile]
        self.n_ref = len(self.mobile)
        self.dev_of = [-1] * self.n_ref          # reference coordinate -> device coordinate (-1: frozen)
        k = 0
        for i, m in enumerate(self.mobile):
            if m:
                self.dev_of[i] = k
                k += 1
        self.n_dev = k
        self.mobile_idx = [i for i, m in enumerate(self.mobile) if m]
        self.frozen_idx = [i for i, m in enumerate(self.mobile) if not m]
        self._cache = {}

    # index tensors per device (state = [q; v])
    def _idx(self, device, what):
        key = (str(device), what)
        t = self._cache.get(key)
        if t is None:
            if what == "state":
                ix = self.mobile_idx + [self.n_ref + i for i in self.mobile_idx]
            elif what == "frozen_q":
                ix = list(self.frozen_idx)
            else:
                ix = list(self.mobile_idx)
            t = torch.tensor(ix, dtype=torch.long, device=device)
            self._cache[key] = t
        return t

    def restrict_state(self, state: torch.Tensor, what: str = "setState", check_frozen: bool = True) -> torch.Tensor:
        """[..., 2 n_ref] in the reference's layout -> [..., 2 n_dev] (a differentiable gather).  Raises on a size mismatch and - the
        model was loaded with the immobile skeletons at their zero configuration - on a frozen POSITION that is not zero."""
        if state.shape[-1] != 2 * self.n_ref:
            raise ValueError(f"{what}: expected {2 * self.n_ref} entries per world (the reference's [q; v] over every skeleton, "
                             f"{self.n_ref - self.n_dev} coordinate(s) of immobile skeletons included); got {state.shape[-1]}")
        if check_frozen and self.frozen_idx:
            fq = state.detach().index_select(-1, self._idx(state.device, "frozen_q"))
            if bool((fq != 0).any()):
                raise ValueError(f"{what}: a coordinate of an immobile skeleton is not zero - the model was loaded with that skeleton welded "
                                 "at its zero configuration (its colliders were placed there)")
        return state.index_select(-1, self._idx(state.device, "state"))

    def expand_state(self, dev_state: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        """[..., 2 n_dev] -> [..., 2 n_ref]: the frozen entries are those of `like` (they pass through a step unchanged; differentiable in both)."""
        full = like.clone()
        ix = self._idx(like.device, "state")
        return full.index_copy(-1, ix, dev_state.to(like.device)) if full.dim() == 1 else full.index_copy(-1, ix, dev_state.to(like.device))

    def expand_grad_state(self, dev_grad: torch.Tensor, grad_next_full: torch.Tensor) -> torch.Tensor:
        """The vector-Jacobian product in the reference's layout: the mobile entries are the device's, the frozen ones the identity's."""
        return self.expand_state(dev_grad, grad_next_full)

    def state_jacobian(self, J_dev: torch.Tensor) -> torch.Tensor:
        """[B, 2 n_dev, 2 n_dev] -> [B, 2 n_ref, 2 n_ref]: identity on the frozen coordinates."""
        B = J_dev.shape[0]
        ix = self._idx(J_dev.device, "state")
        J = torch.eye(2 * self.n_ref, dtype=J_dev.dtype, device=J_dev.device).repeat(B, 1, 1)
        J[:, ix[:, None], ix[None, :]] = J_dev
        return J

    # ---- the action space in the reference's coordinates ----
    def device_action_map(self, ref_map: Sequence[int]) -> List[int]:
        for d in ref_map:
            if not 0 <= int(d) < self.n_ref:
                raise ValueError(f"setActionSpace: coordinate {d} out of range [0, {self.n_ref})")
        return [self.dev_of[int(d)] for d in ref_map if self.dev_of[int(d)] >= 0]

    def action_columns(self, ref_map: Sequence[int]) -> List[int]:
        """columns of an action vector over `ref_map` that drive a device coordinate (the others - forces on immobile skeletons - do nothing)"""
        return [c for c, d in enumerate(ref_map) if self.dev_of[int(d)] >= 0]
