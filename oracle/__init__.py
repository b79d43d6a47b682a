"""oracle — CPU restatement of the reference timestep.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The shipped product (nimblephysics_amd) never does; it fails loudly when its HIP library is absent.

PARITY PINNING (see oracle/oracle.cpp header and DESIGN.md): pinned by the reference's property
tests restated in tests/test_oracle_props.py, by the literal LCP fixtures of
unittests/unit/test_LCPUtils.cpp, and by the reference's own code where it compiles without Eigen, built
from the reference sources where they lie into oracle/_ref (oracle/ref_build.py): the vendored ODE Dantzig
solver, PgsBoxedLcpSolver::solve, dBoxBox and the box-sphere / sphere-sphere narrow phases.  There are no
stored step/gradient outputs in the reference to compare to.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False, native: bool = False, out: str = None) -> str:
    """Compile liboracle.so with g++ (seconds).  `native` adds -march=native for the timed CPU baseline."""
    out = out or os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.cpp", "dynamics.hpp", "spatial.hpp", "contact.hpp", "collision.hpp", "lcp.hpp")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    flags = ["-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]
    if native:
        flags.append("-march=native")
    subprocess.check_call(["g++", *flags, "-Wl,-Bsymbolic", "-o", out, os.path.join(_HERE, "oracle.cpp"), "-lpthread", "-ldl"])
    return out


def _lib(path: str = None):
    global _LIB
    if path is not None:
        return _declare(C.CDLL(path))
    if _LIB is None:
        p = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(p):
            build()
        _LIB = _declare(C.CDLL(p))
    return _LIB


def _declare(lib):
    pd = C.POINTER(C.c_double)
    lib.nbo_create.restype = C.c_void_p
    lib.nbo_create.argtypes = [C.c_void_p]
    lib.nbo_destroy.argtypes = [C.c_void_p]
    lib.nbo_num_dofs.argtypes = [C.c_void_p]
    lib.nbo_step.argtypes = [C.c_void_p, pd, pd, pd, C.POINTER(C.c_uint32)]
    lib.nbo_backprop.argtypes = [C.c_void_p, pd, pd, pd]
    lib.nbo_step_batch.argtypes = [C.c_void_p, C.c_int64, pd, pd, pd, pd, pd, pd, C.POINTER(C.c_uint32), C.c_int,
                                   pd, C.POINTER(C.c_int32), pd, C.POINTER(C.c_int32), C.c_int]
    lib.nbo_state_jacobian.argtypes = [C.c_void_p, pd]
    lib.nbo_action_jacobian.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int, pd]
    lib.nbo_set_lcp_cache.argtypes = [C.c_void_p, pd, C.c_int]
    lib.nbo_get_lcp_cache.argtypes = [C.c_void_p, pd, C.c_int]
    lib.nbo_set_lcp_noise.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int]
    lib.nbo_set_pinv_noise.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    lib.nbo_set_lcp_noise.restype = None
    lib.nbo_set_lcp_forced.argtypes = [C.c_void_p, pd, C.c_int, C.c_int]
    lib.nbo_set_lcp_forced.restype = None
    lib.nbo_set_lcp_cache_slots.argtypes = [C.c_void_p, C.c_int]
    lib.nbo_set_lcp_cache_slots.restype = None
    lib.nbo_set_exact_position_jacobians.argtypes = [C.c_void_p, C.c_int]
    lib.nbo_set_exact_position_jacobians.restype = None
    lib.nbo_get_lcp_cache.restype = C.c_int
    lib.nbo_mass_matrix.argtypes = [C.c_void_p, pd, pd]
    lib.nbo_coriolis_gravity.argtypes = [C.c_void_p, pd, pd, pd]
    lib.nbo_forward_dynamics.argtypes = [C.c_void_p, pd, pd, pd, pd]
    lib.nbo_jac_C.argtypes = [C.c_void_p, pd, pd, C.c_int, pd]
    lib.nbo_jac_Mx.argtypes = [C.c_void_p, pd, pd, pd]
    lib.nbo_impulse_response.argtypes = [C.c_void_p, pd, C.c_int, pd, pd]
    lib.nbo_body_world_transform.argtypes = [C.c_void_p, pd, C.c_int, pd]
    lib.nbo_integrate_positions.argtypes = [C.c_void_p, pd, pd, pd]
    lib.nbo_last_contacts.argtypes = [C.c_void_p, pd, C.c_int]
    lib.nbo_last_contacts.restype = C.c_int
    lib.nbo_last_lcp.argtypes = [C.c_void_p, pd, pd, pd, pd, pd, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
    lib.nbo_last_lcp.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class OracleWorld:
    """One scalar CPU world following the reference's World/BackpropSnapshot API names."""

    def __init__(self, model, lib_path: str = None):
        """model: nimblephysics_amd.model.ModelDescription (welds allowed: the oracle handles 0-DOF joints)."""
        self._lib = _lib(lib_path)
        self.model = model
        desc, self._keep = model.to_desc()
        self._desc = desc
        self._h = self._lib.nbo_create(C.addressof(desc))
        self.n = self._lib.nbo_num_dofs(self._h)
        self.k = len(model.action_map)

    def __del__(self):
        try:
            if self._h:
                self._lib.nbo_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # World::getStateSize / getActionSize
    def getStateSize(self):
        return 2 * self.n

    def getActionSize(self):
        return self.k

    def reset_lcp_cache(self):
        self._lib.nbo_set_lcp_cache(self._h, None, 0)

    def set_lcp_cache(self, x):
        x = _arr(x)
        self._lib.nbo_set_lcp_cache(self._h, _p(x), len(x))

    def get_lcp_cache(self):
        buf = np.zeros(3 * max(self.model.max_contacts, 1) + 8)
        n = self._lib.nbo_get_lcp_cache(self._h, _p(buf), len(buf))
        return buf[:n].copy()

    def step(self, state, action):
        state, action = _arr(state), _arr(action)
        assert state.shape == (2 * self.n,) and action.shape == (self.k,)
        out = np.zeros(2 * self.n)
        st = C.c_uint32(0)
        self._lib.nbo_step(self._h, _p(state), _p(action), _p(out), C.byref(st))
        self.last_status = st.value
        return out

    def backprop(self, grad_next):
        g = _arr(grad_next)
        gs, ga = np.zeros(2 * self.n), np.zeros(self.k)
        rc = self._lib.nbo_backprop(self._h, _p(g), _p(gs), _p(ga))
        assert rc == 0
        return gs, ga

    def getStateJacobian(self):
        """World::getStateJacobian of the last step() (World.cpp:2210-2226): [2n, 2n], out[i, j] = d next[i] / d state[j]."""
        out = np.zeros((2 * self.n, 2 * self.n))
        assert self._lib.nbo_state_jacobian(self._h, _p(out)) == 0
        return out

    def getActionJacobian(self):
        """World::getActionJacobian of the last step() (World.cpp:2229-2243): [2n, k]."""
        amap = np.ascontiguousarray(self.model.action_map, dtype=np.int32)
        out = np.zeros((2 * self.n, self.k))
        assert self._lib.nbo_action_jacobian(self._h, amap.ctypes.data_as(C.POINTER(C.c_int32)), self.k, _p(out)) == 0
        return out

    def step_batch(self, state, action, grad_next=None, threads=1, lcp_in=None, lcp_len_in=None, want_lcp=False):
        """state [B,2n], action [B,k] world-major.  Returns dict(next, grad_state, grad_action, status[, lcp, lcp_len])."""
        state, action = _arr(state), _arr(action)
        B = state.shape[0]
        nxt = np.zeros_like(state)
        status = np.zeros(B, np.uint32)
        gs = ga = None
        gn_p = gs_p = ga_p = None
        if grad_next is not None:
            gn = _arr(grad_next)
            gs, ga = np.zeros_like(state), np.zeros_like(action)
            gn_p, gs_p, ga_p = _p(gn), _p(gs), _p(ga)
        stride = 3 * max(self.model.max_contacts, 1)
        li_p = ll_p = lo_p = lol_p = None
        if lcp_in is not None:
            lcp_in = _arr(lcp_in)
            assert lcp_in.shape == (B, stride), (lcp_in.shape, (B, stride))   # rows of 3 * max_contacts doubles
            lcp_len_in = np.ascontiguousarray(lcp_len_in, dtype=np.int32)
            li_p, ll_p = _p(lcp_in), lcp_len_in.ctypes.data_as(C.POINTER(C.c_int32))
        lcp_out = lcp_len_out = None
        if want_lcp:
            lcp_out = np.zeros((B, stride))
            lcp_len_out = np.zeros(B, np.int32)
            lo_p, lol_p = _p(lcp_out), lcp_len_out.ctypes.data_as(C.POINTER(C.c_int32))
        self._lib.nbo_step_batch(self._h, B, _p(state), _p(action), gn_p, _p(nxt), gs_p, ga_p,
                                 status.ctypes.data_as(C.POINTER(C.c_uint32)), int(threads), li_p, ll_p, lo_p, lol_p, stride)
        return {"next": nxt, "grad_state": gs, "grad_action": ga, "status": status, "lcp": lcp_out, "lcp_len": lcp_len_out}

    # ---- introspection used by the property tests ----
    def mass_matrix(self, q):
        q = _arr(q); out = np.zeros((self.n, self.n))
        self._lib.nbo_mass_matrix(self._h, _p(q), _p(out)); return out

    def coriolis_gravity(self, q, v):
        q, v = _arr(q), _arr(v); out = np.zeros(self.n)
        self._lib.nbo_coriolis_gravity(self._h, _p(q), _p(v), _p(out)); return out

    def forward_dynamics(self, q, v, tau):
        q, v, tau = _arr(q), _arr(v), _arr(tau); out = np.zeros(self.n)
        self._lib.nbo_forward_dynamics(self._h, _p(q), _p(v), _p(tau), _p(out)); return out

    def jac_C(self, q, v, wrt_vel):
        q, v = _arr(q), _arr(v); out = np.zeros((self.n, self.n))
        self._lib.nbo_jac_C(self._h, _p(q), _p(v), int(wrt_vel), _p(out)); return out

    def jac_Mx(self, q, x):
        q, x = _arr(q), _arr(x); out = np.zeros((self.n, self.n))
        self._lib.nbo_jac_Mx(self._h, _p(q), _p(x), _p(out)); return out

    def impulse_response(self, q, body, imp):
        q, imp = _arr(q), _arr(imp); out = np.zeros(self.n)
        self._lib.nbo_impulse_response(self._h, _p(q), int(body), _p(imp), _p(out)); return out

    def body_world_transform(self, q, body):
        q = _arr(q); out = np.zeros(12)
        self._lib.nbo_body_world_transform(self._h, _p(q), int(body), _p(out))
        T = np.eye(4); T[:3, :3] = out[:9].reshape(3, 3); T[:3, 3] = out[9:]; return T

    def integrate_positions(self, q, v):
        q, v = _arr(q), _arr(v); out = np.zeros(self.n)
        self._lib.nbo_integrate_positions(self._h, _p(q), _p(v), _p(out)); return out

    def set_lcp_noise(self, ulps, seed=0, absolute=False):
        """Test instrument, not the reference's behaviour: every entry of the LCP matrix A times 1 + j ulps 2^-52, j in {-1, 0, 1} per
        entry and per solve - the A another order of the same sums could have given; absolute: j ulps 2^-52 max |A| ADDED to every
        non-zero entry instead; absolute="bound": j ulps 2^-52 x (the sum of the magnitudes of the terms of A = J M^-1 J^T / b = -J v
        the entry is the sum of): the first-order rounding-error bound of that entry.  0 switches it off."""
        self._lib.nbo_set_lcp_noise(self._h, int(ulps), int(seed), {False: 0, True: 1, "bound": 3}[absolute])

    def set_pinv_noise(self, ulps, seed=0):
        """Test instrument, not the reference's behaviour: every entry of the pseudo-inverse Q^+ of the BACKWARD pass times 1 + j ulps 2^-52, j in
        {-1, 0, 1} per entry and world - the Q^+ another algorithm of the same accuracy returns.  On a full-rank, ill-conditioned Q the reference's
        imprecise-inverse branch (BackpropSnapshot.cpp:2964-2984) adds Q^+T Q^+ x (the round-off of its own pseudo-inverse): a gradient that moves
        under this instrument is that round-off times cond(Q)^2.  0 switches it off."""
        self._lib.nbo_set_pinv_noise(self._h, int(ulps), int(seed))

    def set_lcp_alternate_a(self, on=True):
        """Test instrument: the LCP matrix recomputed as J M^-1 J^T from the dense inverse mass matrix - the same matrix through another
        valid order of floating-point operations than the reference's impulse tests.  A world whose answer changes under it has no
        answer that is independent of the evaluation order."""
        self._lib.nbo_set_lcp_noise(self._h, 0, 0, 2 if on else 0)

    def set_lcp_cache_slots(self, on=True):
        """The LCP cache (set_lcp_cache / get_lcp_cache / step_batch's lcp_in, lcp out) in the DEVICE's format: three entries per
        constraint - a frictionless contact and a joint-limit row use the first - instead of the reference's 3 / 1 / 1 rows."""
        self._lib.nbo_set_lcp_cache_slots(self._h, 1 if on else 0)

    def set_exact_position_jacobians(self, on=True, doubles=False):
        """Test instrument, not the reference's behaviour: the position-integration Jacobians (posPos / velPos) of free and ball joints by
        exact forward-mode differentiation of expMapRot / logMap (their branches included) instead of the reference's central differences
        (FreeJoint.cpp:950-1007, BallJoint.cpp:351-408), which lose digits like 1 / gap^2 towards a rotation angle of pi.  With it the
        oracle is the reference algorithm with ONE known numerical weakness removed: what tests/parity.py::gradient_tolerance rests on.
        Evaluated in extended precision; doubles=True evaluates the same formulas in doubles, which is good to ~eps / gap^3 only (two
        O(1 / gap) terms cancel) - the accuracy any double-precision analytic derivative of this map, the device's included, can have."""
        self._lib.nbo_set_exact_position_jacobians(self._h, (2 if doubles else 1) if on else 0)

    def set_lcp_forced(self, x=None, cfm_stage=False):
        """Test instrument, not the reference's behaviour: x (one entry per LCP row) stands in for the OUTPUT of the solver stages 1 - 3
        ("had Dantzig ended on this solution") where stage 0 fails; the step reports 0x40000000 when isLCPSolutionValid rejects it.
        Registration, row classes, standardisation and the backward pass stay the reference's.  cfm_stage: x is the output of stage 2 (the
        fallback CFM on the diagonal + PGS) instead of stage 1.  None switches it off."""
        if x is None:
            self._lib.nbo_set_lcp_forced(self._h, None, 0, 0)
        else:
            x = _arr(x); self._lib.nbo_set_lcp_forced(self._h, _p(x), len(x), 1 if cfm_stage else 0)

    def last_contacts(self):
        buf = np.zeros((256, 12))
        c = self._lib.nbo_last_contacts(self._h, _p(buf), 256)
        return buf[:c].copy()

    def last_lcp(self):
        cap = 3 * 128
        A = np.zeros(cap * cap); b = np.zeros(cap); x = np.zeros(cap); lo = np.zeros(cap); hi = np.zeros(cap)
        fi = np.zeros(cap, np.int32); rc = np.zeros(cap, np.int32)
        m = self._lib.nbo_last_lcp(self._h, _p(A), _p(b), _p(x), _p(lo), _p(hi), fi.ctypes.data_as(C.POINTER(C.c_int32)),
                                   rc.ctypes.data_as(C.POINTER(C.c_int32)), cap)
        return {"A": A[:m * m].reshape(m, m).copy(), "b": b[:m].copy(), "x": x[:m].copy(), "lo": lo[:m].copy(),
                "hi": hi[:m].copy(), "findex": fi[:m].copy(), "row_class": rc[:m].copy()}
