"""ctypes binding of libnimble_amd.so (the C ABI of include/nimble_amd.h).

There is deliberately NO fallback: if the HIP library is missing or no GPU is visible, the
product path raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from here.)
"""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBL_LIB_PATH: load another build of the same library (tools/occupancy_sweep.py A/Bs register caps this way)
LIB_PATH = os.environ.get("NBL_LIB_PATH") or os.path.join(_HERE, "libnimble_amd.so")
_lib = None


class NimbleAmdError(RuntimeError):
    pass


def lib():
    """Load libnimble_amd.so (after torch, so both share one HIP runtime: same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- must be loaded first so that its bundled libamdhip64 is the one in the process
    if not os.path.exists(LIB_PATH):
        raise NimbleAmdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "The batched timestep has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    pd, vp = C.POINTER(C.c_double), C.c_void_p
    L.nbl_last_error.restype = C.c_char_p
    L.nbl_version.restype = C.c_int32
    L.nbl_device_count.restype = C.c_int32
    L.nbl_model_create.argtypes = [C.POINTER(_abi.ModelDesc), C.c_int32, C.POINTER(vp)]
    L.nbl_model_create.restype = C.c_int32
    L.nbl_model_destroy.argtypes = [vp]
    L.nbl_model_destroy.restype = None
    for f in ("nbl_model_num_dofs", "nbl_model_num_action", "nbl_model_lcp_rows", "nbl_model_max_contacts"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_int32
    L.nbl_workspace_bytes.argtypes = [vp, C.c_int64]
    L.nbl_workspace_bytes.restype = C.c_size_t
    L.nbl_saved_bytes.argtypes = [vp, C.c_int64]
    L.nbl_saved_bytes.restype = C.c_size_t
    L.nbl_step_forward.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.nbl_step_forward.restype = C.c_int32
    L.nbl_step_backward.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.nbl_step_backward.restype = C.c_int32
    L.nbl_set_body_inertia.argtypes = [vp, C.c_int32, C.c_double, vp, vp]
    L.nbl_set_body_inertia.restype = C.c_int32
    L.nbl_set_body_inertias.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    L.nbl_set_body_inertias.restype = C.c_int32
    L.nbl_set_inertia_params.argtypes = [vp, C.c_int32, vp, vp]
    L.nbl_set_inertia_params.restype = C.c_int32
    L.nbl_set_inertia_params_on.argtypes = [vp, C.c_int32, vp, vp, vp]
    L.nbl_set_inertia_params_on.restype = C.c_int32
    L.nbl_num_inertia_params.argtypes = [vp]
    L.nbl_num_inertia_params.restype = C.c_int32
    L.nbl_backward_inertia.argtypes = [vp, C.c_int64, vp, vp, C.c_int32, vp, C.c_size_t, vp]
    L.nbl_backward_inertia.restype = C.c_int32
    L.nbl_transpose_to_soa.argtypes = [vp, vp, C.c_int64, C.c_int32, vp]
    L.nbl_transpose_to_soa.restype = C.c_int32
    L.nbl_transpose_from_soa.argtypes = [vp, vp, C.c_int64, C.c_int32, vp]
    L.nbl_transpose_from_soa.restype = C.c_int32
    L.nbl_rollout_workspace_bytes.argtypes = [vp, C.c_int64]
    L.nbl_rollout_workspace_bytes.restype = C.c_size_t
    L.nbl_rollout_forward.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, C.c_int64, vp, vp, vp, C.c_int32, vp, C.c_size_t, vp]
    L.nbl_rollout_forward.restype = C.c_int32
    L.nbl_rollout_backward.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.nbl_rollout_backward.restype = C.c_int32
    L.nbl_rollout_backward_inertia.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.nbl_rollout_backward_inertia.restype = C.c_int32
    L.nbl_rollout_checkpoint_bytes.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32]
    L.nbl_rollout_checkpoint_bytes.restype = C.c_size_t
    L.nbl_rollout_forward_checkpointed.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int32, vp, C.c_size_t, vp]
    L.nbl_rollout_forward_checkpointed.restype = C.c_int32
    L.nbl_rollout_backward_checkpointed.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.nbl_rollout_backward_checkpointed.restype = C.c_int32
    L.nbl_set_slices.argtypes = [vp, C.c_int32]
    L.nbl_set_slices.restype = C.c_int32
    L.nbl_slices_for.argtypes = [vp, C.c_int64]
    L.nbl_slices_for.restype = C.c_int32
    L.nbl_set_deferred_join.argtypes = [vp, C.c_int32]
    L.nbl_set_deferred_join.restype = C.c_int32
    L.nbl_slice_stream.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.nbl_slice_stream.restype = C.c_int32
    L.nbl_join_slices.argtypes = [vp, vp]
    L.nbl_join_slices.restype = C.c_int32
    L.nbl_fork_slices.argtypes = [vp, vp]
    L.nbl_fork_slices.restype = C.c_int32
    L.nbl_set_launch_lanes.argtypes = [vp, C.c_int32, C.c_int32]
    L.nbl_set_launch_lanes.restype = C.c_int32
    L.nbl_set_timing.argtypes = [vp, C.c_int32]
    L.nbl_set_timing.restype = C.c_int32
    L.nbl_get_timing.argtypes = [vp, pd, C.POINTER(C.c_int64), pd, C.POINTER(C.c_int64)]
    L.nbl_get_timing.restype = C.c_int32
    L.nbl_kernel_count.restype = C.c_int32
    L.nbl_kernel_name.argtypes = [C.c_int32]
    L.nbl_kernel_name.restype = C.c_char_p
    L.nbl_kernel_timing.argtypes = [vp, C.c_int32, pd, C.POINTER(C.c_int64)]
    L.nbl_kernel_timing.restype = C.c_int32
    L.nbl_selftest_lcp_dantzig.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.nbl_selftest_lcp_dantzig.restype = C.c_int32
    L.nbl_selftest_lcp_dantzig_timed.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp]
    L.nbl_selftest_lcp_dantzig_timed.restype = C.c_int32
    L.nbl_selftest_lcp_cascade.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, C.c_double, vp, vp, vp, vp]
    L.nbl_selftest_lcp_cascade.restype = C.c_int32
    L.nbl_selftest_pinv.argtypes = [C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_int32, vp]
    L.nbl_selftest_pinv.restype = C.c_int32
    L.nbl_selftest_pinv_rows.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_int32, vp]
    L.nbl_selftest_pinv_rows.restype = C.c_int32
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "nbl_last_error", "nbl_version", "nbl_device_count", "nbl_model_create", "nbl_model_destroy",
    "nbl_model_num_dofs", "nbl_model_num_action", "nbl_model_lcp_rows", "nbl_workspace_bytes", "nbl_saved_bytes",
    "nbl_step_forward", "nbl_step_backward", "nbl_transpose_to_soa", "nbl_transpose_from_soa", "nbl_set_timing", "nbl_set_launch_lanes", "nbl_set_slices", "nbl_slices_for", "nbl_set_deferred_join", "nbl_slice_stream", "nbl_join_slices", "nbl_fork_slices", "nbl_rollout_workspace_bytes", "nbl_rollout_forward", "nbl_rollout_backward",
    "nbl_set_body_inertia", "nbl_set_body_inertias", "nbl_set_inertia_params", "nbl_set_inertia_params_on", "nbl_num_inertia_params", "nbl_backward_inertia", "nbl_rollout_backward_inertia",
    "nbl_rollout_checkpoint_bytes", "nbl_rollout_forward_checkpointed", "nbl_rollout_backward_checkpointed",
    "nbl_get_timing", "nbl_kernel_count", "nbl_kernel_name", "nbl_kernel_timing", "nbl_selftest_lcp_dantzig", "nbl_selftest_lcp_dantzig_timed", "nbl_selftest_lcp_cascade", "nbl_selftest_pinv",
    "nbl_model_max_contacts", "nbl_selftest_pinv_rows",
]


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().nbl_last_error()
        raise NimbleAmdError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
