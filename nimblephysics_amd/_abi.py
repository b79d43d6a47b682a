"""ctypes mirror of include/nimble_amd.h (struct nbl_model_desc and constants).

Pure declarations; nothing here touches a GPU or loads a library.
"""
import ctypes as C

JOINT_REVOLUTE = 0
JOINT_PRISMATIC = 1
JOINT_FREE = 2
JOINT_WELD = 3
JOINT_BALL = 4
JOINT_SCREW = 5
JOINT_NAMES = {"revolute": JOINT_REVOLUTE, "prismatic": JOINT_PRISMATIC, "free": JOINT_FREE, "weld": JOINT_WELD, "ball": JOINT_BALL,
               "screw": JOINT_SCREW}
JOINT_NDOF = {JOINT_REVOLUTE: 1, JOINT_PRISMATIC: 1, JOINT_FREE: 6, JOINT_WELD: 0, JOINT_BALL: 3, JOINT_SCREW: 1}

NBL_OK = 0
NBL_E_BADARG = -1
NBL_E_UNSUPPORTED = -2
NBL_E_HIP = -3
NBL_E_WORKSPACE = -4
NBL_E_NOGPU = -5

ST_CONTACT = 0x1
ST_LCP_STAGE0 = 0x2
ST_LCP_PIVOT = 0x4
ST_LCP_PGS = 0x8
ST_LCP_NOFRIC = 0x10
ST_LCP_FAILED = 0x20
ST_NAN = 0x40
ST_CONTACT_OVERFLOW = 0x80
ST_STANDARDIZED = 0x100

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_bodies", C.c_int32),
        ("n_dofs", C.c_int32),
        ("parent", _pi),
        ("joint_type", _pi),
        ("dof_offset", _pi),
        ("T_pj", _pd),
        ("T_cj", _pd),
        ("axis", _pd),
        ("mass", _pd),
        ("com", _pd),
        ("inertia", _pd),
        ("damping", _pd),
        ("spring", _pd),
        ("rest", _pd),
        ("pos_lo", _pd),
        ("pos_hi", _pd),
        ("vel_lo", _pd),
        ("vel_hi", _pd),
        ("force_lo", _pd),
        ("force_hi", _pd),
        ("gravity", C.c_double * 3),
        ("dt", C.c_double),
        ("n_action", C.c_int32),
        ("action_map", _pi),
        ("n_boxes", C.c_int32),
        ("box_body", _pi),
        ("box_T", _pd),
        ("box_size", _pd),
        ("box_mu", _pd),
        ("max_contacts", C.c_int32),
        ("contact_clipping_depth", C.c_double),
        ("fallback_cfm", C.c_double),
        ("box_shape", _pi),
        ("box_restitution", _pd),
        ("penetration_correction", C.c_int32),
        ("body_skeleton", _pi),
        ("pitch", _pd),
        ("dof_limit_enforced", _pi),
        ("body_self_collision", _pi),
        ("box_node", _pi),
        ("box_node_parent", _pi),
    ]
