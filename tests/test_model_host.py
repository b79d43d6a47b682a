"""Host logic: model descriptions, weld merging, action space, C-ABI exports (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

import nimblephysics_amd as na
from nimblephysics_amd import _abi, _lib
from oracle import OracleWorld
from util import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_sizes():
    assert na.single_pendulum().num_dofs == 1
    assert na.cartpole().num_dofs == 2          # "4 DOF" in BASELINE.json is the state size
    assert na.atlas("atlas33").num_dofs == 33   # free root + 27 revolutes, 6 welds
    assert na.atlas("atlas20").num_dofs == 20
    assert na.box_stack().merge_welds().num_dofs == 12
    m = na.atlas("atlas33", ground=True).merge_welds()
    assert len(m.bodies) == 28 and [b.body for b in m.boxes][-1] == -1  # ground box is world-fixed


def test_weld_merge_is_equivalent_rigid_body():
    """Merged model == the same model with explicit 0-DOF weld joints (oracle handles both)."""
    rng = np.random.default_rng(3)
    for name in ("atlas33", "atlas20"):
        md = na.atlas(name)
        a, b = OracleWorld(md), OracleWorld(md.merge_welds())
        n = md.num_dofs
        q = rng.uniform(-0.4, 0.4, n); q[0] -= 1.5
        v = rng.normal(0, 0.5, n); tau = rng.normal(0, 1, n)
        s = np.concatenate([q, v])
        assert rel_err(a.step(s, tau), b.step(s, tau)) < 1e-12
        assert rel_err(a.mass_matrix(q), b.mass_matrix(q)) < 1e-12
        g = rng.normal(0, 1, 2 * n)
        (ga, gta), (gb, gtb) = a.backprop(g), b.backprop(g)
        assert rel_err(ga, gb) < 1e-9 and rel_err(gta, gtb) < 1e-9


def test_action_space_errors():
    md = na.cartpole()
    md.set_action_space([0])
    assert md.action_map == [0]
    with pytest.raises(ValueError):
        md.set_action_space([2])     # reference prints + ignores (World.cpp:2118-2135); we raise
    with pytest.raises(ValueError):
        md.set_action_space([-1])


def test_desc_struct_roundtrip():
    md = na.atlas("atlas20").merge_welds()
    d, keep = md.to_desc()
    assert d.n_dofs == 20 and d.n_bodies == len(md.bodies)
    assert d.dt == 1e-3 and list(d.gravity) == [0.0, -9.81, 0.0]
    offs = [d.dof_offset[i] for i in range(d.n_bodies)]
    assert offs[0] == 0 and offs[1] == 6


def test_cabi_exports_every_declared_symbol():
    """The shared library loads on a CPU-only box and exports every function include/nimble_amd.h declares."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libnimble_amd.so not built (no hipcc in this environment)")
    hdr = open(os.path.join(ROOT, "include", "nimble_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(nbl_[a-z0-9_]+)\s*\(", hdr)))
    L = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in the header but not exported"
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS)
    L.nbl_version.restype = ctypes.c_int32
    assert L.nbl_version() == int(re.search(r"#define NBL_ABI_MINOR (\d+)", hdr).group(1)) == 5   # NBL_ABI_MINOR of include/nimble_amd.h


def test_header_is_plain_c_and_the_python_mirror_matches_its_layout(tmp_path):
    """The drop-in boundary is a C ABI: include/nimble_amd.h must compile as C99 (a cgo / JNI / ctypes binding sees exactly
    this), and the ctypes mirror of nbl_model_desc must have the size and field offsets the C compiler gives the struct."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "nimble_amd.h")
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-pedantic", "-Werror", hdr])
    from nimblephysics_amd import _abi
    fields = [f[0] for f in _abi.ModelDesc._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nimble_amd.h"\nint main(void) {\n  printf("%zu\\n", sizeof(nbl_model_desc));\n'
                   + "".join(f'  printf("%zu\\n", offsetof(nbl_model_desc, {f}));\n' for f in fields) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_abi.ModelDesc)
    for name, off in zip(fields, out[1:]):
        assert getattr(_abi.ModelDesc, name).offset == off, name


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.NimbleAmdError):
        na.World(na.cartpole())


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nimblephysics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src and not re.search(r'#include\s*"[^"]*oracle', src), f
