"""The C ABI used from a plain C program (tests/c_abi_example/main.c: gcc, no Python, no torch, no C++): what a cgo / JNI / N-API
binding of the reference would link against.  Its output is compared with the CPU oracle on the same inputs."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_against_the_oracle(tmp_path):
    import nimblephysics_amd as na
    from oracle import OracleWorld
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no gcc / ROCm runtime")
    libdir = os.path.join(ROOT, "nimblephysics_amd")
    exe = str(tmp_path / "c_abi_example")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_example", "main.c"), "-o", exe, "-L" + libdir, "-lnimble_amd",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + libdir])
    B = 8
    out = np.array([[float(x) for x in line.split()] for line in subprocess.check_output([exe, str(B)]).decode().strip().splitlines()])
    assert out.shape == (B, 5)
    b = np.arange(B)
    s = np.stack([0.3 + 0.1 * b, -0.5 + 0.2 * b], 1)
    a = (0.7 - 0.05 * b)[:, None]
    g = np.tile([1.0, -2.0], (B, 1))
    ref = OracleWorld(na.single_pendulum()).step_batch(s, a, g, threads=1)
    assert np.abs(out[:, 0:2] - ref["next"]).max() < 1e-13
    assert np.abs(out[:, 2:4] - ref["grad_state"]).max() < 1e-10
    assert np.abs(out[:, 4:5] - ref["grad_action"]).max() < 1e-13
