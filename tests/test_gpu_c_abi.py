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


def test_plain_c_ball_joint_rollout_plain_and_checkpointed(tmp_path):
    """tests/c_abi_example/ball_rollout.c: NBL_JOINT_BALL through the C ABI alone, a T-step rollout with all records resident and with
    `segment` of them (nbl_rollout_*_checkpointed): identical bit for bit, and equal to T chained oracle steps and their backprop."""
    import nimblephysics_amd as na
    from oracle import OracleWorld
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no gcc / ROCm runtime")
    libdir = os.path.join(ROOT, "nimblephysics_amd")
    exe = str(tmp_path / "ball_rollout")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_example", "ball_rollout.c"), "-o", exe, "-L" + libdir, "-lnimble_amd",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + libdir])
    B, T, K = 8, 7, 3
    out = np.array([[float(x) for x in line.split()] for line in subprocess.check_output([exe, str(B), str(T), str(K)]).decode().strip().splitlines()])
    assert out.shape == (B, 24)
    assert np.array_equal(out[:, :12], out[:, 12:])                              # checkpointed == plain, bit for bit
    md = na.ModelDescription("ball_pendulum", [na.BodySpec("bob", -1, "ball", "ball", T_pj=na.make_transform((0, 0.5, 0)), T_cj=na.make_transform((0.1, 0.3, -0.05)),
                                                           mass=2.0, com=(0.02, -0.01, 0.03), inertia=(0.4, 0.5, 0.6, 0.01, -0.02, 0.03), damping=(0.3, 0.2, 0.1))],
                             [], gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=0)
    r, b = np.arange(6)[None, :], np.arange(B)[:, None]
    s0 = np.where(r < 3, 0.4, 1.5) * (((r * 7 + b * 3) % 11) / 5.0 - 1.0)
    a = 0.5 - 0.1 * ((np.arange(3)[None, :] + b) % 7)
    for w in range(B):
        worlds = [OracleWorld(md) for _ in range(T)]
        x = s0[w]
        for t in range(T):
            x = worlds[t].step(x, a[w])
        g = np.ones(6)
        for t in reversed(range(T)):
            g, _ = worlds[t].backprop(g)
        assert np.abs(out[w, :6] - x).max() < 1e-12 and np.abs(out[w, 6:12] - g).max() < 1e-8 * max(1.0, np.abs(g).max())
