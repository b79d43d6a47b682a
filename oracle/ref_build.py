#!/usr/bin/env python3
"""Build oracle/_ref/libodelcp_ref.so: the reference's vendored ODE Dantzig LCP solver
(dart/external/odelcpsolver, 9 self-contained .cpp files, no external dependency) compiled with g++
directly from /root/reference — the rest of the reference needs Eigen/libccd/assimp/... and is
unbuildable here.  Output goes to oracle/_ref/ only (git-ignored, shipped by gpurun)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NIMBLE_REFERENCE", "/root/reference")


def main():
    src_dir = os.path.join(REF, "dart", "external", "odelcpsolver")
    if not os.path.isdir(src_dir):
        print("[ref_build] reference absent; keeping any prebuilt oracle/_ref", file=sys.stderr)
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libodelcp_ref.so")
    srcs = sorted(glob.glob(os.path.join(src_dir, "*.cpp"))) + [os.path.join(HERE, "ref_shim.cpp")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return 0
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-w", "-I", REF, "-o", out] + srcs
    print("[ref_build]", " ".join(cmd))
    subprocess.check_call(cmd)
    return 0


if __name__ == "__main__":
    sys.exit(main())
