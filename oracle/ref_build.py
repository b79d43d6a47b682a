#!/usr/bin/env python3
"""Build oracle/_ref/libodelcp_ref.so and oracle/_ref/libdboxbox_ref.so.
libdboxbox_ref.so: the reference's own box-box narrow phase - the ODE-derived dBoxBox with its helpers and the collideBoxBox wrapper,
dart/collision/dart/DARTCollide.cpp from `typedef s_t dVector3[4];` to the end of collideBoxBox - compiled from the reference's file where it
lies: that line range is read at build time into oracle/_ref/ (git-ignored) between ref_boxbox_prelude.hpp (stand-ins for the few Eigen /
collision types the range uses) and ref_boxbox_epilogue.hpp (a C entry point).  Nothing of the reference is stored in this repo.
libodelcp_ref.so: the reference's vendored ODE Dantzig LCP solver
(dart/external/odelcpsolver, 9 self-contained .cpp files, no external dependency) compiled with g++
directly from /root/reference, plus the reference's projected Gauss-Seidel solver: PgsBoxedLcpSolver::solve read from
dart/constraint/PgsBoxedLcpSolver.cpp at build time between ref_pgs_prelude.hpp (the class shell) and ref_pgs_epilogue.hpp (a C entry point) — the rest of the reference needs Eigen/libccd/assimp/... and is
unbuildable here.  Output goes to oracle/_ref/ only (git-ignored, shipped by gpurun)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NIMBLE_REFERENCE", "/root/reference")


def _pgs_translation_unit(out_dir):
    """PgsBoxedLcpSolver::solve read from the reference's file between ref_pgs_prelude.hpp and ref_pgs_epilogue.hpp (deleted after the build)."""
    src = os.path.join(REF, "dart", "constraint", "PgsBoxedLcpSolver.cpp")
    lines = open(src).read().split("\n")
    first = next(i for i, l in enumerate(lines) if l.startswith("bool PgsBoxedLcpSolver::solve("))
    last = next(i for i in range(first, len(lines)) if lines[i] == "}")
    tu = os.path.join(out_dir, "pgs_tu.cpp")
    with open(tu, "w") as f:
        f.write(open(os.path.join(HERE, "ref_pgs_prelude.hpp")).read())
        f.write(f'#line {first + 1} "{src}"\n')
        f.write("\n".join(lines[first:last + 1]) + "\n")
        f.write('#line 1 "ref_pgs_epilogue.hpp"\n')
        f.write(open(os.path.join(HERE, "ref_pgs_epilogue.hpp")).read())
    return tu, src


def main():
    src_dir = os.path.join(REF, "dart", "external", "odelcpsolver")
    if not os.path.isdir(src_dir):
        print("[ref_build] reference absent; keeping any prebuilt oracle/_ref", file=sys.stderr)
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libodelcp_ref.so")
    srcs = sorted(glob.glob(os.path.join(src_dir, "*.cpp"))) + [os.path.join(HERE, "ref_shim.cpp")]
    deps = srcs + [os.path.join(HERE, "ref_pgs_prelude.hpp"), os.path.join(HERE, "ref_pgs_epilogue.hpp"),
                   os.path.join(REF, "dart", "constraint", "PgsBoxedLcpSolver.cpp"), __file__]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in deps):
        return 0
    tu, _ = _pgs_translation_unit(out_dir)
    # -ffp-contract=off: the reference's build has no fused multiply-adds
    cmd = ["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-I", REF, "-o", out] + srcs + [tu]
    print("[ref_build]", " ".join(cmd))
    try:
        subprocess.check_call(cmd)
    finally:
        os.remove(tu)      # the generated translation unit holds reference source: only the shared object stays
    return 0


def build_boxbox():
    src = os.path.join(REF, "dart", "collision", "dart", "DARTCollide.cpp")
    if not os.path.exists(src):
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libdboxbox_ref.so")
    parts = [os.path.join(HERE, "ref_boxbox_prelude.hpp"), os.path.join(HERE, "ref_boxbox_epilogue.hpp")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in parts + [src, __file__]):
        return 0
    lines = open(src).read().split("\n")
    first = next(i for i, l in enumerate(lines) if l.startswith("typedef s_t dVector3[4];"))
    start = next(i for i, l in enumerate(lines) if l.startswith("int collideBoxBox("))
    last = next(i for i in range(start, len(lines)) if lines[i] == "}")        # closing brace of collideBoxBox
    tu = os.path.join(out_dir, "dboxbox_tu.cpp")
    with open(tu, "w") as f:
        f.write(open(parts[0]).read())
        f.write(f'#line {first + 1} "{src}"\n')
        f.write("\n".join(lines[first:last + 1]) + "\n")
        # ... and the analytic sphere narrow phases: collideBoxSphere .. end of collideSphereSphere
        s0 = next(i for i, l in enumerate(lines) if l.startswith("int collideBoxSphere("))
        s1 = next(i for i, l in enumerate(lines) if l.startswith("int collideSphereSphere("))
        s2 = next(i for i in range(s1, len(lines)) if lines[i] == "}")
        f.write(f'#line {s0 + 1} "{src}"\n')
        f.write("\n".join(lines[s0:s2 + 1]) + "\n")
        f.write('#line 1 "ref_boxbox_epilogue.hpp"\n')
        f.write(open(parts[1]).read())
    # -ffp-contract=off: the reference's build has no fused multiply-adds
    cmd = ["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-o", out, tu]
    print("[ref_build]", " ".join(cmd))
    try:
        subprocess.check_call(cmd)
    finally:
        os.remove(tu)      # the generated translation unit holds reference source: only the shared object stays
    return 0


if __name__ == "__main__":
    rc = main()
    sys.exit(rc or build_boxbox())
