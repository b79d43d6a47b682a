#!/usr/bin/env python3
"""Build oracle/_ref/libodelcp_ref.so, libdboxbox_ref.so, libgeometry_ref.so and liblcputils_ref.so (build_lcputils: the reference's LCPUtils).
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
        # ... and the closed-form capsule narrow phases: collideCapsuleCapsule .. end of collideCapsuleSphere
        c0 = next(i for i, l in enumerate(lines) if l.startswith("int collideCapsuleCapsule("))
        c1 = next(i for i, l in enumerate(lines) if l.startswith("int collideCapsuleSphere("))
        c2 = next(i for i in range(c1, len(lines)) if lines[i] == "}")
        f.write(f'#line {c0 + 1} "{src}"\n')
        f.write("\n".join(lines[c0:c2 + 1]) + "\n")
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


GEOMETRY_FUNCTIONS = [   # signature prefix at column 0 of dart/math/Geometry.cpp; each body runs to the next "}" at column 0
    "Eigen::Matrix3s makeSkewSymmetric(", "Eigen::Matrix3s expMapRot(", "Eigen::Matrix3s expMapJac(", "Eigen::Vector3s logMap(const Eigen::Matrix3s& _R)",
    "Eigen::Vector6s AdT(", "Eigen::Vector6s AdR(", "Eigen::Vector6s AdTAngular(", "Eigen::Vector6s AdTLinear(", "Eigen::Vector6s AdInvT(",
    "Eigen::Vector6s AdInvRLinear(", "Eigen::Vector6s ad(", "Eigen::Vector6s dAdT(", "Eigen::Vector6s dAdInvT(", "Eigen::Vector6s dAdInvR(",
    "Eigen::Matrix3s eulerXYZToMatrix(", "Eigen::Matrix3s eulerZYXToMatrix(", "Eigen::Isometry3s expMap(", "Eigen::Isometry3s expAngular(",
    "Eigen::Isometry3s expMapDart(", "Eigen::Vector6s dad(", "Inertia transformInertia(",
    "void dLineClosestApproach(", "Eigen::Vector3s getContactPoint(", "Eigen::Vector3s getContactPointGradient(",
    "Eigen::Vector3s closestPointOnLineGradient(",
]


def _function(lines, prefix):
    first = next(i for i, l in enumerate(lines) if l.startswith(prefix))
    last = next(i for i in range(first, len(lines)) if lines[i] == "}")
    return first, last


def build_geometry():
    """libgeometry_ref.so: the reference's own spatial-algebra primitives (the functions of dart/math/Geometry.cpp listed above, SURVEY.md
    row a19) and its flat-array articulated-body algorithm (dart/dynamics/SimpleFeatherstone.hpp: the three structs / the class;
    SimpleFeatherstone.cpp: emplaceBack, len, forwardDynamics) compiled from the reference's files where they lie, between
    ref_geometry_prelude.hpp (a small fixed-size matrix class under Eigen's names - Eigen itself is not on this machine) and
    ref_geometry_epilogue.hpp (C entry points).  The generated translation unit is deleted after the build."""
    geo = os.path.join(REF, "dart", "math", "Geometry.cpp")
    sfh = os.path.join(REF, "dart", "dynamics", "SimpleFeatherstone.hpp")
    sfc = os.path.join(REF, "dart", "dynamics", "SimpleFeatherstone.cpp")
    if not (os.path.exists(geo) and os.path.exists(sfh) and os.path.exists(sfc)):
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libgeometry_ref.so")
    parts = [os.path.join(HERE, "ref_geometry_prelude.hpp"), os.path.join(HERE, "ref_geometry_epilogue.hpp")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in parts + [geo, sfh, sfc, __file__, os.path.join(HERE, "ref_geometry_shells.hpp")]):
        return 0
    tu = os.path.join(out_dir, "geometry_tu.cpp")
    with open(tu, "w") as f:
        f.write(open(parts[0]).read())
        lines = open(geo).read().split("\n")
        f.write("namespace dart {\nnamespace math {\n")
        eps = next(i for i, l in enumerate(lines) if l.startswith("#define EPSILON_EXPMAP_THETA"))     # the Taylor-branch threshold, as the file has it
        f.write(f'#line {eps + 1} "{geo}"\n' + lines[eps] + "\n")
        for prefix in GEOMETRY_FUNCTIONS:
            a, b = _function(lines, prefix)
            f.write(f'#line {a + 1} "{geo}"\n' + "\n".join(lines[a:b + 1]) + "\n")
        f.write("}  // namespace math\n")
        # SimpleFeatherstone.hpp: `struct JointAndBody` .. the end of `class SimpleFeatherstone` (populateFromSkeleton is only declared)
        lines = open(sfh).read().split("\n")
        a = next(i for i, l in enumerate(lines) if l.startswith("struct JointAndBody"))
        c = next(i for i, l in enumerate(lines) if l.startswith("class SimpleFeatherstone"))
        b = next(i for i in range(c, len(lines)) if lines[i] == "};")
        f.write("namespace dynamics {\nclass Skeleton;\n")
        f.write(f'#line {a + 1} "{sfh}"\n' + "\n".join(lines[a:b + 1]) + "\n")
        # SimpleFeatherstone.cpp: emplaceBack .. the end of forwardDynamics
        lines = open(sfc).read().split("\n")
        a = next(i for i, l in enumerate(lines) if l.startswith("JointAndBody& SimpleFeatherstone::emplaceBack()"))
        c = next(i for i, l in enumerate(lines) if l.startswith("void SimpleFeatherstone::forwardDynamics("))
        b = next(i for i in range(c, len(lines)) if lines[i] == "}")
        f.write(f'#line {a + 1} "{sfc}"\n' + "\n".join(lines[a:b + 1]) + "\n")
        f.write("}  // namespace dynamics\n}  // namespace dart\n")
        # member functions behind class shells: the free joint's SE(3) position integration, the contact tangent basis
        f.write('#line 1 "ref_geometry_shells.hpp"\n' + open(os.path.join(HERE, "ref_geometry_shells.hpp")).read())
        fj = os.path.join(REF, "dart", "dynamics", "FreeJoint.cpp")
        lines = open(fj).read().split("\n")
        f.write("namespace dart {\nnamespace dynamics {\n")
        for prefix in ("Eigen::Vector6s FreeJoint::convertToPositions(", "Eigen::Isometry3s FreeJoint::convertToTransform(",
                       "Eigen::VectorXs FreeJoint::integratePositionsExplicit("):
            a, b = _function(lines, prefix)
            f.write(f'#line {a + 1} "{fj}"\n' + "\n".join(lines[a:b + 1]) + "\n")
        f.write("}  // namespace dynamics\nnamespace constraint {\n")
        cc = os.path.join(REF, "dart", "constraint", "ContactConstraint.cpp")
        lines = open(cc).read().split("\n")
        eps = next(i for i, l in enumerate(lines) if l.startswith("#define DART_CONTACT_CONSTRAINT_EPSILON_SQUARED"))
        f.write(f'#line {eps + 1} "{cc}"\n' + lines[eps] + "\n")
        for sig in ("ContactConstraint::getTangentBasisMatrixODE(", "ContactConstraint::getTangentBasisMatrixODEGradient("):
            a = next(i for i, l in enumerate(lines) if l.startswith(sig)) - 1   # the return type is on the line above
            b = next(i for i in range(a, len(lines)) if lines[i] == "}")
            f.write(f'#line {a + 1} "{cc}"\n' + "\n".join(lines[a:b + 1]) + "\n")
        f.write("}  // namespace constraint\n}  // namespace dart\n")
        f.write('#line 1 "ref_geometry_epilogue.hpp"\n')
        f.write(open(parts[1]).read())
    # -ffp-contract=off: the reference's build has no fused multiply-adds
    cmd = ["g++", "-O2", "-DNDEBUG", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-o", out, tu]
    print("[ref_build]", " ".join(cmd))
    try:
        subprocess.check_call(cmd)
    finally:
        os.remove(tu)      # the generated translation unit holds reference source: only the shared object stays
    return 0


def build_lcputils():
    """liblcputils_ref.so: the reference's own LCPUtils::isLCPSolutionValid, reduce, removeFriction with their helpers mergeLCPColumns and
    dropLCPColumn (dart/constraint/LCPUtils.cpp:12-80, 144-247, 346-549: dynamic Eigen containers, no decomposition) compiled from the
    reference's file where it lies, between ref_lcputils_prelude.hpp (a small dynamic matrix / vector class under Eigen's names + the class
    shell) and ref_lcputils_epilogue.hpp (C entry points).  The generated translation unit is deleted after the build."""
    src = os.path.join(REF, "dart", "constraint", "LCPUtils.cpp")
    if not os.path.exists(src):
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liblcputils_ref.so")
    parts = [os.path.join(HERE, "ref_lcputils_prelude.hpp"), os.path.join(HERE, "ref_lcputils_epilogue.hpp")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in parts + [src, __file__]):
        return 0
    lines = open(src).read().split("\n")
    tu = os.path.join(out_dir, "lcputils_tu.cpp")
    with open(tu, "w") as f:
        f.write(open(parts[0]).read())
        thr = next(i for i, l in enumerate(lines) if l.startswith("#define MERGE_THRESHOLD"))      # the merge threshold, as the file has it
        f.write(f'#line {thr + 1} "{src}"\n' + lines[thr] + "\n")
        f.write("namespace dart {\nnamespace constraint {\n")
        for prefix in ("bool LCPUtils::isLCPSolutionValid(", "Eigen::MatrixXs LCPUtils::reduce(", "Eigen::MatrixXs LCPUtils::removeFriction(",
                       "void LCPUtils::mergeLCPColumns(", "void LCPUtils::dropLCPColumn("):
            a, b = _function(lines, prefix)
            f.write(f'#line {a + 1} "{src}"\n' + "\n".join(lines[a:b + 1]) + "\n")
        f.write("}  // namespace constraint\n}  // namespace dart\n")
        f.write('#line 1 "ref_lcputils_epilogue.hpp"\n')
        f.write(open(parts[1]).read())
    # -ffp-contract=off: the reference's build has no fused multiply-adds; NDEBUG like a release build (the functions assert their own preconditions)
    cmd = ["g++", "-O2", "-DNDEBUG", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-o", out, tu]
    print("[ref_build]", " ".join(cmd))
    try:
        subprocess.check_call(cmd)
    finally:
        os.remove(tu)      # the generated translation unit holds reference source: only the shared object stays
    return 0


if __name__ == "__main__":
    rc = main()
    sys.exit(rc or build_boxbox() or build_geometry() or build_lcputils())
