"""Developer build: recompile only the named instantiations of the library (default: c8) - and, with --timing / --gen-timing, the single-instantiation
libraries with the cycle stamps (tools/cascade_timing.py / tools/gen_timing.py) - in parallel, then relink.  The other instantiations keep their old
objects (same ABI).  The full build is __graft_entry__.build()."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c8"]
objdir = os.path.join(ROOT, "nimblephysics_amd", "_build")
procs = []
for name, flags in ge.VARIANTS:
    if name in names:
        obj = os.path.join(objdir, f"nimble_amd_{name}.o")
        procs.append(subprocess.Popen([ge.HIPCC] + ge.HIPFLAGS + list(flags) + ["-c", os.path.join(ge.CSRC, "nimble_amd.hip"), "-o", obj]))
if "--timing" in sys.argv:
    os.makedirs(os.path.join(ROOT, "tools", "dbg"), exist_ok=True)
    procs.append(subprocess.Popen([ge.HIPCC] + ge.HIPFLAGS + ["-shared", "-DNBL_CASCADE_TIMING", os.path.join(ge.CSRC, "nimble_amd.hip"),
                                   "-o", os.path.join(ROOT, "tools", "dbg", "libnimble_amd_timing.so")]))
if "--gen-timing" in sys.argv:       # the general build's solve kernel with its cycle stamps (tools/gen_timing.py)
    os.makedirs(os.path.join(ROOT, "tools", "dbg"), exist_ok=True)
    procs.append(subprocess.Popen([ge.HIPCC] + ge.HIPFLAGS + ["-shared", "-DNBL_MAXC=64", "-DNBL_GEN_TIMING", os.path.join(ge.CSRC, "nimble_amd.hip"),
                                   "-o", os.path.join(ROOT, "tools", "dbg", "libnimble_amd_gentiming.so")]))
rc = [p.wait() for p in procs]
assert not any(rc), rc
objs = [os.path.join(objdir, f"nimble_amd_{n}.o") for n, _ in ge.VARIANTS] + [os.path.join(objdir, "nimble_amd_dispatch.o")]
subprocess.check_call([ge.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", ge.LIB])
print("relinked", ge.LIB)
