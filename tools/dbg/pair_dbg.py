"""CPU: one collider pair of one mixed-soak world through the oracle's box-box and the reference's compiled dBoxBox.
usage: python tools/dbg/pair_dbg.py <seed> <world> <variant> <boxA> <boxB> [mode]"""
import ctypes as C, os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
src = open(os.path.join(ROOT, "tools", "soak_parity.py")).read().replace("import torch  # noqa: E402", "").replace(
    "from nimblephysics_amd.timestep import timestep  # noqa: E402", "")
mod = types.ModuleType("soak_cpu"); mod.__file__ = os.path.join(ROOT, "tools", "soak_parity.py"); exec(compile(src, "soak_cpu", "exec"), mod.__dict__)
import soak_stress, oracle
from oracle import OracleWorld
pd = C.POINTER(C.c_double)
_p = lambda a: a.ctypes.data_as(pd)
seed, wd, variant, iA, iB = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
mode = sys.argv[6] if len(sys.argv) > 6 else "mix"
md, s, a, g = mod.make_case(seed, 256, variant == "big", variant == "multi", variant == "balls", False)
md, s, a, g = soak_stress.mutator(mode)(seed, md, s, a, g)
ow = OracleWorld(md); n = md.num_dofs
np.set_printoptions(linewidth=200, precision=17)
def wT(bx):
    Tb = ow.body_world_transform(s[wd][:n], bx.body) if bx.body >= 0 else np.eye(4)
    return Tb @ np.asarray(bx.T, dtype=np.float64)
bA, bB = md.boxes[iA], md.boxes[iB]
TA, TB = wT(bA), wT(bB)
print("A: body", bA.body, "size", bA.size, "\n", TA, "\nB: body", bB.body, "size", bB.size, "\n", TB)
print("bodies: parent of", bA.body, "=", md.bodies[bA.body].parent, "; parent of", bB.body, "=", md.bodies[bB.body].parent)
flat = lambda T: np.concatenate([T[:3, :3].reshape(9), T[:3, 3]])
OL = oracle._lib()
for clip in (0.03, 1e9):
    o = np.zeros(16 * 22)
    no = OL.nbo_box_box(_p(flat(TA)), _p(np.array(bA.size, dtype=np.float64)), _p(flat(TB)), _p(np.array(bB.size, dtype=np.float64)), C.c_double(clip), _p(o))
    print(f"oracle box-box (clip {clip:g}):", no)
    for c in o[:no * 22].reshape(no, 22):
        print("   p", c[:3], "n", c[3:6], "depth", c[6], "type", c[7])
path = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libdboxbox_ref.so")
if os.path.exists(path):
    ref = C.CDLL(path); ref.ref_collide_box_box.restype = C.c_int
    import inspect
    r = np.zeros(16 * 22)
    try:
        nr = ref.ref_collide_box_box(_p(flat(TA)), _p(np.array(bA.size, dtype=np.float64)), _p(flat(TB)), _p(np.array(bB.size, dtype=np.float64)), C.c_double(0.03), _p(r))
        print("reference dBoxBox:", nr)
        for c in r[:nr * 22].reshape(nr, 22):
            print("   p", c[:3], "n", c[3:6], "depth", c[6], "type", c[7])
    except Exception as e:
        print("ref call failed", e)
ow.step(s[wd], a[wd]); print("oracle world contacts:", len(ow.last_contacts()))
print(ow.last_contacts())
fl = md.flat()
print("self flags", fl.get("body_self_collision"))
for wtest in (22, 152):
    ow.reset_lcp_cache(); ow.step(s[wtest], a[wtest]); print("world", wtest, "contacts", len(ow.last_contacts()), "status", hex(ow.last_status))
    q = s[wtest][:n]
    print("   q limits check: ", [(d, float(q[d]), float(fl['pos_lo'][d]), float(fl['pos_hi'][d])) for d in range(n) if fl['dof_limit_enforced'][d] and (q[d] <= fl['pos_lo'][d] or q[d] >= fl['pos_hi'][d])])
