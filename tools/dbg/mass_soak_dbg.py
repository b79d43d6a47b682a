"""Debug (GPU box): dL/dmass of random models (ball / free / revolute / prismatic joints, welds) in free fall against central differences of
the oracle step with respect to the same parameters (the helper of tests/test_gpu_mass.py), random bodies and entry types."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import soak_parity
from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
from test_gpu_mass import _check
ok = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, (int(sys.argv[1]) if len(sys.argv) > 1 else 0) + (int(sys.argv[2]) if len(sys.argv) > 2 else 12)):
    case = soak_parity.make_case(seed, 16, balls=True)
    if case is None:
        continue
    md, s, a, g = case
    md.boxes = []; md.max_contacts = 0                      # free fall: the difference quotient of the oracle is clean
    rng = np.random.default_rng(seed)
    movable = [i for i, b in enumerate(md.bodies) if b.joint_type != "weld"]
    entries = [(int(rng.choice(movable)), T(int(rng.choice([0, 1, 3, 4, 5])))) for _ in range(3)]
    entries = list({e[0]: e for e in entries}.values())     # one entry per body
    try:
        _check(md, entries, s, a, seed + 1, tol=2e-5)
        ok += 1
    except AssertionError as e:
        print("seed", seed, "bodies", [(i, md.bodies[i].joint_type, t.name) for i, t in entries], "FAILED", str(e)[:300])
print("models passed:", ok)
