"""Developer check: sha256 of everything one forward + backward step writes (next state, gradients, status, the whole saved record) on the
metric distribution - run it with two builds of the library (NBL_LIB=path) to see whether a change is bit-neutral."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd._lib as _lib
if os.environ.get("NBL_LIB"):
    _lib.LIB_PATH = os.environ["NBL_LIB"]
import nimblephysics_amd as na
from util import contact_inputs
B = 4096
for jn in (0.02, 0.1):
    md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=jn, vel_noise=jn / 2, action_noise=0.1)
    w = na.World(md, device="cuda:0")
    x = w.to_soa(torch.tensor(s, device="cuda:0")); u = w.to_soa(torch.tensor(a, device="cuda:0"))
    g = torch.randn(x.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3)).to("cuda:0")
    nxt, saved, status = w.step_soa(x, u)
    gs, ga = w.backward_soa(saved, g)[:2]
    torch.cuda.synchronize()
    h = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    print(jn, "next", h(nxt), "status", h(status), "grad_state", h(gs), "grad_action", h(ga), "record", h(saved.view(torch.uint8)) if hasattr(saved, "view") else "-")
