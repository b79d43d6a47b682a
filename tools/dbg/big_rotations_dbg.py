import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from test_ball_joint import ball_model
for scale in (1.0, 2.0, 3.0, 6.0):
    md = ball_model(9, True); B = 128; rng = np.random.default_rng(int(scale * 10)); n = md.num_dofs
    s = np.concatenate([rng.normal(0, scale, (B, n)), rng.normal(0, 1.0, (B, n))], 1); a = rng.normal(0, 1, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at); out.backward(torch.tensor(g, device="cuda:0"))
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)
    sc = lambda x: max(np.abs(x).max(), 1e-30)
    e = [np.abs(out.detach().cpu().numpy() - ref["next"]).max() / sc(ref["next"]), np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max() / sc(ref["grad_state"]),
         np.abs(at.grad.cpu().numpy() - ref["grad_action"]).max() / sc(ref["grad_action"])]
    rot = np.linalg.norm(s[:, :3], axis=1)
    print(f"q scale {scale}: |q_root| up to {rot.max():.2f} rad; errors next {e[0]:.1e} grad_state {e[1]:.1e} grad_action {e[2]:.1e}")
    # where are the worst worlds?  distance of the next rotation angle of every exponential-map joint from pi (logMap's singular point)
    from scipy.spatial.transform import Rotation as Rot
    offs = [0] + [o for o in __import__("test_ball_joint")._ball_offsets(md)]
    perw = np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max(1) / sc(ref["grad_state"])
    def gap(w):
        out = []
        for o in offs:
            Rn = Rot.from_rotvec(s[w, o:o + 3]).as_matrix() @ Rot.from_rotvec(s[w, n + o:n + o + 3] * md.dt).as_matrix()
            out.append(abs(np.linalg.norm(Rot.from_matrix(Rn).as_rotvec()) - np.pi))
        return min(out)
    order = np.argsort(-perw)
    print("   worst worlds: err / distance of a joint's next angle from pi:", [(f"{perw[w]:.1e}", f"{gap(w):.1e}") for w in order[:4]],
          " median world:", f"{np.median(perw):.1e}", f"{np.median([gap(w) for w in range(B)]):.1e}")
