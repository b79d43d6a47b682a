"""Debug (GPU box): one soak seed through two builds of the library (NBL_LIB_PATH), world by world: status words, next states and the
oracle's.  usage: python tools/dbg/ab_soak_world.py <seed> <mode> <world>"""
import os, subprocess, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
if len(sys.argv) > 4:      # child: run one build, dump
    import torch, soak_parity
    import nimblephysics_amd as na
    seed, mode = int(sys.argv[1]), sys.argv[2]
    md, s, a, g = soak_parity.make_case(seed, 256, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
    w = na.World(md, device="cuda:0")
    nxt, saved, status = w.step_soa(w.to_soa(torch.tensor(s, device="cuda:0")), w.to_soa(torch.tensor(a, device="cuda:0")), want_saved=True)
    B = 256
    ws = w._workspace(B).view(torch.float64).cpu().numpy()
    nbdev = (ws.size // B - 241 - 8 * 40 - 24 * 8 - 2) // 288      # not needed exactly: search below
    torch.cuda.synchronize()
    sv = saved.view(torch.float64).cpu().numpy()
    n = w.n; total = 5 * n + 1 + 8 * 22 + 24 * 4 + 1 + 8; dense = 24 * 24 * 2 + 2 * n * 24
    wdi = int(sys.argv[3])
    rowsv = sv[:total * B].reshape(total, B)
    dn = sv[total * B + wdi * dense: total * B + (wdi + 1) * dense]
    np.save(sys.argv[4] + ".A.npy", {"A": dn[:576].reshape(24, 24).copy(), "b": rowsv[5 * n + 1 + 176 + 24: 5 * n + 1 + 176 + 48, wdi].copy(), "nc": rowsv[5 * n, wdi],
                                      "x": rowsv[5 * n + 1 + 176: 5 * n + 1 + 176 + 24, wdi].copy(), "contacts": rowsv[5 * n + 1: 5 * n + 1 + 176, wdi].copy()}, allow_pickle=True)
    np.save(sys.argv[4], {"next": w.from_soa(nxt).cpu().numpy(), "status": status.cpu().numpy(), "cache": w.lcp_cache.cpu().numpy(), "ws": ws, "nb": w._L.nbl_model_num_dofs(w._h)}, allow_pickle=True)
    sys.exit(0)
seed, mode, wd = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
outs = {}
for tag, libp in (("sym", os.path.join(ROOT, "nimblephysics_amd", "libnimble_amd.so")), ("qr", os.path.join(ROOT, "tools", "dbg", "libnimble_amd_qr.so"))):
    f = f"/tmp/ab_{tag}.npy"
    subprocess.check_call([sys.executable, __file__, str(seed), mode, str(wd), f], env=dict(os.environ, NBL_LIB_PATH=libp))
    outs[tag] = np.load(f, allow_pickle=True).item()
import soak_parity
from oracle import OracleWorld
md, s, a, g = soak_parity.make_case(seed, 256, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
ref = OracleWorld(md).step_batch(s, a, None, threads=8, want_lcp=True)
d = np.abs(outs["sym"]["next"] - outs["qr"]["next"]).max(1)
print("worlds whose next state differs between the builds (> 1e-9):", np.where(d > 1e-9)[0], d[d > 1e-9])
for tag in ("sym", "qr"):
    o = outs[tag]
    print(tag, "status", hex(int(o["status"][wd])), "x", np.array2string(o["cache"][:, wd], precision=6), "err vs oracle", np.abs(o["next"][wd] - ref["next"][wd]).max())
from nimblephysics_amd.model import ModelDescription
mdw = md.merge_welds() if md.has_welds() else md
nb = len(mdw.bodies) + 2 * sum(1 for b in mdw.bodies if b.joint_type == "ball") + 5 * sum(1 for b in mdw.bodies if b.joint_type == "free" and b.parent >= 0)
for tag in ("sym", "qr"):
    ws = outs[tag]["ws"]; B = 256
    lws = ws[nb * 288 * B:]
    rows = lws[: (lws.size // B) * B].reshape(-1, B)
    print(tag, "device bodies", nb, "X0", np.array2string(rows[0:6, wd], precision=6), "stage x1", np.array2string(rows[144:150, wd], precision=5), "x2", np.array2string(rows[168:174, wd], precision=5),
          "x3", np.array2string(rows[192:198, wd], precision=5), "flags", rows[216, wd], rows[224, wd], rows[232, wd])
dA = np.load("/tmp/ab_qr.npy.A.npy", allow_pickle=True).item()
np.set_printoptions(linewidth=200)
m = 3 * int(dA["nc"]); Ad = dA["A"][:m, :m]; bd = dA["b"][:m]
print("device A (m =", m, "):"); print(np.array2string(Ad, precision=5)); print("b", bd)
print("sym?", np.abs(Ad - Ad.T).max(), "eig", np.linalg.eigvalsh(0.5 * (Ad + Ad.T)))
print("contacts (p, n, depth, type, boxA, boxB):"); print(dA["contacts"].reshape(8, 22)[: m // 3, :10])
np.save(os.path.join(ROOT, "gpurun_out", "dbgA.npy"), dA, allow_pickle=True)
print("oracle status", hex(int(ref["status"][wd])), "x", np.array2string(ref["lcp"][wd][:int(ref["lcp_len"][wd])], precision=6))
