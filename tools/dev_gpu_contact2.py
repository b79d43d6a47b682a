import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import nimblephysics_amd as na
from oracle import OracleWorld
np.set_printoptions(linewidth=220, precision=6, suppress=True)
md = na.atlas("atlas20", ground=True)
w = na.World(md); ow = OracleWorld(md); n = w.n
B = 64
q = np.zeros((B, n)); q[:, 0] = -np.pi/2; q[:, 4] = -0.01
v = np.zeros((B, n)); a = np.zeros((B, n))
s = np.concatenate([q, v], 1)
st = w.to_soa(torch.tensor(s, device="cuda")); at = w.to_soa(torch.tensor(a, device="cuda"))
nxt, saved, status = w.step_soa(st, at)
torch.cuda.synchronize()
sv = saved.view(torch.float64).view(-1, B).cpu().numpy()
MR, MC, CR = 24, 8, 22
L = {}
L["vpre"] = 3*n; L["w"] = 4*n; L["nc"] = 5*n; L["contacts"] = L["nc"]+1; L["x"] = L["contacts"] + MC*CR
L["b"] = L["x"]+MR; L["cls"] = L["b"]+MR; L["cfm"] = L["cls"]+MR; L["A"] = L["cfm"]+1; L["massed"] = L["A"]+MR*MR; L["aall"] = L["massed"]+n*MR
print("rows", sv.shape, "expected", L["aall"]+n*MR)
lane = 0
nC = int(sv[L["nc"], lane]); print("nC", nC, "status", hex(int(status[lane])))
ow.step(s[lane], a[lane]); print("oracle status", hex(ow.last_status))
oc = ow.last_contacts(); ol = ow.last_lcp()
gc = sv[L["contacts"]:L["contacts"]+MC*CR, lane].reshape(MC, CR)
print("gpu contacts\n", gc[:nC, :10]); print("oracle contacts\n", oc[:, :8])
m = 3*nC
gb = sv[L["b"]:L["b"]+m, lane]; print("b gpu", gb); print("b ora", ol["b"])
gA = sv[L["A"]:L["A"]+MR*MR, lane].reshape(MR, MR)[:m,:m]; print("A diff", np.abs(gA-ol["A"]).max(), "A max", np.abs(ol["A"]).max())
print("x gpu", sv[L["x"]:L["x"]+m, lane]); print("x ora", ol["x"])
print("cls gpu", sv[L["cls"]:L["cls"]+m, lane]); print("cls ora", ol["row_class"])
print("vpre diff", np.abs(sv[L["vpre"]:L["vpre"]+n, lane] - 0).max())
print("next v gpu", w.from_soa(nxt).cpu().numpy()[lane, n:]); print("next ora", ow.step(s[lane], a[lane])[n:])
ws = w._ws.view(torch.float64)
nbod = len(w.model.bodies)
wsb = ws[: nbod*132*B].view(nbod*132, B).cpu().numpy()
lw = ws[nbod*132*B : nbod*132*B + (2*24*6 + 2*576)*B].view(-1, B).cpu().numpy()
fa = w.model.boxes[0].body
print("foot body", fa, "V(WS_A)", wsb[fa*132+78: fa*132+84, lane])
print("root V", wsb[0*132+78: 0*132+84, lane])
print("TW foot", wsb[fa*132+120: fa*132+132, lane]); print("oracle TW", ow.body_world_transform(q[lane], fa))
print("JA rows0-2", lw[0:18, lane].reshape(3,6))
print("massed col0 gpu", sv[L["massed"]:L["massed"]+n*MR, lane].reshape(n, MR)[:, 0])
print("aall col0 gpu", sv[L["aall"]:L["aall"]+n*MR, lane].reshape(n, MR)[:, 0])
