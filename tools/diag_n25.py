import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_py as o
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
MARG = (0.9, 0.9, 0.5, 0.9, 0.1)
import os
nseg, sqp = int(os.environ.get("NSEG", "8")), 1
cfg = M.default_config(nseg, sqp, margins=MARG); ocfg = o.default_config(nseg, sqp, margins=MARG)
B = 1; N = 3 * nseg + 1
x0, xf = scenarios.make_batch(B, stream_offset=100)
wx = np.zeros((B, N, 14)); wu = np.zeros((B, N, 7)); wT = np.zeros(B)
for b in range(B): wx[b], wu[b], wT[b] = o.warm_start(ocfg, x0[b], xf[b])
s = M.Solver(cfg, B)
sx, su, sT, info = s.solve(x0, xf, (wx, wu, wT))
ceq = s.debug_fetch(2, 14 * (N - 1)); g = s.debug_fetch(3, 8 * N)
d = o.collocation_defects(nseg, sx[0], su[0], sT[0])[:, :3].reshape(-1)
np.set_printoptions(precision=4, suppress=True, linewidth=220)
print("max |ceq_gpu - defects|", np.abs(ceq - d).max(), "sum gpu", np.abs(ceq).sum(), "sum ref", np.abs(d).sum(), "info", info["viol_l1"][0])
bad = np.where(np.abs(ceq - d) > 1e-8)[0]; print("bad rows", bad)
import ctypes as C
out = np.zeros(160, dtype=np.uint64)
M.lib().mpcmp_debug_stamps(s._ctx, 1, out.ctypes.data_as(C.c_void_p))
ws = out[143:160].view(np.float64)
print("total", ws[0], "wave partials", ws[1:11], "sum", ws[1:11].sum())
a = np.abs(ceq)
print("expected per wave (ceq part)", [a[64 * w:64 * w + 64].sum() for w in range(6)])
