import sys, os, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_py as o
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
MARG = (0.9, 0.9, 0.5, 0.9, 0.1)
nseg, sqp = 8, 1
cfg = M.default_config(nseg, sqp, margins=MARG); ocfg = o.default_config(nseg, sqp, margins=MARG)
B = 1; N = 3 * nseg + 1
x0, xf = scenarios.make_batch(B, stream_offset=100)
wx = np.zeros((B, N, 14)); wu = np.zeros((B, N, 7)); wT = np.zeros(B)
for b in range(B): wx[b], wu[b], wT[b] = o.warm_start(ocfg, x0[b], xf[b])
s = M.Solver(cfg, B)
print("qp only...", flush=True)
p, y, it = s.qp(x0, xf, wx, wu, wT)
print("qp ok", it, flush=True)
print("full solve...", flush=True)
sx, su, sT, info = s.solve(x0, xf, (wx, wu, wT))
print("solve ok", sT, info["viol_l1"], flush=True)
