#!/usr/bin/env python3
"""Phase stamps of k_qp2 (wave 0's view: barrier to barrier) for a prebuilt -DMPCMP_STAMPS library; fixed 700 iterations, no tests.
usage: stamps_lib.py <lib> [B]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_motion_planner_amd.capi as capi
capi._SO = os.path.abspath(sys.argv[1])
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = M.default_config(4, 1, margins=(0.9, 0.9, 0.5, 0.9))
cfg.eps_abs = 0.0; cfg.eps_rel = 0.0; cfg.check_every = 10000
s = M.Solver(cfg, B)
x0, xf = scenarios.make_batch(B)
wx, wu, wT = s.warm_start(x0, xf)
for rep in range(2):
    s.kernel_timing(reset=True)
    p, y, it = s.qp(x0, xf, wx, wu, wT)
    name, ms, launches = s.kernel_timing()
st = np.zeros((B, 160), dtype=np.uint64)
capi.check(capi.lib().mpcmp_debug_stamps(s._ctx, B, st.ctypes.data_as(C.c_void_p)))
st = st.astype(np.float64)
its = st[:, 15]
names = {3: "A(rhs)", 4: "P1", 5: "P2", 6: "P3", 7: "E", 8: "loop end"}
print(sys.argv[1], "%.4f ms/launch" % (ms / launches), "iterations", its.mean())
tot = 0
for k, nm in names.items():
    v = (st[:, k] / its).mean(); tot += v
    print("  %-10s %8.1f cycles / iteration" % (nm, v))
print("  %-10s %8.1f" % ("total", tot))
