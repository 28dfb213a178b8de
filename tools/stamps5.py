#!/usr/bin/env python3
"""Diagnostic: build the library with -DMPCMP_STAMPS and print where k_qp5's cycles go per ADMM iteration (phase totals seen by wave 0,
and the busy part of each phase per wave).  Fixed iteration count (eps = 0), one QP per problem.  Not part of the product path.
usage: stamps5.py [B] [extra hipcc flags ...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
so = os.path.join(ROOT, "tools", "micro", "libmpcmp_stamps.bin")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.environ.get("STAMPS_PREBUILT"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-DMPCMP_STAMPS",
                           *sys.argv[2:], "-o", so, os.path.join(ROOT, "mpc_motion_planner_amd", "csrc", "mpcmp.hip")])
import mpc_motion_planner_amd.capi as capi
capi._SO = so
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = M.default_config(6, 1, margins=(0.9, 0.9, 0.5, 0.9))
cfg.eps_abs = 0.0; cfg.eps_rel = 0.0
if os.environ.get("STAMPS_NOTEST"):
    cfg.check_every = 10000
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
its = np.maximum(st[:, 15], 1)
print("%s B=%d  %.4f ms/launch, mean ADMM iterations %.1f" % (name, B, ms / launches, its.mean()))
names = ["A  rhs = sigma x - q + rho z - y + A^T w  (variable lanes)", "P1 t = G b_J, K_CJ t  (role G)", "P2 y_I = S^-1 r_I  (role S)", "P3 x_J = t - E y_C  (role E)",
         "E  z~ = A x~, projection, duals  (rows, variables)", "termination tests (total / iterations)"]
for k, nm in enumerate(names):
    print("%-62s %9.1f cycles / iteration" % (nm, (st[:, k] / its).mean()))
print("%-62s %9.1f cycles / iteration" % ("loop total", (st[:, 0:6].sum(axis=1) / its).mean()))
busy = st[:, 16:112].reshape(B, 12, 8)
print("per-wave busy cycles / iteration in phases A, P1, P2, P3, E (waves 0-5: role G, 6-8: E (+path), 9-11: S + path)")
for w in range(12):
    print("  wave %2d  " % w + "  ".join("%8.1f" % (busy[:, w, ph] / its).mean() for ph in range(5)))
