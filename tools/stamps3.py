#!/usr/bin/env python3
"""Diagnostic: build the library with -DMPCMP_STAMPS and print where k_qp3's cycles go per ADMM iteration
(phase totals seen by S wave 8, and the busy part of each phase per wave).  Not part of the product path.
usage: stamps3.py [B] [nseg] [narm]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
so = os.path.join(ROOT, "gpurun_out", "libmpcmp_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-DMPCMP_STAMPS", *os.environ.get("MPCMP_EXTRA_DEFS", "").split(),
                       "-o", so, os.path.join(ROOT, "mpc_motion_planner_amd", "csrc", "mpcmp.hip")])
import mpc_motion_planner_amd.capi as capi
capi._SO = so
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 6
narm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
margins = (0.9, 0.9, 0.5, 0.9, 0.05)
cfg = M.default_config(nseg, 1, margins=margins)
if narm == 1:
    s = M.Solver(cfg, B)
    x0, xf = scenarios.make_batch(B, margins)
else:
    import bench_dual14
    s = M.Solver(cfg, B, models=M.arm_models(M.DUAL_BASES))
    x0, xf = bench_dual14.dual_states(B, margins)
jmax = margins[4] * M.default_limits()["jmax"]
warm = s.warm_start_jerk(x0, xf, jmax)
sx, su, sT, info = s.solve(x0, xf, warm)
st = np.zeros((B, 160), dtype=np.uint64)
capi.check(capi.lib().mpcmp_debug_stamps(s._ctx, B, st.ctypes.data_as(C.c_void_p)))
st = st.astype(np.float64)
its = np.maximum(st[:, 15], 1)
print("B=%d nseg=%d narm=%d  mean ADMM iterations %.1f (one SQP iteration)" % (B, nseg, narm, its.mean()))
names = ["A  rhs = sigma x - q + rho z - y + A^T w  (S)", "P1 t = G b_J, K_CJ t  (G)", "P3 y_I = S^-1 r_I  (S)", "P4 x_J = G(b_J - K_JC y_I)  (G, S)",
         "E  z~ = A x~, projection, duals  (S)", "termination tests (total / iterations)"]
for k, nm in enumerate(names):
    print("%-52s %9.1f cycles / iteration" % (nm, (st[:, k] / its).mean()))
print("%-52s %9.1f cycles / iteration" % ("loop total", (st[:, 0:6].sum(axis=1) / its).mean()))
busy = st[:, 16:144].reshape(B, 16, 8)
print("per-wave busy cycles / iteration in phases A, P1, P3, P4, E (waves 0-7: G role, 8-15: S role)")
for w in range(16):
    print("  wave %2d  " % w + "  ".join("%8.1f" % (busy[:, w, ph] / its).mean() for ph in range(5)) + "   | P3 parts: r_I build %7.1f  S^-1 block %7.1f" % ((busy[:, w, 5] / its).mean(), (busy[:, w, 6] / its).mean()))

fs = st[:, 128:135]
print("k_qp3f (factorisation), cycles per QP: prologue %.0f | assembly (all but K_II) %.0f | diagonal, sweep of the interior blocks %.0f | G rows %.0f | K_II assembly + Schur complement %.0f | G scatter + sweep of S %.0f | derived copies + hand-over %.0f | total %.0f"
      % (*fs.mean(axis=0), fs.sum(axis=1).mean()))
