#!/usr/bin/env python3
"""Diagnostic: build the library with -DMPCMP_STAMPS and print where k_qp's cycles go (per ADMM iteration).
Not part of the product path; stamps cost a few % (see MI355X_MICROARCH.md, s_memtime)."""
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
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = M.default_config(nseg, 1, margins=(0.9, 0.9, 0.5, 0.9))
s = M.Solver(cfg, B)
x0, xf = scenarios.make_batch(B)
wx, wu, wT = s.warm_start(x0, xf)
p, y, it = s.qp(x0, xf, wx, wu, wT)
st = np.zeros((B, 160), dtype=np.uint64)
capi.check(capi.lib().mpcmp_debug_stamps(s._ctx, B, st.ctypes.data_as(C.c_void_p)))
st = st.astype(np.float64)
its = st[:, 15]
names = ["setup", "factor J blocks (total)", "sweep S", "A(rhs)", "P1", "P2a+P2b", "P3", "E", "check/loop"]
print("B=%d nseg=%d  mean ADMM iterations %.1f" % (B, nseg, its.mean()))
for k, nm in enumerate(names):
    if k < 3:
        print("%-26s %10.0f cycles (once)" % (nm, st[:, k].mean()))
    else:
        print("%-26s %10.1f cycles / iteration" % (nm, (st[:, k] / its).mean()))
print("loop total %10.1f cycles / iteration" % ((st[:, 3:9].sum(axis=1) / its).mean()))
for k, nm in zip(range(9, 15), ["  assemble K_JJ,K_JC", "  (unused)", "  augmented sweep K_JJ|K_JC", "  role prologue (after the factorisation)", "  (unused)", "  park to HBM"]):
    print("%-26s %10.0f cycles (once, all segment groups)" % (nm, st[:, k].mean()))

# per-wave busy cycles per iteration (k_qp2 only): which role is the critical one in each phase
busy = st[:, 16:144].reshape(B, 16, 8)
if busy.sum() > 0:
    print("per-wave busy cycles / iteration (phase A, P1, P2, P3, E); waves 0-6 role A1, 7-10 role A2, 11-15 role B")
    for w in range(16):
        print("  wave %2d  " % w + "  ".join("%7.1f" % (busy[:, w, ph] / its).mean() for ph in range(5)))

# k_step phases (one SQP iteration through the public solve entry point)
sx, su, sT, info = s.solve(x0, xf, (wx, wu, wT))
st2 = np.zeros((B, 160), dtype=np.uint64)
capi.check(capi.lib().mpcmp_debug_stamps(s._ctx, B, st2.ctypes.data_as(C.c_void_p)))
ks = st2[:, 144:152].astype(np.float64)
if ks.sum() > 0:
    for k, nm in enumerate(["load, mu, l1 at the iterate", "sincos of the trial points", "trial RNEA + FK (9 x N)", "defect / box violation of the trials",
                            "reduction of the 9 merits", "Armijo, update of z and lambda", "re-linearisation (tangent RNEA, Jacobian rows, defects)", "final report"]):
        print("k_step %-52s %9.0f cycles" % (nm, ks[:, k].mean()))
    print("k_step total %9.0f cycles" % ks.sum(axis=1).mean())
    ls = st2[:, 152:156].astype(np.float64)
    for k, nm in enumerate(["sincos of the iterate", "tangent RNEA, one (node, direction) per thread, + FK row", "Jacobian rows -> Gk (global)", "defects -> ceq (global)"]):
        print("   re-linearisation: %-56s %9.0f cycles" % (nm, ls[:, k].mean()))
