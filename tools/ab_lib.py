#!/usr/bin/env python3
"""A/B of two builds of libmpcmp.so on the N=13 QP (one launch of 1024 problems, shipped tolerances): time per launch, and whether the
outputs (p, y, iteration counts) of the second build equal the first's bit for bit.  usage: ab_lib.py <libA> <libB> [eps0] [ce=<check_every>] [qi=<qp_iters>]
(eps0: tolerances zero, i.e. a fixed 700 iterations)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import mpc_motion_planner_amd.capi as capi
capi._SO = %r
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
cfg = M.default_config(4, 1, margins=(0.9, 0.9, 0.5, 0.9))
if %r:
    cfg.eps_abs = 0.0; cfg.eps_rel = 0.0
if %r:
    cfg.check_every = %r
if %r:
    cfg.qp_iters = %r
s = M.Solver(cfg, 1024)
x0, xf = scenarios.make_batch(1024)
wx, wu, wT = s.warm_start(x0, xf)
best = 1e9
for rep in range(6):
    s.kernel_timing(reset=True)
    p, y, it = s.qp(x0, xf, wx, wu, wT)
    name, ms, launches = s.kernel_timing()
    best = min(best, ms / launches)
np.savez(%r, p=p, y=y, it=it)
print("%%s %%.4f ms/launch, mean iterations %%.1f, max %%d" %% (name, best, it.mean(), it.max()))
'''
libs = sys.argv[1:3]
eps0 = "eps0" in sys.argv[3:]
ce = [int(a[3:]) for a in sys.argv[3:] if a.startswith("ce=")]
ce = ce[0] if ce else 0
qi = [int(a[3:]) for a in sys.argv[3:] if a.startswith("qi=")]
qi = qi[0] if qi else 0
outs = []
for i, lib in enumerate(libs):
    out = os.path.join(ROOT, "gpurun_out", "ab_%d.npz" % i)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    r = subprocess.run([sys.executable, "-c", CODE % (ROOT, os.path.abspath(lib), eps0, ce, ce, qi, qi, out)], capture_output=True, text=True)
    print(lib, r.stdout.strip() or r.stderr[-400:])
    outs.append(out)
import numpy as np
a, b = np.load(outs[0]), np.load(outs[1])
for k in ("p", "y", "it"):
    same = np.array_equal(a[k], b[k])
    print(k, "identical" if same else "DIFFERENT: max |d| = %g in %d entries" % (np.abs(a[k].astype(float) - b[k].astype(float)).max(), (a[k] != b[k]).sum()))
