#!/usr/bin/env python3
"""A/B timing of the QP kernels of num_seg 4 at a FIXED ADMM iteration count (eps = 0: every QP runs qp_iters iterations):
one QP per problem, HIP-event time per launch, best of 6.  usage: qpbench.py [B ...]   (env MPCMP_QP13 = 2 | 3 | 4 picks the kernel)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_motion_planner_amd.capi as _capi
if os.environ.get("QPB_LIB"):
    _capi._SO = os.path.abspath(os.environ["QPB_LIB"])      # A/B against another build of the library
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
nseg = int(os.environ.get("QPB_NSEG", "4"))
for B in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    cfg = M.default_config(nseg, 1, margins=(0.9, 0.9, 0.5, 0.9))
    cfg.eps_abs = 0.0; cfg.eps_rel = 0.0
    if os.environ.get("QPB_CE"):
        cfg.check_every = int(os.environ["QPB_CE"])      # (e.g. 10000: no termination test at all)
    if os.environ.get("QPB_ITERS"):
        cfg.qp_iters = int(os.environ["QPB_ITERS"])      # (two settings give the per-iteration cost without the kernel's prologue)
    s = M.Solver(cfg, B)
    x0, xf = scenarios.make_batch(B)
    wx, wu, wT = s.warm_start(x0, xf)
    best = 1e9
    for rep in range(6):
        s.kernel_timing(reset=True)
        p, y, it = s.qp(x0, xf, wx, wu, wT)
        name, ms, launches = s.kernel_timing()
        best = min(best, ms / launches)
    print("QP13=%s nseg=%d B=%5d  %-8s %8.3f ms/launch (best of 6), %6.1f iterations -> %.3f us per problem-iteration-slot" %
          (os.environ.get("MPCMP_QP13", "2"), nseg, B, name, best, np.mean(it), 1e3 * best / np.mean(it)))
