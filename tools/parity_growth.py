#!/usr/bin/env python3
"""Diagnostic: GPU-vs-oracle differences as a function of the SQP iteration count (round-off growth)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
import oracle_py as o
margins = (0.9, 0.9, 0.5, 0.9, 0.1)
B = 6
x0, xf = scenarios.make_batch(B, margins, stream_offset=100)
for sqp in (1, 2, 3, 5, 10, 20):
    cfg = M.default_config(4, sqp, margins=margins); ocfg = o.default_config(4, sqp, margins=margins)
    s = M.Solver(cfg, B)
    wx = np.zeros((B, 13, 14)); wu = np.zeros((B, 13, 7)); wT = np.zeros(B)
    for b in range(B):
        wx[b], wu[b], wT[b] = o.warm_start(ocfg, x0[b], xf[b])
    sx, su, sT, info = s.solve(x0, xf, (wx, wu, wT))
    sx2, su2, sT2, _ = s.solve(x0, xf, (wx, wu, wT))
    dT = []; dx = []
    for b in range(B):
        xs, us, T, oi = o.solve(ocfg, x0[b], xf[b], wx[b], wu[b], wT[b])
        dT.append(abs(sT[b] - T) / T); dx.append(np.abs(sx[b] - xs).max())
    print("sqp %2d  max rel dT %.2e  max dx %.2e  repeat-run identical: %s  alphas %s" % (sqp, max(dT), max(dx), np.array_equal(sx, sx2), info["last_alpha"]))
