#!/usr/bin/env python3
"""Throughput of the non-headline configurations (parity-test cases of BASELINE.json, not bench lines)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
margins = (0.9, 0.9, 0.5, 0.9, 0.1)
for nseg, sqp, B in [(6, 2, 1024), (4, 2, 1024), (4, 20, 1024), (4, 20, 4096), (2, 20, 1024), (1, 20, 1024)]:
    cfg = M.default_config(nseg, sqp, margins=margins)
    s = M.Solver(cfg, B)
    x0, xf = scenarios.make_batch(B, margins)
    s.solve(x0, xf)
    t0 = time.perf_counter(); sx, su, sT, info = s.solve(x0, xf); dt = time.perf_counter() - t0
    print("N=%2d sqp=%2d B=%5d : %8.0f traj/s (host buffers)  ok=%.3f  T_mean=%.3f  defect_med=%.1e  admm/traj=%.0f" % (
        3 * nseg + 1, sqp, B, B / dt, (info["status"] == 0).mean(), sT.mean(), np.median(info["defect_inf"]), info["qp_iters_total"].mean()))
    s.close()
