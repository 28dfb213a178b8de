#!/usr/bin/env python3
"""What the termination tests cost: k_qp2 at a fixed 700 iterations with the test every 25 iterations (product) and never (check_every 1000)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for ce in (25, 1000):
    cfg = M.default_config(4, 1, margins=(0.9, 0.9, 0.5, 0.9))
    cfg.eps_abs = 0.0; cfg.eps_rel = 0.0; cfg.check_every = ce
    s = M.Solver(cfg, B)
    x0, xf = scenarios.make_batch(B)
    wx, wu, wT = s.warm_start(x0, xf)
    best = 1e9
    for rep in range(6):
        s.kernel_timing(reset=True)
        p, y, it = s.qp(x0, xf, wx, wu, wT)
        name, ms, launches = s.kernel_timing()
        best = min(best, ms / launches)
    print("check_every %4d: %-6s %.4f ms per launch (B = %d, %d iterations)" % (ce, name, best, B, int(np.mean(it))))
