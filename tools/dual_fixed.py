#!/usr/bin/env python3
"""Dual-arm N = 25 solve at a FIXED ADMM iteration count (eps = 0: every QP runs qp_iters iterations), one SQP iteration: HIP-event time of the
k_qp3f + k_qp3 launches per launch, best of 4.  usage: QPB_LIB=<lib> dual_fixed.py [B]   (what-if builds: results may be garbage, the timing counts)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_motion_planner_amd.capi as _capi
if os.environ.get("QPB_LIB"):
    _capi._SO = os.path.abspath(os.environ["QPB_LIB"])
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
MARG = (0.9, 0.9, 0.5, 0.9, 0.1)
cfg = M.default_config(8, 1, margins=MARG)
cfg.eps_abs = 0.0; cfg.eps_rel = 0.0
if os.environ.get("QPB_ITERS"):
    cfg.qp_iters = int(os.environ["QPB_ITERS"])
s = M.Solver(cfg, B, models=M.arm_models(M.DUAL_BASES))
a0, af = scenarios.make_batch(B, MARG, stream_offset=60)
b0, bf = scenarios.make_batch(B, MARG, stream_offset=60 + B)
x0 = np.ascontiguousarray(np.concatenate([a0[:, :7], b0[:, :7], a0[:, 7:], b0[:, 7:]], axis=1))
xf = np.ascontiguousarray(np.concatenate([af[:, :7], bf[:, :7], af[:, 7:], bf[:, 7:]], axis=1))
warm = s.warm_start_jerk(x0, xf, MARG[4] * M.default_limits()["jmax"])
best = 1e9
for rep in range(4):
    s.kernel_timing(reset=True)
    s.solve(x0, xf, warm)
    name, ms, launches = s.kernel_timing()
    best = min(best, ms / max(launches, 1))
print("dual N=25 B=%d  %s  %.3f ms per QP launch pair (best of 4), %d iterations" % (B, name, best, cfg.qp_iters))
