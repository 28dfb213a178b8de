#!/usr/bin/env python3
"""QP warm start (mpcmp_config.qp_warm_start) against the cold default, problem by problem: the headline batch (N = 13, 20 SQP) and the shipped depth
(N = 19, 2 SQP), 1,024 seeded problems each.  The mean final time over ALL problems mixes feasible and infeasible iterates (an iterate that violates
its constraints can be faster than any feasible one), so the figures are also given over the problems that are inside every tolerance in BOTH runs.
usage (GPU box): python tools/warm_subset.py > gpurun_out/warm_subset.json"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
MARGINS = (0.9, 0.9, 0.5, 0.9)
out = {}
for name, nseg, sqp in (("batch", 4, 20), ("shipped", 6, 2)):
    B = 1024
    x0, xf = scenarios.make_batch(B, MARGINS)
    res = {}
    for qws in (0, 1):
        cfg = M.default_config(nseg, sqp, margins=MARGINS, qp_warm_start=qws)
        s = M.Solver(cfg, B)
        wx, wu, wT = s.warm_start(x0, xf)
        sx, su, sT, info = s.solve(x0, xf, (wx, wu, wT))
        res[qws] = (sT.copy(), info["status"].copy(), info["qp_iters_total"].copy())
        del s
    (Tc, stc, itc), (Tw, stw, itw) = res[0], res[1]
    bad = 1 | 2 | 4 | 16 | 32
    fc, fw = (stc & bad) == 0, (stw & bad) == 0
    both = fc & fw
    out[name] = {"problems": B, "feasible_cold": int(fc.sum()), "feasible_warm": int(fw.sum()), "feasible_both": int(both.sum()),
                 "feasible_only_cold": int((fc & ~fw).sum()), "feasible_only_warm": int((fw & ~fc).sum()),
                 "T_mean_all": [float(Tc.mean()), float(Tw.mean())], "T_mean_feasible_both": [float(Tc[both].mean()), float(Tw[both].mean())],
                 "T_warm_over_cold_feasible_both": {"median": float(np.median(Tw[both] / Tc[both])), "p05": float(np.quantile(Tw[both] / Tc[both], 0.05)),
                                                    "p95": float(np.quantile(Tw[both] / Tc[both], 0.95)), "mean": float((Tw[both] / Tc[both]).mean())},
                 "T_mean_infeasible_cold": float(Tc[~fc].mean()) if (~fc).any() else None,
                 "admm_iters_mean": [float(itc.mean()), float(itw.mean())]}
print(json.dumps(out, indent=1))
