#!/usr/bin/env python3
"""How much of a k_qp2 launch is tail?  Collect the per-problem ADMM iteration counts of every SQP iteration of the bench
workload and replay the dispatch of 1024 workgroups on 256 CUs (next workgroup to the first free CU, in index order) for
different launch orders.  Diagnostic only."""
import heapq, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
B, K = 1024, 20
x0, xf = scenarios.make_batch(B, (0.9, 0.9, 0.5, 0.9, 0.1))
tot = np.zeros((K + 1, B))
for k in range(1, K + 1):
    cfg = M.default_config(4, k, margins=(0.9, 0.9, 0.5, 0.9, 0.1))
    s = M.Solver(cfg, B)
    _, _, _, info = s.solve(x0, xf)
    tot[k] = info["qp_iters_total"]
its = np.diff(tot, axis=0)                     # [K][B]
FAC, C = 176e3, 3500.0                         # cycles: factorisation, one ADMM iteration
def makespan(order, w):
    cus = [0.0] * 256
    heapq.heapify(cus)
    for b in order:
        t = heapq.heappop(cus); heapq.heappush(cus, t + w[b])
    return max(cus)
res = {"batch order": 0.0, "longest first, previous counts": 0.0, "longest first, decayed max (k_order)": 0.0,
       "longest first, exact counts": 0.0, "lower bound (sum/256)": 0.0}
key = np.zeros(B)
for k in range(K):
    w = FAC + C * its[k]
    res["batch order"] += makespan(range(B), w)
    prev = its[k - 1] if k > 0 else np.zeros(B)
    key = np.maximum(prev, np.floor(key * 29 / 32))          # solver_kernels.hpp: k_order
    res["longest first, previous counts"] += makespan(np.argsort(-prev, kind="stable"), w)
    res["longest first, decayed max (k_order)"] += makespan(np.argsort(-key, kind="stable"), w)
    res["longest first, exact counts"] += makespan(np.argsort(-w, kind="stable"), w)
    res["lower bound (sum/256)"] += w.sum() / 256
print("mean iterations per QP by SQP iteration:", np.round(its.mean(axis=1), 0).astype(int).tolist())
print("fraction of QPs at the 700 cap:", np.round((its >= 700).mean(axis=1), 2).tolist())
for k, v in res.items():
    print("%-34s %8.2f Mcycles per solve  (%.3f of batch order)" % (k, v / 1e6, v / res["batch order"]))
print("corr(its[k], its[k-1]) for k=5..19:", np.round([np.corrcoef(its[k], its[k - 1])[0, 1] for k in range(5, K)], 2).tolist())
cap = its >= 700
print("P(cap now | cap prev) =", round(float((cap[6:] & cap[5:-1]).sum() / cap[5:-1].sum()), 3), " P(cap now | not cap prev) =",
      round(float((cap[6:] & ~cap[5:-1]).sum() / (~cap[5:-1]).sum()), 3))
np.save(os.path.join(ROOT, "gpurun_out", "its.npy"), its)
