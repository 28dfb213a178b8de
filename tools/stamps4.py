#!/usr/bin/env python3
"""Diagnostic: build the library with -DMPCMP_STAMPS and print where k_qp4's cycles go per ADMM iteration (phase totals and busy parts seen by
lane 0 of wave 0 (role A) and of waves 3..5 (role B)), and which workgroups shared a CU in time.  usage: stamps4.py [B]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "gpurun_out", "libmpcmp_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-DMPCMP_STAMPS",
                       *os.environ.get("MPCMP_EXTRA_DEFS", "").split(), "-o", so, os.path.join(ROOT, "mpc_motion_planner_amd", "csrc", "mpcmp.hip")], stderr=subprocess.DEVNULL)
os.environ["MPCMP_QP13"] = "4"
import mpc_motion_planner_amd.capi as capi
capi._SO = so
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = M.default_config(4, 1, margins=(0.9, 0.9, 0.5, 0.9))
cfg.eps_abs = 0.0; cfg.eps_rel = 0.0
s = M.Solver(cfg, B)
x0, xf = scenarios.make_batch(B)
wx, wu, wT = s.warm_start(x0, xf)
p, y, it = s.qp(x0, xf, wx, wu, wT)
st = np.zeros((B, 160), dtype=np.uint64)
capi.check(capi.lib().mpcmp_debug_stamps(s._ctx, B, st.ctypes.data_as(C.c_void_p)))
f = st.astype(np.float64)
its = np.maximum(f[:, 15], 1)
names = ["A", "P1a", "P1b", "P3", "P4a", "P4b", "E", "test"]
print("B=%d iterations %.0f" % (B, its.mean()))
print("role A, wave 0: cycles / iteration per phase (total | busy before the barrier)")
for k in range(8):
    print("  %-4s %8.1f | %8.1f" % (names[k], (f[:, k] / its).mean(), (f[:, 16 + k] / its).mean()))
print("  loop total %.1f" % (f[:, 0:8].sum(axis=1) / its).mean())
for w in range(3):
    print("role B, wave %d: total | busy" % (3 + w))
    for k in range(8):
        print("  %-4s %8.1f | %8.1f" % (names[k], (f[:, 32 + 16 * w + k] / its).mean(), (f[:, 32 + 16 * w + 8 + k] / its).mean()))
hw = st[:, 150]; xcc = st[:, 153]; t0 = st[:, 151].astype(np.int64); t1 = st[:, 152].astype(np.int64)
cu = ((hw >> 8) & 0xF).astype(int); sh = ((hw >> 12) & 1).astype(int); se = ((hw >> 13) & 0x7).astype(int); xc = (xcc & 0xF).astype(int)
key = xc * 1000 + se * 100 + sh * 50 + cu
import collections
groups = collections.defaultdict(list)
for i in range(B): groups[int(key[i])].append(i)
ov = 0; pairs = 0
for k, v in groups.items():
    for a in range(len(v)):
        for c in range(a + 1, len(v)):
            pairs += 1
            lo = max(t0[v[a]], t0[v[c]]); hi = min(t1[v[a]], t1[v[c]])
            if hi > lo: ov += 1
print("distinct (xcc, se, sh, cu): %d; workgroup pairs on one CU: %d, overlapping in time: %d; wall per workgroup %.1f us (100 MHz clock)" %
      (len(groups), pairs, ov, (t1 - t0).mean() / 100.0))
