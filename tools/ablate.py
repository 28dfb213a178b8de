#!/usr/bin/env python3
"""What-if profile of k_qp2: build the library with -DMPCMP_ABL=n for each n (one role's work in one ADMM phase removed),
run ONE QP per problem at a fixed iteration count (eps = 0, so every QP runs qp_iters iterations) and print the kernel time.
The drop against n = 0 is that piece's share of the critical path.  Diagnostic only (results of ablated builds are wrong)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "product build", 1: "B: E (rows+vars)", 2: "A2: E (path rows)", 3: "B: A (rhs gather)", 4: "A1 wave0: T column sum",
         5: "A2: P1 (r_I)", 6: "A1: P1 part of G b", 7: "B: P2 (S^-1 r)", 8: "A1: P3 (x_J)", 9: "A1: P2 part of G b", 11: "candidate build",
         20: "wave order: identity", 21: "wave order: light B pair + light A1 share a SIMD", 22: "wave order: three heavy B on one SIMD (control)",
         23: "wave order: candidate 3", 24: "wave order: candidate 4", 25: "wave order: candidate 5"}
WPERM = {20: "0xFEDCBA9876543210ull", 21: "0xFDCBEA9875436210ull", 22: "0x76FEDA98C543B210ull",
         23: os.environ.get("WPERM23", "0xFEDCBA9876543210ull"), 24: os.environ.get("WPERM24", "0xFEDCBA9876543210ull"), 25: os.environ.get("WPERM25", "0xFEDCBA9876543210ull")}
which = [int(a) for a in sys.argv[1:]] or sorted(NAMES)
B = 1024
import mpc_motion_planner_amd.capi as capi
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
for n in which:
    so = os.path.join(ROOT, "tools", "micro", "libabl%d.bin" % n)
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-falign-loops=64", *(["-DMPCMP_WPERM=" + WPERM[n]] if n in WPERM else ["-DMPCMP_ABL=%d" % n]),
                               "-o", so, os.path.join(ROOT, "mpc_motion_planner_amd", "csrc", "mpcmp.hip")])
if "--build-only" in os.environ.get("ABLATE_MODE", ""):
    sys.exit(0)
base = None
for n in which:
    # one process per library (the ctypes handle is cached per process)
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import mpc_motion_planner_amd.capi as capi
capi._SO = %r
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
cfg = M.default_config(4, 1, margins=(0.9, 0.9, 0.5, 0.9))
cfg.eps_abs = 0.0; cfg.eps_rel = 0.0
s = M.Solver(cfg, %d)
x0, xf = scenarios.make_batch(%d)
wx, wu, wT = s.warm_start(x0, xf)
best = 1e9
for rep in range(6):
    s.kernel_timing(reset=True)
    p, y, it = s.qp(x0, xf, wx, wu, wT)
    name, ms, launches = s.kernel_timing()
    best = min(best, ms / launches)
print("%%.4f %%.2f" %% (best, np.mean(it)))
''' % (ROOT, os.path.join(ROOT, "tools", "micro", "libabl%d.bin" % n), B, B)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    try:
        ms, its = [float(v) for v in out.stdout.strip().split()[-2:]]
    except Exception:
        print(n, "failed", out.stdout[-300:], out.stderr[-600:]); continue
    if n == 0: base = ms
    print("ABL %2d  %-28s  %8.3f ms/launch (best of 6), %6.1f iterations  %s" % (n, NAMES[n], ms, its, "" if base is None or n == 0 else "(%+.1f %%)" % (100 * (ms - base) / base)))
