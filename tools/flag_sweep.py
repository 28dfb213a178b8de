#!/usr/bin/env python3
"""Compiler-flag A/B of k_qp2 at a fixed ADMM iteration count (eps = 0): builds one library per flag set (tools/micro/libflag_<n>.bin, here) and times
one QP launch of 1024 problems with each (on the GPU box), best of 6.  usage: flag_sweep.py build | run"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
SETS = {
    0: [],
    1: ["-mllvm", "-amdgpu-schedule-relaxed-occupancy=true"],
    2: ["-mllvm", "-enable-post-misched=false"],
    3: ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
    4: ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    5: ["-mllvm", "-amdgpu-enable-merge-m0=true"],
    6: ["-mllvm", "-amdgpu-use-divergent-register-indexing=true"],
    7: ["-mllvm", "-misched-prera-direction=bottomup"],
    8: ["-mllvm", "-misched-prera-direction=topdown"],
    9: ["-mllvm", "-amdgpu-igrouplp=false"],
    10: ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule=true"],
    11: ["-mllvm", "-amdgpu-waitcnt-forcelgkm=false"],
}
def so(n): return os.path.join(ROOT, "tools", "micro", "libflag_%d.bin" % n)
if sys.argv[1:] == ["build"]:
    procs = []
    for n, fl in SETS.items():
        procs.append((n, subprocess.Popen(["/opt/rocm/bin/hipcc", *BASE, *fl, "-o", so(n), os.path.join(ROOT, "mpc_motion_planner_amd", "csrc", "mpcmp.hip")],
                                          stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        if len(procs) % 6 == 0:
            for _, p in procs[-6:]: p.wait()
    for n, p in procs:
        err = p.communicate()[1].decode()
        print(n, SETS[n], "ok" if p.returncode == 0 else "FAILED: " + err[-200:])
    sys.exit(0)
for n in SETS:
    if not os.path.exists(so(n)): continue
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import mpc_motion_planner_amd.capi as capi
capi._SO = %r
import mpc_motion_planner_amd as M
from mpc_motion_planner_amd import scenarios
cfg = M.default_config(4, 1, margins=(0.9, 0.9, 0.5, 0.9))
cfg.eps_abs = 0.0; cfg.eps_rel = 0.0
s = M.Solver(cfg, 1024)
x0, xf = scenarios.make_batch(1024)
wx, wu, wT = s.warm_start(x0, xf)
best = 1e9
for rep in range(6):
    s.kernel_timing(reset=True)
    p, y, it = s.qp(x0, xf, wx, wu, wT)
    name, ms, launches = s.kernel_timing()
    best = min(best, ms / launches)
print("%%.4f" %% best)
''' % (ROOT, so(n))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    try:
        print("flags %2d %-60s %s ms/launch" % (n, " ".join(SETS[n]), out.stdout.strip().split()[-1]))
    except Exception:
        print("flags %2d failed: %s" % (n, out.stderr[-300:]))
