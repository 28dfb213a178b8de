#!/usr/bin/env python3
"""What-if profile of k_qp5 (N = 19): builds the library with -DQ5_ABL=n for every n given (0 = product), runs one QP launch of 256 problems at a
fixed 700 ADMM iterations with each, and prints the time per launch: the difference to n = 0 is that piece's share of the critical path.
  1 phase A (rhs of the variables)   2 P1 (G product, K_CJ t)   3 P2 (S^-1 product)   4 P3 (E product)   5 E: path rows   6 E: variables, dynamics rows
usage: ablate5.py build n...   (here, cross-compiles tools/micro/libabl5_<n>.bin)        ablate5.py run n...   (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def lib(n): return os.path.join(ROOT, "tools", "micro", "libabl5_%s.bin" % n)
mode, ns = sys.argv[1], sys.argv[2:]
if mode == "build":
    procs = []
    for n in ns:
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt",
                                       "-DQ5_ABL=%s" % n, *os.environ.get("ABL_EXTRA", "").split(), "-o", lib(n), os.path.join(ROOT, "mpc_motion_planner_amd", "csrc", "mpcmp.hip")], stderr=subprocess.DEVNULL))
        if len(procs) == 4:
            for p in procs: p.wait()
            procs = []
    for p in procs: p.wait()
else:
    for n in ns:
        env = dict(os.environ, QPB_LIB=lib(n), QPB_NSEG="6")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "qpbench.py"), "256"], env=env, capture_output=True, text=True)
        print("Q5_ABL=%s  %s" % (n, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]))
