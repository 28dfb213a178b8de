#!/usr/bin/env python3
"""sha256 (16 hex digits) over the kernel sources mpc_motion_planner_amd/csrc/*.{hpp,hip} + include/mpcmp.h: the identity of the build that
a committed counter profile (profiles/r0N_fp64_counters.json) was measured on.  bench.py compares it with the working tree and reports
`stale` instead of a roofline fraction when they differ (no .git needed: the GPU box has none)."""
import glob, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "mpc_motion_planner_amd", "csrc", "*.hpp")) + glob.glob(os.path.join(root, "mpc_motion_planner_amd", "csrc", "*.hip")))
    files.append(os.path.join(root, "include", "mpcmp.h"))
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
