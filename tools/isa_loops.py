#!/usr/bin/env python3
"""Scratch traffic of a kernel's loops, from the ISA the compiler emits (no GPU needed).

    python tools/isa_loops.py [k_qp5|k_qp3|k_qp2|<mangled prefix>] [--all]

Compiles mpc_motion_planner_amd/csrc/mpcmp.hip for gfx950 with the product build's flags and --save-temps into a
temporary directory, finds every backward branch of the named kernel and prints, per loop, its length in ISA lines, the
barriers inside it and the scratch loads / stores inside it.  DESIGN.md's "no scratch in the hot loops" statements are
this listing (a hot loop = a loop with five barriers: one ADMM iteration of one role).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc_motion_planner_amd", "csrc")
KERNELS = {"k_qp5": "_ZN5mpcmp5k_qp5ILi6E", "k_qp3": "_ZN5mpcmp5k_qp3ILi6ELi1E", "k_qp2": "_ZN5mpcmp5k_qp2ILi4E", "k_qp3f": "_ZN5mpcmp6k_qp3fILi6ELi1ELi5E"}


def compile_isa(tmp):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang",
           "-load-store-opt", "-falign-loops=64", "-DMPCMP_SPLIT_N25", "--save-temps", "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "include"),
           "-c", "-o", os.path.join(tmp, "m.o"), os.path.join(CSRC, "mpcmp.hip")]
    r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    return os.path.join(tmp, "mpcmp-hip-amdgcn-amd-amdhsa-gfx950.s"), r.stderr


def loops(path, prefix, show_all):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(re.escape(prefix) + r"\S*:", l)]
    if not starts:
        sys.exit("no kernel with prefix %s" % prefix)
    for start in starts:
        end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        body = lines[start:end]
        print("%s  (%d ISA lines, %d scratch ops in all)" % (lines[start].split(":")[0], len(body), sum("scratch_" in l for l in body)))
        lab = {}
        for n, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                lab[m.group(1)] = n
        for n, l in enumerate(body):
            m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if not m:
                continue
            t = m.group(1) or m.group(2)
            if lab.get(t, 1 << 30) >= n:
                continue
            seg = body[lab[t]:n]
            nb = sum("s_barrier" in x for x in seg)
            nl = sum("scratch_load" in x for x in seg)
            ns = sum("scratch_store" in x for x in seg)
            if show_all or nb or nl or ns:
                print("  loop at +%-6d len %-5d barriers %-3d scratch loads %-3d stores %-3d%s" %
                      (lab[t], n - lab[t], nb, nl, ns, "   <- one ADMM iteration of a role" if nb == 5 else ""))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "k_qp5"
    with tempfile.TemporaryDirectory() as tmp:
        path, remarks = compile_isa(tmp)
        loops(path, KERNELS.get(name, name), "--all" in sys.argv)


if __name__ == "__main__":
    main()
