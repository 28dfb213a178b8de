#!/usr/bin/env python3
"""Per-phase issue table of the ADMM role loops of a QP kernel, from the ISA (no GPU needed).

    python tools/isa_phases.py [k_qp2|k_qp5|k_qp3_n25] [--raw]

Compiles mpc_motion_planner_amd/csrc/mpcmp.hip for gfx950 with the product flags plus -DMPCMP_NOCHECK (the termination-test
block is compiled out, so the hot path of a role is the straight-line text between its workgroup barriers), finds the role
loops (backward branches whose body holds exactly five s_barrier) and prints, per role and per phase (= the text between two
barriers, in iteration order A, P1, P2, P3, E), the instructions a WAVE of that role issues:

    f64      v_fma / v_fmac / v_mul / v_add / v_max / v_min _f64      4 cycles of a SIMD each (16 FP64 lanes per cycle)
    valu     every other vector ALU instruction (v_mov_dpp, v_cndmask, integer)   4 cycles alone, 2 when another wave interleaves
    lds_r    ds_read_b128 / b64 / b32 ...   LDS-array cycles by MI355X_MICROARCH.md: b128 4, b64 2, b32 2, read2_b64 8
    lds_w    ds_write_b128 8, b64 4, b32 2
    salu     scalar instructions (own issue port)

and `issue` = 4 f64 + 4 valu, the SIMD cycles one wave of the role needs for the phase if nothing overlaps; `lds` = LDS-array
cycles of one wave of the role.  DESIGN.md 9 (spread layout estimate) is computed from this listing and the role -> SIMD map.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc_motion_planner_amd", "csrc")
KERNELS = {"k_qp5": "_ZN5mpcmp5k_qp5ILi6E", "k_qp2": "_ZN5mpcmp5k_qp2ILi4E", "k_qp6": "_ZN5mpcmp5k_qp6ILi4E",
           "k_qp3_n25": "_ZN5mpcmp5k_qp3ILi8ELi2E"}      # (k_qp3_n25: the dual-arm N = 25 loop kernel, from csrc/qp3_n25.hip)
LDS_R = {"ds_read_b128": 4, "ds_read_b64": 2, "ds_read_b32": 2, "ds_read2_b64": 8, "ds_read2_b32": 4, "ds_read_u16": 2, "ds_read_u8": 2, "ds_read_b96": 8,
         "ds_read_u16_d16": 2, "ds_read_u16_d16_hi": 2, "ds_read2st64_b64": 8}
LDS_W = {"ds_write_b128": 8, "ds_write_b64": 4, "ds_write_b32": 2, "ds_write2_b64": 8, "ds_write2_b32": 4, "ds_write_b96": 8, "ds_write_b16": 2}


def compile_isa(tmp, extra, n25=False, n19=False):
    src = "qp3_n25.hip" if n25 else ("qp5_n19.hip" if n19 else "mpcmp.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang",
           "-load-store-opt", "-falign-loops=64", "-DMPCMP_SPLIT_N25", "-DMPCMP_NOCHECK", "--save-temps", "-I", os.path.join(ROOT, "include"),
           "-c", "-o", os.path.join(tmp, "m.o"), os.path.join(CSRC, src)] + extra
    if n25:
        cmd += ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"]      # (as csrc/Makefile builds that translation unit)
    if n19:
        cmd += ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]
    r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    return os.path.join(tmp, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def classify(op):
    if re.match(r"v_(fma|fmac|mul|add|max|min)_f64", op):
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op in LDS_R or op.startswith("ds_read"):
        return "lds_r"
    if op in LDS_W or op.startswith("ds_write"):
        return "lds_w"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "other"


def role_loops(path, prefix):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(re.escape(prefix) + r"\S*:", l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[start:end]
    lab = {}
    for n, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = n
    loops = []
    for n, l in enumerate(body):
        m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if not m:
            continue
        t = m.group(1) or m.group(2)
        if lab.get(t, 1 << 30) >= n:
            continue
        seg = body[lab[t]:n + 1]
        if sum("s_barrier" in x for x in seg) == 5:
            loops.append((lab[t], n, seg))
    # keep the innermost (shortest) loop per start region
    loops.sort(key=lambda x: (x[0], x[1]))
    out = []
    for lo, hi, seg in loops:
        if not any(o[0] <= lo and hi <= o[1] and (o[0], o[1]) != (lo, hi) for o in out):
            out = [o for o in out if not (lo <= o[0] and o[1] <= hi)] + [(lo, hi, seg)]
    return out


def phases(seg):
    """instruction classes between consecutive barriers, cyclically, starting with the text after the LAST barrier of the loop body (the next
    iteration's first phase continues at the loop head)"""
    ops = []
    for s in seg:
        if not s.startswith("\t") or s.strip().startswith((".", ";")):
            continue
        t = s.split()
        if t:
            ops.append(t[0])
    cut = [i for i, o in enumerate(ops) if o.startswith("s_barrier")]
    parts = []
    for k in range(5):
        a = cut[k - 1] + 1 if k > 0 else None
        if k == 0:
            part = ops[cut[-1] + 1:] + ops[:cut[0]]
        else:
            part = ops[a:cut[k]]
        parts.append(part)
    return parts


def summarise(part):
    c = {"f64": 0, "valu": 0, "lds_r": 0, "lds_w": 0, "salu": 0, "vmem": 0, "dpp": 0}
    lds = 0
    for o in part:
        k = classify(o)
        if k in c:
            c[k] += 1
        if "dpp" in o:
            c["dpp"] += 1
        if k == "lds_r":
            lds += LDS_R.get(o, 2)
        if k == "lds_w":
            lds += LDS_W.get(o, 4)
    c["issue"] = 4 * (c["f64"] + c["valu"])
    c["lds"] = lds
    return c


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "k_qp2"
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    with tempfile.TemporaryDirectory() as tmp:
        path = compile_isa(tmp, extra, name.endswith("n25"), name == "k_qp5")
        loops = role_loops(path, KERNELS.get(name, name))
        print("%s: %d role loops (five barriers each)" % (name, len(loops)))
        for r, (lo, hi, seg) in enumerate(loops):
            parts = phases(seg)
            print("role loop %d at +%d (%d ISA lines)" % (r, lo, hi - lo))
            print("   phase    f64  valu (dpp)  lds_r lds_w  salu vmem |  issue cycles   LDS-array cycles")
            tot = {"issue": 0, "lds": 0, "f64": 0, "valu": 0}
            for k, part in enumerate(parts):
                c = summarise(part)
                print("   %d      %5d %5d (%3d) %5d %5d %5d %4d | %8d %14d" % (k, c["f64"], c["valu"], c["dpp"], c["lds_r"], c["lds_w"], c["salu"], c["vmem"], c["issue"], c["lds"]))
                for q in tot:
                    tot[q] += c[q]
            print("   total  %5d %5d                                | %8d %14d" % (tot["f64"], tot["valu"], tot["issue"], tot["lds"]))
            if "--raw" in sys.argv:
                for k, part in enumerate(parts):
                    print("   -- phase %d: %s" % (k, " ".join(part)))


if __name__ == "__main__":
    main()
