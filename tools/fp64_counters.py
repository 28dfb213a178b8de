#!/usr/bin/env python3
"""FP64 flops the QP kernels EXECUTE per ADMM iteration of one trajectory, from one rocprofv3 --pmc pass
(SQ_INSTS_VALU_FMA_F64, _ADD_F64, _MUL_F64, SQ_INSTS_VALU_MFMA_MOPS_F64: wave-level instruction counts) of `bench.py --workload W`:
    flops = 64 lanes x (2 FMA + ADD + MUL) + 512 x MFMA_MOPS      summed over every QP kernel dispatch of the process,
    divided by (problems solved in the process x ADMM iterations per trajectory of the bench line).
usage: fp64_counters.py <workload> <pmc results.db> <bench line of the same process (json file or log)> [<mfma results.db>]
prints one JSON object (tools/profile_round.sh collects them into profiles/r03_fp64_counters.json)."""
import json, os, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import csrc_sha16

def is_qp(name):
    return "k_qp" in name          # k_qp, k_qp2, k_qp3f, k_qp3, k_qp4 (mangled or demangled)


def counters(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    out = {}
    for kn, cn, cnt, sm in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(kn, {})[cn] = (cnt, sm)
    problems = 0
    for name, gx, wx in cur.execute("select name, grid_x, workgroup_x from kernels"):
        if "k_init" in name:
            problems += gx // max(wx, 1)
    return out, problems


def main():
    workload, pmc_db, line_path = sys.argv[1], sys.argv[2], sys.argv[3]
    txt = open(line_path).read()
    line = json.loads([l for l in txt.splitlines() if l.strip().startswith("{")][-1])
    admm = line["roofline"].get("admm_iters_per_traj") or line["roofline"].get("admm_iters_per_resolve")
    c, problems = counters(pmc_db)
    mf = {}
    if len(sys.argv) > 4:
        mf, _ = counters(sys.argv[4])
    tot = {"SQ_INSTS_VALU_FMA_F64": 0.0, "SQ_INSTS_VALU_ADD_F64": 0.0, "SQ_INSTS_VALU_MUL_F64": 0.0, "SQ_INSTS_VALU_MFMA_MOPS_F64": 0.0}
    per_kernel = {}
    for src in (c, mf):
        for kn, d in src.items():
            if not is_qp(kn):
                continue
            for cn in tot:
                if cn in d:
                    tot[cn] += d[cn][1]
                    per_kernel.setdefault(kn[:60], {})[cn] = d[cn][1] / max(d[cn][0], 1)
    flops = 64.0 * (2.0 * tot["SQ_INSTS_VALU_FMA_F64"] + tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"]) + 512.0 * tot["SQ_INSTS_VALU_MFMA_MOPS_F64"]
    print(json.dumps({"workload": workload, "csrc_sha16": csrc_sha16(), "problems_solved_in_process": problems, "admm_iters_per_traj": admm, "counter_sums": tot,
                      "per_dispatch_avg": per_kernel, "fp64_flops_total": flops,
                      "fp64_flops_per_traj_admm_iter": flops / max(problems * admm, 1.0)}))


if __name__ == "__main__":
    main()
