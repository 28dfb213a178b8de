#!/usr/bin/env python3
"""Merge the per-workload outputs of tools/profile_round.sh (gpurun_out/<tag>_fp64_counters.json) into profiles/<round>_fp64_counters.json.
usage: merge_counters.py profiles/r04_fp64_counters.json gpurun_out/r04_v1_batch_fp64_counters.json gpurun_out/r04_v1_shipped_fp64_counters.json ..."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import csrc_sha16
out = {"csrc_sha16": csrc_sha16(),
       "note": "tools/profile_round.sh <tag> [--workload W]: rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 (+ SQ_INSTS_VALU_MFMA_MOPS_F64 "
               "from its own pass) over one bench.py process; tools/fp64_counters.py.  Wave-level instruction counts x 64 lanes (inactive lanes included: an upper "
               "bound of the executed flops).  csrc_sha16 (tools/src_hash.py) names the kernel sources each workload was measured on; bench.py reports `stale` when the tree differs.",
       "workloads": {}}
for f in sys.argv[2:]:
    d = json.load(open(f))
    out["workloads"][d["workload"]] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(sys.argv[1], {k: (v.get("csrc_sha16"), round(v["fp64_flops_per_traj_admm_iter"])) for k, v in out["workloads"].items()})
