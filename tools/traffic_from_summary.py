#!/usr/bin/env python3
"""HBM traffic and MFMA figures of the QP kernels from the text summaries tools/profile_round.sh leaves (profiles/<tag>_<workload>_rocprofv3_kernel_trace_and_pmc.txt).

    python tools/traffic_from_summary.py profiles/r04_traffic.json r04_v2 batch shipped dual14      (after tools/merge_counters.py)

Writes one record per (workload, kernel): FETCH_SIZE / WRITE_SIZE per launch (KB, averages over the dispatches of the PMC pass), the traffic
figure bench.py reports (FETCH_SIZE x 2 + WRITE_SIZE: MI355X_MICROARCH.md, HBM section — gfx950 tallies 128-B requests at 64 B; the accesses here
are 8 B per lane, so the factor is an upper bound), the MFMA counters of k_qp3f, and SQ_WAIT_ANY / SQ_WAVE_CYCLES.  csrc_sha16 names the kernel
sources the passes ran on — taken from profiles/<round>_fp64_counters.json, which tools/profile_round.sh's counter pass stamped on the GPU box in
the same call (tools/src_hash.py) — and bench.py withholds the figures when the tree differs.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QP = ("k_qp2", "k_qp3", "k_qp3f", "k_qp5", "k_qp")
PROBLEMS = {"batch": 1024 / 3.0, "shipped": 512, "dual14": 2048, "rh": 256}      # problems per launch of the bench workloads (parts of a batch on concurrent streams: three at N = 13 from 1024 problems on, else two)


def parse(path):
    ctr, dur = {}, {}
    for line in open(path):
        m = re.match(r"^(\S+)\s+(\S+)\s+dispatches=(\d+)\s+avg=([0-9.eE+-]+)", line)
        if m and m.group(1) in QP:
            ctr.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
            continue
        f = line.split()
        if len(f) >= 6 and f[0] in QP and f[1].isdigit() and f[0] not in dur:      # first table = the kernel-trace pass: name, calls, total ms, avg us, ...
            try:
                dur[f[0]] = float(f[3])
            except ValueError:
                pass
    return ctr, dur


def main():
    out_path, tag, workloads = sys.argv[1], sys.argv[2], sys.argv[3:]
    cj = json.load(open(os.path.join(ROOT, "profiles", tag.split("_")[0] + "_fp64_counters.json")))
    shas = {w: (cj["workloads"].get(w, {}).get("csrc_sha16") or cj.get("csrc_sha16")) for w in workloads}
    if len(set(shas.values())) != 1:
        sys.exit("the workloads were profiled on different kernel sources: %s" % shas)
    out = {"csrc_sha16": list(shas.values())[0], "workloads": {},
           "definition": "traffic_bytes_per_launch = (FETCH_SIZE x 2 + WRITE_SIZE) KB x 1024, averages over the dispatches of the kernel in its own --pmc pass "
                         "(tools/profile_round.sh); problems_per_launch = one half batch; algorithmic bytes: SURVEY 8d (232 B per problem per launch at N = 13)"}
    for w in workloads:
        src = "profiles/%s_%s_rocprofv3_kernel_trace_and_pmc.txt" % (tag, w)
        ctr, dur = parse(os.path.join(ROOT, src))
        rec = {"source": src, "problems_per_launch": PROBLEMS.get(w), "kernels": {}}
        for k, c in ctr.items():
            e = {"avg_us_kernel_trace": dur.get(k)}
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                e.update(fetch_size_kb_per_launch_raw=c["FETCH_SIZE"], write_size_kb_per_launch=c["WRITE_SIZE"],
                         traffic_bytes_per_launch=(2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0)
            if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
                e["sq_wait_any_over_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
            if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_over_idx_active"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64"):
                if n in c:
                    e[n] = c[n]
            rec["kernels"][k] = e
        out["workloads"][w] = rec
    json.dump(out, open(out_path, "w"), indent=1)
    for w, rec in out["workloads"].items():
        for k, e in rec["kernels"].items():
            print(w, k, e.get("avg_us_kernel_trace"), e.get("traffic_bytes_per_launch"), e.get("SQ_INSTS_MFMA"))


if __name__ == "__main__":
    main()
