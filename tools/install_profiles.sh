#!/bin/bash
# After `gpurun -- 'bash tools/profile_round.sh <tag>_batch; ... <tag>_shipped "--workload shipped"; ... <tag>_dual14 "--workload dual14"'`:
# copy the summaries and bench lines into profiles/, rebuild the counter and traffic records (stamped with the kernel-source hash the passes ran on),
# drop the previous tag's files.   usage: tools/install_profiles.sh r04_v4 [r04_v3]
set -e
TAG=$1; OLD=${2:-}
cd "$(dirname "$0")/.."
for w in batch shipped dual14; do
    cp gpurun_out/${TAG}_${w}_summary.txt profiles/${TAG}_${w}_rocprofv3_kernel_trace_and_pmc.txt
    cp gpurun_out/${TAG}_$w/bench.json profiles/${TAG}_${w}_bench.json
done
R=${TAG%%_*}
python3 tools/merge_counters.py profiles/${R}_fp64_counters.json gpurun_out/${TAG}_batch_fp64_counters.json gpurun_out/${TAG}_shipped_fp64_counters.json gpurun_out/${TAG}_dual14_fp64_counters.json
python3 tools/traffic_from_summary.py profiles/${R}_traffic.json ${TAG} batch shipped dual14
if [ -n "$OLD" ]; then rm -f profiles/${OLD}_*; sed -i "s/profiles\/${OLD}_/profiles\/${TAG}_/g" DESIGN.md; fi
