#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02_v1 ["--workload shipped"]'
# Passes (each its own process, MI355X_MICROARCH.md HBM/rocprofv3 section): kernel trace + stats, then one --pmc pass
# per counter group.  Outputs land in gpurun_out/<tag>/ and a text summary in gpurun_out/<tag>_summary.txt; copy the
# summary into profiles/.
set -u
TAG=${1:-r02}
BENCH_ARGS=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" $BENCH_ARGS --no-secondary --steps 5 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o r -- python3 "$ROOT/bench.py" $BENCH_ARGS --no-secondary --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/kt.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"; do
    name=$(echo "$grp" | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$name" -o r -- python3 "$ROOT/bench.py" $BENCH_ARGS --no-secondary --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1
done
python3 "$ROOT/tools/prof_summary.py" $(find "$OUT" -name '*_results.db' | sort) > "$ROOT/gpurun_out/${TAG}_summary.txt" 2>&1
# executed FP64 flops per ADMM iteration of one trajectory, from the FMA/ADD/MUL pass (+ the MFMA pass): one JSON object per profiled workload
WL=$(echo "$BENCH_ARGS" | sed -n 's/.*--workload \([a-z0-9]*\).*/\1/p'); WL=${WL:-batch}
python3 "$ROOT/tools/fp64_counters.py" "$WL" "$(find "$OUT/pmc_SQ_INSTS_VALU_FMA_F64" -name '*_results.db' | head -1)" "$OUT/pmc_SQ_INSTS_VALU_FMA_F64.log" \
    "$(find "$OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES" -name '*_results.db' | head -1)" > "$ROOT/gpurun_out/${TAG}_fp64_counters.json" 2> "$OUT/fp64_counters.err"
# the raw rocprofv3 databases are tens of MB each; gpurun merges at most 64 MiB back: keep the summary, the bench line and the logs
find "$OUT" -name '*_results.db' -delete
find "$OUT" -type d -empty -delete
tail -3 "$OUT/bench.json"
head -8 "$ROOT/gpurun_out/${TAG}_summary.txt"
