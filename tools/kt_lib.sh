#!/bin/bash
# kernel-trace of the batch workload (single stream: exclusive kernel times) with another build of the library: kt_lib.sh tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "$@"; do
  rm -rf $R/gpurun_out/kt_$tag
  MPCMP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$tag -o r -- python3 $R/tools/bench_with_lib.py $R/tools/micro/libht_$tag.bin --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/kt_$tag.log 2>&1
  echo "== $tag"; python3 $R/tools/prof_summary.py $(find $R/gpurun_out/kt_$tag -name "*_results.db" | head -1) 2>&1 | grep -E "^k_step|^k_qp2|^k_init" | cut -c1-100
  find $R/gpurun_out/kt_$tag -name "*_results.db" -delete
done
