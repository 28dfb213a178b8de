cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02g
for l in mpc_motion_planner_amd/libmpcmp.so tools/micro/libv_ilp.bin; do
  n=$(basename $l)
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02g/kt_$n -o r -- python3 $R/tools/bench_with_lib.py $R/$l --workload dual14 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02g/kt_$n.log 2>&1
  python3 $R/tools/prof_summary.py $(find $R/gpurun_out/r02g/kt_$n -name '*_results.db') | head -8
  find $R/gpurun_out/r02g/kt_$n -name '*_results.db' -delete
done
