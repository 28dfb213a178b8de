mkdir -p gpurun_out/r03b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b/pytest_gpu.log 2>&1; tail -6 gpurun_out/r03b/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "F64\|INSTS_VALU_FMA\|INSTS_MFMA" | head -20 > $GRAFT_REPO_ROOT/gpurun_out/r03b/counters_avail.txt; head -20 $GRAFT_REPO_ROOT/gpurun_out/r03b/counters_avail.txt
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r03b/bench_default.json 2> gpurun_out/r03b/bench_default.err; tail -c 3000 gpurun_out/r03b/bench_default.json; tail -3 gpurun_out/r03b/bench_default.err
