mkdir -p gpurun_out/r03e
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi_arm.py -m gpu -q -x -k "qp3 or solve_vs_oracle or multi or dual or gold" > gpurun_out/r03e/pytest.log 2>&1; tail -4 gpurun_out/r03e/pytest.log
timeout 300 python bench.py --workload shipped --no-cpu-baseline > gpurun_out/r03e/bench_shipped.json 2>gpurun_out/r03e/bench_shipped.err; python -c "import json; d=json.load(open('gpurun_out/r03e/bench_shipped.json')); print('shipped', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['admm_iters_per_traj'])"
timeout 300 python tools/stamps3.py 256 6 1 > gpurun_out/r03e/stamps3_6_1.txt 2>&1; grep -v "load-store" gpurun_out/r03e/stamps3_6_1.txt | head -18
