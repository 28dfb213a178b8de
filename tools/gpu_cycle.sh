mkdir -p gpurun_out/r02g
timeout 600 python -m pytest tests/test_gpu_multi_arm.py tests/test_gpu_parity.py -m gpu -q -k "not full_size and (multi or dual or n25 or solve_vs_oracle or qp3 or gold)" > gpurun_out/r02g/pytest.log 2>&1; tail -4 gpurun_out/r02g/pytest.log
timeout 300 python bench.py --workload shipped --no-cpu-baseline > gpurun_out/r02g/bench_shipped.json 2>gpurun_out/r02g/bench_shipped.err; python -c "import json; d=json.load(open('gpurun_out/r02g/bench_shipped.json')); print('shipped', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['admm_iters_per_traj'])"
timeout 600 python bench.py --workload dual14 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02g/bench_dual14.json 2> gpurun_out/r02g/bench_dual14.err; python -c "import json; d=json.load(open('gpurun_out/r02g/bench_dual14.json')); print('dual14', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['admm_iters_per_traj'])"
timeout 300 python tools/stamps3.py 256 6 1 > gpurun_out/r02g/stamps3_6_1.txt 2>&1; head -18 gpurun_out/r02g/stamps3_6_1.txt
timeout 300 python tools/stamps3.py 256 8 2 > gpurun_out/r02g/stamps3_8_2.txt 2>&1; head -18 gpurun_out/r02g/stamps3_8_2.txt
