mkdir -p gpurun_out/r03f
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x > gpurun_out/r03f/pytest.log 2>&1; tail -4 gpurun_out/r03f/pytest.log
timeout 300 python tools/qpbench.py 1 256 1024 2>&1 | grep QP13
timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 5 > gpurun_out/r03f/bench.json 2>gpurun_out/r03f/bench.err; python -c "import json; d=json.load(open('gpurun_out/r03f/bench.json')); print('batch', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['admm_iters_per_traj'])"
