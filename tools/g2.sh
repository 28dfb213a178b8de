mkdir -p gpurun_out/r03a
MPCMP_QP13=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "qp_full or solve_vs_oracle or headline" > gpurun_out/r03a/pytest_qp13_4.log 2>&1; tail -3 gpurun_out/r03a/pytest_qp13_4.log
MPCMP_QP13=4 timeout 300 python tools/qpbench.py 1 256 512 1024 2>&1 | grep QP13
