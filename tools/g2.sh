mkdir -p gpurun_out/r03a
MPCMP_QP13=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "qp_full or solve_vs_oracle or headline" > gpurun_out/r03a/pytest_qp13_4.log 2>&1; tail -15 gpurun_out/r03a/pytest_qp13_4.log
MPCMP_QP13=4 timeout 300 python tools/qpbench.py 256 512 1024 > gpurun_out/r03a/qpb4.txt 2>&1; cat gpurun_out/r03a/qpb4.txt
