mkdir -p gpurun_out/r03a
MPCMP_QP13=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "solve_vs_oracle or qp_full or headline" > gpurun_out/r03a/pytest_qp13_3.log 2>&1; tail -5 gpurun_out/r03a/pytest_qp13_3.log
MPCMP_QP13=2 timeout 300 python tools/qpbench.py 256 512 1024 > gpurun_out/r03a/qpb2.txt 2>&1; cat gpurun_out/r03a/qpb2.txt
MPCMP_QP13=3 timeout 300 python tools/qpbench.py 256 512 1024 > gpurun_out/r03a/qpb3.txt 2>&1; cat gpurun_out/r03a/qpb3.txt
MPCMP_QP13=3 timeout 300 python tools/stamps3.py 256 4 1 > gpurun_out/r03a/stamps3_4_1.txt 2>&1; head -30 gpurun_out/r03a/stamps3_4_1.txt
