ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r03d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in 2 4; do
  export MPCMP_QP13=$q
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $OUT/a$q -o r -- python3 $ROOT/tools/qpbench.py 256 > $OUT/a$q.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM -d $OUT/b$q -o r -- python3 $ROOT/tools/qpbench.py 256 > $OUT/b$q.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 -d $OUT/c$q -o r -- python3 $ROOT/tools/qpbench.py 256 > $OUT/c$q.log 2>&1
done
python3 $ROOT/tools/prof_summary.py $(find $OUT -name '*_results.db' | sort) > $ROOT/gpurun_out/r03d_qp13_counters.txt 2>&1
find $OUT -name '*_results.db' -delete
grep "k_qp" $ROOT/gpurun_out/r03d_qp13_counters.txt | head -70
