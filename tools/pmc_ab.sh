#!/bin/bash
# instruction / wait counters of the N=13 QP kernel for two builds of the library (fixed 700 iterations, 1024 problems)
# usage (through gpurun, from the repo root): tools/pmc_ab.sh <libA> <libB>
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_ab; rm -rf $OUT; mkdir -p $OUT
i=0
for lib in "$@"; do
  export QPB_LIB=$ROOT/$lib
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    n=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $grp -d $OUT/l${i}_$n -o r -- python3 $ROOT/tools/qpbench.py 1024 > $OUT/l${i}_$n.log 2>&1
  done
  i=$((i+1))
done
python3 - <<PY
import sqlite3, glob, os
for i in range($i):
    print("== lib", i)
    for db in sorted(glob.glob("$OUT/l%d_*/**/*_results.db" % i, recursive=True)):
        c = sqlite3.connect(db)
        try:
            for kn, cn, n, av in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1, 2"):
                if "k_qp2" in kn: print("  %-24s %16.0f per dispatch (%d dispatches)" % (cn, av, n))
        except Exception as e:
            print("  query failed on", db, e)
PY
