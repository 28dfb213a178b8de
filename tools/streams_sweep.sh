#!/bin/bash
# diagnostic: throughput versus the number of stream parts a batch is split into (MPCMP_STREAMS)
for s in 1 2 3 4; do
  v=$(MPCMP_STREAMS=$s timeout 300 python bench.py --workload shipped --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
  echo "shipped streams $s: $v"
done
for s in 1 2 4; do
  v=$(MPCMP_STREAMS=$s timeout 600 python bench.py --workload dual14 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
  echo "dual14 streams $s: $v"
done
