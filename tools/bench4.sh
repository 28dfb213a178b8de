mkdir -p gpurun_out/r02g
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/pytest_full.log 2>&1; tail -2 gpurun_out/r02g/pytest_full.log
for w in batch shipped rh; do timeout 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r02g/b_$w.json 2> gpurun_out/r02g/b_$w.err; python -c "import json; d=json.load(open('gpurun_out/r02g/b_$w.json')); print('$w', d['value'])"; done
timeout 600 python bench.py --workload dual14 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02g/b_dual14.json 2> gpurun_out/r02g/b_dual14.err; python -c "import json; d=json.load(open('gpurun_out/r02g/b_dual14.json')); print('dual14', d['value'])"
