mkdir -p gpurun_out/r03g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03g/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03g/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03g/bench_default.json 2> gpurun_out/r03g/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03g/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['canonical_frac'], d['quality']['status_ok_frac'])
print({k:(round(v['value'],1), round(v['roofline'].get('frac') or 0,3)) for k,v in d['secondary'].items()})
PY
