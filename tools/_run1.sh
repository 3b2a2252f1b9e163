set -e
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "pre_canon or mixed_dtypes or slabs or 1024 or 2d_grid" > gpurun_out/tall.log 2>&1 || (tail -40 gpurun_out/tall.log; exit 1)
tail -3 gpurun_out/tall.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || (tail -20 gpurun_out/bench_default.err; exit 1)
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_loops')})
print(d['roofline']['ms'], d['roofline']['frac'], d['roofline'].get('pullback'))
print(d['coherent_input'])
print(d['no_share'], d['config'].get('drop_in_ms_per_step'))
print(d['cpu_baseline'])
PY
