set -e
python -m pytest tests/test_owner_gpu.py tests/test_parity_gpu.py -x -q -k "owner or chunked or point_weight_gradient or poses_than" > gpurun_out/t3.log 2>&1 || (tail -60 gpurun_out/t3.log; exit 1)
tail -2 gpurun_out/t3.log
for cfg in "1000000 256 16" "1000000 256 64" "10000000 256 4" "10000000 256 16" "300000 128 16"; do
  set -- $cfg
  PROBE_ATOMIC=1 python tools/own_probe.py --P $1 --grid $2 --poses $3 --bwd --reps 3 > gpurun_out/p9.log 2>&1 || (tail -30 gpurun_out/p9.log; exit 1)
  python3 - <<PY
import json
t=open('gpurun_out/p9.log').read()
d=json.loads(t[t.index('{'):t.rindex('}')+1])
print("$cfg", 'bwd: chunked', round(d['chunked']['bwd_ms'],4), 'tiled', round(d['tiled']['bwd_ms'],4), 'atomic', round(d['atomic']['bwd_ms'],4), {k:v for k,v in d.items() if k.startswith('rel_') and k!='rel_l2_fwd'})
PY
done
python tools/own_probe.py --P 50000000 --grid 512 --f64 --poses 8 --bwd --reps 3 > gpurun_out/p8.log 2>&1 || (tail -30 gpurun_out/p8.log; exit 1)
python3 - <<'PY'
import json
t=open('gpurun_out/p8.log').read()
d=json.loads(t[t.index('{'):t.rindex('}')+1])
for k in ('chunked','tiled'):
    print('C5-size sorted', k, d[k])
print({k:v for k,v in d.items() if k.startswith('rel')})
PY
