set -e
python -m pytest tests/test_parity_gpu.py -x -q -k "chunked or point_weight_gradient" > gpurun_out/t2.log 2>&1 || (tail -60 gpurun_out/t2.log; exit 1)
tail -3 gpurun_out/t2.log
python tools/own_probe.py --bwd > gpurun_out/p4.log 2>&1 || (tail -30 gpurun_out/p4.log; exit 1)
python3 - <<'PY'
import json
t=open('gpurun_out/p4.log').read()
d=json.loads(t[t.index('{'):t.rindex('}')+1])
for k in ('chunked','tiled'):
    print(k, {a:b for a,b in d[k].items() if 'stages' not in a}, d[k].get('stages'), d[k].get('bwd_stages'))
print({k:v for k,v in d.items() if k.startswith('rel') or k.startswith('max') or k.startswith('bit')})
PY
