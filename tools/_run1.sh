set -e
python -m pytest tests/test_parity_gpu.py -x -q -k "chunked or point_weight_gradient" > gpurun_out/t2.log 2>&1 || (tail -60 gpurun_out/t2.log; exit 1)
tail -2 gpurun_out/t2.log
python tools/own_probe.py --bwd > gpurun_out/p4.log 2>&1 || (tail -30 gpurun_out/p4.log; exit 1)
python3 - <<'PY'
import json
t=open('gpurun_out/p4.log').read()
d=json.loads(t[t.index('{'):t.rindex('}')+1])
print(d['chunked'].get('bwd_stages'))
print({k:v for k,v in d.items() if k.startswith('rel')})
PY
