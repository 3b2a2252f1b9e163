set -e
python -m pytest tests/test_parity_gpu.py -x -q -k "chunked or point_weight_gradient" > gpurun_out/t2.log 2>&1 || (tail -60 gpurun_out/t2.log; exit 1)
tail -3 gpurun_out/t2.log
DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/libdpr_stats.so python tools/own_probe.py > gpurun_out/p3.log 2>&1 || (tail -30 gpurun_out/p3.log; exit 1)
python tools/own_probe.py --bwd > gpurun_out/p4.log 2>&1 || (tail -30 gpurun_out/p4.log; exit 1)
python3 - <<'PY'
import json
t=open('gpurun_out/p3.log').read()
d=json.loads(t[t.index('{'):t.rindex('}')+1])
print({k:v for k,v in d['stats'].items() if k in ('visits_per_point','touch_per_point','lane_util','items_with_work','list_entries','slabs','buckets')})
print({k:v for k,v in d['items'].items() if k!='top'})
t=open('gpurun_out/p4.log').read()
d=json.loads(t[t.index('{'):t.rindex('}')+1])
for k in ('chunked','tiled'):
    print(k, {a:b for a,b in d[k].items() if 'stages' not in a}, d[k].get('stages'), d[k].get('bwd_stages'))
print({k:v for k,v in d.items() if k.startswith('rel') or k.startswith('max') or k.startswith('bit')})
PY
