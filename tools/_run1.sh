set -e
for cfg in "1000000 256 16" "1000000 256 64" "10000000 256 4" "10000000 256 16" "3000000 256 8" "300000 128 16"; do
  set -- $cfg
  PROBE_ATOMIC=1 python tools/own_probe.py --P $1 --grid $2 --poses $3 --bwd --reps 3 > gpurun_out/p9.log 2>&1 || (tail -30 gpurun_out/p9.log; exit 1)
  python3 - <<PY
import json
t=open('gpurun_out/p9.log').read()
d=json.loads(t[t.index('{'):t.rindex('}')+1])
print("$cfg", 'bwd: chunked', round(d['chunked']['bwd_ms'],4), 'tiled', round(d['tiled']['bwd_ms'],4), 'atomic', round(d['atomic']['bwd_ms'],4))
PY
done
