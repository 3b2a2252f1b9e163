#!/bin/bash
# A/B over values of one environment knob, same box, same binary:
#   tools/ab_env.sh VAR "v1 v2 ..." OUTDIR -- bench.py args
# prints raster_ms / pullback_ms / ms_per_step of each line
VAR=$1; VALS=$2; OUT=gpurun_out/$3; shift 4
mkdir -p $OUT
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-scaling-reference "$@" > $OUT/$VAR.$v.json 2> $OUT/$VAR.$v.err || exit 1
  python3 - $OUT/$VAR.$v.json $VAR $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(sys.argv[2], sys.argv[3], "ms_per_step", d["ms_per_step"], "fwd_ms", r.get("ms"), "stages", {k: round(v, 1) for k, v in (r.get("stage_us") or {}).items()} if isinstance(r.get("stage_us"), dict) else "")
PY
done
