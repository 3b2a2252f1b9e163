#!/bin/bash
# A/B of one build-time macro of one translation unit, same box:
#   tools/ab_build.sh FILE.hip "FLAGS_A|FLAGS_B|..." OUTDIR -- bench.py args
# (the first variant should be the committed build: its objects travel with the snapshot)
F=$1; IFS='|' read -ra VARS <<< "$2"; OUT=gpurun_out/$3; shift 4
mkdir -p $OUT
C=diffpointrasterisation.jl_amd/csrc
i=0
for fl in "${VARS[@]}"; do
  ( cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $fl -c $F -o ${F%.hip}.o && make ../libdpr.so >/dev/null ) || exit 1
  if [ -n "$AB_SCRIPT" ]; then  # any other timing script instead of bench.py
    echo "[$fl]"; timeout -k 10 300 python3 $AB_SCRIPT "$@" 2> $OUT/v$i.err || exit 1
    i=$((i+1)); continue
  fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-scaling-reference "$@" > $OUT/v$i.json 2> $OUT/v$i.err || exit 1
  python3 - $OUT/v$i.json "$fl" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("[%s]" % sys.argv[2], "ms_per_step", d["ms_per_step"], "fwd_ms", r.get("ms"), "pullback_ms", (r.get("pullback") or {}).get("ms"))
PY
  i=$((i+1))
done
