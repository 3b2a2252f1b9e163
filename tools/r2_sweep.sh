for rep in 1 2; do
for v in "" head; do
  if [ -n "$v" ]; then export DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$v.so; else unset DPR_LIB_OVERRIDE; fi
  echo "== [$v]"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['ms'], d['roofline']['pullback']['ms'], d['roofline']['stages'])"
done; done
