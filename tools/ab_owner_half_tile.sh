set -e
cd $GRAFT_REPO_ROOT
L=diffpointrasterisation.jl_amd
for lib in libdpr.so libdpr_half.so; do
  for args in "--poses 1" "--poses 8" "--poses 1 --dist uniform" "--poses 8 --dist uniform" "--poses 4 --dist tight" "--poses 8 --P 3000000" "--poses 2 --f64"; do
    echo "== $lib $args"
    DPR_LIB_OVERRIDE=$PWD/$L/$lib timeout -k 10 120 python tools/own_probe.py $args --reps 10 | python -c "
import json,sys
r=json.load(sys.stdin)
print(json.dumps({'chunked':r['chunked'],'tiled_fwd_ms':r['tiled']['fwd_ms'],'rel_l2':r['rel_l2_fwd'],'bit_equal':r['bit_equal_fwd']}))"
  done
done
for cd in 512 1024 2048; do
  echo "== half cap_div $cd"
  DPR_OWN_CAP_DIV=$cd DPR_LIB_OVERRIDE=$PWD/$L/libdpr_half.so timeout -k 10 120 python tools/own_probe.py --poses 1 --reps 10 | python -c "
import json,sys
r=json.load(sys.stdin); print(json.dumps(r['chunked']))"
done
