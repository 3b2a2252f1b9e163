for v in "" noflush nolds nodirect; do
  if [ -n "$v" ]; then export DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$v.so; else unset DPR_LIB_OVERRIDE; fi
  echo "== [$v]"; python tools/chunkown_sweep.py 10000000,512,16 2>&1 | grep -v amdgpu
done
