for v in noop6 noop7; do
  export DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$v.so
  echo "== [$v]"; python tools/stage_probe.py --P 1000000 --grid 128 128 128 2>&1 | grep fwd; python tools/stage_probe.py --P 10000000 --grid 256 256 256 2>&1 | grep fwd
done
