mkdir -p gpurun_out/r2b; rm -f gpurun_out/r2b/sweep2.txt
run() { echo "== $1 blocked=$2" >> gpurun_out/r2b/sweep2.txt
  if [ -n "$1" ]; then export DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$1.so; else unset DPR_LIB_OVERRIDE; fi
  for o in random morton; do DPR_SPLAT_BLOCKED=$2 python tools/stage_probe.py --P 10000000 --grid 256 256 256 --order $o 2>&1 | grep fwd >> gpurun_out/r2b/sweep2.txt || exit 1; done; }
run "" 1; run "" 0; run pf1 1; run pf4 1; run pf8 1; run pf4 0; run pf8 0
cat gpurun_out/r2b/sweep2.txt
