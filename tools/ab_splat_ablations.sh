#!/bin/bash
# k_tile_splat ablation builds (make ../libdpr_abl{1,2,3,7}.so: timing only, wrong results), interleaved twice
cd "${GRAFT_REPO_ROOT:-.}"
L=$PWD/diffpointrasterisation.jl_amd
for pass in 1 2; do
  for lib in libdpr.so libdpr_abl1.so libdpr_abl2.so libdpr_abl7.so libdpr_abl3.so; do
    [ -f $L/$lib ] || continue
    DPR_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/c3_stage_probe.py --reps 30 2>/dev/null | tail -1
  done
done
