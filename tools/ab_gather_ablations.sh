#!/bin/bash
# k_tile_gather ablation builds (make ../libdpr_abl4.so ../libdpr_abl5.so ../libdpr_abl6.so: timing only, wrong results)
cd "${GRAFT_REPO_ROOT:-.}"
L=$PWD/diffpointrasterisation.jl_amd
for pass in 1 2; do
  for lib in libdpr.so libdpr_abl4.so libdpr_abl5.so libdpr_abl6.so; do
    [ -f $L/$lib ] || continue
    DPR_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/c3_stage_probe.py --reps 30 2>/dev/null | tail -1
  done
done
