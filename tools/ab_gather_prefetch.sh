#!/bin/bash
# A/B of k_tile_gather's record prefetch depth (DPR_GATHER_PF builds), interleaved twice on one box
cd "${GRAFT_REPO_ROOT:-.}"
L=$PWD/diffpointrasterisation.jl_amd
for pass in 1 2; do
  for lib in libdpr.so libdpr_gpf2.so libdpr_gpf3.so; do
    [ -f $L/$lib ] || continue
    DPR_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/c3_stage_probe.py --reps 30 2>/dev/null | tail -1
  done
done
