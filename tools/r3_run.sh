#!/bin/bash
# usage: bash tools/r3_run.sh <tag> <step> [<step> ...]   -- steps: tests bench_c3 bench_c3_<variant> bench_c4 bench_c5 regret ...
# Every step runs under its own timeout; a step that is killed at its limit ends the script
# (no further GPU step after a hang).  Outputs under gpurun_out/<tag>/.
set -o pipefail
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
run() {  # run <name> <limit_s> <cmd...>
  local name=$1 lim=$2; shift 2
  echo "== $name"
  timeout -k 10 $lim "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/steps.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; exit 1; fi
}
for step in "$@"; do
  case $step in
    tests)      run tests 900 python -m pytest tests -m gpu -x -q ;;
    tests_all)  run tests 900 python -m pytest tests -m gpu -q ;;
    tests_k)    run tests_k 600 python -m pytest tests -m gpu -q -k "$K" ;;
    tests_kv)   DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$VARIANT.so run tests_kv 600 python -m pytest tests -m gpu -q -k "$K" ;;
    fuzz_share_v) DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$VARIANT.so run fuzz_share_v 900 python tools/fuzz_share.py 60 ;;
    bench_c3)   run bench_c3 300 python bench.py --steps 20 --warmup 5 --cpu-budget 5 ;;
    bench_c2_cap*) v=${step#bench_c2_cap}; DPR_CAP_MIN=$v run bench_c2_cap$v 300 python bench.py --config C2 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary ;;
    fwdonly_*)  v=${step#fwdonly_}; DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$v.so run fwdonly_$v 300 python tools/time_call.py 1e7,256x256x256,1,fwd,tiled 1e7,256x256x256,1,fwd,tiled 1e6,128x128x128,1,fwd,tiled 5e6,256x256x256,1,fwd,tiled ;;
    fwdonly)    run fwdonly 300 python tools/time_call.py 1e7,256x256x256,1,fwd,tiled 1e7,256x256x256,1,fwd,tiled 1e6,128x128x128,1,fwd,tiled 5e6,256x256x256,1,fwd,tiled ;;
    bench_c3_unblocked) DPR_SPLAT_BLOCKED=0 run bench_c3_unblocked 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-reference ;;
    bench_c3_*) v=${step#bench_c3_}; DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$v.so run bench_c3_$v 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-reference ;;
    bench_c4_*) v=${step#bench_c4_}; DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$v.so run bench_c4_$v 300 python bench.py --config C4 --poses 64 --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_c2)   run bench_c2 300 python bench.py --config C2 --steps 30 --warmup 3 --no-cpu-baseline ;;
    bench_c4)   run bench_c4 300 python bench.py --config C4 --poses 64 --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_c5)   run bench_c5 300 python bench.py --config C5 --poses 8 --steps 3 --warmup 1 --no-cpu-baseline ;;
    bench_c5_64) run bench_c5_64 900 python bench.py --config C5 --poses 64 --steps 2 --warmup 1 --no-cpu-baseline ;;
    bench_c5_noshare) run bench_c5_noshare 300 python bench.py --config C5 --poses 8 --steps 3 --warmup 1 --no-cpu-baseline --no-share-binning ;;
    rehearse2)  DPR_BENCH_BACKEND=gloo run rehearse2 600 python bench.py --gpus 2 --poses 8 --steps 2 --warmup 1 ;;
    rehearse2_c3) DPR_BENCH_BACKEND=gloo run rehearse2_c3 600 python bench.py --gpus 2 --config C3 --shard points --steps 2 --warmup 1 ;;
    torchrun2)  DPR_BENCH_BACKEND=gloo run torchrun2 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --poses 8 --steps 2 --warmup 1 ;;
    fuzz_big)   run fuzz_big 900 python tools/fuzz_big.py 60 ;;
    fuzz_share) run fuzz_share 900 python tools/fuzz_share.py 60 ;;
    fuzz_co)    run fuzz_co 900 python tools/fuzz_chunkown.py 600 ;;
    fuzz_more)  run fuzz_more 900 python tools/fuzz_more.py ;;
    timecall_cap*) v=${step#timecall_cap}; DPR_CAP_MIN=$v run timecall_cap$v 600 python tools/time_call.py $SPECS ;;
    timecall)   run timecall 600 python tools/time_call.py $SPECS ;;
    timecall_v) DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/variants_libdpr_$VARIANT.so run timecall_v 600 python tools/time_call.py $SPECS ;;
    regret)     run regret 1000 python tools/auto_regret.py ;;
    regret_quick) run regret 600 python tools/auto_regret.py --quick ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
echo all steps done
