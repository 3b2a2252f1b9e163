#!/bin/bash
# Runs on the GPU box (gpurun): bench lines + rocprofv3 kernel stats + HBM traffic PMC passes.
# Outputs under gpurun_out/final4/ ; tools/summarise_profiles_r04.py turns them into profiles/r04_*.
# Every profiled program is `python ...` itself after `--` (no wrapper that would re-exec).
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out/final4
rm -rf $O && mkdir -p $O
step() { echo "== $1"; }
step bench_default
timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json || exit 1
step bench_c2
timeout -k 10 300 python bench.py --config C2 --steps 30 --warmup 3 --cpu-budget 5 2>/dev/null | tail -1 > $O/bench_c2.json || exit 1
step bench_c4
timeout -k 10 300 python bench.py --config C4 --poses 64 --steps 5 --warmup 2 --cpu-budget 5 2>/dev/null | tail -1 > $O/bench_c4_64poses.json || exit 1
step bench_c5
timeout -k 10 300 python bench.py --config C5 --poses 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_8poses.json || exit 1
step bench_c4_coherent
timeout -k 10 300 python bench.py --config C4 --poses 64 --order hilbert --coherent --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_64poses_coherent.json || exit 1
ARGS="--no-cpu-baseline --no-secondary --no-scaling-reference"
cd /tmp && export TMPDIR=/tmp
step stats_c3
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $ROOT/bench.py --steps 30 --warmup 3 $ARGS > $O/stats.log 2>&1 || exit 1
grep '^{"metric"' $O/stats.log | tail -1 > $O/bench_under_rocprof.json
step stats_coherent
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_coherent -- python $ROOT/bench.py --steps 30 --warmup 3 $ARGS --order hilbert --coherent > $O/stats_coherent.log 2>&1 || exit 1
step stats_c4
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python $ROOT/bench.py --config C4 --poses 64 --steps 5 --warmup 2 $ARGS > $O/stats_c4.log 2>&1 || exit 1
step stats_c5
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python $ROOT/bench.py --config C5 --poses 8 --steps 3 --warmup 1 $ARGS > $O/stats_c5.log 2>&1 || exit 1
for mode in random coherent; do
  if [ $mode = coherent ]; then M="--order hilbert --coherent"; else M=""; fi
  step fetch_$mode
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$mode -- python $ROOT/bench.py --steps 3 --warmup 1 $ARGS $M > $O/fetch_$mode.log 2>&1 || exit 1
  step write_$mode
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$mode -- python $ROOT/bench.py --steps 3 --warmup 1 $ARGS $M > $O/write_$mode.log 2>&1 || exit 1
done
cd $ROOT
for cfg in c3 c3coh c4; do
  case $cfg in c3) M="";; c3coh) M="--order hilbert --coherent";; c4) M="--config C4 --poses 64";; esac
  step sq_$cfg
  SQ_OUT=final4/sq_$cfg bash tools/r04_sq_pmc.sh $M || exit 1
  step sqi_$cfg
  SQ_OUT=final4/sqi_$cfg bash tools/r04_sq_insts.sh $M || exit 1
done
cd $ROOT
step other_configs
timeout -k 10 600 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > $O/other_configs.txt
ls $O
echo collect done
