#!/bin/bash
# Runs on the GPU box (gpurun): bench lines + rocprofv3 kernel stats + HBM traffic PMC passes of round 6.
# Outputs under gpurun_out/final6/ ; tools/summarise_profiles_r06.py turns them into profiles/r06_*.
# Every profiled program is `python ...` itself after `--` (no wrapper that would re-exec).
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out/final6
PART=${1:-all}   # "a" = bench lines + kernel stats, "b" = counter passes + the rest (two gpurun calls)
mkdir -p $O
step() { echo "== $1 $(date +%T)"; }
ARGS="--no-cpu-baseline --no-secondary --no-scaling-reference"
if [ $PART != b ]; then
step bench_default
timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json || exit 1
step bench_c2
timeout -k 10 300 python bench.py --config C2 --steps 30 --warmup 3 --cpu-budget 5 2>/dev/null | tail -1 > $O/bench_c2.json || exit 1
step bench_c4
timeout -k 10 300 python bench.py --config C4 --poses 64 --steps 5 --warmup 2 --cpu-budget 5 2>/dev/null | tail -1 > $O/bench_c4_64poses.json || exit 1
step bench_c5
timeout -k 10 300 python bench.py --config C5 --poses 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_8poses.json || exit 1
step bench_c5_coherent
timeout -k 10 300 python bench.py --config C5 --poses 8 --order hilbert --coherent --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_8poses_coherent.json || exit 1
step bench_c5_full
timeout -k 10 600 python bench.py --config C5 --poses 64 --steps 2 --warmup 1 --cpu-budget 15 2>/dev/null | tail -1 > $O/bench_c5_full.json || exit 1
step bench_c3_coherent_auto
timeout -k 10 300 python bench.py --order hilbert --coherent --steps 30 --warmup 5 --no-cpu-baseline --no-scaling-reference 2>/dev/null | tail -1 > $O/bench_c3_coherent_auto.json || exit 1
step bench_c3_coherent_chunked
timeout -k 10 300 python bench.py --order hilbert --coherent --algo chunked --steps 30 --warmup 5 --no-cpu-baseline --no-scaling-reference 2>/dev/null | tail -1 > $O/bench_c3_coherent_chunked.json || exit 1
cd /tmp && export TMPDIR=/tmp
step stats_c3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $ROOT/bench.py --steps 30 --warmup 3 $ARGS > $O/stats.log 2>&1 || exit 1
grep '^{"metric"' $O/stats.log | tail -1 > $O/bench_under_rocprof.json
step stats_coherent_auto
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_coh_auto -- python $ROOT/bench.py --steps 30 --warmup 3 $ARGS --order hilbert --coherent > $O/stats_coh_auto.log 2>&1 || exit 1
step stats_coherent_chunked
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_coh_chunked -- python $ROOT/bench.py --steps 30 --warmup 3 $ARGS --order hilbert --coherent --algo chunked > $O/stats_coh_chunked.log 2>&1 || exit 1
step stats_c4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python $ROOT/bench.py --config C4 --poses 64 --steps 5 --warmup 2 $ARGS > $O/stats_c4.log 2>&1 || exit 1
step stats_c5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python $ROOT/bench.py --config C5 --poses 8 --steps 3 --warmup 1 $ARGS > $O/stats_c5.log 2>&1 || exit 1
fi
if [ $PART = a ]; then echo collect part a done; exit 0; fi
cd /tmp && export TMPDIR=/tmp
for mode in random coh_auto coh_chunked; do
  case $mode in random) M="";; coh_auto) M="--order hilbert --coherent";; coh_chunked) M="--order hilbert --coherent --algo chunked";; esac
  step fetch_$mode
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$mode -- python $ROOT/bench.py --steps 3 --warmup 1 $ARGS $M > $O/fetch_$mode.log 2>&1 || exit 1
  step write_$mode
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$mode -- python $ROOT/bench.py --steps 3 --warmup 1 $ARGS $M > $O/write_$mode.log 2>&1 || exit 1
done
for mode in random coh_chunked; do
  case $mode in random) M="";; coh_chunked) M="--order hilbert --coherent --algo chunked";; esac
  step sq_$mode
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq_$mode -- python3 $ROOT/bench.py --steps 3 --warmup 1 $ARGS $M > $O/sq_$mode.log 2>&1 || exit 1
  step sqi_$mode
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/sqi_$mode -- python3 $ROOT/bench.py --steps 3 --warmup 1 $ARGS $M > $O/sqi_$mode.log 2>&1 || exit 1
done
step fetch_c5
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5 -- python $ROOT/bench.py --config C5 --poses 8 --steps 2 --warmup 1 $ARGS > $O/fetch_c5.log 2>&1 || exit 1
step write_c5
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5 -- python $ROOT/bench.py --config C5 --poses 8 --steps 2 --warmup 1 $ARGS > $O/write_c5.log 2>&1 || exit 1
cd $ROOT
step ablations
P=$ROOT/diffpointrasterisation.jl_amd
(python tools/c3_stage_probe.py --tag shipped
 for n in 1 2 3; do [ -f $P/libdpr_abl$n.so ] && DPR_LIB_OVERRIDE=$P/libdpr_abl$n.so python tools/c3_stage_probe.py --tag tile_splat_ablation_$n; done
 python tools/c3_stage_probe.py --coherent --tag shipped_coherent_tiled
 python tools/c3_stage_probe.py --coherent --algo chunked --tag shipped_coherent_chunked
) 2>/dev/null > $O/ablations.jsonl
step microbench
[ -x tools/microbench_valu ] && ./tools/microbench_valu > $O/microbench_valu.txt 2>&1
[ -x tools/microbench_write ] && ./tools/microbench_write > $O/microbench_write.txt 2>&1
step readme_timings
timeout -k 10 300 python tools/readme_timings.py 2>&1 | grep -v amdgpu.ids > $O/readme_timings.txt
step other_configs
timeout -k 10 600 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > $O/other_configs.txt
ls $O
echo collect done
