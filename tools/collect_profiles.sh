#!/bin/bash
# Runs on the GPU box (gpurun): bench line + rocprofv3 kernel stats + HBM traffic PMC passes.
# Outputs under gpurun_out/final/ ; tools/summarise_profiles.py turns them into profiles/*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python bench.py --steps 30 --warmup 3 2>/dev/null | tail -1 > $O/bench_default.json
ARGS="--no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 30 --warmup 3 $ARGS > $O/stats.log 2>&1
tail -1 $O/stats.log > $O/bench_under_rocprof.json
for order in random morton; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$order -- python bench.py --steps 3 --warmup 1 $ARGS --order $order > $O/fetch_$order.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$order -- python bench.py --steps 3 --warmup 1 $ARGS --order $order > $O/write_$order.log 2>&1
done
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > $O/other_configs.txt
ls $O
