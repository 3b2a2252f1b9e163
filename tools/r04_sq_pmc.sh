#!/bin/bash
# SQ counters (one pass of 8) of the C3 step's kernels: rocprofv3 --pmc ... -- python3 bench.py
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out/${SQ_OUT:-sq4}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-scaling-reference "$@" > $O/sq.log 2>&1 || exit 1
echo done
