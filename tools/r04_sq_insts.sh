#!/bin/bash
# executed-instruction counters (one pass) of a bench command: rocprofv3 --pmc ... -- python3 bench.py
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out/${SQ_OUT:-sqi}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/sq -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-scaling-reference "$@" > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
echo done
