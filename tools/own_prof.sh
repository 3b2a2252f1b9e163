#!/bin/bash
# rocprofv3 passes over tools/own_probe.py (the owner-computes 3-D path): kernel trace + two SQ counter
# passes; summaries under gpurun_out/$OUT.   usage: bash tools/own_prof.sh [probe args]
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out/${OUT:-ownprof}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $ROOT/tools/own_probe.py --reps 5 "$@" > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq1 -- python3 $ROOT/tools/own_probe.py --reps 5 "$@" > $O/sq1.log 2>&1 || { tail -5 $O/sq1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $O/sq2 -- python3 $ROOT/tools/own_probe.py --reps 5 "$@" > $O/sq2.log 2>&1 || { tail -5 $O/sq2.log; exit 1; }
cd $ROOT
python3 tools/own_prof_summary.py $O > $O/summary.txt
cat $O/summary.txt
