# usage: bash tools/r2_stats.sh <tag> <python script + args>   -- per-kernel averages of one run
O=$PWD/gpurun_out/$1; mkdir -p $O; ROOT=$PWD; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python "$@" > $O/stats.log 2>&1 || exit 1
cd $ROOT
python - <<PY
import csv,glob
for fn in glob.glob("$O/stats/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
