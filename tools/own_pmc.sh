#!/bin/bash
# one rocprofv3 --pmc pass over tools/own_probe.py; usage: PMC="C1 C2 .." OUT=name bash tools/own_pmc.sh [probe args]
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out/${OUT:-ownpmc}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 90 rocprofv3 --pmc $PMC --output-format csv -d $O/pmc -- python3 $ROOT/tools/own_probe.py --reps 5 "$@" > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
cd $ROOT
python3 - $O <<'PY'
import collections, csv, glob, os, sys
O = sys.argv[1]
f = glob.glob(os.path.join(O, "pmc", "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")
    if k.startswith("dpr::k_own") or k.startswith("dpr::k_tile") or k.startswith("dpr::k_bin") or k.startswith("dpr::k_unperm"):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()})
PY
