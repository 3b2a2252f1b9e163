O=$PWD/gpurun_out/$1; mkdir -p $O; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python $ROOT/tools/co_probe.py 10000000 512 16 > $O/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc2 -- python $ROOT/tools/co_probe.py 10000000 512 16 > $O/pmc2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $ROOT/tools/co_probe.py 10000000 512 16 > $O/stats.log 2>&1 || exit 1
cd $ROOT
python - <<PY
import csv,glob,collections
for p in ("pmc1","pmc2"):
    f=glob.glob("$O/%s/**/*counter_collection.csv"%p,recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); ids=collections.defaultdict(set)
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); ids[k].add(r["Dispatch_Id"])
    for k in sorted(acc):
        if "k_co" not in k: continue
        nd=len(ids[k]); print(p,k,"disp",nd," ".join("%s=%.3g"%(c,v/nd) for c,v in sorted(acc[k].items())))
for fn in glob.glob("$O/stats/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        if "dpr" in r["Name"] or "rocprim" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
