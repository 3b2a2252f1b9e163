# PMC passes over one forward (C3) -- usage: bash tools/r2_pmc.sh <outdir> [order]
O=$PWD/gpurun_out/$1; ORDER=${2:-random}; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python $ROOT/tools/stage_probe.py --P 10000000 --grid 256 256 256 --order $ORDER > $O/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc2 -- python $ROOT/tools/stage_probe.py --P 10000000 --grid 256 256 256 --order $ORDER > $O/pmc2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc3 -- python $ROOT/tools/stage_probe.py --P 10000000 --grid 256 256 256 --order $ORDER > $O/pmc3.log 2>&1 || exit 1
cd $ROOT
python - <<PY
import csv,glob,collections
for p in ("pmc1","pmc2","pmc3"):
    f=glob.glob("$O/%s/**/*counter_collection.csv"%p,recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k in acc:
        # count dispatches = distinct Dispatch_Id
        pass
    ids=collections.defaultdict(set)
    for fn in f:
        for r in csv.DictReader(open(fn)):
            ids[r["Kernel_Name"].split("(")[0][:40]].add(r["Dispatch_Id"])
    for k in sorted(acc):
        if "dpr" not in k: continue
        nd=len(ids[k])
        print(p,k,"disp",nd," ".join("%s=%.3g"%(c,v/nd) for c,v in sorted(acc[k].items())))
PY
