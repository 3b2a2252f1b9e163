O=$PWD/gpurun_out/$1; mkdir -p $O; ROOT=$PWD; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $ROOT/tools/stage_probe.py "$@" > $O/trace.log 2>&1 || exit 1
cd $ROOT
python - <<PY
import csv,glob
rows=[]
for fn in glob.glob("$O/trace/**/*kernel_trace.csv",recursive=True):
    rows+=list(csv.DictReader(open(fn)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find the last forward sequence: k_count ... k_halo_gather
idx=[i for i,r in enumerate(rows) if "k_halo_gather" in r["Kernel_Name"]]
i1=idx[len(idx)//2]; i0=i1
while "k_count" not in rows[i0]["Kernel_Name"]: i0-=1
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1+2]:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
    print(f"{r['Kernel_Name'].split('(')[0][:44]:44s} start {s/1e3:8.1f} us  dur {(e-s)/1e3:7.1f} us  end {e/1e3:8.1f}  grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')} lds {r.get('LDS_Block_Size','?')} vgpr {r.get('VGPR_Count','?')}")
PY
