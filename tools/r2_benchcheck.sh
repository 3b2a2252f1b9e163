mkdir -p gpurun_out/r2d; O=gpurun_out/r2d
python bench.py --steps 5 --warmup 2 --cpu-budget 3 > $O/n1.json 2> $O/n1.err || { tail -5 $O/n1.err; exit 1; }
python bench.py --config C1 --steps 5 --warmup 2 --cpu-budget 1 > $O/c1.json 2> $O/c1.err || { tail -5 $O/c1.err; exit 1; }
python bench.py --config C2 --steps 5 --warmup 2 --cpu-budget 2 > $O/c2.json 2> $O/c2.err || { tail -5 $O/c2.err; exit 1; }
python bench.py --config C4 --poses 16 --steps 3 --warmup 1 --cpu-budget 2 > $O/c4.json 2> $O/c4.err || { tail -5 $O/c4.err; exit 1; }
python bench.py --config C5 --poses 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/c5.json 2> $O/c5.err || { tail -5 $O/c5.err; exit 1; }
python bench.py --gpus 2 --steps 2 --warmup 1 > $O/g2_nccl.json 2> $O/g2_nccl.err; echo "nccl on 1 GPU rc=$?" >> $O/summary.txt; tail -2 $O/g2_nccl.err >> $O/summary.txt
DPR_BENCH_BACKEND=gloo python bench.py --gpus 2 --poses 8 --steps 2 --warmup 1 > $O/g2_gloo.json 2> $O/g2_gloo.err || { tail -5 $O/g2_gloo.err; exit 1; }
DPR_BENCH_BACKEND=gloo python bench.py --gpus 2 --config C3 --shard points --steps 2 --warmup 1 > $O/g2_pts.json 2> $O/g2_pts.err || { tail -5 $O/g2_pts.err; exit 1; }
DPR_BENCH_BACKEND=gloo python bench.py --gpus 2 --config C3 --shard poses --steps 2 --warmup 1 > $O/g2_weak.json 2> $O/g2_weak.err || { tail -5 $O/g2_weak.err; exit 1; }
for f in n1 c1 c2 c4 c5 g2_gloo g2_pts g2_weak; do echo "== $f"; python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}, d['config']['workload'][:40], d['roofline']['frac'], d.get('no_share'), d.get('cpu_baseline',{}).get('value'))
"; done
cat $O/summary.txt
