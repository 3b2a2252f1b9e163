for d in gauss tight uniform; do python3 bench.py --config C4 --poses 64 --dist $d --no-cpu-baseline --no-secondary --no-scaling-reference 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('$d',d['ms_per_step'],r['ms'],r['pullback']['ms'])"; done
