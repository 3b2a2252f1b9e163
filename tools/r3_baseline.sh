#!/bin/bash
# round-3 baseline: where the time goes at C3 / C4 / C5 before this round's changes
set -o pipefail
O=gpurun_out/r3_base; mkdir -p $O
python bench.py --steps 20 --warmup 3 --cpu-budget 3 2>/dev/null | tail -1 > $O/bench_c3.json || exit 1
echo c3 done
python tools/stage_probe.py --P 50000000 --grid 512 512 512 --dtype f64 --order random 2>&1 | grep -v amdgpu.ids > $O/c5_stage_random.txt || exit 1
python tools/stage_probe.py --P 50000000 --grid 512 512 512 --dtype f64 --order hilbert 2>&1 | grep -v amdgpu.ids > $O/c5_stage_hilbert.txt || exit 1
echo c5 stages done
python tools/bench_configs.py C5 C3 2>&1 | grep -v amdgpu.ids > $O/configs.txt || exit 1
echo configs done
python - > $O/c5_chunked3d.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, dpr_amd
from tests import data as D
dev = torch.device("cuda:0")
def t_ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in e]))
rng = np.random.default_rng(0)
for (P, n, dt, B) in ((50_000_000, 512, torch.float64, 8), (10_000_000, 256, torch.float32, 1), (10_000_000, 256, torch.float32, 8)):
    npdt = np.float32 if dt == torch.float32 else np.float64
    pts = (0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32)).astype(npdt)
    tp = dpr_amd.sort_points(torch.as_tensor(pts, device=dev))[0]
    R = torch.as_tensor(D.random_rotations(rng, B).astype(npdt), device=dev)
    t = torch.as_tensor((0.1 * rng.normal(size=(B, 3))).astype(npdt), device=dev)
    grid = (n,) * 3
    out = dpr_amd.empty_grid(grid, B, dt, dev)
    g = torch.randn((B,) + grid, device=dev, dtype=dt).permute(3, 2, 1, 0)
    for algo in ("chunked", "tiled"):
        kw = dict(coherent_points=True) if algo == "tiled" else {}
        ws = torch.empty(max(16, dpr_amd.workspace_bytes("pullback", grid, P, B, 3, dt, algo, **kw), dpr_amd.workspace_bytes("raster", grid, P, B, 3, dt, algo, **kw)), dtype=torch.uint8, device=dev)
        f = t_ms(lambda: dpr_amd.raster_(out, tp, R, t, algo=algo, workspace=ws, **kw))
        b = t_ms(lambda: dpr_amd.raster_pullback_(g, tp, R, t, algo=algo, workspace=ws, **kw))
        print(f"P={P} {n}^3 B={B} sorted algo={algo}: fwd {f:.3f} ms  bwd {b:.3f} ms", flush=True)
        del ws
    del out, g, tp; torch.cuda.empty_cache()
PY
echo chunked3d done
python bench.py --config C4 --poses 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4.json || exit 1
python bench.py --config C5 --poses 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json || exit 1
echo all done
