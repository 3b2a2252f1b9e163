"""Crossover sweep for the AUTO heuristic: forward / pullback time per algorithm vs P."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from bench import morton_order
from tests import data as D

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def t_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in e]))
grids = [(128,)*3, (256,)*3, (512, 512)]
for grid in grids:
    n_out = len(grid)
    for P in [30_000, 100_000, 300_000, 1_000_000, 3_000_000, 10_000_000]:
        pts = (0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32))
        for order in ["random", "morton"]:
            p = pts[morton_order(pts)] if order == "morton" else pts
            tp = torch.as_tensor(p, device=dev)
            R = torch.as_tensor(D.random_rotations(rng, 1)[:, :n_out].astype(np.float32), device=dev)
            t = torch.zeros(1, n_out, device=dev)
            g = torch.randn((1,) + tuple(reversed(grid)), device=dev).permute(*reversed(range(n_out + 1)))
            out = dpr_amd.empty_grid(grid, 1, torch.float32, dev)
            row = []
            for algo in ["atomic", "tiled", "chunked"]:
                ws = torch.empty(max(16, dpr_amd.workspace_bytes("pullback", grid, P, 1, 3, torch.float32, algo)), dtype=torch.uint8, device=dev)
                f = t_ms(lambda: dpr_amd.raster_(out, tp, R, t, algo=algo, workspace=ws))
                b = t_ms(lambda: dpr_amd.raster_pullback_(g, tp, R, t, algo=algo, workspace=ws))
                row.append(f"{algo}: fwd {f*1e3:7.0f}us bwd {b*1e3:7.0f}us")
            print(f"grid {'x'.join(map(str,grid)):11s} P={P:9d} {order:6s} | " + " | ".join(row), flush=True)
