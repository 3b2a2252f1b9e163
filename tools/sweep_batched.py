"""Crossover sweep for the AUTO heuristic with batched poses: per-pose time, atomic vs tiled."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def t_ms(fn, reps=7):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in e]))
for grid in [(64,) * 3, (128,) * 3, (256,) * 3, (128, 128), (512, 512)]:
    n_out = len(grid)
    for B in [4, 16, 64]:
        for P in [3_000, 10_000, 30_000, 100_000, 300_000, 1_000_000]:
            tp = torch.as_tensor(0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32), device=dev)
            R = torch.as_tensor(D.random_rotations(rng, B)[:, :n_out].astype(np.float32), device=dev)
            t = torch.zeros(B, n_out, device=dev)
            g = torch.randn((B,) + tuple(reversed(grid)), device=dev).permute(*reversed(range(n_out + 1)))
            out = dpr_amd.empty_grid(grid, B, torch.float32, dev)
            row = []
            for algo in ["atomic", "tiled"]:
                ws = torch.empty(max(16, dpr_amd.workspace_bytes("pullback", grid, P, B, 3, torch.float32, algo)), dtype=torch.uint8, device=dev)
                f = t_ms(lambda: dpr_amd.raster_(out, tp, R, t, algo=algo, workspace=ws))
                b = t_ms(lambda: dpr_amd.raster_pullback_(g, tp, R, t, algo=algo, workspace=ws))
                row.append(f"{algo}: fwd {f / B * 1e3:7.1f} bwd {b / B * 1e3:7.1f}")
            print(f"grid {'x'.join(map(str, grid)):11s} B={B:3d} P={P:8d} us/pose | " + " | ".join(row), flush=True)
