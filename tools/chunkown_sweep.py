"""2-D grids, batched poses: atomic / tiled / chunked (internal sort) / chunked on pre-sorted
points with DPR_FLAG_COHERENT_POINTS.  ms per call (fwd, bwd)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D
dev = torch.device("cuda:0")
def t_ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in e]))
cases = [(10_000_000, 512, 64), (10_000_000, 512, 16), (10_000_000, 512, 4), (1_000_000, 512, 64), (1_000_000, 128, 16),
         (200_000, 128, 64), (100_000, 128, 64), (100_000, 1024, 64), (10_000_000, 512, 1)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for P, n, B in cases:
    rng = np.random.default_rng(0)
    pts = torch.as_tensor(0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32), device=dev)
    spts, _ = dpr_amd.sort_points(pts)
    R = torch.as_tensor(D.random_rotations(rng, B)[:, :2].astype(np.float32), device=dev)
    t = torch.as_tensor((0.1 * rng.normal(size=(B, 2))).astype(np.float32), device=dev)
    g = torch.randn((B, n, n), device=dev).permute(2, 1, 0)
    out = dpr_amd.empty_grid((n, n), B, torch.float32, dev)
    line = f"P={P:>9} grid={n}^2 B={B:>3}:"
    for name, algo, p, kw in (("atomic", "atomic", pts, {}), ("tiled", "tiled", pts, {}), ("chunked", "chunked", pts, {}),
                              ("tiled/coherent", "tiled", spts, dict(coherent_points=True)),
                              ("chunked/coherent", "chunked", spts, dict(coherent_points=True))):
        if name == "atomic" and P * B > 2e8:
            continue
        wsb = max(16, dpr_amd.workspace_bytes("pullback", (n, n), P, B, 3, torch.float32, algo, **kw))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        f = t_ms(lambda: dpr_amd.raster_(out, p, R, t, algo=algo, workspace=ws, **kw))
        b = t_ms(lambda: dpr_amd.raster_pullback_(g, p, R, t, algo=algo, workspace=ws, **kw))
        line += f"  {name} {f:.3f}/{b:.3f}"
        del ws
    print(line + f"   auto={dpr_amd.resolve_algo('raster', (n, n), P, B, 3)}", flush=True)
