"""C3-like single pose on a Hilbert-sorted cloud: every algorithm's forward / pullback call time (ms)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in e]))
rng = np.random.default_rng(0)
pts = torch.as_tensor(0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32), device=dev)
spts, _ = dpr_amd.sort_points(pts)
R = torch.as_tensor(D.random_rotations(rng, 1)[0].astype(np.float32), device=dev)
t = torch.zeros(3, device=dev)
g = torch.randn((n, n, n), device=dev).permute(2, 1, 0)
out = dpr_amd.empty_grid((n,) * 3, None, torch.float32, dev)
for name, algo, p, kw in (("atomic/random", "atomic", pts, {}), ("atomic/sorted", "atomic", spts, {}),
                          ("tiled/random", "tiled", pts, {}), ("tiled/sorted+flag", "tiled", spts, dict(coherent_points=True))):
    wsb = max(16, dpr_amd.workspace_bytes("pullback", (n,) * 3, P, 1, 3, torch.float32, algo, **kw))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    f = t_ms(lambda: dpr_amd.raster_(out, p, R, t, algo=algo, workspace=ws, **kw))
    b = t_ms(lambda: dpr_amd.raster_pullback_(g, p, R, t, algo=algo, workspace=ws, **kw))
    print(f"P={P} grid={n}^3 {name:22s} fwd {f:.3f} ms  bwd {b:.3f} ms", flush=True)
