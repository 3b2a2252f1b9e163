"""Stage breakdown of one pose of an arbitrary config (diagnostic)."""
import sys, os, argparse
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from bench import morton_order
from tests import data as D
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=10_000_000)
ap.add_argument("--grid", type=int, nargs="+", default=[512, 512])
ap.add_argument("--dtype", default="f32")
ap.add_argument("--order", default="random")
ap.add_argument("--algo", default="tiled")
ap.add_argument("--coherent", action="store_true", help="pass coherent_points=True (local binning)")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float32 if a.dtype == "f32" else torch.float64
npdt = np.float32 if a.dtype == "f32" else np.float64
rng = np.random.default_rng(0)
pts = 0.4 * rng.standard_normal(size=(a.P, 3), dtype=np.float32)
if a.order == "morton":
    pts = pts[morton_order(pts)]
grid = tuple(a.grid); n_out = len(grid)
tp = torch.as_tensor(pts.astype(npdt), device=dev)
if a.order == "hilbert":
    tp, _ = dpr_amd.sort_points(tp)
kw = dict(coherent_points=True) if a.coherent else {}
names = "tiled_local" if (a.coherent and a.algo == "tiled") else a.algo
R = torch.as_tensor(D.random_rotations(rng, 1)[:, :n_out].astype(npdt), device=dev)
t = torch.zeros(1, n_out, device=dev, dtype=dt)
g = torch.randn((1,) + tuple(reversed(grid)), device=dev, dtype=dt).permute(*reversed(range(n_out + 1)))
out = dpr_amd.empty_grid(grid, 1, dt, dev)
ws = torch.empty(max(16, dpr_amd.workspace_bytes("pullback", grid, a.P, 1, 3, dt, a.algo, **kw)), dtype=torch.uint8, device=dev)
fwd = lambda: dpr_amd.raster_(out, tp, R, t, algo=a.algo, workspace=ws, keep_binning=True, **kw)
bwd = lambda: dpr_amd.raster_pullback_(g, tp, R, t, algo=a.algo, workspace=ws, reuse_binning=True, **kw)
fwd(); bwd(); torch.cuda.synchronize()
sf = dpr_amd.stage_times(fwd, "raster", names, 10)
sb = dpr_amd.stage_times(bwd, "pullback", names, 10, prepare=fwd)
print(f"P={a.P} grid={grid} {a.dtype} {a.order} {names}")
print(" fwd", {k: round(v * 1e3) for k, v in sf.items()})
print(" bwd", {k: round(v * 1e3) for k, v in sb.items()})
