"""Training-step comparison at one pose: raster_ -> [ds_dout = 2 (out - target) on the device;
raster_pullback_]  vs  raster_ -> raster_residual_pullback_ (sensitivity formed in-kernel)."""
import sys, os, argparse
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=10_000_000)
ap.add_argument("--grid", type=int, nargs="+", default=[256, 256, 256])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float32
rng = np.random.default_rng(0)
grid = tuple(a.grid); n_out = len(grid)
tp = torch.as_tensor(0.4 * rng.standard_normal(size=(a.P, 3), dtype=np.float32), device=dev)
R = torch.as_tensor(D.random_rotations(rng, 1)[:, :n_out].astype(np.float32), device=dev)
t = torch.zeros(1, n_out, device=dev, dtype=dt)
tgt = torch.randn((1,) + tuple(reversed(grid)), device=dev, dtype=dt).permute(*reversed(range(n_out + 1)))
out = dpr_amd.empty_grid(grid, 1, dt, dev)
g = torch.empty_like(tgt)
ws = torch.empty(dpr_amd.workspace_bytes("pullback", grid, a.P, 1, 3, dt, "tiled"), dtype=torch.uint8, device=dev)
outs = dict(ds_dpoints=torch.empty(a.P, 3, device=dev), ds_dpoint_weight=torch.empty(a.P, device=dev))
loss = torch.empty(1, device=dev)
fwd = lambda: dpr_amd.raster_(out, tp, R, t, algo="tiled", workspace=ws, keep_binning=True)
def unfused():
    fwd()
    torch.sub(out, tgt, out=g); g.mul_(2.0)
    l = (g * g).sum() * 0.25  # the loss value a training loop also wants
    return dpr_amd.raster_pullback_(g, tp, R, t, algo="tiled", workspace=ws, reuse_binning=True, **outs)
def fused():
    fwd()
    return dpr_amd.raster_residual_pullback_(out, tgt, tp, R, t, algo="tiled", workspace=ws,
                                             reuse_binning=True, loss=loss, **outs)
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
a_, b_ = unfused(), fused()[0]
assert torch.equal(a_.points, b_.points)
tu, tf, t0 = timeit(unfused), timeit(fused), timeit(fwd)
print(f"P={a.P} grid={grid}: forward {t0:.3f} ms | step unfused {tu:.3f} ms | fused {tf:.3f} ms "
      f"({a.P / tf / 1e6:.1f} vs {a.P / tu / 1e6:.1f} G pts/s)")
