"""Forward over several poses of a Hilbert-sorted cloud on a 3-D grid: owner-computes tiles (DPR_ALGO_CHUNKED,
csrc/dpr_owner.hip) against the tiled pipeline with local binning, both with DPR_FLAG_COHERENT_POINTS.
Three kinds of cloud (the host cannot tell them apart): Gaussian 0.4 sigma, uniform, clustered 0.1 sigma.
Usage: owner_batch_sweep.py [--f64] > profiles/r05_owner_batch_sweep.txt"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpr_amd
from tests import data as D
ap = argparse.ArgumentParser()
ap.add_argument("--f64", action="store_true")
ap.add_argument("--quick", action="store_true")
ap.add_argument("--sparse", action="store_true", help="only the rows where explicit `chunked` means the chunk lists")
a = ap.parse_args()
dev = torch.device("cuda:0")
tdt = torch.float64 if a.f64 else torch.float32

def cloud(dist, P):
    g = torch.Generator(device=dev); g.manual_seed(0)
    if dist == "uniform":
        return (1.1 * torch.rand(P, 3, device=dev, generator=g) - 0.55).to(tdt)
    return ((0.1 if dist == "tight" else 0.4) * torch.randn(P, 3, device=dev, generator=g)).to(tdt)

def time_fwd(algo, grid, pts, R, t, B):
    kw = dict(coherent_points=True)
    P = pts.shape[0]
    try:
        need = dpr_amd.workspace_bytes("raster", grid, P, B, 3, tdt, algo, **kw)
    except dpr_amd.DprError:
        return float("nan")
    ws = torch.empty(max(16, need), dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(grid, B, tdt, dev)
    f = lambda: dpr_amd.raster_(out, pts, R, t, None, None, None, algo=algo, workspace=ws, **kw)
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

print(f"# {'cloud':8s} {'P':>9s} {'grid':>5s} {'B':>3s} {'pts/voxel':>9s} {'owner ms':>9s} {'tiled ms':>9s} {'tiled/owner':>11s}   ({'fp64' if a.f64 else 'fp32'})", flush=True)
Ps = [1_000_000, 3_000_000, 10_000_000, 30_000_000] if not a.quick else [3_000_000]
if a.sparse:
    Ps = [300_000, 1_000_000, 3_000_000, 10_000_000]
for dist in ("gauss", "uniform", "tight"):
    for P in Ps:
        pts = dpr_amd.sort_points(cloud(dist, P))[0]
        for n in (128, 256, 512):
            grid = (n, n, n)
            for B in (2, 4, 8, 16):
                if n ** 3 * B * (8 if a.f64 else 4) > 20e9 or (a.f64 and P > 10_000_000):
                    continue
                if a.sparse and not (B >= 4 and P * 10 <= n ** 3):
                    continue
                rng = np.random.default_rng(B)
                R = torch.as_tensor(D.random_rotations(rng, B, 3), device=dev).to(tdt)
                t = torch.as_tensor(0.1 * rng.normal(size=(B, 3)), device=dev).to(tdt)
                to, tt = time_fwd("chunked", grid, pts, R, t, B), time_fwd("tiled", grid, pts, R, t, B)
                print(f"  {dist:8s} {P:9d} {n:5d} {B:3d} {P / n ** 3:9.3f} {to:9.3f} {tt:9.3f} {tt / to:11.2f}", flush=True)
        del pts
        torch.cuda.empty_cache()
