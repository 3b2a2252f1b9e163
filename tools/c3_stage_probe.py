#!/usr/bin/env python3
"""Per-stage times of the C3 step (10 M points 0.4*N(0,I) -> 256^3 fp32, one pose, tiled KEEP / REUSE pair,
as bench.py runs it) for the library named by DPR_LIB_OVERRIDE -- the A/B harness of the round-6 kernel
experiments (ablation builds: `make -C diffpointrasterisation.jl_amd/csrc ../libdpr_abl<N>.so`).
  python tools/c3_stage_probe.py [--coherent] [--P 10000000] [--grid 256] [--reps 30] [--tag name]
Prints one JSON line: {"tag", "raster": {stage: ms}, "pullback": {stage: ms}, "pair_ms": event-timed step}."""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpr_amd
from tests import data as D

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=10_000_000)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--coherent", action="store_true")
ap.add_argument("--algo", default="tiled")
ap.add_argument("--tag", default=os.path.basename(os.environ.get("DPR_LIB_OVERRIDE", "libdpr.so")))
ap.add_argument("--check", action="store_true", help="compare `out` with the default library's (forward parity of a variant)")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
points = 0.4 * torch.randn((a.P, 3), device=dev, generator=g)
prng = np.random.default_rng(1)
R = torch.as_tensor(D.random_rotations(prng, 1, 3)[0], device=dev, dtype=torch.float32)
t = torch.as_tensor(0.1 * prng.normal(size=3), device=dev, dtype=torch.float32)
kw = {}
if a.coherent:
    points = dpr_amd.sort_points(points)[0]
    kw = dict(coherent_points=True)
grid = (a.grid,) * 3
share = a.algo == "tiled"
ws = torch.empty(max(16, dpr_amd.workspace_bytes("pullback", grid, a.P, 1, 3, torch.float32, a.algo, sharing=share, **kw),
                     dpr_amd.workspace_bytes("raster", grid, a.P, 1, 3, torch.float32, a.algo, sharing=share, **kw)),
                 dtype=torch.uint8, device=dev)
out = dpr_amd.empty_grid(grid, None, torch.float32, dev)
ds = dpr_amd.empty_grid(grid, None, torch.float32, dev)
ds.copy_(torch.randn(grid, device=dev, generator=g))
d_pts = torch.empty_like(points)
skw = dict(keep_binning=True) if share else {}
rkw = dict(reuse_binning=True) if share else {}
fwd = lambda: dpr_amd.raster_(out, points, R, t, None, None, None, algo=a.algo, workspace=ws, **skw, **kw)
bwd = lambda: dpr_amd.raster_pullback_(ds, points, R, t, None, None, None, ds_dpoints=d_pts, algo=a.algo, workspace=ws,
                                       **rkw, **kw)
for _ in range(3):
    fwd(); bwd()
torch.cuda.synchronize()
name = lambda op: ("tiled_local" if a.coherent else "tiled") if a.algo == "tiled" else a.algo
res = {"tag": a.tag, "coherent": a.coherent, "algo": a.algo}
res["raster"] = {k: round(v, 4) for k, v in dpr_amd.stage_times(fwd, "raster", name("raster"), a.reps).items()}
res["pullback"] = {k: round(v, 4) for k, v in
                   dpr_amd.stage_times(bwd, "pullback", name("pullback"), a.reps, prepare=fwd if share else None).items()}
# the pair, event-timed
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
for e0, e1 in ev:
    e0.record(); fwd(); bwd(); e1.record()
torch.cuda.synchronize()
res["pair_ms"] = round(float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])), 4)
res["out_sum"] = float(out.double().sum())
res["out_abs_sum"] = float(out.double().abs().sum())
res["dpts_abs_sum"] = float(d_pts.double().abs().sum())
print(json.dumps(res))
