#!/usr/bin/env python3
"""Stage times of the C3 forward (10 M points 0.4*N(0,I) -> 256^3 fp32, one pose, tiled, as-generated order) as a PLAIN call
(12-byte records, no slot map) and as a DPR_FLAG_KEEP_BINNING call (16-byte records + slot map): does the record size show?"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpr_amd
from tests import data as D

dev = torch.device("cuda:0")
P, grid = 10_000_000, (256,) * 3
g = torch.Generator(device=dev).manual_seed(1234)
points = 0.4 * torch.randn((P, 3), device=dev, generator=g)
prng = np.random.default_rng(1)
R = torch.as_tensor(D.random_rotations(prng, 1, 3)[0], device=dev, dtype=torch.float32)
t = torch.as_tensor(0.1 * prng.normal(size=3), device=dev, dtype=torch.float32)
ws = torch.empty(max(dpr_amd.workspace_bytes(op, grid, P, 1, 3, torch.float32, "tiled", sharing=True) for op in ("raster", "pullback")),
                 dtype=torch.uint8, device=dev)
out = dpr_amd.empty_grid(grid, None, torch.float32, dev)
res = {}
for rep in range(2):
    for name, keep in (("plain", False), ("keep", True)):
        f = lambda: dpr_amd.raster_(out, points, R, t, algo="tiled", workspace=ws, keep_binning=keep)
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        st = dpr_amd.stage_times(f, "raster", "tiled", 30)
        res.setdefault(name, []).append({k: round(v * 1e3, 1) for k, v in st.items()})
print(json.dumps(res))
