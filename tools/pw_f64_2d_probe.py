#!/usr/bin/env python3
"""2 M points -> 512^2 fp64, 8 poses, sorted + coherent, with and without point weights (for rocprofv3)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr  # noqa: E402
from tests import data as D  # noqa: E402
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
P, B = 2_000_000, 8
pts = torch.from_numpy((0.4 * rng.standard_normal(size=(P, 3))).astype(np.float64)).to(dev)
R = torch.from_numpy(D.random_rotations(rng, B, 3)[:, :2, :].astype(np.float64)).to(dev)
t = torch.from_numpy((0.1 * rng.normal(size=(B, 2))).astype(np.float64)).to(dev)
pw = torch.rand(P, device=dev, dtype=torch.float64) + 0.5
p, _, w = dpr.sort_points(pts, pw)
out = dpr.empty_grid((512, 512), B, torch.float64, dev)
for ww in (None, w):
    for _ in range(5):
        dpr.raster_(out, p, R, t, None, None, ww, coherent_points=True)
    torch.cuda.synchronize()
