#!/usr/bin/env python3
"""Only the sorted + coherent + point-weights case of tools/time_point_weights_c3.py (for rocprofv3)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import dpr_amd as dpr  # noqa: E402
dev = torch.device("cuda:0")
pts = torch.from_numpy(bench.synth_points("C3")).to(dev)
R, t = bench.synth_poses("C3", 1, 1)
R, t = torch.from_numpy(R[0]).to(dev), torch.from_numpy(t[0]).to(dev)
pw = torch.rand(pts.shape[0], device=dev) + 0.5
g = torch.randn(256, 256, 256, device=dev)
p, perm, w = dpr.sort_points(pts, pw)
out = dpr.raster((256, 256, 256), p, R, t, point_weight=w, coherent_points=True)
for _ in range(5):
    dpr.raster_(out, p, R, t, point_weight=w, coherent_points=True)
    dpr.raster_pullback_(g, p, R, t, point_weight=w, coherent_points=True)
torch.cuda.synchronize()
