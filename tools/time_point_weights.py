#!/usr/bin/env python3
"""C4 share (10 M points -> 512^2, 64 poses) with and without point weights: forward and pullback
times of the chunk-owner path (AUTO).  The bench configs carry no point weights; this is the check
that the HAS_PW instantiations keep up."""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import dpr_amd as dpr  # noqa: E402

dev = torch.device("cuda:0")
pts = torch.from_numpy(bench.synth_points("C4")).to(dev)
R, t = bench.synth_poses("C4", 64, 1)
R, t = torch.from_numpy(R).to(dev), torch.from_numpy(t).to(dev)
pw = torch.rand(pts.shape[0], device=dev) + 0.5
g = torch.randn(512, 512, 64, device=dev).permute(2, 1, 0).contiguous().permute(2, 1, 0)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, w in (("no point weights", None), ("point weights", pw)):
    out = dpr.raster((512, 512), pts, R, t, point_weight=w)
    f = timed(lambda: dpr.raster_(out, pts, R, t, point_weight=w))
    b = timed(lambda: dpr.raster_pullback_(out, pts, R, t, point_weight=w))
    print(f"{name:18s} raster {f:7.3f} ms   pullback {b:7.3f} ms", flush=True)
