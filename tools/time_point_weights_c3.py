#!/usr/bin/env python3
"""C3 (10 M points -> 256^3, one pose) with and without point weights: forward / pullback times of
the tiled path (AUTO), random and Hilbert-sorted + coherent.  The bench configs carry no point
weights; this is the check that the HAS_PW instantiations keep up."""
import sys
import time

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


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for order in ("random", "sorted+coherent"):
    p, w = pts, pw
    kw = {}
    if order != "random":
        p, perm, w = dpr.sort_points(pts, pw)
        kw = dict(coherent_points=True)
    for name, ww in (("no point weights", None), ("point weights", w)):
        out = dpr.raster((256, 256, 256), p, R, t, point_weight=ww, **kw)
        f = timed(lambda: dpr.raster_(out, p, R, t, point_weight=ww, **kw))
        b = timed(lambda: dpr.raster_pullback_(g, p, R, t, point_weight=ww, **kw))
        print(f"{order:16s} {name:18s} raster {f:7.4f} ms   pullback {b:7.4f} ms", flush=True)
