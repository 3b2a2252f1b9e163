#!/usr/bin/env python3
"""Throughput over odd shapes (grids that are no multiple of the tiles, tiny and huge grids, tight
clusters, clouds mostly outside the grid): ns per point-pose of raster / pullback with AUTO, to spot
a shape that falls far behind its neighbours."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr  # noqa: E402
from tests import data as D  # noqa: E402

dev = torch.device("cuda:0")
GRIDS = [(64, 64, 64), (100, 100, 100), (250, 250, 250), (256, 256, 256), (384, 384, 384), (512, 512, 512),
         (640, 200, 72), (1000, 1000), (4096, 4096), (300, 200), (2048, 64), (33, 17)]
CLOUDS = {"gauss0.4": lambda rng, P: 0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32),
          "tight0.02": lambda rng, P: 0.02 * rng.standard_normal(size=(P, 3), dtype=np.float32) + np.float32(0.3),
          "wide2.0": lambda rng, P: 2.0 * rng.standard_normal(size=(P, 3), dtype=np.float32)}


def timed(fn, n=5):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for P in (1_000_000, 10_000_000):
    for cname, mk in CLOUDS.items():
        rng = np.random.default_rng(5)
        pts = torch.from_numpy(mk(rng, P)).to(dev)
        for grid in GRIDS:
            for B in (1, 8):
                n_out = len(grid)
                G = int(np.prod(grid))
                if G * B * 4 > 3e9:
                    continue
                R = torch.from_numpy(D.random_rotations(rng, B, 3)[:, :n_out, :].astype(np.float32)).to(dev)
                t = torch.from_numpy((0.05 * rng.normal(size=(B, n_out))).astype(np.float32)).to(dev)
                out = dpr.empty_grid(grid, B, torch.float32, dev)
                g = dpr.empty_grid(grid, B, torch.float32, dev).normal_()
                f = timed(lambda: dpr.raster_(out, pts, R, t))
                b = timed(lambda: dpr.raster_pullback_(g, pts, R, t))
                af = dpr.resolve_algo("raster", grid, P, B, 3)
                ab = dpr.resolve_algo("pullback", grid, P, B, 3)
                print(f"P={P:>8d} {cname:10s} grid={'x'.join(map(str, grid)):>12s} B={B} raster {f:8.3f} ms {1e6 * f / (P * B):7.3f} ns/pp [{af:7s}]  "
                      f"pullback {b:8.3f} ms {1e6 * b / (P * B):7.3f} ns/pp [{ab:7s}]", flush=True)
                del out, g
        del pts
        torch.cuda.empty_cache()
