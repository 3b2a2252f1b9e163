#!/usr/bin/env python3
"""Times the variants the bench configs do not carry -- point weights, all optional arguments,
pre-sorted + coherent input -- over a handful of shapes and prints them next to the plain call, so
that an instantiation that falls behind its siblings shows up (AUTO everywhere)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr  # noqa: E402
from tests import data as D  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # P, n_in, grid, B, dtype
    (1_000_000, 3, (128, 128, 128), 1, torch.float32),
    (5_000_000, 3, (256, 256, 256), 4, torch.float32),
    (10_000_000, 3, (256, 256, 256), 1, torch.float64),
    (20_000_000, 3, (512, 512, 512), 2, torch.float64),
    (5_000_000, 3, (512, 512), 16, torch.float32),
    (2_000_000, 2, (512, 512), 8, torch.float32),
    (2_000_000, 3, (512, 512), 8, torch.float64),
]


def timed(fn, n=5):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for P, n_in, grid, B, dt in SHAPES:
    rng = np.random.default_rng(3)
    npdt = np.float32 if dt == torch.float32 else np.float64
    n_out = len(grid)
    pts = torch.from_numpy((0.4 * rng.standard_normal(size=(P, n_in), dtype=np.float32)).astype(npdt)).to(dev)
    R = torch.from_numpy(D.random_rotations(rng, B, n_in)[:, :n_out, :].astype(npdt)).to(dev)
    t = torch.from_numpy((0.1 * rng.normal(size=(B, n_out))).astype(npdt)).to(dev)
    pw = torch.rand(P, device=dev, dtype=dt) + 0.5
    ow = torch.rand(B, device=dev, dtype=dt) + 0.5
    bg = torch.rand(B, device=dev, dtype=dt)
    g = dpr.empty_grid(grid, B, dt, dev).normal_()
    base = None
    for order in ("random", "sorted+coherent"):
        p, w, kw = pts, pw, {}
        if order != "random":
            p, _, w = dpr.sort_points(pts, pw)
            kw = dict(coherent_points=True)
        for name, args in (("plain", (None, None, None)), ("point weights", (None, None, w)), ("bg + ow + pw", (bg, ow, w))):
            out = dpr.empty_grid(grid, B, dt, dev)
            f = timed(lambda: dpr.raster_(out, p, R, t, *args, **kw))
            b = timed(lambda: dpr.raster_pullback_(g, p, R, t, *args, **kw))
            if name == "plain":
                base = (f, b)
            flag = "  <-- " if (f > 1.35 * base[0] or b > 1.35 * base[1]) else ""
            print(f"P={P:>9d} n_in={n_in} grid={'x'.join(map(str, grid)):>11s} B={B:<3d} {str(dt)[6:]:8s} {order:15s} {name:14s} "
                  f"raster {f:8.3f} ms ({f / base[0]:4.2f}x)  pullback {b:8.3f} ms ({b / base[1]:4.2f}x){flag}", flush=True)
    del pts, g, pw
    torch.cuda.empty_cache()
