#!/usr/bin/env python3
"""The raster (KEEP_BINNING) + pullback (REUSE_BINNING) pair -- what an rrule runs -- with and
without point weights / optional arguments, C3 and C4-share shapes (AUTO)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr  # noqa: E402
from tests import data as D  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(10_000_000, (256, 256, 256), 1, torch.float32), (10_000_000, (512, 512), 64, torch.float32),
          (20_000_000, (512, 512, 512), 4, torch.float64), (1_000_000, (128, 128, 128), 1, torch.float32)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ONLY = [int(x) for x in sys.argv[1:]]  # shape indices (all when empty)
for P, grid, B, dt in ([SHAPES[i] for i in ONLY] if ONLY else SHAPES):
    rng = np.random.default_rng(3)
    npdt = np.float32 if dt == torch.float32 else np.float64
    n_out = len(grid)
    pts = torch.from_numpy((0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32)).astype(npdt)).to(dev)
    R = torch.from_numpy(D.random_rotations(rng, B, 3)[:, :n_out, :].astype(npdt)).to(dev)
    t = torch.from_numpy((0.1 * rng.normal(size=(B, n_out))).astype(npdt)).to(dev)
    pw = torch.rand(P, device=dev, dtype=dt) + 0.5
    ow = torch.rand(B, device=dev, dtype=dt) + 0.5
    bg = torch.rand(B, device=dev, dtype=dt)
    g = dpr.empty_grid(grid, B, dt, dev).normal_()
    out = dpr.empty_grid(grid, B, dt, dev)
    base = None
    for order in ("random", "sorted+coherent"):
        p, w, kw = pts, pw, {}
        if order != "random":
            p, _, w = dpr.sort_points(pts, pw)
            kw = dict(coherent_points=True)
        for name, args in (("plain", (None, None, None)), ("point weights", (None, None, w)), ("bg + ow + pw", (bg, ow, w))):
            need = max(dpr.workspace_bytes(op, grid, P, B, 3, dt, "auto", sharing=True, **kw) for op in ("raster", "pullback"))
            ws = torch.zeros(max(need, 16), dtype=torch.uint8, device=dev)

            def step():
                dpr.raster_(out, p, R, t, *args, workspace=ws, keep_binning=True, **kw)
                dpr.raster_pullback_(g, p, R, t, *args, workspace=ws, reuse_binning=True, **kw)
            s = timed(step)
            if name == "plain":
                base = s
            flag = "  <-- " if s > 1.3 * base else ""
            print(f"P={P:>9d} grid={'x'.join(map(str, grid)):>11s} B={B:<3d} {str(dt)[6:]:8s} {order:15s} {name:14s} step {s:8.3f} ms ({s / base:4.2f}x){flag}", flush=True)
            del ws
    del pts, g, pw, out
    torch.cuda.empty_cache()
