#!/usr/bin/env python3
"""Few points on large grids: every algorithm next to AUTO's choice (the regret table stops at 256^3 / 1024^2)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr  # noqa: E402
from tests import data as D  # noqa: E402

dev = torch.device("cuda:0")
MID = "--mid" in sys.argv  # fp32, the regret table's grids, batch sizes between its columns
F64 = "--f64" in sys.argv  # fp64 data on the regret table's grid sizes instead of fp32 on large grids
DT, NPDT = (torch.float64, np.float64) if F64 else (torch.float32, np.float32)


def timed(fn, n=5):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for P in (100_000, 300_000, 1_000_000, 3_000_000, 10_000_000):
    rng = np.random.default_rng(5)
    pts = torch.from_numpy((0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32)).astype(NPDT)).to(dev)
    for grid in (((128, 128, 128), (256, 256, 256), (512, 512), (1024, 1024)) if MID else
                 ((128, 128, 128), (256, 256, 256), (384, 384, 384), (512, 512), (1024, 1024), (2048, 2048)) if F64 else
                 ((256, 256, 256), (384, 384, 384), (512, 512, 512), (768, 768, 768), (2048, 2048), (4096, 4096))):
        for B in ((2, 8, 32) if MID else (1, 8) if F64 else (1, 4)):
            n_out = len(grid)
            if int(np.prod(grid)) * B * (8 if F64 else 4) > 4e9:
                continue
            R = torch.from_numpy(D.random_rotations(rng, B, 3)[:, :n_out, :].astype(NPDT)).to(dev)
            t = torch.from_numpy((0.05 * rng.normal(size=(B, n_out))).astype(NPDT)).to(dev)
            out = dpr.empty_grid(grid, B, DT, dev)
            g = dpr.empty_grid(grid, B, DT, dev).normal_()
            res = {}
            for algo in ("atomic", "tiled", "chunked"):
                try:
                    res[algo] = (timed(lambda: dpr.raster_(out, pts, R, t, algo=algo)),
                                 timed(lambda: dpr.raster_pullback_(g, pts, R, t, algo=algo)))
                except Exception:
                    res[algo] = (float("nan"), float("nan"))
            af = dpr.resolve_algo("raster", grid, P, B, 3)
            ab = dpr.resolve_algo("pullback", grid, P, B, 3)
            bf = min(v[0] for v in res.values() if v[0] == v[0])
            bb = min(v[1] for v in res.values() if v[1] == v[1])
            print(f"P={P:>8d} grid={'x'.join(map(str, grid)):>12s} B={B}  fwd a/t/c {res['atomic'][0]:7.3f} {res['tiled'][0]:7.3f} {res['chunked'][0]:7.3f} auto={af:7s} regret {res[af][0] / bf:4.2f} | "
                  f"bwd a/t/c {res['atomic'][1]:7.3f} {res['tiled'][1]:7.3f} {res['chunked'][1]:7.3f} auto={ab:7s} regret {res[ab][1] / bb:4.2f}", flush=True)
            del out, g
    del pts
    torch.cuda.empty_cache()
