#!/usr/bin/env python3
"""How many (chunk, pose) pairs of C4's share take the chunk-owner forward's wide path, and how large
their footprints are (the bounding-box projection the kernels use, without their rounding slack)."""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import dpr_amd as dpr  # noqa: E402

dev = torch.device("cuda:0")
pts = torch.from_numpy(bench.synth_points("C4")).to(dev)
R, t = bench.synth_poses("C4", 64, 1)
R = torch.from_numpy(R).to(dev)
ps, _ = dpr.sort_points(pts)
P = ps.shape[0]
nc = (P + 4095) // 4096
pad = nc * 4096 - P
if pad:
    ps = torch.cat([ps, ps[-1:].expand(pad, 3)])
ch = ps.view(nc, 4096, 3)
lo, hi = ch.min(1).values, ch.max(1).values
h = 0.5 * (hi - lo)  # (nc, 3)
ph = torch.einsum("bij,cj->cbi", R.abs(), h)  # (nc, 64, 2) half extents in [-1, 1] units
ext = ph * 512 + 4.0  # pixels (n / 2 per unit, both sides) + slack
cells = ext[..., 0] * ext[..., 1]
tot = cells.numel()
for cap in (8192, 9984, 19968, 39936):
    print(f"footprint > {cap:6d} cells: {(cells > cap).sum().item():7d} of {tot} pairs ({100.0 * (cells > cap).float().mean().item():.2f} %)")
wide = cells > 9984
print("chunks with any wide pose:", wide.any(1).sum().item(), "of", nc)
print("cells of wide pairs: median %.0f, 90%% %.0f, max %.0f" % tuple(np.percentile(cells[wide].cpu().numpy(), [50, 90, 100])))
print("median footprint of the rest: %.0f cells" % cells[~wide].median().item())
