import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import dpr_amd
from tests import data as D
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
grid = (256, 256, 256)
tp = torch.as_tensor(0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32), device=dev)
R = torch.as_tensor(D.random_rotations(rng, 1).astype(np.float32), device=dev)
t = torch.zeros(1, 3, device=dev)
out = dpr_amd.empty_grid(grid, 1, torch.float32, dev)
ws = torch.empty(dpr_amd.workspace_bytes("raster", grid, P, 1, 3, torch.float32, "tiled"), dtype=torch.uint8, device=dev)
for _ in range(3):
    dpr_amd.raster_(out, tp, R, t, algo="tiled", workspace=ws)
torch.cuda.synchronize()
buf = np.zeros((8192, 8), dtype=np.uint64)
assert dpr_amd.lib().dpr_debug_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
tr = buf[buf[:, 0] > 0]
t0 = tr[:, 0].min()
full = tr[tr[:, 5] > 0]
us = (full[:, [0, 1, 2, 5]].astype(np.int64) - int(t0)) / 100.0
print("blocks started", len(tr), "blocks that did work", len(full), "span", us[:, 3].max(), "us")
for k, name in enumerate(["n_split load", "setup (tile_parts, bg, origin)", "voxel loop"]):
    d = us[:, k + 1] - us[:, k]
    print(f"{name:32s} mean {d.mean():6.2f} p50 {np.median(d):6.2f} p95 {np.percentile(d,95):6.2f} max {d.max():6.2f}")
st = np.sort((tr[:, 0].astype(np.int64) - int(t0)) / 100.0)
print("block start times: #0 %.1f, 25%% %.1f, 50%% %.1f, 75%% %.1f, last %.1f" % (st[0], st[len(st)//4], st[len(st)//2], st[3*len(st)//4], st[-1]))
