"""Per-block phase timestamps of k_tile_splat (needs the DPR_TRACE variant library)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
rc = dpr_amd.lib().dpr_debug_trace(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0
tr = buf[buf[:, 5] > 0]
t0 = tr[:, 0].min()
us = (tr[:, :6].astype(np.int64) - int(t0)) / 100.0  # 100 MHz
print("blocks traced", len(tr), " kernel span", us[:, 5].max(), "us")
names = ["fetch item", "zero+issue prefetch", "record loop", "barrier wait", "flush"]
for k in range(5):
    d = us[:, k + 1] - us[:, k]
    print(f"{names[k]:22s} mean {d.mean():6.2f} us  p50 {np.median(d):6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f}")
tot = us[:, 5] - us[:, 0]
print(f"block total            mean {tot.mean():6.2f} us  p95 {np.percentile(tot, 95):6.2f}")
# concurrency: how many blocks are alive over time
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 5], -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
for q in (0.1, 0.3, 0.5, 0.7, 0.9):
    i = int(q * len(ev)); print(f"  t={ev[i,0]:6.1f} us alive blocks {int(alive[i])}")
nrec = tr[:, 7].astype(np.int64)
print("records per item: mean", nrec.mean(), "max", nrec.max())
order = np.argsort(us[:, 0])
print("start time of block #0/#511/#512/#1024/#2047 (by start order):", [round(float(us[order[i], 0]), 1) for i in (0, 511, 512, 1024, min(2047, len(us) - 1))])
