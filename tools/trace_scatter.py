"""Per-block, per-round phase timestamps of k_scatter_wc (needs the variant library built from
tools/trace_scatter_instrument.py)."""
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
buf = np.zeros((512, 8, 8), dtype=np.uint64)
rc = dpr_amd.lib().dpr_debug_trace(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0
t0 = buf[:, 0, 0][buf[:, 0, 0] > 0].min()
us = (buf.astype(np.int64) - int(t0)) / 100.0  # 100 MHz
live = buf[:, :, 7] > 0
names = ["classify (+LDS atomics)", "barrier 1", "scan (2 barriers inside)", "place into LDS + slot stores", "barrier 4", "write-out + cursors", "barrier 5"]
for r in range(5):
    m = live[:, r]
    if not m.any(): continue
    print(f"round {r}: blocks {m.sum()}  start mean {us[m, r, 0].mean():7.2f} us")
    for k in range(7):
        d = us[m, r, k + 1] - us[m, r, k]
        print(f"   {names[k]:30s} mean {d.mean():6.2f}  p50 {np.median(d):6.2f}  p95 {np.percentile(d, 95):6.2f}")
    tot = us[m, r, 7] - us[m, r, 0]
    print(f"   round total                    mean {tot.mean():6.2f}  p95 {np.percentile(tot, 95):6.2f}")
m = live[:, 0]
last = np.array([us[b, live[b].nonzero()[0].max(), 7] for b in range(512) if live[b].any()])
print("block end: mean", last.mean(), "max", last.max(), " first-wave blocks (start<5us):", (us[m, 0, 0] < 5).sum())
