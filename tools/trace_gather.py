"""Per-block phase timestamps of k_tile_gather (diagnostic build, see tools/trace_splat_instrument.py)."""
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
g = torch.randn(1, 256, 256, 256, device=dev).permute(3, 2, 1, 0)
ws = torch.empty(dpr_amd.workspace_bytes("pullback", grid, P, 1, 3, torch.float32, "tiled"), dtype=torch.uint8, device=dev)
for _ in range(3):
    dpr_amd.raster_pullback_(g, tp, R, t, algo="tiled", workspace=ws)
torch.cuda.synchronize()
buf = np.zeros((8192, 8), dtype=np.uint64)
assert dpr_amd.lib().dpr_debug_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
tr = buf[buf[:, 4] > 0]
t0 = tr[:, 0].min()
us = (tr[:, :5].astype(np.int64) - int(t0)) / 100.0
print("blocks traced", len(tr), " kernel span", us[:, 4].max(), "us")
for k, name in enumerate(["fetch item", "stage ds_dout tile", "record loop (wave 0)", "reduce + partials"]):
    d = us[:, k + 1] - us[:, k]
    print(f"{name:24s} mean {d.mean():6.2f} us  p50 {np.median(d):6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f}")
tot = us[:, 4] - us[:, 0]
print(f"block total              mean {tot.mean():6.2f} us  p95 {np.percentile(tot, 95):6.2f}   sum/span = {tot.sum() / us[:, 4].max():.0f} blocks alive on average")
order = np.argsort(us[:, 0])
print("start of block #0/#767/#768/#1536/last:", [round(float(us[order[i], 0]), 1) for i in (0, 767, 768, 1536, len(us) - 1)])
