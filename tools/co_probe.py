"""One chunk-owner forward + pullback (diagnostic for rocprofv3)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D
dev = torch.device("cuda:0")
P, n, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
pts = torch.as_tensor(0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32), device=dev)
spts, _ = dpr_amd.sort_points(pts)
R = torch.as_tensor(D.random_rotations(rng, B)[:, :2].astype(np.float32), device=dev)
t = torch.as_tensor((0.1 * rng.normal(size=(B, 2))).astype(np.float32), device=dev)
g = torch.randn((B, n, n), device=dev).permute(2, 1, 0)
out = dpr_amd.empty_grid((n, n), B, torch.float32, dev)
ws = torch.empty(dpr_amd.workspace_bytes("pullback", (n, n), P, B, 3, torch.float32, "chunked", coherent_points=True), dtype=torch.uint8, device=dev)
for _ in range(3):
    dpr_amd.raster_(out, spts, R, t, algo="chunked", workspace=ws, coherent_points=True)
    dpr_amd.raster_pullback_(g, spts, R, t, algo="chunked", workspace=ws, coherent_points=True)
torch.cuda.synchronize()
