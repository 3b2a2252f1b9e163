import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, dpr_amd
from tests import data as D
dev = torch.device("cuda:0")
P, n, B = 3_000_000, 512, 160
rng = np.random.default_rng(0)
pts = torch.as_tensor(0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32), device=dev)
R = torch.as_tensor(D.random_rotations(rng, B)[:, :2].astype(np.float32), device=dev)
t = torch.as_tensor((0.1 * rng.normal(size=(B, 2))).astype(np.float32), device=dev)
g = torch.randn((B, n, n), device=dev).permute(2, 1, 0)
out = dpr_amd.raster((n, n), pts, R, t, algo="chunked")
pb = dpr_amd.raster_pullback_(g, pts, R, t, algo="chunked")
acc = torch.zeros_like(pb.points); accw = torch.zeros_like(pb.point_weight)
for lo in range(0, B, 40):
    hi = min(B, lo + 40)
    o2 = dpr_amd.raster((n, n), pts, R[lo:hi], t[lo:hi], algo="tiled")
    d = (out[..., lo:hi] - o2).abs().max().item()
    assert d < 2e-4 * max(1.0, o2.abs().max().item()), ("out", lo, d)
    p2 = dpr_amd.raster_pullback_(g[..., lo:hi], pts, R[lo:hi], t[lo:hi], algo="tiled")
    acc += p2.points; accw += p2.point_weight
    assert (pb.rotation[lo:hi] - p2.rotation).abs().max().item() < 1e-3 * max(1.0, p2.rotation.abs().max().item())
print("points grad max diff", (pb.points - acc).abs().max().item(), "scale", acc.abs().max().item())
assert (pb.points - acc).abs().max().item() < 1e-3 * acc.abs().max().item()
assert (pb.point_weight - accw).abs().max().item() < 1e-3 * accw.abs().max().item()
print("ok: 160 poses in slices == sum of tiled calls")
