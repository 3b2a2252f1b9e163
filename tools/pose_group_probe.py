"""Batched poses: per-pose pipeline (DPR_POSE_GROUP=1) vs pose groups, one GPU."""
import sys, os, argparse
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=10_000_000)
ap.add_argument("--grid", type=int, nargs="+", default=[512, 512])
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4, 8, 16])
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float32
rng = np.random.default_rng(0)
grid = tuple(a.grid); n_out = len(grid); B = a.B
tp = torch.as_tensor(0.4 * rng.standard_normal(size=(a.P, 3), dtype=np.float32), device=dev)
R = torch.as_tensor(D.random_rotations(rng, B)[:, :n_out].astype(np.float32), device=dev)
t = torch.as_tensor((0.1 * rng.normal(size=(B, n_out))).astype(np.float32), device=dev)
g = torch.randn((B,) + tuple(reversed(grid)), device=dev, dtype=dt).permute(*reversed(range(n_out + 1)))
out = dpr_amd.empty_grid(grid, B, dt, dev)
def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in e:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    return float(np.median([x.elapsed_time(y) for x, y in e]))
ref = None
for grp in a.groups:
    wsb = max(dpr_amd.workspace_bytes("pullback", grid, a.P, B, 3, dt, "tiled", max_pose_group=grp), 16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    f = t_ms(lambda: dpr_amd.raster_(out, tp, R, t, algo="tiled", workspace=ws, max_pose_group=grp))
    pb = dpr_amd.raster_pullback_(g, tp, R, t, algo="tiled", workspace=ws, max_pose_group=grp)
    b = t_ms(lambda: dpr_amd.raster_pullback_(g, tp, R, t, algo="tiled", workspace=ws, max_pose_group=grp))
    chk = (float(out.double().sum()), float(pb.points.double().abs().sum()))
    print(f"P={a.P} grid={grid} B={B} group<={grp:2d}: fwd {f / B * 1e3:7.1f} us/pose  bwd {b / B * 1e3:7.1f} us/pose  "
          f"({a.P * B / (f + b) / 1e6:6.1f} G point-poses/s fwd+bwd)  ws {wsb / 2**20:.0f} MiB  chk {chk[0]:.6e} {chk[1]:.6e}", flush=True)
    del ws
