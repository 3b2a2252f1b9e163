"""The rows of the reference's own "Timings" table (/root/reference README.md:185-197; copied as
data into BASELINE.md section 1) on this GPU: `raster` and `raster_pullback!` with algo="auto",
ms per call, next to the published A100 / 8-thread CPU figures.  The README gives neither element
type nor point distribution; both fp32 and fp64 are run, points 0.4*N(0,I), random rotations."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D

dev = torch.device("cuda:0")
def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in e]))

# points, images, grid, published ms: (cpu1 fwd, cpu8 fwd, a100 fwd, cpu1 bwd, cpu8 bwd, a100 bwd)
ROWS = [(10_000, 64, (128, 128), (341, 73, 15, 37, 10, 1)),
        (10_000, 64, (1024, 1024), (387, 101, 16, 78, 24, 2)),
        (100_000, 64, (128, 128), (3313, 741, 153, 374, 117, 9)),
        (100_000, 64, (1024, 1024), (3499, 821, 154, 469, 173, 10)),
        (100_000, 1, (1024, 1024, 1024), (493, 420, 24, 265, 269, 17))]
print(f"{'points':>8s} {'images':>6s} {'grid':>8s} {'dtype':>5s} {'algo':>15s} | {'fwd ms':>9s} {'A100':>6s} {'CPUx8':>6s} | {'bwd ms':>9s} {'A100':>6s} {'CPUx8':>6s}")
for P, B, grid, pub in ROWS:
    for dt, npdt in ((torch.float32, np.float32), (torch.float64, np.float64)):
        rng = np.random.default_rng(0)
        n_out = len(grid)
        pts = torch.as_tensor((0.4 * rng.standard_normal(size=(P, 3))).astype(npdt), device=dev)
        R = torch.as_tensor(D.random_rotations(rng, B)[:, :n_out].astype(npdt), device=dev)
        t = torch.as_tensor((0.1 * rng.normal(size=(B, n_out))).astype(npdt), device=dev)
        out = dpr_amd.empty_grid(grid, B, dt, dev)
        g = dpr_amd.empty_grid(grid, B, dt, dev)
        g.normal_()
        wsb = max(16, *(dpr_amd.workspace_bytes(op, grid, P, B, 3, dt, "auto") for op in ("raster", "pullback")))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        f = t_ms(lambda: dpr_amd.raster_(out, pts, R, t, algo="auto", workspace=ws))
        b = t_ms(lambda: dpr_amd.raster_pullback_(g, pts, R, t, algo="auto", workspace=ws))
        algo = dpr_amd.resolve_algo("raster", grid, P, B, 3) + "/" + dpr_amd.resolve_algo("pullback", grid, P, B, 3)
        gs = "x".join(str(x) for x in grid[:1]) + ("^%d" % n_out)
        print(f"{P:8d} {B:6d} {gs:>8s} {'f32' if dt == torch.float32 else 'f64':>5s} {algo:>15s} | {f:9.3f} {pub[2]:6d} {pub[1]:6d} | {b:9.3f} {pub[5]:6d} {pub[4]:6d}", flush=True)
        del out, g, ws
        torch.cuda.empty_cache()
