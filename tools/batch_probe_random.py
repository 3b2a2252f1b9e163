"""Forward and pullback over a batch of poses of a cloud in AS-GENERATED order on a 3-D grid: DPR_ALGO_CHUNKED (sorts
inside the call from 8 poses on) against the tiled pipeline and the ATOMIC kernel.
Usage: batch_probe_random.py [--f64] [--fwd]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr
dev = "cuda"
dt = torch.float64 if "--f64" in sys.argv else torch.float32

FWD = "--fwd" in sys.argv
def run(pts, n, B, algo):
    P = pts.shape[0]
    g0 = torch.Generator(device=dev); g0.manual_seed(B)
    R = torch.linalg.qr(torch.randn(B, 3, 3, device=dev, dtype=dt, generator=g0))[0]
    t = 0.05 * torch.randn(B, 3, device=dev, dtype=dt, generator=g0)
    g = torch.randn(B, n, n, n, device=dev, dtype=dt, generator=g0).permute(3, 2, 1, 0)
    op = "raster" if FWD else "pullback"
    ws = torch.empty(max(16, dpr.workspace_bytes(op, (n, n, n), P, B, 3, dt, algo)), dtype=torch.uint8, device=dev)
    if FWD:
        f = lambda: dpr.raster_(g, pts, R, t, None, None, workspace=ws, algo=algo)  # (g doubles as `out`)
    else:
        f = lambda: dpr.raster_pullback_(g, pts, R, t, None, None, workspace=ws, algo=algo)
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 5
    e0.record()
    for _ in range(N): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N

print(f"# {'forward' if FWD else 'pullback'}: {'P':>9s} {'grid':>5s} {'B':>3s} {'chunked':>9s} {'tiled':>9s} {'atomic':>9s}  AUTO   ({str(dt)[6:]}, as-generated order)", flush=True)
SPARSE = "--sparse" in sys.argv  # (forward: clouds sparse on the grid -- the chunk lists behind the sort)
for P in ((300_000, 1_000_000) if SPARSE else (1_000_000, 3_000_000, 10_000_000)):
    g0 = torch.Generator(device=dev); g0.manual_seed(0)
    pts = (0.4 * torch.randn(P, 3, device=dev, generator=g0)).to(dt)
    if "--uniform" in sys.argv:
        pts = (1.1 * torch.rand(P, 3, device=dev, generator=g0) - 0.55).to(dt)
    if "--tight" in sys.argv:
        pts = (0.1 * torch.randn(P, 3, device=dev, generator=g0)).to(dt)
    for n in ((256, 384) if SPARSE else (128, 256)):
        for B in (8, 16, 32, 64):
            if n ** 3 * B * (8 if dt == torch.float64 else 4) > 12e9:
                continue
            ts = [run(pts, n, B, a) for a in ("chunked", "tiled", "atomic")]
            print(f"  {P:9d} {n:5d} {B:3d} {ts[0]:9.3f} {ts[1]:9.3f} {ts[2]:9.3f}  {dpr.resolve_algo('raster' if FWD else 'pullback', (n, n, n), P, B, 3)}", flush=True)
