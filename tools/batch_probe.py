"""Pullback over a batch of poses of a Hilbert-sorted cloud on a 3-D grid (DPR_FLAG_COHERENT_POINTS): the direct
kernels of DPR_ALGO_CHUNKED (fp32: pose loop inside; fp64: a launch per pose) against the tiled pipeline and the
direct kernel of DPR_ALGO_ATOMIC.  Usage: batch_probe.py [--f64]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr
dev = "cuda"
dt = torch.float64 if "--f64" in sys.argv else torch.float32

def run(pts, n, B, algo):
    P = pts.shape[0]
    g0 = torch.Generator(device=dev); g0.manual_seed(B)
    R = torch.linalg.qr(torch.randn(B, 3, 3, device=dev, dtype=dt, generator=g0))[0]
    t = 0.05 * torch.randn(B, 3, device=dev, dtype=dt, generator=g0)
    g = torch.randn(B, n, n, n, device=dev, dtype=dt, generator=g0).permute(3, 2, 1, 0)  # grid layout, batch last
    kw = dict(algo=algo, coherent_points=True)
    ws = torch.empty(max(16, dpr.workspace_bytes("pullback", (n, n, n), P, B, 3, dt, algo, coherent_points=True)),
                     dtype=torch.uint8, device=dev)
    f = lambda: dpr.raster_pullback_(g, pts, R, t, None, None, workspace=ws, **kw)
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 5
    e0.record()
    for _ in range(N): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N

print(f"# {'P':>9s} {'grid':>5s} {'B':>3s} {'chunked':>9s} {'tiled':>9s} {'atomic':>9s}  AUTO   ({str(dt)[6:]})", flush=True)
for P in (1_000_000, 3_000_000, 10_000_000):
    g0 = torch.Generator(device=dev); g0.manual_seed(0)
    pts = dpr.sort_points((0.4 * torch.randn(P, 3, device=dev, generator=g0)).to(dt))[0]
    for n in (128, 256):
        for B in (4, 16, 32, 64):
            if n ** 3 * B * (8 if dt == torch.float64 else 4) > 12e9:
                continue
            ts = [run(pts, n, B, a) for a in ("chunked", "tiled", "atomic")]
            auto = dpr.resolve_algo("pullback", (n, n, n), P, B, 3, coherent_points=True)
            print(f"  {P:9d} {n:5d} {B:3d} {ts[0]:9.3f} {ts[1]:9.3f} {ts[2]:9.3f}  {auto}", flush=True)
