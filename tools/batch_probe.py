"""Timing of the batched direct 3-D pullback (DPR_ALGO_CHUNKED, coherent cloud) on a few shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd as dpr
torch.manual_seed(0)
dev = "cuda"

def run(P, n, B, dt, algo="chunked"):
    pts = (0.4 * torch.randn(P, 3, device=dev, dtype=dt)).clamp(-1.2, 1.2)
    pts, _ = dpr.sort_points(pts)
    R = torch.linalg.qr(torch.randn(B, 3, 3, device=dev, dtype=dt))[0]
    t = 0.05 * torch.randn(B, 3, device=dev, dtype=dt)
    bg = torch.zeros(B, device=dev, dtype=dt); ow = torch.ones(B, device=dev, dtype=dt)
    g = torch.randn(B, n, n, n, device=dev, dtype=dt).permute(3, 2, 1, 0)  # grid layout: axis 1 fastest, batch last
    kw = dict(algo=algo, coherent_points=True)
    ws = torch.empty(max(16, dpr.workspace_bytes("pullback", (n, n, n), P, B, 3, dt, algo, coherent_points=True)),
                     dtype=torch.uint8, device=dev)
    f = lambda: dpr.raster_pullback_(g, pts, R, t, bg, ow, workspace=ws, **kw)
    out = f(); torch.cuda.synchronize()
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N = 10
    for _ in range(N): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N, out

for (P, n, B, dt) in [(10_000_000, 256, 4, torch.float32), (10_000_000, 256, 16, torch.float32),
                      (1_000_000, 128, 16, torch.float32), (10_000_000, 256, 4, torch.float64),
                      (50_000_000, 512, 8, torch.float64)]:
    ms, out = run(P, n, B, dt)
    print(f"P={P:>9d} n={n} B={B:>2d} {str(dt)[6:]:8s} chunked pullback {ms:8.3f} ms", flush=True)
    if len(sys.argv) > 1 and P <= 10_000_000:
        ms2, out2 = run(P, n, B, dt, "tiled")
        err = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(out, out2))
        print(f"    tiled {ms2:8.3f} ms   max rel diff {err:.2e}", flush=True)
    del out
    torch.cuda.empty_cache()
