"""One-GPU timings of the BASELINE.json configs at their one-GPU-of-8 shares (not bench lines; see
profiles/).  C2: 1M pts -> 128^3 f32, 1 pose.  C3: 10M -> 256^3.  C4: 10M pts -> 512^2 f32, the 64
poses one of 8 GPUs owns.  C5: 50M pts -> 512^3 f64, the 8 poses one of 8 GPUs owns.
"sorted" = Hilbert-sorted with dpr_sort_points + DPR_FLAG_COHERENT_POINTS."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from bench import morton_order
from tests import data as D

dev = torch.device("cuda:0")
def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in e]))

def run(name, P, grid, B, dt, order, algo="auto"):
    rng = np.random.default_rng(0)
    npdt = np.float32 if dt == torch.float32 else np.float64
    pts = (0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32))
    n_out = len(grid)
    tp = torch.as_tensor(pts.astype(npdt), device=dev)
    kw = {}
    if order == "sorted":
        tp = dpr_amd.sort_points(tp)[0]
        kw = dict(coherent_points=True)
    R = torch.as_tensor(D.random_rotations(rng, B)[:, :n_out].astype(npdt), device=dev)
    t = torch.as_tensor((0.1 * rng.normal(size=(B, n_out))).astype(npdt), device=dev)
    g = torch.randn((B,) + tuple(reversed(grid)), device=dev, dtype=dt).permute(*reversed(range(n_out + 1)))
    out = dpr_amd.empty_grid(grid, B, dt, dev)
    wsb = max(16, dpr_amd.workspace_bytes("pullback", grid, P, B, 3, dt, algo, **kw), dpr_amd.workspace_bytes("raster", grid, P, B, 3, dt, algo, **kw))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    f = t_ms(lambda: dpr_amd.raster_(out, tp, R, t, algo=algo, workspace=ws, **kw))
    b = t_ms(lambda: dpr_amd.raster_pullback_(g, tp, R, t, algo=algo, workspace=ws, **kw))
    if algo == "auto":
        algo = dpr_amd.resolve_algo("raster", grid, P, B, 3, **kw) + "/" + dpr_amd.resolve_algo("pullback", grid, P, B, 3, **kw)
    pp = P * B
    print(f"{name:38s} {order:6s} algo={algo:15s} fwd {f:9.3f} ms ({pp / f / 1e6:8.2f} G point-poses/s)  bwd {b:9.3f} ms ({pp / b / 1e6:8.2f} G point-poses/s)  workspace {wsb / 2**20:7.0f} MiB", flush=True)
    del out, g, ws, tp
    torch.cuda.empty_cache()

only = sys.argv[1:]  # e.g. "C5" "C4": run only the lines whose name starts with one of these
_run = run
def run(name, *a, **k):
    if not only or any(name.startswith(o) for o in only):
        _run(name, *a, **k)
for order in ("random", "sorted"):
    run("C2 1M -> 128^3 f32, B=1", 1_000_000, (128,) * 3, 1, torch.float32, order)
    run("C3 10M -> 256^3 f32, B=1", 10_000_000, (256,) * 3, 1, torch.float32, order)
    run("C4 10M -> 512^2 f32, B=64 (1 GPU of 8)", 10_000_000, (512, 512), 64, torch.float32, order)
    run("C4 same, algo=tiled", 10_000_000, (512, 512), 64, torch.float32, order, "tiled")
    run("C5 50M -> 512^3 f64, B=8 (1 GPU of 8)", 50_000_000, (512,) * 3, 8, torch.float64, order)
    run("C5 50M -> 512^3 f64, B=1", 50_000_000, (512,) * 3, 1, torch.float64, order)
    if order == "sorted":  # the 3-D DPR_ALGO_CHUNKED paths on their intended input
        run("C3 same, algo=chunked", 10_000_000, (256,) * 3, 1, torch.float32, order, "chunked")
        run("C5 B=8 same, algo=chunked", 50_000_000, (512,) * 3, 8, torch.float64, order, "chunked")
        run("10M -> 256^3 f32, B=16", 10_000_000, (256,) * 3, 16, torch.float32, order)
        run("10M -> 256^3 f32, B=16, algo=tiled", 10_000_000, (256,) * 3, 16, torch.float32, order, "tiled")
