"""Times single calls: python tools/time_call.py "P,grid,B,op,algo[,dtype[,order[,cloud]]]" ...
grid like 128x128x128; op fwd|bwd; algo auto|atomic|tiled|chunked; dtype f32|f64; order random|sorted;
cloud gauss (0.4 sigma) | tight (0.1 sigma) | uniform.
Prints one line per spec (median of 9 after a warm-up, hipEvents on the current stream).  A/B against
another build with DPR_LIB_OVERRIDE=<libdpr variant>."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
from tests import data as D

dev = torch.device("cuda:0")

def t_ms(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

for spec in sys.argv[1:]:
    f = spec.split(",")
    P, grid, B, op, algo = int(float(f[0])), tuple(int(x) for x in f[1].split("x")), int(f[2]), f[3], f[4]
    dt = torch.float64 if len(f) > 5 and f[5] == "f64" else torch.float32
    order = f[6] if len(f) > 6 else "random"
    npdt = np.float64 if dt == torch.float64 else np.float32
    rng = np.random.default_rng(0)
    n_out = len(grid)
    cloud = f[7] if len(f) > 7 else "gauss"
    if cloud == "uniform":
        pts = rng.uniform(-0.55, 0.55, size=(P, 3)).astype(npdt)
    else:
        pts = ((0.1 if cloud == "tight" else 0.4) * rng.standard_normal(size=(P, 3))).astype(npdt)
    tp = torch.as_tensor(pts, device=dev)
    if order == "sorted":
        tp = dpr_amd.sort_points(tp)[0]
    R = torch.as_tensor(D.random_rotations(rng, B)[:, :n_out].astype(npdt), device=dev)
    t = torch.as_tensor((0.1 * rng.standard_normal(size=(B, n_out))).astype(npdt), device=dev)
    ws = torch.empty(max(16, dpr_amd.workspace_bytes("pullback", grid, P, B, 3, dt, algo)), dtype=torch.uint8, device=dev)
    if op == "fwd":
        out = dpr_amd.empty_grid(grid, B, dt, dev)
        ms = t_ms(lambda: dpr_amd.raster_(out, tp, R, t, algo=algo, workspace=ws))
    else:
        g = torch.randn((B,) + tuple(reversed(grid)), device=dev, dtype=dt).permute(*reversed(range(n_out + 1)))
        ms = t_ms(lambda: dpr_amd.raster_pullback_(g, tp, R, t, algo=algo, workspace=ws))
    res = dpr_amd.resolve_algo("raster" if op == "fwd" else "pullback", grid, P, B, 3) if algo == "auto" else algo
    print(f"{spec:48s} -> {res:8s} {ms:9.4f} ms  ({P * B / ms * 1e-6:9.1f} G point-poses/s)", flush=True)
