import sys, os, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, dpr_amd, bench
dev = torch.device("cuda:0")
pts = torch.as_tensor(bench.synth_points("C2"), device=dev)
R, t = bench.synth_poses("C2", 1, seed=1)
R = torch.as_tensor(R[0], device=dev); t = torch.as_tensor(t[0], device=dev)
grid = (128,)*3
out = dpr_amd.empty_grid(grid, None, torch.float32, dev)
ws = torch.empty(dpr_amd.workspace_bytes("raster", grid, 1_000_000, 1, 3, torch.float32, "tiled"), dtype=torch.uint8, device=dev)
def step(): dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws)
for _ in range(2000): step()
torch.cuda.synchronize()
for mode in ("gc on", "gc off"):
    if mode == "gc off": gc.disable()
    ts = []
    for l in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, "loops ms:", " ".join(f"{x:.2f}" for x in ts))
    # per-step CPU enqueue time
    t0 = time.perf_counter()
    for _ in range(300): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(mode, f"enqueue {1e3*(t1-t0)/300:.4f} ms/step, total {1e3*(t2-t0)/300:.4f} ms/step")
