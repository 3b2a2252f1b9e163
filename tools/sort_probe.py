"""Time dpr_sort_points (diagnostic)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd
dev = torch.device("cuda:0")
for P in (1_000_000, 10_000_000, 50_000_000):
    pts = 0.4 * torch.randn(P, 3, device=dev)
    dpr_amd.sort_points(pts); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in e:
        a.record(); dpr_amd.sort_points(pts); b.record()
    torch.cuda.synchronize()
    print(f"P={P}: sort_points {np.median([a.elapsed_time(b) for a, b in e]):.3f} ms (includes torch allocations)")
