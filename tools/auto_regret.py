"""AUTO regret table: for {Gaussian 0.4 sigma, uniform, clustered 0.1 sigma}
x P in {1e4, 1e5, 1e6, 1e7} x B in {1, 4, 16, 64} x {128^2, 512^2, 1024^2, 128^3, 256^3} (fp32,
3-D points, as-generated order) time every algorithm that applies and look up what
DPR_ALGO_AUTO resolves to; regret = t(AUTO's choice) / t(best).  A second section repeats the
3-D grids with a Hilbert-sorted cloud and DPR_FLAG_COHERENT_POINTS (the regime of the 3-D chunk
lists).  Usage: python tools/auto_regret.py [--quick] > profiles/r03_auto_regret.txt"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpr_amd  # noqa: E402
from tests import data as D  # noqa: E402

dev = torch.device("cuda:0")


def t_ms(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def cloud(kind, P, rng):
    if kind == "gauss0.4":
        return (0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32))
    if kind == "uniform":
        return (1.6 * rng.random(size=(P, 3), dtype=np.float32) - 0.8)
    return (0.1 * rng.standard_normal(size=(P, 3), dtype=np.float32))  # clustered


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    kinds = ["gauss0.4", "uniform", "cluster0.1"]
    Ps = [10_000, 100_000, 1_000_000, 10_000_000]
    Bs = [1, 4, 16, 64]
    grids = [(128, 128), (512, 512), (1024, 1024), (128, 128, 128), (256, 256, 256)]
    if a.quick:
        kinds, Ps, Bs = kinds[:1], [100_000, 1_000_000], [1, 16]
    worst = {"raster": (1.0, ""), "pullback": (1.0, "")}
    print("# times in ms (median of 3); '*' marks what DPR_ALGO_AUTO resolves to; regret = t(AUTO) / t(best)")
    print("# columns: cloud P B grid | forward: atomic tiled chunked regret | pullback: atomic tiled chunked regret")
    for section in ("random", "coherent"):
        print(f"## point order: {section}" + (" (Hilbert-sorted once, DPR_FLAG_COHERENT_POINTS; 3-D grids)" if section == "coherent" else ""))
        for kind in kinds:
            for P in Ps:
                rng = np.random.default_rng(0)
                pts = torch.as_tensor(cloud(kind, P, rng), device=dev)
                kw = {}
                if section == "coherent":
                    pts = dpr_amd.sort_points(pts)[0]
                    kw = dict(coherent_points=True)
                for B in Bs:
                    for grid in grids:
                        n_out = len(grid)
                        if section == "coherent" and n_out == 2:
                            continue
                        prng = np.random.default_rng(1)
                        R = torch.as_tensor(D.random_rotations(prng, B)[:, :n_out].astype(np.float32), device=dev)
                        t = torch.as_tensor((0.1 * prng.normal(size=(B, n_out))).astype(np.float32), device=dev)
                        out = dpr_amd.empty_grid(grid, B, torch.float32, dev)
                        g = torch.randn((B,) + tuple(reversed(grid)), device=dev).permute(*reversed(range(n_out + 1)))
                        res = {}
                        for op in ("raster", "pullback"):
                            auto = dpr_amd.resolve_algo(op, grid, P, B, 3, **kw)
                            times = {}
                            for algo in ("atomic", "tiled", "chunked"):
                                if algo == "chunked" and n_out == 3 and section == "random" and (B < 8 or P < 200_000):
                                    continue  # (without the coherence flag the 3-D paths sort inside the call from 8 poses on)
                                try:
                                    need = dpr_amd.workspace_bytes(op, grid, P, B, 3, torch.float32, algo, **kw)
                                except dpr_amd.DprError:
                                    continue
                                ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
                                if op == "raster":
                                    fn = lambda: dpr_amd.raster_(out, pts, R, t, algo=algo, workspace=ws, **kw)
                                else:
                                    fn = lambda: dpr_amd.raster_pullback_(g, pts, R, t, algo=algo, workspace=ws, **kw)
                                times[algo] = t_ms(fn)
                                del ws
                            best = min(times.values())
                            regret = times[auto] / best
                            tag = f"{kind} P={P} B={B} {'x'.join(map(str, grid))} {section}"
                            if regret > worst[op][0]:
                                worst[op] = (regret, tag)
                            cells = " ".join((f"{times[k]:9.3f}" + ("*" if k == auto else " ")) if k in times else "        - "
                                             for k in ("atomic", "tiled", "chunked"))
                            res[op] = f"{cells} {regret:5.2f}"
                        print(f"{kind:10s} {P:9d} {B:3d} {'x'.join(map(str, grid)):12s} | {res['raster']} | {res['pullback']}", flush=True)
                        del out, g
                del pts
                torch.cuda.empty_cache()
    for op in ("raster", "pullback"):
        print(f"# max regret {op}: {worst[op][0]:.2f} at {worst[op][1]}")


if __name__ == "__main__":
    main()
