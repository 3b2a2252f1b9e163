#!/usr/bin/env python3
"""Probe of the owner-computes 3-D path (DPR_ALGO_CHUNKED on 3-D grids): parity against the tiled
path on the same Hilbert-sorted cloud and per-stage times.
  python tools/own_probe.py [--P 10000000] [--grid 256] [--dist gauss|uniform|tight] [--f64] [--poses 1]"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpr_amd
from tests import data as D

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=10_000_000)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--dist", default="gauss")
ap.add_argument("--f64", action="store_true")
ap.add_argument("--poses", type=int, default=1)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--pw", action="store_true")
ap.add_argument("--no-sort", action="store_true")
ap.add_argument("--bwd", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
tdt = torch.float64 if a.f64 else torch.float32
rng = np.random.default_rng(0)
pts = 0.4 * rng.standard_normal(size=(a.P, 3), dtype=np.float32)
if a.dist == "uniform":
    pts = (1.1 * rng.random(size=(a.P, 3), dtype=np.float32) - 0.55)
if a.dist == "tight":
    pts = 0.1 * rng.standard_normal(size=(a.P, 3), dtype=np.float32)
prng = np.random.default_rng(1)
R = D.random_rotations(prng, a.poses, 3)
t = 0.1 * prng.normal(size=(a.poses, 3))
points = torch.as_tensor(pts, device=dev).to(tdt)
pw = None
if a.pw:
    pw = torch.rand(a.P, device=dev, dtype=tdt)
if not a.no_sort:
    if pw is None:
        points = dpr_amd.sort_points(points)[0]
    else:
        points, _, pw = dpr_amd.sort_points(points, pw)
grid = (a.grid,) * 3
single = a.poses == 1
Rt = torch.as_tensor(R[0] if single else R, device=dev).to(tdt)
tt = torch.as_tensor(t[0] if single else t, device=dev).to(tdt)
res = {}
outs = {}
ap_algos = [("chunked", {}), ("tiled", dict(coherent_points=True))]
if os.environ.get("PROBE_ATOMIC"):
    ap_algos.append(("atomic", {}))
for algo, kw in ap_algos:
    ws = torch.empty(max(16, dpr_amd.workspace_bytes("pullback", grid, a.P, a.poses, 3, tdt, algo, **kw),
                         dpr_amd.workspace_bytes("raster", grid, a.P, a.poses, 3, tdt, algo, **kw)),
                     dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(grid, None if single else a.poses, tdt, dev)
    f = lambda: dpr_amd.raster_(out, points, Rt, tt, None, None, pw, algo=algo, workspace=ws, **kw)
    f(); torch.cuda.synchronize()
    outs[algo] = out.clone()
    if algo == "chunked" and os.environ.get("DPR_LIB_OVERRIDE", "").endswith("stats.so"):
        c = ws[0:128].cpu().numpy().view(np.uint32)
        res["stats"] = {"list_entries": int(c[0]), "slabs": int(c[1]), "split_tiles": int(c[2]), "overflow": int(c[3]),
                        "visits": int(c[4]), "touching": int(c[5]), "batches": int(c[6]), "lanes_taken": int(c[7]),
                        "items_with_work": int(c[8]), "l0_test_rounds": int(c[9]),
                        "sum_item_us": c[10] * 0.01, "max_item_us": c[11] * 0.01,
                        "phase_sum_us": {"setup": c[12] * 0.01, "zero+barrier": c[13] * 0.01,
                                         "wave0_main": c[14] * 0.01, "wave0_wait_pool_barrier": c[15] * 0.01},
                        "visits_per_point": c[4] / a.P, "touch_per_point": c[5] / a.P,
                        "lane_util": c[7] / max(1, 64 * c[6]), "buckets": [int(x) for x in c[16:32]]}
        # per-item records at the end of the slab area (= end of the raster workspace layout)
        need = dpr_amd.workspace_bytes("raster", grid, a.P, a.poses, 3, tdt, algo)
        nit = int(sum(c[16:32]))
        rec = ws[need - 32 * nit: need].cpu().numpy().view(np.uint32).reshape(-1, 8)[::-1].astype(np.int64)
        dt, vis, t0 = rec[:, 0] * 0.01, rec[:, 1], rec[:, 4]
        res.setdefault("phases2", {})["pool+final_sum_us"] = float(rec[:, 6].sum() * 0.01)
        res["phases2"]["setup_sum_us"] = float(rec[:, 7].sum() * 0.01)
        t0 = (t0 - t0.min()) * 0.01
        order = np.argsort(-dt)
        res["items"] = {"n": nit, "sum_us": float(dt.sum()), "max_us": float(dt.max()),
                        "span_us": float((t0 + dt).max()),
                        "top": [[float(dt[i]), int(vis[i]), int(rec[i, 2]), int(rec[i, 3] & 0xffff), int(rec[i, 3] >> 16), int(rec[i, 5]), float(t0[i])] for i in order[:12]],
                        "us_per_kvisit_median": float(np.median(dt[vis > 2000] / (vis[vis > 2000] / 1000.0))),
                        "start_hist_us": np.histogram(t0, bins=10)[0].tolist(),
                        "end_hist_us": np.histogram(t0 + dt, bins=10)[0].tolist()}
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for e0, e1 in evs:
        e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    res[algo] = {"fwd_ms": float(np.median([e0.elapsed_time(e1) for e0, e1 in evs])), "ws_MB": ws.numel() / 1e6}
    if single:
        name = "tiled_local" if algo == "tiled" else algo
        st = dpr_amd.stage_times(f, "raster", name, 10)
        res[algo]["stages"] = {k: round(v, 4) for k, v in st.items()}
    if a.bwd:
        g = torch.randn(tuple(reversed(grid)) if single else (a.poses,) + tuple(reversed(grid)), device=dev, dtype=tdt,
                        generator=torch.Generator(device=dev).manual_seed(2))
        g = g.permute(*reversed(range(g.ndim)))
        dp = torch.empty(a.P, 3, device=dev, dtype=tdt); dw = torch.empty(a.P, device=dev, dtype=tdt)
        fb = lambda: dpr_amd.raster_pullback_(g, points, Rt, tt, None, None, pw, ds_dpoints=dp, ds_dpoint_weight=dw,
                                              algo=algo, workspace=ws, **kw)
        r = fb(); torch.cuda.synchronize()
        outs[algo + "_pb"] = [x.clone() for x in r]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for e0, e1 in evs:
            e0.record(); fb(); e1.record()
        torch.cuda.synchronize()
        res[algo]["bwd_ms"] = float(np.median([e0.elapsed_time(e1) for e0, e1 in evs]))
        if single:
            st = dpr_amd.stage_times(fb, "pullback", "tiled_local" if algo == "tiled" else algo, 10)
            res[algo]["bwd_stages"] = {k: round(v, 4) for k, v in st.items()}
    del ws
d = (outs["chunked"].double() - outs["tiled"].double())
res["max_abs_diff_fwd"] = float(d.abs().max()); res["rel_l2_fwd"] = float(d.norm() / outs["tiled"].double().norm())
res["bit_equal_fwd"] = bool(torch.equal(outs["chunked"], outs["tiled"]))
if a.bwd:
    names = ["points", "rotation", "translation", "background", "out_weight", "point_weight"]
    for n, x, y in zip(names, outs["chunked_pb"], outs["tiled_pb"]):
        res["rel_" + n] = float((x.double() - y.double()).norm() / max(float(y.double().norm()), 1e-300))
print(json.dumps(res, indent=1))
