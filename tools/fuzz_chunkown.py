"""Seeded fuzz of DPR_ALGO_CHUNKED on 2-D grids (chunk-owned tiles, dpr_chunkown.hip) against the
oracle: cloud sizes around the chunk / slice boundaries, anisotropic images from tiny to sparse
(wide footprints, the work-list kernel), 1 ... 70 poses (pose slices), both point dimensions and
element types, optional arguments, pre-sorted input with the coherence flag, keep / reuse."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpr_amd
import tests.test_parity_gpu as tp
from tests import data as D
from oracle import oracle
oracle.build()
dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fails = 0
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.default_rng(9000 + seed)
    n_in = int(rng.choice([2, 3]))
    npdt, tdt = tp.DTYPES[rng.integers(2)]
    P = int(rng.choice([1, 700, 4095, 4096, 4097, 8193, 30_000, 120_001, 300_000]))
    B = int(rng.choice([1, 2, 5, 16, 33, 64, 70]))
    if P * B > 6_000_000:
        B = max(1, 6_000_000 // P)
    grid = (int(rng.choice([3, 17, 64, 150, 512, 900])), int(rng.choice([4, 31, 128, 333, 700])))
    spread = float(rng.choice([0.02, 0.4, 1.5]))
    pts = (spread * rng.normal(size=(P, n_in))).astype(npdt)
    if seed % 5 == 0 and P > 10:
        pts[:: max(1, P // 7)] = np.nan
    R = D.random_rotations(rng, B, n_in)[:, :2, :].astype(npdt)
    if seed % 4 == 0:
        R *= float(rng.choice([0.3, 2.5]))  # not a rotation: the footprint bound must still hold
    t = (float(rng.choice([0.0, 0.2, 1.1])) * rng.normal(size=(B, 2))).astype(npdt)
    use = rng.integers(0, 2, size=3).astype(bool)
    bg = rng.normal(size=B).astype(npdt) if use[0] else None
    ow = rng.uniform(0.5, 3, size=B).astype(npdt) if use[1] else None
    pw = rng.uniform(0.1, 2, size=P).astype(npdt) if use[2] else None
    g = np.asfortranarray(rng.normal(size=grid + (B,)).astype(npdt))
    mode = ["plain", "coherent", "keep_reuse"][seed % 3]
    try:
        ref_out = oracle.raster(grid, pts, R, t, bg, ow, pw, dtype=npdt)
        ref_pb = oracle.raster_pullback(g, pts, R, t, ow, pw, dtype=npdt)
        dp, dR, dt_, dbg, dow, dpw = (tp.T(x, dev) for x in (pts, R, t, bg, ow, pw))
        perm = None
        kw = {}
        if mode == "coherent":
            if pw is not None:
                dp, perm, dpw = dpr_amd.sort_points(dp, dpw)
            else:
                dp, perm = dpr_amd.sort_points(dp)
            kw = dict(coherent_points=True)
        need = max(16, *(dpr_amd.workspace_bytes(op, grid, P, B, n_in, tdt, "chunked", **kw) for op in ("raster", "pullback")))
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        out = dpr_amd.empty_grid(grid, B, tdt, dev)
        dpr_amd.raster_(out, dp, dR, dt_, dbg, dow, dpw, algo="chunked", workspace=ws,
                        keep_binning=(mode == "keep_reuse"), **kw)
        pb = dpr_amd.raster_pullback_(tp.grid_to_dev(g, dev), dp, dR, dt_, dbg, dow, dpw, algo="chunked",
                                      workspace=ws, reuse_binning=(mode == "keep_reuse"), **kw)
        if perm is not None:
            bp = torch.empty_like(pb.points); bp.index_copy_(0, perm.long(), pb.points)
            bw = torch.empty_like(pb.point_weight); bw.index_copy_(0, perm.long(), pb.point_weight)
            pb = pb._replace(points=bp, point_weight=bw)
        # NaN points: zero gradient in both
        tp._compare(ref_out, ref_pb, out, pb, npdt)
    except AssertionError as e:
        fails += 1
        print("FAIL seed", seed, dict(P=P, B=B, grid=grid, n_in=n_in, dt=str(tdt), mode=mode, spread=spread), str(e)[:300], flush=True)
    if seed % 20 == 19:
        print(f"seed {seed} done, {time.time() - t0:.0f} s, fails {fails}", flush=True)
print("done, fails =", fails)
sys.exit(1 if fails else 0)
