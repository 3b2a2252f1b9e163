"""Seeded fuzz of DPR_FLAG_KEEP_BINNING / DPR_FLAG_REUSE_BINNING on the tiled path with BATCHES
(every pose keeps its binning; one un-permute pass over the batch; sorted copy reused on grids of
more than 4096 tiles) and single poses, HIP path vs the oracle.  Usage: fuzz_share.py [n_seeds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpr_amd
import tests.test_parity_gpu as tp
from tests import data as D
from oracle import oracle
oracle.build()
dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fails = 0
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.default_rng(7000 + seed)
    n_in, n_out = tp.SHAPES[rng.integers(3)]
    npdt, tdt = tp.DTYPES[rng.integers(2)]
    P = int(rng.choice([1, 900, 4097, 50_000, 210_000, 400_000]))
    B = int(rng.choice([1, 2, 3, 5, 9]))
    if n_out == 3:
        grid = tuple(int(x) for x in rng.choice([17, 40, 96, 200, 340], size=3))
        while int(np.prod(grid)) * B > 60_000_000:
            grid = tuple(max(8, x // 2) for x in grid)
    else:
        grid = tuple(int(x) for x in rng.choice([9, 64, 300, 1100], size=2))
    coherent = bool(rng.integers(2)) and n_in == 3
    spread = float(rng.choice([0.05, 0.4, 1.2]))
    pts = (spread * rng.normal(size=(P, n_in))).astype(npdt)
    if seed % 6 == 0 and P > 10:
        pts[:: max(1, P // 5)] = np.nan
    R = D.random_rotations(rng, B, n_in)[:, :n_out, :].astype(npdt)
    t = (0.2 * rng.normal(size=(B, n_out))).astype(npdt)
    use = rng.integers(0, 2, size=3).astype(bool)
    bg = rng.normal(size=B).astype(npdt) if use[0] else None
    ow = rng.uniform(0.5, 3, size=B).astype(npdt) if use[1] else None
    pw = rng.uniform(0.1, 2, size=P).astype(npdt) if use[2] else None
    g = np.asfortranarray(rng.normal(size=grid + (B,)).astype(npdt))
    ref_out = oracle.raster(grid, pts, R, t, bg, ow, pw, dtype=npdt, threaded=True)
    ref_pb = oracle.raster_pullback(g, pts, R, t, ow, pw, dtype=npdt)
    T = tp.T
    dp, dpw = T(pts, dev), T(pw, dev)
    perm, kw = None, {}
    if coherent:
        if pw is not None:
            dp, perm, dpw = dpr_amd.sort_points(dp, dpw)
        else:
            dp, perm = dpr_amd.sort_points(dp)
        kw = dict(coherent_points=True)
    try:
        need = max(16, *(dpr_amd.workspace_bytes(op, grid, P, B, n_in, tdt, "tiled", sharing=True, **kw)
                         for op in ("raster", "pullback")))
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        out = dpr_amd.empty_grid(grid, B, tdt, dev)
        dpr_amd.raster_(out, dp, T(R, dev), T(t, dev), T(bg, dev), T(ow, dev), dpw, algo="tiled",
                        workspace=ws, keep_binning=True, **kw)
        pb = dpr_amd.raster_pullback_(tp.grid_to_dev(g, dev), dp, T(R, dev), T(t, dev), T(bg, dev),
                                      T(ow, dev), dpw, algo="tiled", workspace=ws, reuse_binning=True, **kw)
        if perm is not None:
            bp = torch.empty_like(pb.points); bp.index_copy_(0, perm.long(), pb.points)
            bw = torch.empty_like(pb.point_weight); bw.index_copy_(0, perm.long(), pb.point_weight)
            pb = pb._replace(points=bp, point_weight=bw)
        tp._compare(ref_out, ref_pb, out, pb, npdt)
    except (AssertionError, dpr_amd.DprError) as e:
        fails += 1
        print(f"FAIL seed {seed} P={P} B={B} grid={grid} {npdt.__name__} coherent={coherent}: {str(e)[:300]}", flush=True)
    if seed % 10 == 9:
        print(f"seed {seed} done, {time.time() - t0:.0f} s, fails {fails}", flush=True)
print("done, fails =", fails)
sys.exit(1 if fails else 0)
