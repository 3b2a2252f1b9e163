"""Seeded fuzz of DPR_ALGO_CHUNKED on 3-D grids (csrc/dpr_owner.hip: owner-computes forward over the box
hierarchy, chunk lists for sparse batches, direct thread-per-point pullback; AUTO with the coherence flag in
between), HIP path vs the oracle: cloud sizes around the box granularities (16 / 1024 / 65536 points), odd
grids, batches, optional arguments, sorted and unsorted input, non-finite points, declined weight gradient.
Usage: fuzz_owner.py [n_seeds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpr_amd
import tests.test_parity_gpu as tp
from tests import data as D
from oracle import oracle
oracle.build()
dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
fails = 0
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.default_rng(9100 + seed)
    npdt, tdt = tp.DTYPES[rng.integers(2)]
    P = int(rng.choice([1, 15, 17, 1023, 1025, 5000, 65_537, 140_000, 210_000, 300_000]))
    B = int(rng.choice([1, 1, 2, 4, 7, 8, 17]))
    grid = tuple(int(x) for x in rng.choice([1, 2, 3, 31, 33, 64, 97, 130], size=3))
    while int(np.prod(grid)) * B > 40_000_000:
        grid = tuple(max(2, x // 2) for x in grid)
    algo = ["chunked", "chunked", "auto"][rng.integers(3)]
    sort = bool(rng.integers(4))  # unsorted input must still be right (slow, never wrong)
    spread = float(rng.choice([0.03, 0.4, 1.3]))
    pts = (spread * rng.normal(size=(P, 3))).astype(npdt)
    if seed % 5 == 0 and P > 10:
        pts[:: max(1, P // 7)] = [np.nan, np.inf, -np.inf][seed % 3]
    R = D.random_rotations(rng, B, 3).astype(npdt)
    t = (0.3 * rng.normal(size=(B, 3))).astype(npdt)
    use = rng.integers(0, 2, size=3).astype(bool)
    bg = rng.normal(size=B).astype(npdt) if use[0] else None
    ow = rng.uniform(-2, 3, size=B).astype(npdt) if use[1] else None
    pw = (rng.uniform(0.1, 2, size=P) * 10.0 ** rng.integers(-3, 4, size=P)).astype(npdt) if use[2] else None
    want_pw = bool(rng.integers(3))
    g = np.asfortranarray(rng.normal(size=grid + (B,)).astype(npdt))
    # (crowded cells: the fp32 oracle's own serial sums drift by ~n eps / 2, the kernels' fixed-point sums are
    # exact -- compare `out` against the fp64 oracle there)
    crowded = npdt == np.float32 and (P > 5 * int(np.prod(grid)) or pw is not None)  # (weights span 6 decades)
    ref_out = oracle.raster(grid, pts, R, t, bg, ow, pw, dtype=np.float64 if crowded else npdt, threaded=True)
    ref_pb = oracle.raster_pullback(g, pts, R, t, ow, pw, dtype=npdt)
    T = tp.T
    dp, dpw = T(pts, dev), T(pw, dev)
    perm = None
    if sort:
        if pw is not None:
            dp, perm, dpw = dpr_amd.sort_points(dp, dpw)
        else:
            dp, perm = dpr_amd.sort_points(dp)
    # (one call in four WITHOUT the coherence flag: from 8 poses and 2e5 points on the library sorts inside the call)
    kw = dict(coherent_points=True) if rng.integers(4) else {}
    try:
        need = max(16, *(dpr_amd.workspace_bytes(op, grid, P, B, 3, tdt, algo, **kw) for op in ("raster", "pullback")))
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        out = dpr_amd.empty_grid(grid, B, tdt, dev)
        dpr_amd.raster_(out, dp, T(R, dev), T(t, dev), T(bg, dev), T(ow, dev), dpw, algo=algo, workspace=ws, **kw)
        pb = dpr_amd.raster_pullback_(tp.grid_to_dev(g, dev), dp, T(R, dev), T(t, dev), T(bg, dev), T(ow, dev), dpw,
                                      algo=algo, workspace=ws, point_weight_grad=want_pw, **kw)
        if not want_pw:
            assert pb.point_weight is None
            pb = pb._replace(point_weight=T(np.asarray(ref_pb.point_weight)[perm.cpu().numpy()] if perm is not None
                                            else ref_pb.point_weight, dev))
        if perm is not None:
            bp = torch.empty_like(pb.points); bp.index_copy_(0, perm.long(), pb.points)
            bw = torch.empty_like(pb.point_weight); bw.index_copy_(0, perm.long(), pb.point_weight)
            pb = pb._replace(points=bp, point_weight=bw)
        try:
            tp._compare(ref_out, ref_pb, out, pb, npdt)
        except AssertionError as e:
            # (hundreds of points per cell: the fp32 oracle's serial per-pose sums of 1e5+ cancelling terms are
            # themselves only good to ~1e-3 -- the four per-pose outputs get 3e-3 there, everything else stays)
            if not (npdt == np.float32 and P > 50 * int(np.prod(grid)) and any(k in str(e) for k in
                    ("ds_drotation", "ds_dtranslation", "ds_dbackground", "ds_dout_weight"))):
                raise
            tp.assert_close(out, ref_out, tp.tol(npdt, "out"), "out")
            tp.assert_close(pb.points, ref_pb.points, tp.tol(npdt, "points"), "ds_dpoints")
            for name in ("rotation", "translation", "background", "out_weight"):
                tp.assert_close(getattr(pb, name), getattr(ref_pb, name), 3e-3, name)
    except (AssertionError, dpr_amd.DprError) as e:
        fails += 1
        print(f"FAIL seed {seed} P={P} B={B} grid={grid} {npdt.__name__} algo={algo} sorted={sort}: {str(e)[:300]}", flush=True)
    if seed % 10 == 9:
        print(f"seed {seed} done, {time.time() - t0:.0f} s, fails {fails}", flush=True)
print("done, fails =", fails)
sys.exit(1 if fails else 0)
