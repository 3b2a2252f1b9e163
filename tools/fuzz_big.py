"""Larger seeded fuzz around the AUTO thresholds (P up to 1.2M, batches up to 20, all optional
arguments), HIP path vs the oracle; complements tests/test_parity_gpu.py's small-P fuzz."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpr_amd
import tests.test_parity_gpu as tp
from tests import data as D
from oracle import oracle
oracle.build()
dev = torch.device("cuda:0")
fails = 0
t0 = time.time()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for seed in range(n_seeds):
    rng = np.random.default_rng(5000 + seed)
    n_in, n_out = tp.SHAPES[rng.integers(3)]
    npdt, tdt = tp.DTYPES[rng.integers(2)]
    algo = ["auto", "tiled", "atomic"][rng.integers(3)]
    P = int(rng.choice([70_000, 260_000, 333_333, 650_000, 1_200_000]))
    B = int(rng.choice([1, 2, 5, 8, 20]))
    if P * B > 6_000_000:
        B = max(1, 6_000_000 // P)
    grid = tuple(int(g) for g in rng.integers(40, 300 if n_out == 2 else 140, size=n_out))
    spread = float(rng.choice([0.1, 0.4, 0.8]))
    pts = (spread * rng.normal(size=(P, n_in))).astype(npdt)
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
    try:
        out = dpr_amd.raster(grid, T(pts, dev), T(R, dev), T(t, dev), T(bg, dev), T(ow, dev), T(pw, dev), algo=algo)
        pb = dpr_amd.raster_pullback_(tp.grid_to_dev(g, dev), T(pts, dev), T(R, dev), T(t, dev), T(bg, dev), T(ow, dev), T(pw, dev), algo=algo)
        tp._compare(ref_out, ref_pb, out.reshape(ref_out.shape) if out.ndim != ref_out.ndim else out,
                    type(pb)(*[x.reshape(np.shape(r)) for x, r in zip(pb, ref_pb)]), npdt)
    except AssertionError as e:
        fails += 1
        print("FAIL seed", seed, (n_in, n_out), npdt.__name__, algo, P, B, grid, str(e)[:160], flush=True)
    if seed % 10 == 9:
        print(f"  {seed + 1} seeds, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print("done, fails =", fails)
