"""Seeded synthetic fixtures with the distributions of the reference's test/data.jl
(which sets no RNG seed, so only the distributions can be reproduced):
points 0.4*randn (test/data.jl:22,27); rotations uniform on SO(3) (:29-30);
projections = first two rows (:13-16,:43); translations 0.1*randn (:54-56,:63);
backgrounds 1..B (:72); weights 10*rand (:75); point weights normalised rand (:77-84);
ds_dout randn (test/cuda.jl:46).  B chosen uneven w.r.t. a worker count like
batch_size_for_test (:5-11)."""
from types import SimpleNamespace

import numpy as np


def uneven_batch(n_workers: int) -> int:
    b = n_workers + 1
    while n_workers > 1 and b % n_workers == 0:
        b += 1
    return b


def random_rotations(rng, batch, n=3):
    if n == 1:  # "rotations" of the line: +-1
        return rng.choice([-1.0, 1.0], size=(batch, 1, 1))
    if n == 2:
        th = rng.uniform(0, 2 * np.pi, batch)
        return np.stack([np.stack([np.cos(th), -np.sin(th)], -1),
                         np.stack([np.sin(th), np.cos(th)], -1)], -2)
    q = rng.normal(size=(batch, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.empty((batch, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def random_isometries(rng, batch, n_out, n_in):
    """(batch, n_out, n_in) matrices with orthonormal rows (n_out <= n_in: projections of a rotation) or orthonormal
    columns (n_out > n_in: embeddings) -- the dimension pairs beyond the reference's own data generator."""
    m = max(n_out, n_in)
    q = np.linalg.qr(rng.normal(size=(batch, m, m)))[0]
    return q[:, :n_out, :n_in]


def make(n_points=10, n_in=3, n_out=3, batch=3, grid_n=8, seed=0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    points = 0.4 * rng.normal(size=(n_points, n_in))
    if n_in > 3 or n_out > n_in:
        rot = random_isometries(rng, batch, n_out, n_in)
    else:
        rot = random_rotations(rng, batch, n_in)[:, :n_out, :]
    trans = 0.1 * rng.normal(size=(batch, n_out))
    backgrounds = np.arange(1, batch + 1, dtype=np.float64)
    weights = 10 * rng.uniform(size=batch)
    pw = rng.uniform(size=n_points)
    pw = pw / max(pw.sum(), 1e-300)
    grid = tuple(grid_n) if isinstance(grid_n, (tuple, list)) else (grid_n,) * n_out
    assert len(grid) == n_out
    ds_dout = np.asfortranarray(rng.normal(size=grid + (batch,)))
    c = lambda a: np.ascontiguousarray(a, dtype=dtype)
    return SimpleNamespace(points=c(points), rotations=c(rot), translations=c(trans),
                           backgrounds=c(backgrounds), weights=c(weights), point_weights=c(pw),
                           grid=grid, ds_dout=np.asfortranarray(ds_dout, dtype=dtype),
                           batch=batch, n_in=n_in, n_out=n_out, n_points=n_points)


def isapprox(a, b, rtol=None):
    """Julia's `≈` for arrays: norm(a-b) <= rtol*max(norm(a),norm(b)),
    rtol = sqrt(eps(T)) (test/util.jl:13,22)."""
    a = np.asarray(a); b = np.asarray(b)
    if rtol is None:
        rtol = float(np.sqrt(np.finfo(np.result_type(a, b)).eps))
    na, nb = np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) <= rtol * max(na, nb)
