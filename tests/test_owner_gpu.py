"""GPU parity tests of the 3-D DPR_ALGO_CHUNKED paths (csrc/dpr_owner.hip, csrc/dpr_chunked.hip)
against the CPU oracle: owner-computes tiles over the box hierarchy (forward), chunk lists for
sparse clouds over several poses (forward), the direct thread-per-point pullback.  Edge cases of the
machinery: clouds that are not a multiple of 16 / 1024 points, heavy tiles split into parts and
combined from slabs, candidate lists that overflow their slot (every chunk becomes a candidate),
grids one or two cells wide, non-finite points and weights, batches beyond one plan group.
Semantics: /root/reference/src/raster.jl:36-66, src/raster_pullback.jl:39-72, :85-148."""
import numpy as np
import pytest
import torch

import dpr_amd
from tests import data as D
from tests.test_parity_gpu import T, assert_close, grid_to_dev, tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    dpr_amd.lib()
    return torch.device("cuda:0")


def hilbert_like_order(pts):
    """A coherent order without the library's sort: lexicographic on coarse cells (any coherent order
    exercises the same paths)."""
    q = np.floor((np.clip(pts, -1, 1) * 0.5 + 0.5) * 64).astype(np.int64)
    return np.lexsort((pts[:, 0], q[:, 0], q[:, 1], q[:, 2]))


def run_case(oracle, dev, npdt, tdt, grid, n_points, batch, order="sorted", pw="rand", seed=0, scale=0.4,
             fwd_tol=None):
    rng = np.random.default_rng(seed)
    d = D.make(n_points=max(n_points, 1), n_in=3, n_out=3, batch=batch, grid_n=grid, seed=seed, dtype=npdt)
    pts = (scale / 0.4 * d.points[:n_points]).astype(npdt)
    if order == "sorted" and n_points > 1:
        pts = np.ascontiguousarray(pts[hilbert_like_order(pts)])
    w = None
    if pw == "rand":
        w = (rng.uniform(0.2, 2.0, size=n_points) * rng.choice([-1.0, 1.0], size=n_points)).astype(npdt)
    ref_out = oracle.raster(d.grid, pts, d.rotations, d.translations, d.backgrounds, d.weights, w, dtype=npdt)
    ref_pb = oracle.raster_pullback(d.ds_dout, pts, d.rotations, d.translations, d.weights, w, dtype=npdt)
    args = (T(pts, dev), T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev), T(d.weights, dev),
            T(w, dev))
    out = dpr_amd.raster(d.grid, *args, algo="chunked")
    pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout, dev), *args, algo="chunked")
    assert_close(out, ref_out, fwd_tol or tol(npdt, "out"), "out")
    assert_close(pb.points, ref_pb.points, tol(npdt, "points"), "ds_dpoints")
    assert_close(pb.point_weight, ref_pb.point_weight, tol(npdt, "points"), "ds_dpoint_weight")
    for name in ("rotation", "translation", "background", "out_weight"):
        assert_close(getattr(pb, name), getattr(ref_pb, name), tol(npdt, "pose"), name)
    return d, pts, w, out, pb


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
@pytest.mark.parametrize("n_points", [0, 1, 15, 16, 17, 1023, 1024, 1025, 4097, 70_001])
def test_cloud_sizes_around_the_box_granularity(oracle, dev, npdt, tdt, n_points):
    run_case(oracle, dev, npdt, tdt, (40, 33, 29), n_points, 2, seed=n_points)


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
@pytest.mark.parametrize("grid", [(1, 40, 40), (2, 37, 19), (3, 3, 3), (70, 1, 1), (33, 65, 15), (96, 96, 96)])
def test_odd_grid_shapes(oracle, dev, npdt, tdt, grid):
    """One and two cells along x (the pullback's x-pair gathers need two: a one-cell row falls back to the
    direct kernels of DPR_ALGO_ATOMIC), tiles with remainders in every axis."""
    run_case(oracle, dev, npdt, tdt, grid, 20_000, 3, seed=3)


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
@pytest.mark.parametrize("pw", [None, "rand"])
def test_heavy_tiles_are_split_into_parts_and_combined(oracle, dev, npdt, tdt, pw):
    """150 000 points within +-0.1: two or three tiles of the 128^3 grid hold them all, far above the
    8192 visits at which a tile is split; the parts' raw 64-bit tiles are summed from slabs -- for
    fp32 data as integers: the same bits whatever the split and the order of the points."""
    d, pts, w, out, _ = run_case(oracle, dev, npdt, tdt, (128, 128, 128), 150_000, 2, pw=pw, seed=5, scale=0.035)
    if npdt == np.float32:  # exact sums: the split leaves no trace -- any order of the points gives the same bits
        perm = np.random.default_rng(3).permutation(len(pts))
        out2 = dpr_amd.raster(d.grid, T(np.ascontiguousarray(pts[perm]), dev), T(d.rotations, dev),
                              T(d.translations, dev), T(d.backgrounds, dev), T(d.weights, dev),
                              None if w is None else T(np.ascontiguousarray(w[perm]), dev), algo="chunked")
        assert torch.equal(out, out2)


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
def test_incoherent_cloud_overflows_the_candidate_lists(oracle, dev, npdt, tdt):
    """300 000 points in RANDOM order on 128^3 (160 tiles): every chunk of 1024 points covers the whole
    grid, a tile's 293 candidates do not fit its slot of 256 -- the tile takes every chunk as a
    candidate.  Slow, never wrong; the direct pullback does not care."""
    run_case(oracle, dev, npdt, tdt, (128, 128, 128), 300_000, 1, order="random", seed=7)


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
def test_more_poses_than_one_plan_group_and_sparse_regime(oracle, dev, npdt, tdt):
    """37 poses: three plan groups of 16 on the owner path (dense cloud); the same batch of a cloud that
    is sparse on its grid runs the chunk lists (P * 10 <= G, B >= 4)."""
    run_case(oracle, dev, npdt, tdt, (48, 40, 36), 30_000, 37, seed=9)   # dense: owner-computes tiles
    run_case(oracle, dev, npdt, tdt, (64, 64, 64), 20_000, 5, seed=10)   # sparse: chunk lists
    # 70 poses: the fp32 pullback's in-kernel pose loop takes 64 per launch, the second launch ADDS its point
    # gradients to the first one's; five plan groups of the forward
    run_case(oracle, dev, npdt, tdt, (33, 40, 30), 8_000, 70, seed=11)


@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_non_finite_points_and_weights(oracle, dev, bad):
    """A NaN / Inf coordinate is a point without a cell (zero gradient, no contribution); a NaN / Inf
    weight switches its tiles to f64 sums and lands where the reference's arithmetic puts it."""
    npdt = np.float32
    d = D.make(n_points=40_000, n_in=3, n_out=3, batch=1, grid_n=48, seed=11, dtype=npdt)
    pts = np.ascontiguousarray(d.points[hilbert_like_order(d.points)])
    pts[[5, 1024, 30_000], [0, 1, 2]] = bad
    w = np.random.default_rng(1).uniform(0.5, 1.5, size=len(pts)).astype(npdt)
    ref = oracle.raster(d.grid, pts, d.rotations, d.translations, None, d.weights, w, dtype=npdt)
    out = dpr_amd.raster(d.grid, T(pts, dev), T(d.rotations, dev), T(d.translations, dev), None, T(d.weights, dev),
                         T(w, dev), algo="chunked")
    assert_close(out, ref, 5e-5, "out with non-finite points")
    pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout, dev), T(pts, dev), T(d.rotations, dev),
                                  T(d.translations, dev), None, T(d.weights, dev), T(w, dev), algo="chunked")
    refpb = oracle.raster_pullback(d.ds_dout, pts, d.rotations, d.translations, d.weights, w, dtype=npdt)
    assert_close(pb.points, refpb.points, 1e-4, "ds_dpoints")
    assert float(pb.points[5].abs().sum()) == 0.0
    # (the per-pose sums multiply by the point itself: 0 * NaN of a rejected point must not reach them --
    # found by tools/fuzz_owner.py in round 5)
    assert_close(pb.rotation, refpb.rotation, 1e-3, "ds_drotation")
    assert_close(pb.translation, refpb.translation, 1e-3, "ds_dtranslation")
    assert_close(pb.out_weight, refpb.out_weight, 1e-3, "ds_dout_weight")
    for npdt2, B in ((np.float32, 3), (np.float64, 2)):  # the batched kernels (pose loop inside / per pose)
        db = D.make(n_points=20_000, n_in=3, n_out=3, batch=B, grid_n=32, seed=12, dtype=npdt2)
        pb_pts = np.ascontiguousarray(db.points[hilbert_like_order(db.points)])
        pb_pts[[7, 2000, 19_999]] = bad
        got = dpr_amd.raster_pullback_(grid_to_dev(db.ds_dout, dev), T(pb_pts, dev), T(db.rotations, dev),
                                       T(db.translations, dev), None, T(db.weights, dev), None, algo="chunked",
                                       coherent_points=True)
        want = oracle.raster_pullback(db.ds_dout, pb_pts, db.rotations, db.translations, db.weights, None, dtype=npdt2)
        for name in ("points", "rotation", "translation", "out_weight", "background"):
            assert_close(getattr(got, name), getattr(want, name), tol(npdt2, "pose"), f"batched {name}")
    w2 = w.copy()
    w2[777] = bad
    ref2 = oracle.raster(d.grid, pts, d.rotations, d.translations, None, d.weights, w2, dtype=npdt)
    out2 = dpr_amd.raster(d.grid, T(pts, dev), T(d.rotations, dev), T(d.translations, dev), None,
                          T(d.weights, dev), T(w2, dev), algo="chunked").cpu().numpy()
    assert np.array_equal(np.isnan(out2), np.isnan(ref2))
    assert np.array_equal(np.isposinf(out2), np.isposinf(ref2))
    fin = np.isfinite(ref2)
    assert_close(out2[fin], ref2[fin], 5e-5, "finite cells")


def test_owner_forward_is_independent_of_the_point_order(dev):
    """fp32: 64-bit fixed-point sums per tile, parts of split tiles added as integers -- `out` is the
    same bit pattern for any order of the points (and so for any split into parts)."""
    d = D.make(n_points=120_000, n_in=3, n_out=3, batch=1, grid_n=64, seed=13, dtype=np.float32)
    pts = np.ascontiguousarray(d.points[hilbert_like_order(d.points)])
    perm = np.random.default_rng(2).permutation(len(pts))
    a = dpr_amd.raster(d.grid, T(pts, dev), T(d.rotations, dev), T(d.translations, dev), None, T(d.weights, dev),
                       None, algo="chunked")
    b = dpr_amd.raster(d.grid, T(np.ascontiguousarray(pts[perm]), dev), T(d.rotations, dev), T(d.translations, dev),
                       None, T(d.weights, dev), None, algo="chunked")
    assert torch.equal(a, b)


def test_auto_takes_the_direct_pullback_for_a_coherent_cloud(oracle, dev):
    """DPR_ALGO_AUTO + DPR_FLAG_COHERENT_POINTS, one pose on a 3-D grid: the pullback is the direct
    kernel, a KEEP / REUSE pair shares nothing (flags dropped) and still works."""
    d = D.make(n_points=60_000, n_in=3, n_out=3, batch=1, grid_n=64, seed=15, dtype=np.float32)
    pts = np.ascontiguousarray(d.points[hilbert_like_order(d.points)])
    assert dpr_amd.resolve_algo("pullback", d.grid, len(pts), 1, 3, coherent_points=True) == "chunked"
    assert not dpr_amd.sharing_effective(d.grid, len(pts), 1, 3, coherent_points=True)
    ws = torch.empty(max(dpr_amd.workspace_bytes(op, d.grid, len(pts), 1, 3, torch.float32, "auto",
                                                 coherent_points=True, sharing=True)
                         for op in ("raster", "pullback")) + 16, dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(d.grid, None, torch.float32, dev)
    args = (T(pts, dev), T(d.rotations[0], dev), T(d.translations[0], dev))
    dpr_amd.raster_(out, *args, workspace=ws, keep_binning=True, coherent_points=True)
    pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout[..., 0], dev), *args, workspace=ws, reuse_binning=True,
                                  coherent_points=True)
    ref = oracle.raster_pullback(d.ds_dout, pts, d.rotations, d.translations, dtype=np.float32)
    assert_close(pb.points, ref.points, 1e-4, "ds_dpoints")
    assert_close(pb.rotation, ref.rotation[0], 1e-3, "ds_drotation")
    assert_close(out, oracle.raster(d.grid, pts, d.rotations, d.translations, dtype=np.float32)[..., 0], 5e-5, "out")


def test_auto_takes_the_owner_forward_for_dense_coherent_batches(oracle, dev):
    """DPR_ALGO_AUTO + DPR_FLAG_COHERENT_POINTS, two poses of a dense cloud (0.4-2 points per voxel) on a grid of
    >= 1024 owner tiles: forward = owner-computes tiles, pullback = direct gathers, a KEEP / REUSE pair shares
    nothing (flags dropped) and still works; both against the oracle at a size where the rule applies."""
    grid = (256, 256, 224)  # 8 x 8 x 16 = 1024 tiles of 32 x 32 x 14 cells
    P, B = 6_000_000, 2
    rng = np.random.default_rng(21)
    pts = dpr_amd.sort_points(T((0.4 * rng.normal(size=(P, 3))).astype(np.float32), dev))[0]
    assert dpr_amd.resolve_algo("raster", grid, P, B, 3, coherent_points=True) == "chunked"
    assert dpr_amd.resolve_algo("raster", grid, P, B, 3, coherent_points=True, sharing=True) == "chunked"
    assert dpr_amd.resolve_algo("pullback", grid, P, B, 3, coherent_points=True, sharing=True) == "chunked"
    assert dpr_amd.resolve_algo("raster", grid, P, 1, 3, coherent_points=True) == "tiled"  # one pose: tiled forward
    R = D.random_rotations(rng, B, 3).astype(np.float32)
    t = (0.05 * rng.normal(size=(B, 3))).astype(np.float32)
    ow = rng.uniform(0.5, 2.0, size=B).astype(np.float32)
    bg = rng.normal(size=B).astype(np.float32)
    ws = torch.empty(max(dpr_amd.workspace_bytes(op, grid, P, B, 3, torch.float32, "auto", coherent_points=True,
                                                 sharing=True) for op in ("raster", "pullback")) + 16,
                     dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(grid, B, torch.float32, dev)
    args = (pts, T(R, dev), T(t, dev), T(bg, dev), T(ow, dev))
    dpr_amd.raster_(out, *args, workspace=ws, keep_binning=True, coherent_points=True)
    pts_h = pts.cpu().numpy()
    ref = oracle.raster(grid, pts_h, R, t, bg, ow, None, dtype=np.float32, threaded=True)
    assert_close(out, ref, 5e-5, "out")
    g = np.asfortranarray(rng.normal(size=grid + (B,)).astype(np.float32))
    pb = dpr_amd.raster_pullback_(grid_to_dev(g, dev), *args, workspace=ws, reuse_binning=True, coherent_points=True,
                                  point_weight_grad=False)
    sub = np.arange(0, P, 97)  # (the serial fp32 oracle pullback on a subsample; the sums over all points below)
    refpb = oracle.raster_pullback(g, pts_h[sub], R, t, ow, None, dtype=np.float32)
    assert_close(pb.points[torch.as_tensor(sub, device=dev)], refpb.points, 1e-4, "ds_dpoints (subsample)")
    # (the fp32 oracle: its cell choice is bit-identical to the kernels'; against fp64 arithmetic ~20 of the 3.6e7
    # (point, pose, axis) triples pick another cell, which shows at 2e-3 in these cancellation-heavy sums)
    full = oracle.raster_pullback(g, pts_h, R, t, ow, None, dtype=np.float32, threaded=True)
    assert_close(pb.rotation, full.rotation, 1e-3, "ds_drotation")
    assert_close(pb.translation, full.translation, 1e-3, "ds_dtranslation")
    assert_close(pb.background, full.background, 1e-3, "ds_dbackground")


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
@pytest.mark.parametrize("with_pw,want_pw", [(True, True), (False, False)])
def test_unsorted_batch_is_sorted_inside_the_pullback(oracle, dev, npdt, tdt, with_pw, want_pw):
    """DPR_ALGO_CHUNKED pullback on a 3-D grid WITHOUT the coherence flag, >= 8 poses and >= 2e5 points: the cloud
    is Hilbert-sorted into the workspace, the direct kernels run on the sorted copy and the point gradients come
    back in the caller's order (non-finite points included)."""
    P, B, grid = 210_000, 9, (40, 33, 29)
    d = D.make(n_points=P, n_in=3, n_out=3, batch=B, grid_n=40, seed=23, dtype=npdt)
    rng = np.random.default_rng(3)
    pts = d.points.copy()
    pts[[11, 100_000, P - 1]] = np.nan
    g = np.asfortranarray(rng.normal(size=grid + (B,)).astype(npdt))
    pw = rng.uniform(0.5, 1.5, size=P).astype(npdt) if with_pw else None
    need = dpr_amd.workspace_bytes("pullback", grid, P, B, 3, tdt, "chunked")
    assert need > 2 * P * 3 * np.dtype(npdt).itemsize  # (sorted copy + sorted gradients: the sorting variant)
    assert dpr_amd.workspace_bytes("pullback", grid, P, B, 3, tdt, "chunked", coherent_points=True) < need // 2
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    pb = dpr_amd.raster_pullback_(grid_to_dev(g, dev), T(pts, dev), T(d.rotations, dev), T(d.translations, dev), None,
                                  T(d.weights, dev), T(pw, dev), algo="chunked", workspace=ws, point_weight_grad=want_pw)
    ref = oracle.raster_pullback(g, pts, d.rotations, d.translations, d.weights, pw, dtype=npdt)
    assert_close(pb.points, ref.points, tol(npdt, "points"), "ds_dpoints")
    assert float(pb.points[11].abs().sum()) == 0.0
    if want_pw:
        assert_close(pb.point_weight, ref.point_weight, tol(npdt, "points"), "ds_dpoint_weight")
    else:
        assert pb.point_weight is None
    for name in ("rotation", "translation", "out_weight", "background"):
        assert_close(getattr(pb, name), getattr(ref, name), tol(npdt, "pose"), name)
    # AUTO takes this path for 16+ poses of 3e6+ points in any order
    assert dpr_amd.resolve_algo("pullback", (256,) * 3, 10_000_000, 16, 3) == "chunked"
    assert dpr_amd.resolve_algo("pullback", (256,) * 3, 10_000_000, 8, 3) == "chunked"
    assert dpr_amd.resolve_algo("pullback", (256,) * 3, 5_000_000, 8, 3) == "tiled"
    assert dpr_amd.resolve_algo("pullback", (256,) * 3, 1_000_000, 16, 3) != "chunked"


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
def test_unsorted_batch_is_sorted_inside_the_forward(oracle, dev, npdt, tdt):
    """DPR_ALGO_CHUNKED forward on a 3-D grid WITHOUT the coherence flag, >= 8 poses of >= 2e5 points that are not
    sparse on the grid: sorted into the workspace, owner tiles on the sorted copy; the caller's arrays are not
    touched and `out` does not depend on the order (fp32: bit for bit)."""
    P, B, grid = 230_000, 8, (48, 40, 36)
    d = D.make(n_points=P, n_in=3, n_out=3, batch=B, grid_n=48, seed=29, dtype=npdt)
    rng = np.random.default_rng(4)
    pts = d.points.copy()
    pts[[3, 99_999]] = np.inf
    pw = rng.uniform(0.5, 1.5, size=P).astype(npdt)
    need = dpr_amd.workspace_bytes("raster", grid, P, B, 3, tdt, "chunked")
    assert need > dpr_amd.workspace_bytes("raster", grid, P, B, 3, tdt, "chunked", coherent_points=True) + P * 3 * np.dtype(npdt).itemsize
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(grid, B, tdt, dev)
    tp, tw = T(pts, dev), T(pw, dev)
    before = tp.clone()
    dpr_amd.raster_(out, tp, T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev), T(d.weights, dev), tw,
                    algo="chunked", workspace=ws)
    assert torch.equal(torch.nan_to_num(tp), torch.nan_to_num(before))
    ref = oracle.raster(grid, pts, d.rotations, d.translations, d.backgrounds, d.weights, pw, dtype=npdt, threaded=True)
    assert_close(out, ref, tol(npdt, "out"), "out")
    if npdt == np.float32:
        sp, _, sw = dpr_amd.sort_points(tp, tw)
        out2 = dpr_amd.raster((grid), sp, T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev),
                              T(d.weights, dev), sw, algo="chunked", coherent_points=True)
        assert torch.equal(out, out2)
    # the same cloud SPARSE on a larger grid (P * 10 <= G): the chunk lists behind the same sort
    grid2 = (160, 160, 96)
    need2 = dpr_amd.workspace_bytes("raster", grid2, P, B, 3, tdt, "chunked")
    assert need2 > dpr_amd.workspace_bytes("raster", grid2, P, B, 3, tdt, "chunked", coherent_points=True) + P * 3 * np.dtype(npdt).itemsize
    out3 = dpr_amd.empty_grid(grid2, B, tdt, dev)
    dpr_amd.raster_(out3, tp, T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev), T(d.weights, dev), tw,
                    algo="chunked", workspace=torch.empty(need2, dtype=torch.uint8, device=dev))
    ref3 = oracle.raster(grid2, pts, d.rotations, d.translations, d.backgrounds, d.weights, pw, dtype=npdt, threaded=True)
    assert_close(out3, ref3, tol(npdt, "out"), "out (sparse)")
    assert dpr_amd.resolve_algo("raster", (256,) * 3, 1_000_000, 16, 3) == "chunked"
    assert dpr_amd.resolve_algo("raster", (256,) * 3, 1_000_000, 8, 3) == "tiled"
