"""GPU parity tests: the HIP path (through the C ABI, via the host mirror) against the CPU
oracle, the reference's golden vectors, and size-independent properties at BASELINE sizes.

Tolerances:
  fp64  norm-wise rtol 1e-10  (the reference's own `≈` would allow sqrt(eps) = 1.5e-8)
  fp32  vs the fp32 oracle (identical per-contribution arithmetic, hence identical cell
        choice; different summation order -- the oracle sums sequentially in fp32, the
        tiled path accumulates in fp64): norm-wise 5e-5 for `out`, 1e-4 for ds_dpoints /
        ds_dpoint_weight, 1e-3 for the per-pose sums (rotation / translation / out_weight /
        background); `out` is additionally checked against the fp64 oracle on the same
        fp32 inputs (continuous across cell flips) at 5e-5
"""
import numpy as np
import pytest
import torch

import dpr_amd
from tests import data as D

pytestmark = pytest.mark.gpu

ALGOS = ["atomic", "tiled", "chunked"]
DTYPES = [(np.float64, torch.float64), (np.float32, torch.float32)]
SHAPES = [(2, 2), (3, 3), (3, 2)]  # (n_in, n_out)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    # fail loudly if the extension is missing: there is no fallback
    dpr_amd.lib()
    return torch.device("cuda:0")


def T(a, dev, tdt=None):
    if a is None:
        return None
    t = torch.as_tensor(np.ascontiguousarray(a), device=dev)
    return t if tdt is None else t.to(tdt)


def grid_to_dev(a, dev):
    """numpy [i1..iN, b] (any order) -> device tensor with the reference memory order."""
    return dpr_amd.to_grid_layout(torch.as_tensor(np.ascontiguousarray(a), device=dev))


def tol(npdt, kind):
    if npdt == np.float64:
        return 1e-10
    return {"out": 5e-5, "points": 1e-4, "pose": 1e-3}[kind]


def assert_close(actual, expected, rtol, what=""):
    a = actual.detach().cpu().numpy() if isinstance(actual, torch.Tensor) else np.asarray(actual)
    e = np.asarray(expected)
    assert a.shape == e.shape, f"{what}: shape {a.shape} != {e.shape}"
    na, ne = np.linalg.norm(a.ravel()), np.linalg.norm(e.ravel())
    err = np.linalg.norm((a.astype(np.float64) - e.astype(np.float64)).ravel())
    assert err <= rtol * max(na, ne) + 1e-300, f"{what}: |a-e|={err:.3e} > {rtol:g}*{max(na, ne):.3e}"


# ------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("npdt,tdt", DTYPES)
def test_forward_known_answers(dev, golden, algo, npdt, tdt):
    """src/raster.jl:143-309 and README.md:41-68 through the HIP path (single-pose API)."""
    for case in golden["forward"]:
        pts = T(np.array(case["points"], dtype=npdt), dev)
        R = T(np.array(case["rotation"], dtype=npdt), dev)
        t = T(np.array(case["translation"], dtype=npdt), dev)
        args = [pts, R, t]
        if case["out_weight"] is not None:
            args += [case["background"], case["out_weight"]]
            if case["point_weight"] is not None:
                args += [T(np.array(case["point_weight"], dtype=npdt), dev)]
        out = dpr_amd.raster(tuple(case["grid_size"]), *args, algo=algo)
        assert out.shape == (5, 5) and out.dtype == tdt
        np.testing.assert_allclose(out.cpu().numpy(), np.array(case["expected"]), rtol=0,
                                   atol=1e-12 if npdt == np.float64 else 1e-5,
                                   err_msg=case["name"])


@pytest.mark.parametrize("algo", ALGOS)
def test_readme_gradient_example(dev, golden, algo):
    """README.md:84-183 (values printed to 6 digits)."""
    g = golden["readme_gradient"]
    pts = T(np.array(g["points"]), dev)
    R = T(np.array(g["rotation"]), dev)
    t = T(np.array(g["translation"]), dev)
    target = T(np.array(g["target_image"]), dev)
    out = dpr_amd.raster((5, 5), pts, R, t, algo=algo)
    ds_dout = 2.0 * (target - out)
    np.testing.assert_allclose(ds_dout.cpu().numpy(), np.array(g["ds_dout"]), rtol=2e-5, atol=2e-6)
    pb = dpr_amd.raster_pullback_(ds_dout, pts, R, t, algo=algo)
    full = -np.array(g["ds_dpoints_zygote_full_precision_negated"])
    np.testing.assert_allclose(pb.points.cpu().numpy(), full, rtol=0, atol=2e-5)
    np.testing.assert_allclose(pb.rotation.cpu().numpy(), np.array(g["ds_drotation"]), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pb.translation.cpu().numpy(), np.array(g["ds_dtranslation"]), rtol=0, atol=2e-5)
    assert pb.background.ndim == 0 and pb.out_weight.ndim == 0  # single pose -> scalars
    np.testing.assert_allclose(float(pb.background), float(ds_dout.sum()), rtol=1e-12)


# ------------------------------------------------------------------ oracle parity
def _run_both(oracle, dev, d, npdt, algo, with_optional=True):
    opt = (d.backgrounds, d.weights, d.point_weights) if with_optional else (None, None, None)
    ref_out = oracle.raster(d.grid, d.points, d.rotations, d.translations, *opt, dtype=npdt)
    ref_pb = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, opt[1],
                                    opt[2], dtype=npdt)
    out = dpr_amd.raster(d.grid, T(d.points, dev), T(d.rotations, dev), T(d.translations, dev),
                         T(opt[0], dev), T(opt[1], dev), T(opt[2], dev), algo=algo)
    pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout, dev), T(d.points, dev),
                                  T(d.rotations, dev), T(d.translations, dev), T(opt[0], dev),
                                  T(opt[1], dev), T(opt[2], dev), algo=algo)
    if npdt == np.float32:
        ref64 = oracle.raster(d.grid, d.points, d.rotations, d.translations, *opt, dtype=np.float64)
        assert_close(out, ref64.astype(np.float32), 5e-5, "out vs fp64 oracle")
    return ref_out, ref_pb, out, pb


def _compare(ref_out, ref_pb, out, pb, npdt):
    assert_close(out, ref_out, tol(npdt, "out"), "out")
    assert_close(pb.points, ref_pb.points, tol(npdt, "points"), "ds_dpoints")
    assert_close(pb.point_weight, ref_pb.point_weight, tol(npdt, "points"), "ds_dpoint_weight")
    assert_close(pb.rotation, ref_pb.rotation, tol(npdt, "pose"), "ds_drotation")
    assert_close(pb.translation, ref_pb.translation, tol(npdt, "pose"), "ds_dtranslation")
    assert_close(pb.background, ref_pb.background, tol(npdt, "pose"), "ds_dbackground")
    assert_close(pb.out_weight, ref_pb.out_weight, tol(npdt, "pose"), "ds_dout_weight")


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out", SHAPES)
@pytest.mark.parametrize("n_points", [10, 1000, 100_000])
def test_device_equals_oracle(oracle, dev, algo, npdt, tdt, n_in, n_out, n_points):
    """Counterpart of test/cuda.jl:2-74 (`cuda_cpu_agree`): 8^N grids, uneven batch,
    the distributions of test/data.jl, all optional arguments given."""
    d = D.make(n_points=n_points, n_in=n_in, n_out=n_out, batch=D.uneven_batch(4), seed=11,
               dtype=npdt)
    _compare(*_run_both(oracle, dev, d, npdt, algo), npdt)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out,n_points,batch,keep", [
    (3, 3, 30_000, 1, False), (3, 3, 30_000, 3, False), (3, 2, 30_000, 5, False), (2, 2, 3_000, 2, False),
    (3, 3, 30_000, 1, True), (3, 2, 30_000, 5, True)])
def test_pullback_without_the_point_weight_gradient(oracle, dev, algo, npdt, tdt, n_in, n_out, n_points,
                                                    batch, keep):
    """DPR_FLAG_NO_POINT_WEIGHT_GRAD (`point_weight_grad=False`): ds_dpoint_weight is neither
    allocated nor written -- the tangent the reference's rrule drops when point_weight was defaulted
    (ext/DiffPointRasterisationChainRulesCoreExt.jl:23,70) -- and the five other outputs are those
    of the plain call (oracle parity), with and without the KEEP / REUSE pairing."""
    if keep and (algo == "atomic" or (algo == "chunked" and n_out == 3 and batch > 16)):
        pytest.skip("nothing to keep on this path")
    d = D.make(n_points=n_points, n_in=n_in, n_out=n_out, batch=batch, grid_n=40 if n_out == 3 else 96,
               seed=41, dtype=npdt)
    single = batch == 1
    args = (T(d.points, dev), T(d.rotations[0] if single else d.rotations, dev),
            T(d.translations[0] if single else d.translations, dev), None,
            T(d.weights[0] if single else d.weights, dev), None)
    g = grid_to_dev(d.ds_dout[..., 0] if single else d.ds_dout, dev)
    kw = {}
    if keep:
        ws = torch.empty(max(dpr_amd.workspace_bytes(op, d.grid, n_points, batch, n_in, tdt, algo, sharing=True)
                             for op in ("raster", "pullback")), dtype=torch.uint8, device=dev)
        out = dpr_amd.empty_grid(d.grid, None if single else batch, tdt, dev)
        dpr_amd.raster_(out, *args, algo=algo, workspace=ws, keep_binning=True)
        kw = dict(workspace=ws, reuse_binning=True)
    pb = dpr_amd.raster_pullback_(g, *args, algo=algo, point_weight_grad=False, **kw)
    assert pb.point_weight is None
    ref = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights, None, dtype=npdt)
    sq = (lambda a: a[0]) if single else (lambda a: a)
    assert_close(pb.points, ref.points, tol(npdt, "points"), "ds_dpoints")
    assert_close(pb.rotation, sq(ref.rotation), tol(npdt, "pose"), "ds_drotation")
    assert_close(pb.translation, sq(ref.translation), tol(npdt, "pose"), "ds_dtranslation")
    assert_close(pb.background, sq(ref.background), tol(npdt, "pose"), "ds_dbackground")
    assert_close(pb.out_weight, sq(ref.out_weight), tol(npdt, "pose"), "ds_dout_weight")
    with pytest.raises(ValueError):
        dpr_amd.raster_pullback_(g, *args, algo=algo, point_weight_grad=False,
                                 ds_dpoint_weight=torch.empty(n_points, dtype=tdt, device=dev))


@pytest.mark.parametrize("algo", ["auto", "atomic"])
@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out", [(1, 1), (2, 1), (3, 1)])
@pytest.mark.parametrize("n_points,grid_n", [(10, 8), (30_000, 100), (400_000, 3001)])  # (line grids:
# a few hundred fp32 contributions per cell at most, the direct kernels accumulate in fp32)
def test_other_dimension_pairs_on_the_direct_kernels(oracle, dev, algo, npdt, tdt, n_in, n_out,
                                                     n_points, grid_n):
    """The reference is generic in (N_in, N_out) (src/raster.jl:5-13, src/util.jl:26-27); beyond
    the three shapes its tests use, every other pair runs on the direct kernels: AUTO
    resolves to them, the other algorithms are refused."""
    d = D.make(n_points=n_points, n_in=n_in, n_out=n_out, batch=D.uneven_batch(4), grid_n=grid_n,
               seed=19, dtype=npdt)
    assert dpr_amd.resolve_algo("raster", d.grid, n_points, d.batch, n_in, sharing=True) == "atomic"
    _compare(*_run_both(oracle, dev, d, npdt, algo), npdt)
    with pytest.raises(dpr_amd.DprError):
        dpr_amd.raster(d.grid, T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), algo="tiled")


@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out,grid_n", [(1, 2, 40), (1, 3, 20), (2, 3, 24), (4, 1, 300), (4, 2, 50), (4, 3, 21),
                                               (4, 4, 11), (3, 4, 9), (2, 4, (7, 9, 5, 6)), (1, 4, 8)])
def test_embeddings_and_four_dimensions_on_the_direct_kernels(oracle, dev, npdt, tdt, n_in, n_out, grid_n):
    """1 <= N_in, N_out <= 4 in ANY combination (round 6): embeddings with N_out > N_in and 4-D points / grids, which
    the reference's generic signatures (src/raster.jl:5-13) admit and its tests never exercise; forward and pullback
    against the oracle (itself checked for these pairs against a numpy restatement and central differences,
    tests/test_oracle_golden.py), optional arguments included, uneven batch."""
    d = D.make(n_points=20_000, n_in=n_in, n_out=n_out, batch=D.uneven_batch(4), grid_n=grid_n, seed=23, dtype=npdt)
    assert dpr_amd.resolve_algo("raster", d.grid, d.n_points, d.batch, n_in) == "atomic"
    assert dpr_amd.resolve_algo("pullback", d.grid, d.n_points, d.batch, n_in, sharing=True) == "atomic"
    _compare(*_run_both(oracle, dev, d, npdt, "auto"), npdt)
    with pytest.raises(dpr_amd.DprError):
        dpr_amd.raster(d.grid, T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), algo="chunked")


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out,grid_n", [(3, 3, 37), (3, 2, 100), (2, 2, 33), (3, 3, 64)])
def test_device_equals_oracle_odd_grids_defaults(oracle, dev, algo, npdt, tdt, n_in, n_out, grid_n):
    """Non-power-of-two grids (tile remainders) and defaulted optional arguments."""
    d = D.make(n_points=20_000, n_in=n_in, n_out=n_out, batch=2, grid_n=grid_n, seed=12, dtype=npdt)
    _compare(*_run_both(oracle, dev, d, npdt, algo, with_optional=False), npdt)


@pytest.mark.parametrize("algo", ALGOS)
def test_anisotropic_grid(oracle, dev, algo):
    rng = np.random.default_rng(5)
    grid = (20, 7, 45)
    pts = 0.5 * rng.normal(size=(5000, 3))
    R = D.random_rotations(rng, 3)
    t = 0.1 * rng.normal(size=(3, 3))
    g = np.asfortranarray(rng.normal(size=grid + (3,)))
    ref = oracle.raster(grid, pts, R, t)
    out = dpr_amd.raster(grid, T(pts, dev), T(R, dev), T(t, dev), algo=algo)
    assert_close(out, ref, 1e-10, "out")
    rp = oracle.raster_pullback(g, pts, R, t)
    pb = dpr_amd.raster_pullback_(grid_to_dev(g, dev), T(pts, dev), T(R, dev), T(t, dev), algo=algo)
    for a, e, n in zip(pb, rp, pb._fields):
        assert_close(a, e, 1e-10, n)


# ------------------------------------------------------------------ structural properties
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n_in,n_out", [(3, 3), (3, 2)])
def test_batched_equals_loop_of_singles(dev, algo, n_in, n_out):
    """src/raster.jl:383-431, src/raster_pullback.jl:271-345 on the device."""
    d = D.make(n_points=20_000, n_in=n_in, n_out=n_out, batch=D.uneven_batch(4), seed=13)
    pts, Rs, ts = T(d.points, dev), T(d.rotations, dev), T(d.translations, dev)
    bgs, ws, pw = T(d.backgrounds, dev), T(d.weights, dev), T(d.point_weights, dev)
    out_b = dpr_amd.raster(d.grid, pts, Rs, ts, bgs, ws, pw, algo=algo)
    g = grid_to_dev(d.ds_dout, dev)
    pb_b = dpr_amd.raster_pullback_(g, pts, Rs, ts, bgs, ws, pw, algo=algo)
    sum_pts = torch.zeros_like(pb_b.points)
    sum_pw = torch.zeros_like(pb_b.point_weight)
    for b in range(d.batch):
        out_i = dpr_amd.raster(d.grid, pts, Rs[b], ts[b], float(bgs[b]), float(ws[b]), pw, algo=algo)
        assert out_i.shape == tuple(d.grid)
        assert_close(out_b[..., b], out_i.cpu().numpy(), 1e-12, "out")
        pi = dpr_amd.raster_pullback_(g[..., b], pts, Rs[b], ts[b], float(bgs[b]), float(ws[b]),
                                      pw, algo=algo)
        assert_close(pb_b.rotation[b], pi.rotation.cpu().numpy(), 1e-10, "rotation")
        assert_close(pb_b.translation[b], pi.translation.cpu().numpy(), 1e-10, "translation")
        assert_close(pb_b.background[b], pi.background.cpu().numpy(), 1e-10, "background")
        assert_close(pb_b.out_weight[b], pi.out_weight.cpu().numpy(), 1e-10, "out_weight")
        sum_pts += pi.points
        sum_pw += pi.point_weight
    assert_close(pb_b.points, sum_pts.cpu().numpy(), 1e-10, "points")
    assert_close(pb_b.point_weight, sum_pw.cpu().numpy(), 1e-10, "point_weight")


@pytest.mark.parametrize("algo", ALGOS)
def test_defaults_equal_explicit_and_preallocated_outputs(dev, algo):
    """src/interface.jl:414-595 + kwargs outputs returned by identity and overwritten."""
    d = D.make(n_points=3000, n_in=3, n_out=3, batch=3, seed=14)
    pts, Rs, ts = T(d.points, dev), T(d.rotations, dev), T(d.translations, dev)
    a = dpr_amd.raster(d.grid, pts, Rs, ts, algo=algo)
    b = dpr_amd.raster(d.grid, pts, Rs, ts, torch.zeros(3, device=dev, dtype=torch.float64),
                       torch.ones(3, device=dev, dtype=torch.float64),
                       torch.ones(3000, device=dev, dtype=torch.float64), algo=algo)
    assert_close(a, b.cpu().numpy(), 1e-13)
    out = dpr_amd.empty_grid(d.grid, 3, torch.float64, dev)
    out.fill_(123.0)  # must be fully overwritten
    ret = dpr_amd.raster_(out, pts, Rs, ts, algo=algo)
    assert ret is out
    assert_close(out, a.cpu().numpy(), 1e-13)
    g = grid_to_dev(d.ds_dout, dev)
    pa = dpr_amd.raster_pullback_(g, pts, Rs, ts, algo=algo)
    garbage = lambda *s: torch.full(s, 7.0, device=dev, dtype=torch.float64)
    bufs = dict(ds_dpoints=garbage(3000, 3), ds_drotation=garbage(3, 3, 3).transpose(1, 2),
                ds_dtranslation=garbage(3, 3), ds_dbackground=garbage(3),
                ds_dout_weight=garbage(3), ds_dpoint_weight=garbage(3000))
    pb = dpr_amd.raster_pullback_(g, pts, Rs, ts, algo=algo, **bufs)
    assert pb.points is bufs["ds_dpoints"] and pb.point_weight is bufs["ds_dpoint_weight"]
    assert pb.translation.data_ptr() == bufs["ds_dtranslation"].data_ptr()
    assert pb.rotation.data_ptr() == bufs["ds_drotation"].data_ptr()
    for x, y, n in zip(pa, pb, pa._fields):
        assert_close(x, y.cpu().numpy(), 1e-10, n)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n_in,n_out", SHAPES)
def test_pullback_matches_central_differences_on_device(dev, algo, n_in, n_out):
    """Device counterpart of test_rrule (test/chainrules.jl:2-90; the CUDA version is
    commented out as failing in the reference, test/cuda.jl:76-93)."""
    d = D.make(n_points=10, n_in=n_in, n_out=n_out, batch=3, seed=15)
    g = grid_to_dev(d.ds_dout, dev)
    base = dict(points=T(d.points, dev), rot=T(d.rotations, dev), tr=T(d.translations, dev),
                bg=T(d.backgrounds, dev), ow=T(d.weights, dev), pw=T(d.point_weights, dev))

    def f(**over):
        a = dict(base); a.update(over)
        out = dpr_amd.raster(d.grid, a["points"], a["rot"], a["tr"], a["bg"], a["ow"], a["pw"], algo=algo)
        return float((g * out).sum())

    pb = dpr_amd.raster_pullback_(g, base["points"], base["rot"], base["tr"], base["bg"],
                                  base["ow"], base["pw"], algo=algo)
    h = 1e-6
    for name, grad in [("points", pb.points), ("rot", pb.rotation), ("tr", pb.translation),
                       ("bg", pb.background), ("ow", pb.out_weight), ("pw", pb.point_weight)]:
        arr = base[name].cpu().numpy()
        fd = np.zeros_like(arr)
        for idx in np.ndindex(arr.shape):
            ap = arr.copy(); ap[idx] += h
            am = arr.copy(); am[idx] -= h
            fd[idx] = (f(**{name: T(ap, dev)}) - f(**{name: T(am, dev)})) / (2 * h)
        np.testing.assert_allclose(grad.cpu().numpy(), fd, rtol=1e-5, atol=1e-6, err_msg=name)


@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out", SHAPES)
@pytest.mark.parametrize("with_pw", [False, True])
@pytest.mark.parametrize("algo", ["tiled", "chunked"])
def test_pullback_reusing_forward_binning(oracle, dev, algo, npdt, tdt, n_in, n_out, with_pw):
    """DPR_FLAG_KEEP_BINNING / DPR_FLAG_REUSE_BINNING: the pullback consumes the tile binning
    the forward call left in the workspace (the rrule pairing, ChainRulesCoreExt.jl:6-27).
    Some points lie outside the grid: their gradients must come back as zeros."""
    d = D.make(n_points=30_000, n_in=n_in, n_out=n_out, batch=1, grid_n=40, seed=21, dtype=npdt)
    d.points[::7] *= 4.0  # a good fraction far outside (-1, 1)
    pw = d.point_weights if with_pw else None
    ws = torch.empty(dpr_amd.workspace_bytes("raster", d.grid, d.n_points, 1, n_in, tdt, algo),
                     dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(d.grid, 1, tdt, dev)
    args = (T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev),
            T(d.weights, dev), T(pw, dev))
    dpr_amd.raster_(out, *args, algo=algo, workspace=ws, keep_binning=True)
    pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout, dev), *args, algo=algo, workspace=ws,
                                  reuse_binning=True)
    ref_out = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds,
                            d.weights, pw, dtype=npdt)
    ref_pb = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                    pw, dtype=npdt)
    _compare(ref_out, ref_pb, out, pb, npdt)
    with pytest.raises(dpr_amd.DprError):  # flags are a tiled-path, single-pose feature
        dpr_amd.raster_(out, *args, algo="atomic", workspace=ws, keep_binning=True)


@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_points,batch,grid,coherent", [
    (30_000, 3, (40, 40, 40), False),       # a small batch: B copies of the per-pose lists
    (30_000, 5, (90, 70), False),           # 2-D grid on the tiled path (pose groups are off)
    (210_000, 4, (336, 336, 336), False),   # > 4096 tiles, B >= 4: the cloud is sorted inside the
                                            # call -- the REUSE pullback reuses the sorted copy too
    (60_000, 3, (64, 48, 40), True),        # local binning (coherent flag) per pose
])
def test_batched_pullback_reusing_forward_binning(oracle, dev, npdt, tdt, n_points, batch, grid, coherent):
    """DPR_FLAG_KEEP_BINNING / REUSE_BINNING with B > 1 on the tiled path: every pose keeps its
    own binning (the per-pose part of the workspace exists B times), the pullback consumes all
    of them; a stale pose header gives NaN."""
    n_out = len(grid)
    d = D.make(n_points=n_points, n_in=3, n_out=n_out, batch=batch, grid_n=grid, seed=31, dtype=npdt)
    d.points[::9] *= 3.0  # some points outside the grid
    kw = dict(coherent_points=True) if coherent else {}
    need = max(dpr_amd.workspace_bytes(op, d.grid, n_points, batch, 3, tdt, "tiled", sharing=True, **kw)
               for op in ("raster", "pullback"))
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(d.grid, batch, tdt, dev)
    args = (T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev),
            T(d.weights, dev), T(d.point_weights, dev))
    g = grid_to_dev(d.ds_dout, dev)
    dpr_amd.raster_(out, *args, algo="tiled", workspace=ws, keep_binning=True, **kw)
    pb = dpr_amd.raster_pullback_(g, *args, algo="tiled", workspace=ws, reuse_binning=True, **kw)
    ref_out = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds,
                            d.weights, d.point_weights, dtype=npdt)
    ref_pb = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                    d.point_weights, dtype=npdt)
    _compare(ref_out, ref_pb, out, pb, npdt)
    # consumed: a second reuse finds no valid binning
    pb2 = dpr_amd.raster_pullback_(g, *args, algo="tiled", workspace=ws, reuse_binning=True, **kw)
    assert bool(torch.isnan(pb2.points).all()) and bool(torch.isnan(pb2.rotation).all())
    # one pose changed between the calls: that pose's outputs (and the summed point gradients) NaN
    dpr_amd.raster_(out, *args, algo="tiled", workspace=ws, keep_binning=True, **kw)
    t2 = args[2].clone()
    t2[batch - 1] += 0.01
    pb3 = dpr_amd.raster_pullback_(g, args[0], args[1], t2, *args[3:], algo="tiled", workspace=ws,
                                   reuse_binning=True, **kw)
    assert bool(torch.isnan(pb3.points).all()) and bool(torch.isnan(pb3.rotation[batch - 1]).all())
    assert not bool(torch.isnan(pb3.rotation[0]).any())


def test_reuse_binning_is_validated_on_the_device(oracle, dev):
    """DPR_FLAG_REUSE_BINNING without a matching DPR_FLAG_KEEP_BINNING forward (stale workspace,
    other pose, other points, binning already consumed) must not read through stale lists:
    every output comes back NaN instead of silently wrong; the matching pair is exact."""
    d = D.make(n_points=300_000, n_in=3, n_out=3, batch=1, grid_n=64, seed=17, dtype=np.float32)
    pts, R, t = T(d.points, dev), T(d.rotations[0], dev), T(d.translations[0], dev)
    g = grid_to_dev(d.ds_dout[..., 0], dev)
    ws = torch.zeros(dpr_amd.workspace_bytes("pullback", d.grid, d.n_points, 1, 3, torch.float32,
                                             "tiled"), dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(d.grid, None, torch.float32, dev)
    isnan = lambda pb: all(bool(torch.isnan(x).all()) for x in pb)
    # (a) nothing was kept (zeroed workspace)
    assert isnan(dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws, reuse_binning=True))
    # (b) a forward WITHOUT keep_binning
    dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws)
    assert isnan(dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws, reuse_binning=True))
    # (c) the matching pair works ...
    dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws, keep_binning=True)
    pb = dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws, reuse_binning=True)
    ref = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, dtype=np.float32)
    assert_close(pb.points, ref.points, 1e-4, "ds_dpoints")
    assert_close(pb.rotation, ref.rotation[0], 1e-3, "ds_drotation")
    # ... once: the gradient records have overwritten the binning
    assert isnan(dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws, reuse_binning=True))
    # (d) kept for another pose / another point buffer
    dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws, keep_binning=True)
    assert isnan(dpr_amd.raster_pullback_(g, pts, R, t + 0.01, algo="tiled", workspace=ws,
                                          reuse_binning=True))
    dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws, keep_binning=True)
    assert isnan(dpr_amd.raster_pullback_(g, pts.clone(), R, t, algo="tiled", workspace=ws,
                                          reuse_binning=True))
    # (e) kept in ANOTHER WORKSPACE LAYOUT: coherent_points (local binning) on one call of the
    # pair only -- the lists then live at other offsets, in another form (round-2 advisor finding)
    ws_c = torch.zeros(dpr_amd.workspace_bytes("pullback", d.grid, d.n_points, 1, 3, torch.float32,
                                               "tiled", coherent_points=True),
                       dtype=torch.uint8, device=dev)
    dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws_c, keep_binning=True,
                    coherent_points=True)
    assert isnan(dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws_c,
                                          reuse_binning=True))
    dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws_c, keep_binning=True)
    assert isnan(dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws_c,
                                          reuse_binning=True, coherent_points=True))
    dpr_amd.raster_(out, pts, R, t, algo="tiled", workspace=ws_c, keep_binning=True,
                    coherent_points=True)
    pb_c = dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws_c,
                                    reuse_binning=True, coherent_points=True)
    assert_close(pb_c.points, ref.points, 1e-4, "ds_dpoints (coherent pair)")
    # and a pullback that re-bins is unaffected by whatever the workspace holds
    pb2 = dpr_amd.raster_pullback_(g, pts, R, t, algo="tiled", workspace=ws)
    assert_close(pb2.points, ref.points, 1e-4, "ds_dpoints (own binning)")


@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out,grid_n", [(3, 3, 96), (3, 2, 200), (2, 2, 150)])
def test_chunked_on_spatially_sorted_points(oracle, dev, npdt, tdt, n_in, n_out, grid_n):
    """DPR_ALGO_CHUNKED on its intended input: Morton-sorted points (chunk lists are used,
    few chunks diverted), several poses, grids of many tiles with remainders."""
    d = D.make(n_points=60_000, n_in=n_in, n_out=n_out, batch=3, grid_n=grid_n, seed=23, dtype=npdt)
    q = np.clip(((d.points.astype(np.float64) * 0.5 + 0.5) * 1024).astype(np.int64), 0, 1023)
    code = np.zeros(len(q), dtype=np.int64)
    for bit in range(10):
        for k in range(n_in):
            code |= ((q[:, k] >> bit) & 1) << (n_in * bit + k)
    order = np.argsort(code, kind="stable")
    d.points = np.ascontiguousarray(d.points[order])
    d.point_weights = np.ascontiguousarray(d.point_weights[order])
    _compare(*_run_both(oracle, dev, d, npdt, "chunked"), npdt)
    # the tiled path on the same sorted cloud, NOT flagged coherent: k_count finds few bins per
    # slice and the tile scan hands k_tile_splat the blocked record assignment (a cloud in random
    # order, as in most tests here, gets the strided one)
    _compare(*_run_both(oracle, dev, d, npdt, "tiled"), npdt)


@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out,grid_n", [(3, 3, 70), (3, 2, 90), (2, 2, 64), (3, 3, (130, 70, 70))])
def test_heavy_tiles_are_split(oracle, dev, npdt, tdt, n_in, n_out, grid_n):
    """A tightly clustered cloud puts far more than the split threshold (4096 records) into a
    handful of tiles: the tile kernels then run several work items per tile and the parts are
    combined from overflow slabs.  Forward, pullback and the binning-reuse
    pairing against the oracle.  On the 130 x 70 x 70 grid the cluster sits on a tile corner in all
    three axes (voxel 65 | 35 | 35 against tile edges at 64 | 32 | 32): split tiles whose x, y, z
    and diagonal neighbours are split tiles too, two poses in one pose group."""
    n_points = 150_000 if isinstance(grid_n, tuple) else 40_000
    d = D.make(n_points=n_points, n_in=n_in, n_out=n_out, batch=2, grid_n=grid_n, seed=27, dtype=npdt)
    d.points = (d.points * npdt(0.12)).astype(npdt)  # ~ +-0.15: a few tiles hold everything
    d.points[::50] *= npdt(8.0)                      # plus some stragglers elsewhere
    _compare(*_run_both(oracle, dev, d, npdt, "tiled"), npdt)
    ws = torch.empty(dpr_amd.workspace_bytes("raster", d.grid, d.n_points, 1, n_in, tdt, "tiled"),
                     dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(d.grid, 1, tdt, dev)
    args = (T(d.points, dev), T(d.rotations[:1], dev), T(d.translations[:1], dev),
            T(d.backgrounds[:1], dev), T(d.weights[:1], dev), T(d.point_weights, dev))
    dpr_amd.raster_(out, *args, algo="tiled", workspace=ws, keep_binning=True)
    pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout[..., :1], dev), *args, algo="tiled",
                                  workspace=ws, reuse_binning=True)
    ref_out = oracle.raster(d.grid, d.points, d.rotations[:1], d.translations[:1],
                            d.backgrounds[:1], d.weights[:1], d.point_weights, dtype=npdt)
    ref_pb = oracle.raster_pullback(d.ds_dout[..., :1], d.points, d.rotations[:1],
                                    d.translations[:1], d.weights[:1], d.point_weights, dtype=npdt)
    _compare(ref_out, ref_pb, out, pb, npdt)


# ------------------------------------------------------------------ residual pullback
def test_readme_gradient_example_fused(dev, golden):
    """README.md:151-183 with the sensitivity formed inside the pullback kernels: scale = -2
    reproduces the README's `ds_dout = 2 .* (target .- raster(...))` literally."""
    g = golden["readme_gradient"]
    pts = T(np.array(g["points"]), dev)
    R = T(np.array(g["rotation"]), dev)
    t = T(np.array(g["translation"]), dev)
    target = T(np.array(g["target_image"]), dev)
    for algo in ("atomic", "tiled"):
        out = dpr_amd.raster((5, 5), pts, R, t, algo=algo)
        pb, loss = dpr_amd.raster_residual_pullback_(out, target, pts, R, t, scale=-2.0, algo=algo)
        full = -np.array(g["ds_dpoints_zygote_full_precision_negated"])
        np.testing.assert_allclose(pb.points.cpu().numpy(), full, rtol=0, atol=2e-5)
        np.testing.assert_allclose(pb.rotation.cpu().numpy(), np.array(g["ds_drotation"]),
                                   rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(pb.translation.cpu().numpy(), np.array(g["ds_dtranslation"]),
                                   rtol=0, atol=2e-5)
        assert loss.ndim == 0
        np.testing.assert_allclose(float(loss), float(((out - target) ** 2).sum()), rtol=1e-12)


@pytest.mark.parametrize("algo", ["atomic", "tiled"])
@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out", SHAPES)
def test_residual_pullback_equals_oracle_recipe(oracle, dev, algo, npdt, tdt, n_in, n_out):
    """dpr_raster_residual_pullback_* against the oracle's host recipe (ds_dout = scale *
    (out - target), then raster_pullback; README.md:151-165) and against the un-fused device
    path, batched, with every optional argument and points outside the grid."""
    d = D.make(n_points=20_000, n_in=n_in, n_out=n_out, batch=3, grid_n=36, seed=33, dtype=npdt)
    d.points[::9] *= 3.0
    rng = np.random.default_rng(34)
    target = np.asfortranarray(rng.normal(size=d.grid + (d.batch,)).astype(npdt))
    args = (T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev),
            T(d.weights, dev), T(d.point_weights, dev))
    out = dpr_amd.raster(d.grid, *args, algo=algo)
    tgt = grid_to_dev(target, dev)
    loss_buf = torch.empty(d.batch, dtype=tdt, device=dev)
    pb, loss = dpr_amd.raster_residual_pullback_(out, tgt, *args, scale=2.0, loss=loss_buf, algo=algo)
    assert loss.data_ptr() == loss_buf.data_ptr()
    # (1) oracle recipe on the DEVICE's `out` (isolates the pullback from forward rounding)
    ref_pb, ref_loss = oracle.residual_pullback(out.cpu().numpy(), target, d.points, d.rotations,
                                                d.translations, d.weights, d.point_weights,
                                                scale=2.0, dtype=npdt)
    assert_close(pb.points, ref_pb.points, tol(npdt, "points"), "ds_dpoints")
    assert_close(pb.point_weight, ref_pb.point_weight, tol(npdt, "points"), "ds_dpoint_weight")
    assert_close(pb.rotation, ref_pb.rotation, tol(npdt, "pose"), "ds_drotation")
    assert_close(pb.translation, ref_pb.translation, tol(npdt, "pose"), "ds_dtranslation")
    assert_close(pb.background, ref_pb.background, tol(npdt, "pose"), "ds_dbackground")
    assert_close(pb.out_weight, ref_pb.out_weight, tol(npdt, "pose"), "ds_dout_weight")
    assert_close(loss, ref_loss.astype(npdt), tol(npdt, "out"), "loss")
    # (2) the un-fused device path on the same sensitivity: the tiled kernels see bit-identical
    # values, so the per-point results agree bitwise (the per-pose sums depend on the record
    # order inside a tile, which the binning's atomics do not fix)
    unfused = dpr_amd.raster_pullback_(2.0 * (out - tgt), *args, algo=algo)
    if algo == "tiled":
        assert torch.equal(pb.points, unfused.points)
        assert torch.equal(pb.point_weight, unfused.point_weight)
        assert_close(pb.rotation, unfused.rotation.cpu().numpy(), tol(npdt, "pose"), "rot")
        assert_close(pb.background, unfused.background.cpu().numpy(), tol(npdt, "pose"), "bg")
    else:  # global atomics: same values, unordered sums
        assert_close(pb.points, unfused.points.cpu().numpy(), tol(npdt, "points"), "pts")
        assert_close(pb.rotation, unfused.rotation.cpu().numpy(), tol(npdt, "pose"), "rot")


def test_residual_pullback_reuses_forward_binning_and_rejects_chunked(oracle, dev):
    """The training-step pairing: raster_(keep_binning) -> residual pullback (reuse_binning),
    one pose, clustered cloud so that split tiles are covered too."""
    npdt, tdt = np.float32, torch.float32
    d = D.make(n_points=60_000, n_in=3, n_out=3, batch=1, grid_n=70, seed=35, dtype=npdt)
    d.points[: 40_000] *= npdt(0.1)
    rng = np.random.default_rng(36)
    target = np.asfortranarray(rng.normal(size=d.grid + (1,)).astype(npdt))
    args = (T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev),
            T(d.weights, dev), None)
    ws = torch.empty(dpr_amd.workspace_bytes("raster", d.grid, d.n_points, 1, 3, tdt, "tiled"),
                     dtype=torch.uint8, device=dev)
    out = dpr_amd.empty_grid(d.grid, 1, tdt, dev)
    dpr_amd.raster_(out, *args, algo="tiled", workspace=ws, keep_binning=True)
    tgt = grid_to_dev(target, dev)
    pb, loss = dpr_amd.raster_residual_pullback_(out, tgt, *args, algo="tiled", workspace=ws,
                                                 reuse_binning=True)
    ref_pb, ref_loss = oracle.residual_pullback(out.cpu().numpy(), target, d.points, d.rotations,
                                                d.translations, d.weights, None, dtype=npdt)
    assert_close(pb.points, ref_pb.points, 1e-4, "ds_dpoints")
    assert_close(pb.rotation, ref_pb.rotation, 1e-3, "ds_drotation")
    assert_close(pb.background, ref_pb.background, 1e-3, "ds_dbackground")
    assert_close(loss, ref_loss.astype(npdt), 5e-5, "loss")
    with pytest.raises(dpr_amd.DprError):
        dpr_amd.raster_residual_pullback_(out, tgt, *args, algo="chunked")
    with pytest.raises(dpr_amd.DimensionMismatch):
        dpr_amd.raster_residual_pullback_(out, tgt[1:], *args)


# ------------------------------------------------------------------ pose groups
@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in,n_out,grid_n,batch", [(3, 2, 96, 21), (3, 3, 40, 19), (2, 2, 64, 7)])
@pytest.mark.parametrize("with_pw", [False, True])
def test_pose_groups_equal_oracle(oracle, dev, npdt, tdt, n_in, n_out, grid_n, batch, with_pw):
    """Grids with few tiles bin up to 16 poses together ((pose, tile) bins):
    odd batch sizes decompose into groups of 16 / 4 / 2 / 1 poses.  Forward, pullback and the
    residual pullback against the oracle; a clustered cloud makes some (pose, tile) bins split."""
    d = D.make(n_points=30_000, n_in=n_in, n_out=n_out, batch=batch, grid_n=grid_n, seed=41, dtype=npdt)
    d.points[:20_000] *= npdt(0.3)
    d.points[::11] *= npdt(5.0)
    if not with_pw:
        d.point_weights = None
    _compare(*_run_both(oracle, dev, d, npdt, "tiled"), npdt)
    args = (T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev),
            T(d.weights, dev), T(d.point_weights, dev))
    out = dpr_amd.raster(d.grid, *args, algo="tiled")
    target = np.asfortranarray(np.random.default_rng(42).normal(size=d.grid + (batch,)).astype(npdt))
    pb, loss = dpr_amd.raster_residual_pullback_(out, grid_to_dev(target, dev), *args, algo="tiled")
    ref_pb, ref_loss = oracle.residual_pullback(out.cpu().numpy(), target, d.points, d.rotations,
                                                d.translations, d.weights, d.point_weights,
                                                dtype=npdt)
    assert_close(pb.points, ref_pb.points, tol(npdt, "points"), "ds_dpoints")
    assert_close(pb.rotation, ref_pb.rotation, tol(npdt, "pose"), "ds_drotation")
    assert_close(pb.background, ref_pb.background, tol(npdt, "pose"), "ds_dbackground")
    assert_close(loss, ref_loss.astype(npdt), tol(npdt, "out"), "loss")


# ------------------------------------------------------------------ autograd rule
@pytest.mark.parametrize("algo", ["atomic", "tiled"])
@pytest.mark.parametrize("n_in,n_out,batched", [(3, 3, False), (3, 2, True), (2, 2, True)])
def test_autograd_rule_gradcheck(dev, algo, n_in, n_out, batched):
    """`raster_ad` is the torch-autograd counterpart of the reference's rrule
    (ext/DiffPointRasterisationChainRulesCoreExt.jl:6-27, :47-74, tested there with
    test_rrule = finite differences, test/chainrules.jl): gradcheck in fp64 over every
    differentiable argument, single pose and batch."""
    d = D.make(n_points=14, n_in=n_in, n_out=n_out, batch=3 if batched else 1, grid_n=6, seed=61,
               dtype=np.float64)
    req = lambda a: T(a, dev).clone().requires_grad_(True)
    pts = req(d.points * 0.8)
    if batched:
        R, t, bg, ow = req(d.rotations), req(d.translations), req(d.backgrounds), req(d.weights)
    else:
        R, t = req(d.rotations[0]), req(d.translations[0])
        bg, ow = req(d.backgrounds[0]), req(d.weights[0])
    pw = req(d.point_weights * 10)
    fn = lambda *a: dpr_amd.raster_ad((6,) * n_out, *a, algo=algo)
    assert torch.autograd.gradcheck(fn, (pts, R, t, bg, ow, pw), eps=1e-6, atol=1e-6, rtol=1e-5,
                                    nondet_tol=1e-9)
    # defaults are constants: only the three mandatory arguments get gradients
    assert torch.autograd.gradcheck(fn, (pts, R, t), eps=1e-6, atol=1e-6, rtol=1e-5,
                                    nondet_tol=1e-9)


def test_autograd_rule_matches_the_explicit_pullback(oracle, dev):
    """loss = sum((raster(...) - target)^2); loss.backward() through `raster_ad` (single pose,
    tiled: the backward pass reuses the forward's binning) gives the gradients of the explicit
    `raster_residual_pullback_` and of the oracle recipe; a second backward pass through the
    same graph re-bins and gives the same result."""
    npdt, tdt = np.float32, torch.float32
    d = D.make(n_points=40_000, n_in=3, n_out=3, batch=1, grid_n=48, seed=62, dtype=npdt)
    target = np.asfortranarray(np.random.default_rng(63).normal(size=d.grid + (1,)).astype(npdt))
    pts = T(d.points, dev).requires_grad_(True)
    R = T(d.rotations[0], dev).requires_grad_(True)
    t = T(d.translations[0], dev).requires_grad_(True)
    ow = T(d.weights[:1], dev)[0].clone().requires_grad_(True)
    tgt = grid_to_dev(target, dev)[..., 0]
    out = dpr_amd.raster_ad(d.grid, pts, R, t, 0.5, ow, algo="tiled")
    loss = ((out - tgt) ** 2).sum()
    loss.backward(retain_graph=True)
    g1 = [x.grad.clone() for x in (pts, R, t, ow)]
    for x in (pts, R, t, ow):
        x.grad = None
    loss.backward()  # the kept binning is gone: this pass bins again
    g2 = [x.grad.clone() for x in (pts, R, t, ow)]
    pb, lval = dpr_amd.raster_residual_pullback_(out.detach(), tgt, pts.detach(), R.detach(),
                                                 t.detach(), 0.5, ow.detach(), algo="tiled")
    ref_pb, ref_loss = oracle.residual_pullback(out.detach().cpu().numpy()[..., None], target,
                                                d.points, d.rotations, d.translations,
                                                d.weights, None, dtype=npdt)
    for ga, gb in zip(g1, g2):
        assert_close(ga, gb.cpu().numpy(), 1e-4, "first vs second backward pass")
    assert_close(g1[0], pb.points.cpu().numpy(), 1e-5, "autograd vs explicit: points")
    assert_close(g1[1], pb.rotation.cpu().numpy(), 1e-3, "autograd vs explicit: rotation")
    assert_close(g1[0], ref_pb.points, 1e-4, "autograd vs oracle: points")
    assert_close(g1[2], ref_pb.translation[0], 1e-3, "autograd vs oracle: translation")
    assert_close(g1[3], ref_pb.out_weight[0], 1e-3, "autograd vs oracle: out_weight")
    assert abs(float(loss.detach()) - float(ref_loss[0])) <= 5e-5 * float(ref_loss[0])
    assert abs(float(lval) - float(ref_loss[0])) <= 5e-5 * float(ref_loss[0])


@pytest.mark.parametrize("n_points,n_out,grid,batch", [
    (150_000, 2, (96, 96), 12),        # AUTO -> the chunk-owner pair: the sorted cloud is shared
    (1_000_000, 3, (272, 272, 272), 4),  # AUTO -> tiled pair on a grid too large for pose groups
                                         # (2890 tiles; dense enough for the tiled pullback: 320
                                         # points per tile): every pose keeps its binning
])
def test_autograd_rule_on_batches_that_share(oracle, dev, n_points, n_out, grid, batch):
    """`raster_ad` on a BATCH: where the library shares between a raster call and its pullback
    (dpr_resolve_flags_ex) the rule keeps a workspace between forward and backward; gradients
    against the oracle, and a second backward pass (binning consumed) gives the same."""
    npdt = np.float32
    d = D.make(n_points=n_points, n_in=3, n_out=n_out, batch=batch, grid_n=grid, seed=71, dtype=npdt)
    assert dpr_amd.sharing_effective(d.grid, n_points, batch, 3)
    pts = T(d.points, dev).requires_grad_(True)
    R = T(d.rotations, dev).requires_grad_(True)
    t = T(d.translations, dev).requires_grad_(True)
    ow = T(d.weights, dev).requires_grad_(True)
    g = grid_to_dev(d.ds_dout, dev)
    out = dpr_amd.raster_ad(d.grid, pts, R, t, None, ow)
    loss = (out * g).sum()
    loss.backward(retain_graph=True)
    g1 = [x.grad.clone() for x in (pts, R, t, ow)]
    for x in (pts, R, t, ow):
        x.grad = None
    loss.backward()
    g2 = [x.grad.clone() for x in (pts, R, t, ow)]
    ref = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights, None,
                                 dtype=npdt)
    for ga, gb in zip(g1, g2):
        assert_close(ga, gb.cpu().numpy(), 1e-4, "first vs second backward pass")
    assert_close(g1[0], ref.points, 1e-4, "points")
    assert_close(g1[1], ref.rotation, 1e-3, "rotation")
    assert_close(g1[2], ref.translation, 1e-3, "translation")
    assert_close(g1[3], ref.out_weight, 1e-3, "out_weight")


# ------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("npdt,tdt", DTYPES)
def test_edge_points(oracle, dev, algo, npdt, tdt):
    """Border neighbours dropped individually (src/raster.jl:62); far / non-finite points
    contribute nothing and get zero gradients."""
    pts = np.array([[0.999, 0.0], [-0.999, -0.999], [0.0, 0.999], [5.0, 0.0], [np.nan, 0.0],
                    [1e30, 0.0], [-1.2, 0.3], [0.3, np.inf]], dtype=npdt)
    rng = np.random.default_rng(1)
    R = D.random_rotations(rng, 2, 2).astype(npdt)
    R[0] = np.eye(2)
    t = np.zeros((2, 2), dtype=npdt)
    g = np.asfortranarray(rng.normal(size=(6, 6, 2)).astype(npdt))
    ref = oracle.raster((6, 6), pts, R, t, dtype=npdt)
    out = dpr_amd.raster((6, 6), T(pts, dev), T(R, dev), T(t, dev), algo=algo)
    assert torch.isfinite(out).all()
    assert_close(out, ref, tol(npdt, "out"))
    rp = oracle.raster_pullback(g, pts, R, t, dtype=npdt)
    pb = dpr_amd.raster_pullback_(grid_to_dev(g, dev), T(pts, dev), T(R, dev), T(t, dev), algo=algo)
    for a, e, n in zip(pb, rp, pb._fields):
        assert torch.isfinite(a).all(), n
        assert_close(a, e, tol(npdt, "pose"), n)
    assert (pb.points[3:6] == 0).all() and (pb.point_weight[3:6] == 0).all()


def test_empty_cloud_with_the_coherent_flag(dev):
    """P = 0 on the local-binning path (the sub-chunk kernel has nothing to load): background only,
    zero gradients of the right shapes -- single pose and a batch on a grid of more than 4096 tiles."""
    for grid, B in (((40, 40, 40), 1), ((320, 320, 328), 5)):
        R = torch.eye(3, device=dev, dtype=torch.float32)[None].repeat(B, 1, 1)
        t = torch.zeros(B, 3, device=dev, dtype=torch.float32)
        empty = torch.zeros(0, 3, device=dev, dtype=torch.float32)
        bg = torch.arange(1, B + 1, device=dev, dtype=torch.float32)
        out = dpr_amd.raster(grid, empty, R, t, bg, algo="tiled", coherent_points=True)
        for b in range(B):
            assert (out[..., b] == float(b + 1)).all()
        g = torch.ones((B,) + tuple(reversed(grid)), device=dev, dtype=torch.float32).permute(3, 2, 1, 0)
        pb = dpr_amd.raster_pullback_(g, empty, R, t, bg, algo="tiled", coherent_points=True)
        assert pb.points.shape == (0, 3) and float(pb.rotation.abs().max()) == 0.0
        assert torch.allclose(pb.background, torch.full((B,), float(np.prod(grid)), device=dev))


@pytest.mark.parametrize("algo", ALGOS)
def test_empty_and_tiny_inputs(dev, algo):
    R = torch.eye(3, device=dev, dtype=torch.float64)[None].repeat(2, 1, 1)
    t = torch.zeros(2, 3, device=dev, dtype=torch.float64)
    empty = torch.zeros(0, 3, device=dev, dtype=torch.float64)
    bg = torch.tensor([2.0, -1.0], device=dev, dtype=torch.float64)
    out = dpr_amd.raster((4, 4, 4), empty, R, t, bg, algo=algo)
    assert (out[..., 0] == 2.0).all() and (out[..., 1] == -1.0).all()
    g = dpr_amd.empty_grid((4, 4, 4), 2, torch.float64, dev)
    g.fill_(1.0)
    pb = dpr_amd.raster_pullback_(g, empty, R, t, algo=algo)
    assert pb.points.shape == (0, 3) and pb.point_weight.shape == (0,)
    assert (pb.background == 64.0).all() and (pb.rotation == 0).all() and (pb.out_weight == 0).all()
    one = torch.zeros(1, 3, device=dev, dtype=torch.float64)
    out = dpr_amd.raster((4, 4, 4), one, R[0], t[0], algo=algo)
    assert abs(float(out.sum()) - 1.0) < 1e-12


def test_dimension_errors(dev):
    """src/raster.jl:14-23, src/interface.jl:137-162: raised before any launch."""
    f64 = dict(device=dev, dtype=torch.float64)
    pts = torch.zeros(5, 3, **f64)
    R = torch.eye(3, **f64)[None].repeat(2, 1, 1)
    t = torch.zeros(2, 3, **f64)
    with pytest.raises(dpr_amd.DimensionMismatch, match="Column dimension"):
        dpr_amd.raster((4, 4, 4), torch.zeros(5, 2, **f64), R, t)
    with pytest.raises(dpr_amd.DimensionMismatch, match="Row dimension"):
        dpr_amd.raster((4, 4, 4), pts, R, torch.zeros(2, 2, **f64))
    with pytest.raises(dpr_amd.DimensionMismatch):
        dpr_amd.raster((4, 4, 4), pts, R, torch.zeros(3, 3, **f64))
    with pytest.raises(dpr_amd.DimensionMismatch):
        dpr_amd.raster((4, 4, 4), pts, R, t, torch.zeros(3, **f64))
    with pytest.raises(dpr_amd.DimensionMismatch):
        dpr_amd.raster((4, 4, 4), pts, R, t, None, None, torch.ones(4, **f64))
    with pytest.raises(dpr_amd.DimensionMismatch):
        dpr_amd.raster_(dpr_amd.empty_grid((4, 4), 2, torch.float64, dev), pts, R, t)
    with pytest.raises(dpr_amd.DimensionMismatch):
        dpr_amd.raster_pullback_(dpr_amd.empty_grid((4, 4, 4), 3, torch.float64, dev), pts, R, t)
    # an embedding 2 -> 3 is a legal pair since round 6 (1 <= N_in, N_out <= 4: the reference's generic
    # signatures, src/raster.jl:5-13) ...
    out = dpr_amd.raster((4, 4, 4), torch.zeros(5, 2, **f64), torch.zeros(1, 3, 2, **f64),
                         torch.zeros(1, 3, **f64))
    assert tuple(out.shape)[-3:] == (4, 4, 4) or tuple(out.shape)[:3] == (4, 4, 4)
    # ... five dimensions are rejected by the library itself (DPR_ERR_UNSUPPORTED_DIMS)
    with pytest.raises(dpr_amd.DprError):
        dpr_amd.raster((4, 4, 4), torch.zeros(5, 5, **f64), torch.zeros(1, 3, 5, **f64),
                       torch.zeros(1, 3, **f64))


def test_mixed_dtypes_promote(dev):
    """promote_type over arguments (src/interface.jl:63-64); Bool rotation like I(2)."""
    pts = torch.rand(100, 2, device=dev, dtype=torch.float32) - 0.5
    out = dpr_amd.raster((8, 8), pts, torch.eye(2, device=dev, dtype=torch.bool),
                         torch.zeros(2, device=dev, dtype=torch.float32))
    assert out.dtype == torch.float32 and abs(float(out.sum()) - 100.0) < 1e-3
    out = dpr_amd.raster((8, 8), pts, torch.eye(2, device=dev, dtype=torch.float64),
                         torch.zeros(2, device=dev, dtype=torch.float32))
    assert out.dtype == torch.float64


# ------------------------------------------------------------------ BASELINE-size properties
def _ball_points(n, dev, dtype, radius=0.85, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    v = torch.randn(n, 3, device=dev, dtype=dtype, generator=g)
    v = v / v.norm(dim=1, keepdim=True)
    r = radius * torch.rand(n, 1, device=dev, dtype=dtype, generator=g) ** (1.0 / 3.0)
    return v * r


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("config", ["C2", "C3"])
def test_full_size_properties(dev, algo, config):
    """BASELINE.json configs 2/3 (1M -> 128^3, 10M -> 256^3, fp32, one pose): properties
    that do not need the oracle.  Points inside a ball of radius 0.85 stay interior under
    any rotation + |t|<=0.1, so (i) total mass = out_weight * sum(point_weight) + bg*G,
    (ii) linearity in out_weight, (iii) with constant ds_dout = c the position gradient
    vanishes and d point_weight = c*out_weight, (iv) ds_dbackground = sum(ds_dout)."""
    P, n = (1_000_000, 128) if config == "C2" else (10_000_000, 256)
    f32 = dict(device=dev, dtype=torch.float32)
    pts = _ball_points(P, dev, torch.float32)
    rng = np.random.default_rng(1)
    R = T(D.random_rotations(rng, 1).astype(np.float32), dev)
    t = T((0.05 * rng.normal(size=(1, 3))).clip(-0.1, 0.1).astype(np.float32), dev)
    ow = torch.tensor([1.5], **f32)
    bg = torch.tensor([0.25], **f32)
    out = dpr_amd.raster((n, n, n), pts, R, t, bg, ow, algo=algo)
    G = n ** 3
    total = float(out.double().sum())
    assert abs(total - (1.5 * P + 0.25 * G)) <= 2e-4 * (1.5 * P + 0.25 * G)
    out2 = dpr_amd.raster((n, n, n), pts, R, t, bg, 2 * ow, algo=algo)
    lin = (out2 - bg) - 2 * (out - bg)
    assert float(lin.abs().max()) <= 1e-3 * float((out - bg).abs().max())
    # pullback with constant sensitivity
    g = dpr_amd.empty_grid((n, n, n), 1, torch.float32, dev)
    g.fill_(0.5)
    pb = dpr_amd.raster_pullback_(g, pts, R, t, bg, ow, algo=algo)
    assert float(pb.points.abs().max()) <= 1e-3 * n  # exact 0 up to fp32 rounding * scale
    assert_close(pb.point_weight, np.full(P, 0.75, dtype=np.float32), 1e-5)
    assert abs(float(pb.background[0]) - 0.5 * G) <= 1e-3 * 0.5 * G
    assert abs(float(pb.out_weight[0]) - 0.5 * P) <= 1e-3 * 0.5 * P
    # random sensitivity: background gradient is the plain sum
    g2 = torch.randn(1, n, n, n, **f32).permute(3, 2, 1, 0)
    pb2 = dpr_amd.raster_pullback_(g2, pts, R, t, bg, ow, algo=algo)
    assert abs(float(pb2.background[0]) - float(g2.double().sum())) <= 1e-3 * np.sqrt(G)
    # <g2, out - bg> = ow * d/d(ow) => adjoint identity between forward and pullback
    lhs = float((g2.double() * (out.double() - 0.25)).sum())
    rhs = 1.5 * float(pb2.out_weight[0])
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs), np.sqrt(P))


def test_batched_projection_share_properties(dev):
    """BASELINE.json config 4 at a one-GPU share (10M points -> 512^2 orthographic projection,
    8 of the 64 poses a GPU owns; the poses are binned as one pose group): mass per pose,
    batch == loop of singles, pose groups == per-pose pipeline == direct kernels, and the
    adjoint identity <g, out - bg> = ow * d/d(ow) per pose."""
    P, n, B = 10_000_000, 512, 8
    f32 = dict(device=dev, dtype=torch.float32)
    pts = _ball_points(P, dev, torch.float32)
    rng = np.random.default_rng(5)
    R = T(D.random_rotations(rng, B)[:, :2, :].astype(np.float32), dev)
    t = T((0.05 * rng.normal(size=(B, 2))).clip(-0.1, 0.1).astype(np.float32), dev)
    ow = torch.linspace(0.5, 2.0, B, **f32)
    bg = torch.linspace(-1.0, 1.0, B, **f32)
    out = dpr_amd.raster((n, n), pts, R, t, bg, ow, algo="tiled")
    sums = out.double().sum(dim=(0, 1)).cpu().numpy()
    expect = ow.double().cpu().numpy() * P + bg.double().cpu().numpy() * n * n
    np.testing.assert_allclose(sums, expect, rtol=2e-4)
    single = dpr_amd.raster((n, n), pts, R[5], t[5], float(bg[5]), float(ow[5]), algo="tiled")
    assert_close(out[..., 5], single.cpu().numpy(), 2e-5, "pose 5 of the batch vs single call")
    g = torch.randn(B, n, n, **f32).permute(2, 1, 0)
    pb = dpr_amd.raster_pullback_(g, pts, R, t, bg, ow, algo="tiled")
    lhs = (g.double() * (out.double() - bg.double())).sum(dim=(0, 1)).cpu().numpy()
    rhs = (ow.double() * pb.out_weight.double()).cpu().numpy()
    np.testing.assert_allclose(lhs, rhs, rtol=2e-3, atol=2e-3 * np.sqrt(P))
    np.testing.assert_allclose(pb.background.double().cpu().numpy(),
                               g.double().sum(dim=(0, 1)).cpu().numpy(), rtol=0, atol=1e-3 * n)
    # the same batch through the per-pose pipeline and through the direct kernels
    out1 = dpr_amd.raster((n, n), pts, R, t, bg, ow, algo="tiled", max_pose_group=1)
    pb1 = dpr_amd.raster_pullback_(g, pts, R, t, bg, ow, algo="tiled", max_pose_group=1)
    assert (dpr_amd.workspace_bytes("raster", (n, n), P, B, 3, torch.float32, "tiled", max_pose_group=1)
            < dpr_amd.workspace_bytes("raster", (n, n), P, B, 3, torch.float32, "tiled") / 3)
    assert_close(out, out1.cpu().numpy(), 2e-5, "pose groups vs per-pose pipeline: out")
    assert_close(pb.points, pb1.points.cpu().numpy(), 1e-4, "pose groups vs per-pose: ds_dpoints")
    assert_close(pb.rotation, pb1.rotation.cpu().numpy(), 1e-3, "pose groups vs per-pose: ds_drotation")
    pb_a = dpr_amd.raster_pullback_(g, pts, R, t, bg, ow, algo="atomic")
    assert_close(pb.points, pb_a.points.cpu().numpy(), 1e-4, "tiled vs atomic: ds_dpoints")
    assert_close(pb.translation, pb_a.translation.cpu().numpy(), 1e-3, "tiled vs atomic: ds_dtranslation")


@pytest.mark.parametrize("algo", ["tiled", "chunked"])
def test_large_grid_fp64_properties(dev, algo):
    """BASELINE.json config 5 shape per pose (512^3 fp64 grid, 16384 tiles: the binning
    kernels need > 48 KiB of dynamic LDS) with 2 M points: mass conservation, constant-
    sensitivity pullback and the forward/pullback adjoint identity."""
    P, n = 2_000_000, 512
    f64 = dict(device=dev, dtype=torch.float64)
    pts = _ball_points(P, dev, torch.float64, seed=3)
    rng = np.random.default_rng(2)
    R = T(D.random_rotations(rng, 1), dev)
    t = T((0.05 * rng.normal(size=(1, 3))).clip(-0.1, 0.1), dev)
    ow = torch.tensor([0.75], **f64)
    out = dpr_amd.raster((n, n, n), pts, R, t, None, ow, algo=algo)
    assert abs(float(out.sum()) - 0.75 * P) <= 1e-9 * P
    g = torch.randn(1, n, n, n, **f64).permute(3, 2, 1, 0)
    pb = dpr_amd.raster_pullback_(g, pts, R, t, None, ow, algo=algo)
    assert abs(float(pb.background[0]) - float(g.sum())) <= 1e-8 * float(g.abs().sum())
    lhs = float((g * out).sum())
    rhs = 0.75 * float(pb.out_weight[0])
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1.0)
    del out, g
    gc = dpr_amd.empty_grid((n, n, n), 1, torch.float64, dev)
    gc.fill_(2.0)
    pb = dpr_amd.raster_pullback_(gc, pts, R, t, None, ow, algo=algo)
    assert float(pb.points.abs().max()) <= 1e-9 * n
    assert_close(pb.point_weight, np.full(P, 1.5), 1e-12)


def _hilbert_keys(q, bits):
    """Skilling's AxesToTranspose + interleave on integer coordinates q (P, n), numpy."""
    X = [q[:, j].astype(np.int64).copy() for j in range(q.shape[1])]
    n = len(X)
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        Pm = Q - 1
        for i in range(n):
            hit = (X[i] & Q) != 0
            X0_new = np.where(hit, X[0] ^ Pm, X[0])
            t = (X[0] ^ X[i]) & Pm
            X0_new = np.where(hit, X0_new, X[0] ^ t)
            Xi_new = np.where(hit, X[i], X[i] ^ t)
            if i == 0:  # t == 0 for i == 0: only the `hit` branch changes X[0]
                X[0] = X0_new
            else:
                X[0], X[i] = X0_new, Xi_new
        Q >>= 1
    for i in range(1, n):
        X[i] = X[i] ^ X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[n - 1] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    X = [x ^ t for x in X]
    key = np.zeros_like(X[0])
    for bit in range(bits):
        for j in range(n):
            key |= ((X[j] >> bit) & 1) << (n * bit + (n - 1 - j))
    return key


@pytest.mark.parametrize("npdt,tdt", DTYPES)
@pytest.mark.parametrize("n_in", [2, 3])
def test_sort_points_is_a_morton_permutation(oracle, dev, npdt, tdt, n_in):
    """dpr_sort_points_*: a permutation, Hilbert keys non-decreasing, weights follow, runs of
    consecutive points are compact, and the rasterisation of the sorted cloud equals that of
    the original (order-independent up to rounding); gradients map back through perm."""
    rng = np.random.default_rng(9)
    P = 50_000
    pts = (0.45 * rng.normal(size=(P, n_in))).astype(npdt)
    pts[:5] = [[np.nan] * n_in, [5.0] * n_in, [-5.0] * n_in, [0.999] * n_in, [-1.0] * n_in]
    pw = rng.uniform(size=P).astype(npdt)
    sp, perm, spw = dpr_amd.sort_points(T(pts, dev), T(pw, dev))
    perm_h = perm.cpu().numpy().astype(np.int64)
    assert sorted(perm_h.tolist()) == list(range(P))
    np.testing.assert_array_equal(sp.cpu().numpy(), pts[perm_h])
    np.testing.assert_array_equal(spw.cpu().numpy(), pw[perm_h])
    bits = 10 if n_in == 3 else 12  # (the public sort: 30-bit keys in 3-D)
    x = (pts[perm_h].astype(npdt) * npdt(0.5) + npdt(0.5)) * npdt(1 << bits)
    q = np.where(~(x > 0), 0, np.where(x >= (1 << bits) - 1, (1 << bits) - 1, np.nan_to_num(x).astype(np.int64))).astype(np.int64)
    key = _hilbert_keys(q, bits)
    assert (np.diff(key) >= 0).all()
    # the point of a Hilbert order: every run of consecutive points is a compact blob (no jumps)
    body = pts[perm_h][np.isfinite(pts[perm_h]).all(axis=1) & (np.abs(pts[perm_h]) < 1).all(axis=1)]
    run = 512
    ext = np.array([np.ptp(body[i:i + run], axis=0).max() for i in range(0, len(body) - run, 97)])
    ideal = 2.0 * (run / len(body)) ** (1.0 / n_in)  # side of a cube holding `run` uniform points
    assert np.median(ext) < 6 * ideal and ext.max() < 40 * ideal
    # same image, gradients map back through perm
    fin = np.isfinite(pts).all(axis=1)
    pts[~fin] = 7.0
    d = D.make(n_points=10, n_in=n_in, n_out=2 if n_in == 2 else 3, batch=1, grid_n=24, seed=3, dtype=npdt)
    sp, perm = dpr_amd.sort_points(T(pts, dev))
    a = dpr_amd.raster(d.grid, T(pts, dev), T(d.rotations, dev), T(d.translations, dev))
    b = dpr_amd.raster(d.grid, sp, T(d.rotations, dev), T(d.translations, dev))
    assert_close(b, a.cpu().numpy(), tol(npdt, "out"), "sorted vs original image")
    g = grid_to_dev(D.make(n_points=1, n_in=n_in, n_out=len(d.grid), batch=1, grid_n=24, seed=4, dtype=npdt).ds_dout, dev)
    pa = dpr_amd.raster_pullback_(g, T(pts, dev), T(d.rotations, dev), T(d.translations, dev))
    pb = dpr_amd.raster_pullback_(g, sp, T(d.rotations, dev), T(d.translations, dev))
    back = torch.empty_like(pb.points)
    back.index_copy_(0, perm.long(), pb.points)
    assert_close(back, pa.points.cpu().numpy(), tol(npdt, "points"), "gradients through perm")


@pytest.mark.parametrize("algo", ALGOS)
def test_runs_on_the_callers_stream_and_in_hip_graphs(oracle, dev, algo):
    """The library only enqueues on the stream it is given (no sync, no allocation, no global
    state): (i) results on a side stream match, (ii) forward + pullback can be captured into
    a HIP graph and replayed on new input contents."""
    d = D.make(n_points=20_000, n_in=3, n_out=3, batch=1, grid_n=48, seed=33, dtype=np.float32)
    pts = T(d.points, dev)
    R, t = T(d.rotations, dev), T(d.translations, dev)
    g = grid_to_dev(d.ds_dout, dev)
    out = dpr_amd.empty_grid(d.grid, 1, torch.float32, dev)
    ws = torch.empty(max(16, *(dpr_amd.workspace_bytes(op, d.grid, d.n_points, 1, 3, torch.float32, algo)
                               for op in ("raster", "pullback"))),  # one buffer for both calls
                     dtype=torch.uint8, device=dev)
    d_pts = torch.empty(d.n_points, 3, device=dev)
    d_pw = torch.empty(d.n_points, device=dev)

    def run():
        dpr_amd.raster_(out, pts, R, t, algo=algo, workspace=ws)
        return dpr_amd.raster_pullback_(g, pts, R, t, algo=algo, workspace=ws, ds_dpoints=d_pts,
                                        ds_dpoint_weight=d_pw)

    ref_out = oracle.raster(d.grid, d.points, d.rotations, d.translations, dtype=np.float32)
    ref_pb = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations,
                                    dtype=np.float32)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        pb = run()
    side.synchronize()
    _compare(ref_out, ref_pb, out, pb, np.float32)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run()  # warm-up outside capture
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        pb = run()
    # new contents in the same buffers, then replay
    pts2 = d.points[::-1].copy() * np.float32(0.9)
    pts.copy_(T(pts2, dev))
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    ref_out2 = oracle.raster(d.grid, pts2, d.rotations, d.translations, dtype=np.float32)
    ref_pb2 = oracle.raster_pullback(d.ds_dout, pts2, d.rotations, d.translations,
                                     dtype=np.float32)
    _compare(ref_out2, ref_pb2, out, pb, np.float32)


def test_concurrent_host_threads_on_their_own_streams(oracle, dev):
    """SURVEY.md 8b "Threading": the library is re-entrant -- several host threads, each with
    its own stream, workspace and outputs, call it at the same time (ctypes drops the GIL
    during the call) and every thread gets the oracle's answer; error strings are per thread."""
    import threading

    cases = []
    for k, (algo, n_out, grid_n) in enumerate([("tiled", 3, 44), ("atomic", 3, 30), ("tiled", 2, 120),
                                                ("chunked", 3, 40)]):
        d = D.make(n_points=25_000 + 1000 * k, n_in=3, n_out=n_out, batch=2, grid_n=grid_n,
                   seed=50 + k, dtype=np.float32)
        cases.append((algo, d))
    results, errors = [None] * len(cases), []

    def worker(i):
        try:
            algo, d = cases[i]
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                args = (T(d.points, dev), T(d.rotations, dev), T(d.translations, dev),
                        T(d.backgrounds, dev), T(d.weights, dev), T(d.point_weights, dev))
                g = grid_to_dev(d.ds_dout, dev)
                for _ in range(5):  # keep the threads overlapping for a while
                    out = dpr_amd.raster(d.grid, *args, algo=algo)
                    pb = dpr_amd.raster_pullback_(g, *args, algo=algo)
                if i == 0:  # a failing call in one thread must not disturb the others
                    with pytest.raises(dpr_amd.DprError):
                        dpr_amd.raster_(out, *args, algo="atomic", keep_binning=True,
                                        workspace=torch.empty(1 << 20, dtype=torch.uint8, device=dev))
            stream.synchronize()
            results[i] = (out, pb)
        except Exception as e:  # surfaced in the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(cases))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for (algo, d), (out, pb) in zip(cases, results):
        ref_out = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds,
                                d.weights, d.point_weights, dtype=np.float32)
        ref_pb = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                        d.point_weights, dtype=np.float32)
        _compare(ref_out, ref_pb, out, pb, np.float32)


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_against_oracle(oracle, dev, seed):
    """Seeded fuzz over shapes the fixed tests do not hit: random point counts (incl. sizes
    around the block / chunk / sub-chunk boundaries), anisotropic grids with tile remainders,
    batch sizes, optional arguments, dtypes and algorithms; partly out-of-range clouds."""
    rng = np.random.default_rng(1000 + seed)
    n_in, n_out = SHAPES[rng.integers(len(SHAPES))]
    npdt, tdt = DTYPES[rng.integers(2)]
    algo = ALGOS[rng.integers(len(ALGOS))]
    P = int(rng.choice([1, 63, 64, 65, 255, 257, 1023, 1025, 4095, 4097, 8193, 20481, 50_000]))
    B = int(rng.choice([1, 2, 3, 4, 5, 9, 17, 33]))  # pose groups of 1 ... 16 (+ remainders)
    grid = tuple(int(g) for g in rng.integers(3, 150 if n_out == 2 else 90, size=n_out))
    spread = float(rng.choice([0.05, 0.4, 0.9]))
    pts = (spread * rng.normal(size=(P, n_in))).astype(npdt)
    R = D.random_rotations(rng, B, n_in)[:, :n_out, :].astype(npdt)
    t = (0.2 * rng.normal(size=(B, n_out))).astype(npdt)
    use = rng.integers(0, 2, size=3).astype(bool)
    bg = rng.normal(size=B).astype(npdt) if use[0] else None
    ow = (rng.uniform(0.5, 3, size=B)).astype(npdt) if use[1] else None
    pw = rng.uniform(0.1, 2, size=P).astype(npdt) if use[2] else None
    g = np.asfortranarray(rng.normal(size=grid + (B,)).astype(npdt))
    ref_out = oracle.raster(grid, pts, R, t, bg, ow, pw, dtype=npdt)
    ref_pb = oracle.raster_pullback(g, pts, R, t, ow, pw, dtype=npdt)
    out = dpr_amd.raster(grid, T(pts, dev), T(R, dev), T(t, dev), T(bg, dev), T(ow, dev),
                         T(pw, dev), algo=algo)
    pb = dpr_amd.raster_pullback_(grid_to_dev(g, dev), T(pts, dev), T(R, dev), T(t, dev),
                                  T(bg, dev), T(ow, dev), T(pw, dev), algo=algo)
    _compare(ref_out, ref_pb, out, pb, npdt)
    if seed % 3 == 0 and algo != "chunked":  # the fused squared-error pullback on the same case
        target = np.asfortranarray(rng.normal(size=grid + (B,)).astype(npdt))
        rp, loss = dpr_amd.raster_residual_pullback_(out, grid_to_dev(target, dev), T(pts, dev),
                                                     T(R, dev), T(t, dev), T(bg, dev), T(ow, dev),
                                                     T(pw, dev), scale=-2.0, algo=algo)
        o = out.cpu().numpy().reshape(grid + (B,))
        ref_rp, ref_loss = oracle.residual_pullback(o, target, pts, R, t, ow, pw, scale=-2.0,
                                                    dtype=npdt)
        assert_close(rp.points, ref_rp.points, tol(npdt, "points"), "residual ds_dpoints")
        assert_close(rp.rotation, ref_rp.rotation, tol(npdt, "pose"), "residual ds_drotation")
        assert_close(rp.out_weight, ref_rp.out_weight, tol(npdt, "pose"), "residual ds_dout_weight")
        assert_close(loss, ref_loss.astype(npdt).reshape(loss.shape), tol(npdt, "out"), "loss")


def test_residual_pullback_under_auto_where_the_plain_pullback_takes_the_direct_3d_kernels(oracle, dev):
    """Round 5's advisor finding: AUTO sends 3-D pullbacks of coherent clouds (and of large batches in any
    order) to the direct gather kernels of DPR_ALGO_CHUNKED, which have no residual variant -- the
    residual entry points (always AUTO through the plain ABI) failed with DPR_ERR_UNSUPPORTED_ALGO on
    such shapes.  DPR_OP_RESIDUAL_PULLBACK now resolves among the algorithms that have one."""
    # (a) one pose of a coherent cloud: plain pullback -> chunked, residual pullback -> not chunked
    d = D.make(n_points=50_000, n_in=3, n_out=3, batch=1, grid_n=64, seed=71, dtype=np.float32)
    assert dpr_amd.resolve_algo("pullback", d.grid, d.n_points, 1, 3, coherent_points=True) == "chunked"
    assert dpr_amd.resolve_algo("residual_pullback", d.grid, d.n_points, 1, 3, coherent_points=True) != "chunked"
    spts, _ = dpr_amd.sort_points(T(d.points, dev))
    pts = spts.cpu().numpy()
    R, t, ow = d.rotations, d.translations, d.weights
    out = dpr_amd.raster(d.grid, spts, T(R, dev), T(t, dev), None, T(ow, dev), None, coherent_points=True)
    target = np.asfortranarray(np.random.default_rng(72).normal(size=d.grid + (1,)).astype(np.float32))
    rp, loss = dpr_amd.raster_residual_pullback_(out, grid_to_dev(target, dev), spts, T(R, dev), T(t, dev),
                                                 None, T(ow, dev), None, scale=-2.0, coherent_points=True)
    o = out.cpu().numpy().reshape(d.grid + (1,))
    ref_rp, ref_loss = oracle.residual_pullback(o, target, pts, R, t, ow, None, scale=-2.0, dtype=np.float32)
    assert_close(rp.points, ref_rp.points, tol(np.float32, "points"), "residual ds_dpoints")
    assert_close(rp.rotation, ref_rp.rotation, tol(np.float32, "pose"), "residual ds_drotation")
    assert_close(loss, ref_loss.astype(np.float32).reshape(loss.shape), tol(np.float32, "out"), "loss")
    # (b) a batch above the sort-inside-the-call thresholds, cloud in any order: AUTO == explicit tiled
    P, B, grid = 3_000_000, 16, (128, 128, 128)
    assert dpr_amd.resolve_algo("pullback", grid, P, B, 3) == "chunked"
    assert dpr_amd.resolve_algo("residual_pullback", grid, P, B, 3) != "chunked"
    assert (dpr_amd.workspace_bytes("residual_pullback", grid, P, B, 3) !=
            dpr_amd.workspace_bytes("pullback", grid, P, B, 3))
    g = torch.Generator(device=dev).manual_seed(73)
    pts = 0.4 * torch.randn((P, 3), device=dev, generator=g)
    rng = np.random.default_rng(74)
    R = T(D.random_rotations(rng, B, 3).astype(np.float32), dev)
    t = T((0.1 * rng.normal(size=(B, 3))).astype(np.float32), dev)
    out = dpr_amd.raster(grid, pts, R, t)
    target = dpr_amd.to_grid_layout(torch.randn(grid + (B,), device=dev, generator=g))
    auto, loss_a = dpr_amd.raster_residual_pullback_(out, target, pts, R, t, scale=2.0)
    tiled, loss_t = dpr_amd.raster_residual_pullback_(out, target, pts, R, t, scale=2.0, algo="tiled")
    plain = dpr_amd.raster_pullback_(2.0 * (out - target), pts, R, t)  # (AUTO: the direct kernels)
    assert_close(auto.points, tiled.points.cpu().numpy(), 1e-6, "AUTO vs tiled residual ds_dpoints")
    assert_close(auto.points, plain.points.cpu().numpy(), 1e-4, "residual vs plain ds_dpoints")
    assert_close(auto.rotation, plain.rotation.cpu().numpy(), 1e-3, "residual vs plain ds_drotation")
    assert_close(loss_a, loss_t.cpu().numpy(), 1e-6, "loss")


# ------------------------------------------------------------------ fixed-point LDS accumulators
def test_tiled_fp32_forward_is_independent_of_the_point_order(dev):
    """The fp32 tile kernels accumulate in 64-bit fixed point: the sums are exact integers, so a
    permutation of the cloud gives the same `out` BIT FOR BIT (no tile of this cloud is split
    into parts -- parts are added in floating point)."""
    d = D.make(n_points=200_000, n_in=3, n_out=3, batch=1, grid_n=96, seed=5, dtype=np.float32)
    # uniform in a cube: ~2500 points per tile, below the 4096 records at which a tile is split
    d.points = np.random.default_rng(5).uniform(-0.9, 0.9, size=d.points.shape).astype(np.float32)
    pts, pw = T(d.points, dev), T(d.point_weights, dev)
    R, t = T(d.rotations, dev), T(d.translations, dev)
    ow = T(d.weights, dev)
    perm = torch.randperm(pts.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    for weights in (None, pw):
        a = dpr_amd.raster(d.grid, pts, R, t, None, ow, weights, algo="tiled")
        b = dpr_amd.raster(d.grid, pts[perm].contiguous(), R, t, None, ow,
                           None if weights is None else weights[perm].contiguous(), algo="tiled")
        assert torch.equal(a, b)


@pytest.mark.parametrize("coherent", [False, True])
def test_tiled_fp32_point_weights_over_many_orders_of_magnitude(oracle, dev, coherent):
    """Fixed-point scale = 2^k / max|weight|: weights 2^-14 and more below the largest lose low
    bits of THEIR mantissa, never more than 2^-38 of the largest weight -- norm-wise parity and a
    max-abs bound relative to the largest weight."""
    d = D.make(n_points=200_000, n_in=3, n_out=3, batch=1, grid_n=64, seed=6, dtype=np.float32)
    rng = np.random.default_rng(7)
    pw = (10.0 ** rng.uniform(-12, 3, size=d.n_points)).astype(np.float32)
    pw[::2] *= -1  # signed
    pts = d.points
    if coherent:
        pts = pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))]  # any coherent order will do
    ref64 = oracle.raster(d.grid, pts, d.rotations, d.translations, None, d.weights, pw, dtype=np.float64)
    ref32 = oracle.raster(d.grid, pts, d.rotations, d.translations, None, d.weights, pw, dtype=np.float32)
    out = dpr_amd.raster(d.grid, T(pts, dev), T(d.rotations, dev), T(d.translations, dev), None,
                         T(d.weights, dev), T(pw, dev), algo="tiled", coherent_points=coherent)
    assert_close(out, ref64.astype(np.float32), 5e-5, "out vs fp64 oracle")
    # against the fp32 oracle (identical contributions, sequential fp32 sums): a few ulp of the
    # largest values
    err = np.abs(out.cpu().numpy().astype(np.float64) - ref32.astype(np.float64)).max()
    assert err <= 2e-6 * np.abs(ref32).max(), err


def _weight_field(kind, x01, rng):
    """Positive point weights over 40-50 binary orders as a function of the position x01 in [0, 1]
    along axis 0: `slabs` -- five regions 2^0, 2^-10, ... 2^-40 (times [1, 2)): whole tiles / chunks
    hold ONLY small weights; `smooth` -- 2^(-50 x01); `random` -- 2^U(-50, 0) per point."""
    if kind == "slabs":
        return (2.0 ** (-10.0 * np.floor(np.clip(x01, 0, 0.999) * 5)) * rng.uniform(1, 2, x01.shape)).astype(np.float32)
    if kind == "smooth":
        return (2.0 ** (-50.0 * x01)).astype(np.float32)
    return (2.0 ** rng.uniform(-50, 0, x01.shape)).astype(np.float32)


@pytest.mark.parametrize("kind", ["slabs", "smooth", "random"])
@pytest.mark.parametrize("path", ["tiled", "tiled_coherent", "owner", "chunkown2d", "atomic"])
def test_fp32_cells_reached_only_by_small_weights_keep_their_relative_precision(oracle, dev, path, kind):
    """The reference adds float contributions with atomics (src/raster.jl:62-64): a cell that only
    points of weight 2^-40 reach still carries 24 good bits.  The fp32 forwards here accumulate in
    64-bit fixed point scaled by the LARGEST weight of a scope (call / tile / chunk); where the non-zero
    weights of that scope span more than 2^10 they fall back to f64 atomics (fix_guard_range), so the
    same holds: PER-CELL relative error <= 1e-5 against the fp32 oracle on every cell that is not itself
    a far corner of its contributions (value >= 2^-12 of the local weight; for `random` weights: every
    non-zero cell).  Before round 6 the scale came from the largest weight of the whole call and such
    cells came out as exactly 0."""
    rng = np.random.default_rng(61)
    n_out = 2 if path == "chunkown2d" else 3
    grid = (160, 96) if n_out == 2 else (160, 48, 40)
    P, B = 300_000, (3 if n_out == 2 else 1)
    pts = rng.uniform(-0.95, 0.95, size=(P, 3)).astype(np.float32)
    pw = _weight_field(kind, (pts[:, 0].astype(np.float64) + 1) / 2, rng)
    R = np.broadcast_to(np.eye(n_out, 3, dtype=np.float32), (B, n_out, 3)).copy()
    t = np.zeros((B, n_out), dtype=np.float32)
    t[:, 1:] = 0.01 * np.arange(B, dtype=np.float32)[:, None]  # (poses differ; axis 0 stays put)
    ow = np.full(B, 3.0, dtype=np.float32)
    tp, tw = T(pts, dev), T(pw, dev)
    coherent = path in ("tiled_coherent", "owner", "chunkown2d")
    if coherent:
        res = dpr_amd.sort_points(tp, tw)
        tp, tw = res[0], res[2]
        pts, pw = tp.cpu().numpy(), tw.cpu().numpy()
    algo = {"tiled": "tiled", "tiled_coherent": "tiled", "owner": "chunked", "chunkown2d": "chunked",
            "atomic": "atomic"}[path]
    out = dpr_amd.raster(grid, tp, T(R, dev), T(t, dev), None, T(ow, dev), tw, algo=algo,
                         coherent_points=coherent).cpu().numpy().astype(np.float64)
    ref = oracle.raster(grid, pts, R, t, None, ow, pw, dtype=np.float32).astype(np.float64)
    out, ref = out.reshape(ref.shape), ref
    # local weight scale of a cell from its index along axis 0 (axis 0 is the fastest: column-major)
    ix = np.arange(grid[0], dtype=np.float64)
    x01 = (ix + 0.5) / grid[0]
    if kind == "slabs":
        wloc = 2.0 ** (-10.0 * np.floor(np.clip(x01, 0, 0.999) * 5))
    elif kind == "smooth":
        wloc = 2.0 ** (-50.0 * x01)
    else:
        wloc = np.zeros_like(x01)
    wloc = wloc.reshape((grid[0],) + (1,) * (ref.ndim - 1))
    mask = (ref > 0) & (ref >= 2.0 ** -12 * 3.0 * wloc)
    assert mask.sum() > 0.5 * (ref > 0).sum(), "the threshold must keep most cells"
    rel = np.abs(out - ref)[mask] / ref[mask]
    assert rel.max() <= 1e-5, (f"{path}/{kind}: worst per-cell relative error {rel.max():.3e} "
                               f"({(rel > 1e-5).sum()} of {mask.sum()} cells)")
    # and nothing that should be there is missing
    assert ((out == 0) & mask).sum() == 0


@pytest.mark.parametrize("mode", ["coherent2d", "coherent3d", "sort_inside"])
@pytest.mark.parametrize("bad", [None, np.nan])
def test_tiled_fp32_batch_scale_covers_weights_only_later_poses_see(oracle, dev, mode, bad):
    """Local binning publishes ONE max |point_weight| for all poses of a local batch (the
    fixed-point scale of k_tile_splat_runs).  The heaviest points -- and one NaN weight -- are
    outside the grid under pose 0 and inside under the later poses: the maximum must cover them
    (round 4's kernel looked at pose 0's valid points only and corrupted the later poses)."""
    # grids with more than 2048 tiles per pose (no pose groups: local batches); sort_inside: more
    # than 4096 tiles, B >= 4, >= 2e5 points
    grid = {"coherent2d": (1472, 1472), "coherent3d": (320, 256, 256), "sort_inside": (2112, 2112)}[mode]
    n_out = len(grid)
    n_points, batch = 220_000, 4
    d = D.make(n_points=n_points, n_in=3, n_out=n_out, batch=batch, grid_n=grid, seed=31, dtype=np.float32)
    rng = np.random.default_rng(32)
    pts = rng.uniform(-0.45, 0.45, size=(n_points, 3)).astype(np.float32)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))]  # coherent in memory
    R = np.broadcast_to(np.eye(n_out, 3, dtype=np.float32), (batch, n_out, 3)).copy()
    t = np.zeros((batch, n_out), dtype=np.float32)
    t[0, 0] = 0.7  # pose 0 pushes x > 0.3 out of the grid; the other poses keep everything
    pw = rng.uniform(0.5, 1.0, size=n_points).astype(np.float32)
    heavy = pts[:, 0] > 0.35
    pw[heavy] *= np.float32(3.0e4)  # 15 bits above the weights pose 0 sees
    if bad is not None:
        pw[np.flatnonzero(heavy)[7]] = bad
    ref = oracle.raster(grid, pts, R, t, None, d.weights, pw, dtype=np.float32)
    out = dpr_amd.raster(grid, T(pts, dev), T(R, dev), T(t, dev), None, T(d.weights, dev), T(pw, dev),
                         algo="tiled", coherent_points=mode.startswith("coherent")).cpu().numpy()
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    for b in range(batch):
        fin = np.isfinite(ref[..., b])
        assert_close(out[..., b][fin], ref[..., b][fin], 5e-5, f"pose {b}")


@pytest.mark.parametrize("bad", [np.inf, np.nan])
@pytest.mark.parametrize("where", ["point_weight", "out_weight"])
def test_tiled_fp32_non_finite_weights_fall_back_to_ieee_sums(oracle, dev, bad, where):
    """A NaN / Inf weight switches the tile kernels from fixed point to f64 atomics: NaN and Inf
    land where the reference's arithmetic puts them, finite voxels stay right."""
    d = D.make(n_points=50_000, n_in=3, n_out=3, batch=1, grid_n=48, seed=8, dtype=np.float32)
    pw, ow = d.point_weights.copy(), d.weights.copy()
    if where == "point_weight":
        pw[123] = bad
    else:
        ow[0] = bad
    ref = oracle.raster(d.grid, d.points, d.rotations, d.translations, None, ow, pw, dtype=np.float32)
    out = dpr_amd.raster(d.grid, T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), None,
                         T(ow, dev), T(pw, dev), algo="tiled").cpu().numpy()
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    assert np.array_equal(np.isposinf(out), np.isposinf(ref))
    assert np.array_equal(np.isneginf(out), np.isneginf(ref))
    fin = np.isfinite(ref)
    assert fin.any() or where == "out_weight"
    if fin.any():
        assert_close(out[fin], ref[fin], 5e-5)


def test_pre_canonicalised_rotation(oracle, dev):
    """`column_major_rotation` hands the pose over in the C ABI's memory order once; the calls then
    skip their transpose kernels.  Same results as the tensor form, single pose and batch, forward
    and pullback; a snapshot (later writes to the source tensor are not seen)."""
    d = D.make(n_points=5_000, n_in=3, n_out=3, batch=2, grid_n=16, seed=51, dtype=np.float64)
    pts = T(d.points, dev)
    g = grid_to_dev(d.ds_dout, dev)
    for single in (True, False):
        R = T(d.rotations[0] if single else d.rotations, dev).clone()
        t = T(d.translations[0] if single else d.translations, dev)
        Rc = dpr_amd.column_major_rotation(R)
        assert Rc.shape == tuple(R.shape) and Rc.ndim == R.ndim
        a, b = dpr_amd.raster(d.grid, pts, R, t), dpr_amd.raster(d.grid, pts, Rc, t)
        assert_close(a, b.cpu().numpy(), 1e-13)  # (global atomics: equal up to the summation order)
        gg = g[..., 0] if single else g
        pa = dpr_amd.raster_pullback_(gg, pts, R, t, algo="atomic")
        pb = dpr_amd.raster_pullback_(gg, pts, Rc, t, algo="atomic")
        assert torch.equal(pa.points, pb.points)
        assert_close(pa.rotation, pb.rotation.cpu().numpy(), 1e-12)  # (per-pose sums: float atomics)
        R.mul_(0.5)  # the snapshot keeps the old pose
        assert_close(dpr_amd.raster(d.grid, pts, Rc, t), a.cpu().numpy(), 1e-13)
    with pytest.raises(dpr_amd.DimensionMismatch):
        dpr_amd.raster(d.grid, pts, dpr_amd.column_major_rotation(T(d.rotations[0, :2], dev)), T(d.translations[0], dev))


def test_autograd_rule_with_mixed_dtypes_and_strided_views(dev):
    """raster_ad keeps the forward's binning for its backward pass; the device-side check of that
    pairing compares buffer addresses, so the rule must hand the SAME canonical tensors to both
    calls when points / point_weight need a dtype promotion or are non-contiguous views."""
    torch.manual_seed(3)
    P = 400_000  # AUTO -> tiled: the pair shares
    base = (0.4 * torch.randn(P, 6, device=dev, dtype=torch.float32))
    pts32 = base[:, ::2].detach().requires_grad_(True)           # strided view, fp32
    pw_all = torch.rand(2 * P, device=dev, dtype=torch.float32)
    pw32 = pw_all[::2].detach().requires_grad_(True)             # strided view, fp32
    R = torch.eye(3, device=dev, dtype=torch.float64).requires_grad_(True)   # fp64 pose: promotion
    t = torch.zeros(3, device=dev, dtype=torch.float64).requires_grad_(True)
    out = dpr_amd.raster_ad((64, 64, 64), pts32, R, t, 0.0, 1.0, pw32)
    assert out.dtype == torch.float64
    g = torch.randn_like(out)
    (out * g).sum().backward()
    for name, x in (("points", pts32), ("rotation", R), ("translation", t), ("point_weight", pw32)):
        assert x.grad is not None and torch.isfinite(x.grad).all(), name
    assert pts32.grad.dtype == torch.float32 and pw32.grad.shape == pw32.shape
    # same numbers as the explicit pullback on canonical inputs
    pb = dpr_amd.raster_pullback_(g, pts32.detach().double().contiguous(), R.detach(), t.detach(), None,
                                  None, pw32.detach().double().contiguous())
    assert_close(pts32.grad, pb.points.float().cpu().numpy(), 1e-6, "points.grad")
    assert_close(R.grad, pb.rotation.cpu().numpy(), 1e-9, "rotation.grad")


# ------------------------------------------------------------------ grids beyond 32768 tiles: slabs
def _assert_close_on_device(a, e, rtol, what):
    """norm-wise comparison of two large device tensors without float64 copies on the host"""
    err2 = na2 = ne2 = 0.0
    af, ef = a.reshape(-1), e.reshape(-1)
    step = 1 << 26
    for i in range(0, af.numel(), step):
        x, y = af[i:i + step].double(), ef[i:i + step].double()
        err2 += float(((x - y) ** 2).sum())
        na2 += float((x * x).sum())
        ne2 += float((y * y).sum())
    assert err2 ** 0.5 <= rtol * max(na2, ne2) ** 0.5, f"{what}: |a-e|={err2 ** 0.5:.3e} vs {max(na2, ne2) ** 0.5:.3e}"


_SLAB_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import dpr_amd
from oracle import oracle
from tests import data as D
from tests.test_parity_gpu import T, assert_close, grid_to_dev, tol
dev = torch.device("cuda:0")
cases = [(3, 3, (100, 40, 90), 3), (3, 3, (64, 64, 200), 1), (2, 2, (300, 400), 2), (3, 2, (260, 500), 3)]
for npdt in (np.float64, np.float32):
    for n_in, n_out, grid, B in cases:
        d = D.make(n_points=60_000, n_in=n_in, n_out=n_out, batch=B, grid_n=grid, seed=21, dtype=npdt)
        d.points[:30_000] *= 0.05     # a dense blob: tiles above the split threshold (4096 records)
        d.points[:, n_in - 1] *= 2.0  # spread along the slab axis (some points leave the grid)
        ref = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds, d.weights,
                            d.point_weights, dtype=npdt)
        rp = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                    d.point_weights, dtype=npdt)
        args = [T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), T(d.backgrounds, dev),
                T(d.weights, dev), T(d.point_weights, dev)]
        for coherent in (False, True):
            out = dpr_amd.raster(d.grid, *args, algo="tiled", coherent_points=coherent)
            assert_close(out, ref, tol(npdt, "out"), f"out {grid} {npdt.__name__}")
            pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout, dev), *args, algo="tiled",
                                          coherent_points=coherent)
            for name, kind in (("points", "points"), ("point_weight", "points"), ("rotation", "pose"),
                               ("translation", "pose"), ("background", "pose"), ("out_weight", "pose")):
                assert_close(getattr(pb, name), getattr(rp, name), tol(npdt, kind), f"{name} {grid}")
print("SLABS_OK")
"""


def test_tiled_on_grids_processed_in_slabs(dev):
    """More than kMaxTiles tiles: the tiled path walks slabs of tile layers along the last axis
    (forward: each slab re-bins the top layer of the slab below as a ghost layer for its halo;
    pullback: the slabs add up point gradients and per-pose sums).  With DPR_MAX_TILES=24 (read
    once per process, hence the child process) small grids are cut into 3-13 slabs: forward +
    pullback against the oracle in both element types, batches, a split tile, the coherent flag."""
    import os
    import subprocess
    import sys

    from tests.conftest import ROOT

    env = dict(os.environ, DPR_MAX_TILES="24")
    r = subprocess.run([sys.executable, "-c", _SLAB_SCRIPT, ROOT], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "SLABS_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("n_in,n_out,grid", [(2, 2, (8192, 4100)), (3, 2, (8192, 4100))])
def test_tiled_on_a_2d_grid_of_more_than_32768_tiles(oracle, dev, n_in, n_out, grid):
    """256 x 129 = 33 024 tiles of 32 x 32 pixels: two slabs at the library's own bound."""
    npdt = np.float32
    d = D.make(n_points=300_000, n_in=n_in, n_out=n_out, batch=2, grid_n=grid, seed=21, dtype=npdt)
    d.points[:, n_in - 1] *= 2.2  # spread along the slab axis
    ref = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds, d.weights,
                        d.point_weights, dtype=npdt)
    out = dpr_amd.raster(d.grid, T(d.points, dev), T(d.rotations, dev), T(d.translations, dev),
                         T(d.backgrounds, dev), T(d.weights, dev), T(d.point_weights, dev), algo="tiled")
    _assert_close_on_device(out, grid_to_dev(ref, dev), tol(npdt, "out"), "out")
    rp = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                d.point_weights, dtype=npdt)
    pb = dpr_amd.raster_pullback_(grid_to_dev(d.ds_dout, dev), T(d.points, dev), T(d.rotations, dev),
                                  T(d.translations, dev), T(d.backgrounds, dev), T(d.weights, dev),
                                  T(d.point_weights, dev), algo="tiled")
    assert_close(pb.points, rp.points, tol(npdt, "points"), "ds_dpoints")
    assert_close(pb.rotation, rp.rotation, tol(npdt, "pose"), "ds_drotation")
    assert_close(pb.background, rp.background, tol(npdt, "pose"), "ds_dbackground")


def test_1024_cube_fp32_forward_auto_is_tiled_and_matches_the_oracle(oracle, dev):
    """The reference README's largest grid (README.md:193: 1024^3): 131 072 tiles, four slabs.
    AUTO takes the tiled path for the forward where the cloud is dense enough on the grid (~60
    points per tile: 10^7 -> 1024^3), the direct kernels below (10^6 points: 2x faster there,
    profiles/r04_sparse_grids.txt); 10^6 points through the slabs (algo="tiled") against the
    oracle, compared on the device (4.3 GB per grid)."""
    n, P = 1024, 1_000_000
    assert dpr_amd.resolve_algo("raster", (n, n, n), 10_000_000, 1, 3) == "tiled"
    assert dpr_amd.resolve_algo("raster", (n, n, n), P, 1, 3) == "atomic"
    assert dpr_amd.resolve_algo("raster", (n, n, n), 100_000, 1, 3) == "atomic"   # README row: 1e5 points
    d = D.make(n_points=P, n_in=3, n_out=3, batch=1, grid_n=n, seed=22, dtype=np.float32)
    ref = oracle.raster(d.grid, d.points, d.rotations, d.translations, None, d.weights, dtype=np.float32,
                        threaded=True)
    out = dpr_amd.raster(d.grid, T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), None,
                         T(d.weights, dev), algo="tiled")
    _assert_close_on_device(out, grid_to_dev(ref, dev), 5e-5, "out (1024^3)")


@pytest.mark.parametrize("npdt,tdt", DTYPES)
def test_chunk_owner_pullback_with_large_nearly_equal_and_infinite_sensitivities(oracle, dev, npdt, tdt):
    """Round 4's advisor: `k_co_gather` forms the bilinear gradient from DIFFERENCES of the four neighbours
    (g10 - g00, ...) and adds the 4096 terms of a (chunk, pose) as an f32 tree.  (a) `ds_dout` = 1e4 + noise:
    the differences cancel five digits -- the point gradients still meet the tolerance against the oracle's
    product form (they are compared on the scale of the sensitivities, as everywhere).  (b) an Inf cell: the
    difference form gives NaN (Inf - Inf) where the reference's sum of products gives +-Inf -- pinned here as
    'non-finite exactly where the oracle is non-finite, equal everywhere else'."""
    d = D.make(n_points=20_000, n_in=3, n_out=2, batch=5, grid_n=64, seed=77, dtype=npdt)
    rng = np.random.default_rng(5)
    g = np.asfortranarray((1.0e4 + rng.normal(size=d.ds_dout.shape)).astype(npdt))
    args = (T(d.points, dev), T(d.rotations, dev), T(d.translations, dev), None, T(d.weights, dev), None)
    pb = dpr_amd.raster_pullback_(grid_to_dev(g, dev), *args, algo="chunked")
    ref = oracle.raster_pullback(g, d.points, d.rotations, d.translations, d.weights, None, dtype=npdt)
    scale = 1.0e4  # |ds_dout|: the sums below are O(scale * gradient of the weights)
    for name in ("points", "rotation", "translation", "out_weight", "background"):
        a, e = getattr(pb, name).cpu().numpy(), np.asarray(getattr(ref, name))
        err = np.abs(a - e).max()
        bound = tol(npdt, "pose") * max(np.abs(e).max(), scale * (1 if name == "points" else len(d.points) ** 0.5))
        assert err <= bound, f"{name}: {err} > {bound}"
    g2 = np.asfortranarray(rng.normal(size=d.ds_dout.shape).astype(npdt))
    g2[20, 31, 2] = np.inf
    pb2 = dpr_amd.raster_pullback_(grid_to_dev(g2, dev), *args, algo="chunked")
    ref2 = oracle.raster_pullback(g2, d.points, d.rotations, d.translations, d.weights, None, dtype=npdt)
    for name in ("points", "rotation", "translation", "out_weight", "background"):
        a, e = getattr(pb2, name).cpu().numpy(), np.asarray(getattr(ref2, name))
        fin = np.isfinite(e)
        assert np.array_equal(np.isfinite(a), fin), f"{name}: non-finite entries differ from the oracle's"
        if fin.any():
            assert_close(a[fin], e[fin], tol(npdt, "pose"), name + " (finite entries)")
