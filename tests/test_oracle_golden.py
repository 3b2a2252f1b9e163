"""Pins the CPU oracle against the reference's own known-answer tests
(fixtures in tests/golden/reference_known_answers.json).

Counterparts of the reference test items:
  "raster correctness"      src/raster.jl:110-310
  "digitstuple"/"voxel_shifts"  src/util.jl:10-46
  README examples           README.md:41-68, 84-183
  "raster batched consistency"  src/raster.jl:383-431
  "raster_pullback! threaded"   src/raster_pullback.jl:271-345
  test_rrule (finite differences)  test/chainrules.jl:2-90
"""
import numpy as np
import pytest

from tests import data as D


def _fwd_case(oracle, case, dtype):
    pts = np.array(case["points"], dtype=dtype)
    R = np.array(case["rotation"], dtype=dtype)[None]
    t = np.array(case["translation"], dtype=dtype)[None]
    bg = None if case["background"] is None else np.array([case["background"]], dtype=dtype)
    ow = None if case["out_weight"] is None else np.array([case["out_weight"]], dtype=dtype)
    pw = None if case["point_weight"] is None else np.array(case["point_weight"], dtype=dtype)
    out = oracle.raster(tuple(case["grid_size"]), pts, R, t, bg, ow, pw, dtype=dtype)
    return out[..., 0]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_forward_known_answers(oracle, golden, dtype):
    for case in golden["forward"]:
        out = _fwd_case(oracle, case, dtype)
        expected = np.array(case["expected"], dtype=dtype)
        tol = 1e-12 if dtype == np.float64 else 1e-5
        np.testing.assert_allclose(out, expected, rtol=0, atol=tol, err_msg=case["name"])


def test_neighbour_order(oracle, golden):
    for n, expected in golden["neighbour_order"]["voxel_shifts"].items():
        np.testing.assert_array_equal(oracle.voxel_shifts(int(n)), np.array(expected))
    for item in golden["neighbour_order"]["digitstuple"]:
        shifts = oracle.voxel_shifts(item["N"])
        np.testing.assert_array_equal(shifts[item["k"]], np.array(item["expected"]))


def test_readme_gradient_example(oracle, golden):
    g = golden["readme_gradient"]
    pts = np.array(g["points"])
    R = np.array(g["rotation"])[None]
    t = np.array(g["translation"])[None]
    target = np.array(g["target_image"])
    out = oracle.raster((5, 5), pts, R, t)[..., 0]
    ds_dout = 2.0 * (target - out)
    # README.md:151-157 prints ds_dout to 6 significant digits
    np.testing.assert_allclose(ds_dout, np.array(g["ds_dout"]), rtol=2e-5, atol=2e-6)
    pb = oracle.raster_pullback(ds_dout[..., None], pts, R, t)
    full = -np.array(g["ds_dpoints_zygote_full_precision_negated"])
    # limited by the 6-digit printout of target_image (README.md:84-90)
    np.testing.assert_allclose(pb.points, full, rtol=0, atol=2e-5)
    np.testing.assert_allclose(pb.points, np.array(g["ds_dpoints"]), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pb.rotation[0], np.array(g["ds_drotation"]), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pb.translation[0], np.array(g["ds_dtranslation"]), rtol=0, atol=2e-5)
    np.testing.assert_allclose(pb.background[0], ds_dout.sum(), rtol=1e-12)


@pytest.mark.parametrize("n_out", [3, 2])
def test_batched_equals_loop_of_singles(oracle, n_out):
    d = D.make(n_points=2000, n_out=n_out, batch=5, seed=3)
    out_b = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds,
                          d.weights, d.point_weights)
    for b in range(d.batch):
        out_i = oracle.raster(d.grid, d.points, d.rotations[b:b + 1], d.translations[b:b + 1],
                              d.backgrounds[b:b + 1], d.weights[b:b + 1], d.point_weights)
        np.testing.assert_allclose(out_b[..., b], out_i[..., 0], rtol=1e-13, atol=1e-13)
    ds_dout = d.ds_dout
    pb = oracle.raster_pullback(ds_dout, d.points, d.rotations, d.translations, d.weights,
                                d.point_weights)
    sum_pts = np.zeros_like(pb.points)
    sum_pw = np.zeros_like(pb.point_weight)
    for b in range(d.batch):
        pi = oracle.raster_pullback(ds_dout[..., b:b + 1], d.points, d.rotations[b:b + 1],
                                    d.translations[b:b + 1], d.weights[b:b + 1], d.point_weights)
        np.testing.assert_allclose(pb.rotation[b], pi.rotation[0], rtol=1e-12)
        np.testing.assert_allclose(pb.translation[b], pi.translation[0], rtol=1e-12)
        np.testing.assert_allclose(pb.background[b], pi.background[0], rtol=1e-12)
        np.testing.assert_allclose(pb.out_weight[b], pi.out_weight[0], rtol=1e-12)
        sum_pts += pi.points
        sum_pw += pi.point_weight
    np.testing.assert_allclose(pb.points, sum_pts, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(pb.point_weight, sum_pw, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("n_out", [3, 2])
def test_threaded_port_matches_serial(oracle, n_out):
    d = D.make(n_points=5000, n_out=n_out, batch=D.uneven_batch(4), seed=4)
    a = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds, d.weights,
                      d.point_weights)
    b = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds, d.weights,
                      d.point_weights, threaded=True)
    assert D.isapprox(a, b)
    pa = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                d.point_weights)
    pb = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                d.point_weights, threaded=True, n_threads=4)
    for x, y in zip(pa, pb):
        assert D.isapprox(x, y)


def test_defaults_equal_explicit(oracle):
    """src/interface.jl:414-595: defaulted args == explicit zeros/ones."""
    d = D.make(n_points=300, n_out=3, batch=3, seed=5)
    a = oracle.raster(d.grid, d.points, d.rotations, d.translations)
    b = oracle.raster(d.grid, d.points, d.rotations, d.translations, np.zeros(3), np.ones(3),
                      np.ones(300))
    np.testing.assert_array_equal(a, b)
    pa = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations)
    pb = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, np.ones(3),
                                np.ones(300))
    for x, y in zip(pa, pb):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("n_out", [3, 2])
def test_pullback_matches_central_differences(oracle, n_out):
    """Counterpart of ChainRulesTestUtils.test_rrule (test/chainrules.jl:2-90):
    <ds_dout, d raster / d theta> by central differences vs the pullback.
    10 points, 8^N grid, Float64 (test/data.jl:18-23)."""
    d = D.make(n_points=10, n_out=n_out, batch=3, seed=6)
    g = d.ds_dout

    def f(points=d.points, rot=d.rotations, tr=d.translations, bg=d.backgrounds, ow=d.weights,
          pw=d.point_weights):
        return float(np.sum(g * oracle.raster(d.grid, points, rot, tr, bg, ow, pw)))

    pb = oracle.raster_pullback(g, d.points, d.rotations, d.translations, d.weights,
                                d.point_weights)
    h = 1e-6

    def fd(name, arr):
        grad = np.zeros_like(arr)
        it = np.nditer(arr, flags=["multi_index"])
        for _ in it:
            idx = it.multi_index
            ap = arr.copy(); ap[idx] += h
            am = arr.copy(); am[idx] -= h
            grad[idx] = (f(**{name: ap}) - f(**{name: am})) / (2 * h)
        return grad

    np.testing.assert_allclose(pb.points, fd("points", d.points), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pb.rotation, fd("rot", d.rotations), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pb.translation, fd("tr", d.translations), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pb.background, fd("bg", d.backgrounds), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pb.out_weight, fd("ow", d.weights), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pb.point_weight, fd("pw", d.point_weights), rtol=1e-5, atol=1e-6)


def test_edge_semantics(oracle):
    """Out-of-range neighbours are dropped individually (src/raster.jl:62);
    far-out / non-finite points contribute nothing."""
    pts = np.array([[0.999, 0.0], [-0.999, -0.999], [5.0, 0.0], [np.nan, 0.0], [1e30, 0.0]])
    R = np.eye(2)[None]
    t = np.zeros((1, 2))
    out = oracle.raster((4, 4), pts, R, t)[..., 0]
    # point 0: coord_1 = 1.999*2 = 3.998 -> ref=4 (1-based), upper neighbour 5 dropped
    assert out.sum() < 2.0 and out.sum() > 0.0
    assert np.isfinite(out).all()
    pb = oracle.raster_pullback(np.ones((4, 4, 1)), pts, R, t)
    assert np.isfinite(pb.points).all()
    np.testing.assert_array_equal(pb.points[2:], 0.0)
    np.testing.assert_array_equal(pb.point_weight[2:], 0.0)


def test_empty_inputs(oracle):
    R = np.eye(3)[None]
    t = np.zeros((1, 3))
    out = oracle.raster((4, 4, 4), np.zeros((0, 3)), R, t, np.array([2.0]))
    np.testing.assert_array_equal(out, 2.0)
    pb = oracle.raster_pullback(np.ones((4, 4, 4, 1)), np.zeros((0, 3)), R, t)
    assert pb.points.shape == (0, 3) and pb.background[0] == 64.0


@pytest.mark.parametrize("n_in", [1, 2, 3])
def test_one_dimensional_grids_follow_appendix_a(oracle, n_in):
    """N_out = 1 has no literal in the reference's tests; the oracle's dimension-generic code is
    checked here against a direct numpy restatement of SURVEY.md Appendix A for a line grid:
    coord = (R p + t + 1) n / 2, ref = ceil(coord - 1/2), delta = coord - (ref - 1/2), weights
    (1 - delta) / delta on the cells ref - 1 / ref (1-based), out-of-range neighbours dropped
    individually (src/raster.jl:62), a point with no in-range neighbour ignored."""
    rng = np.random.default_rng(4)
    n, P, B = 9, 300, 3
    pts = 0.7 * rng.normal(size=(P, n_in))
    R = rng.normal(size=(B, 1, n_in))
    t = 0.2 * rng.normal(size=(B, 1))
    bg, ow, pw = rng.normal(size=B), rng.uniform(0.5, 2, size=B), rng.uniform(size=P)
    expect = np.zeros((n, B))
    for b in range(B):
        expect[:, b] = bg[b]
        coord = (pts @ R[b, 0] + t[b, 0] + 1.0) * (n / 2)
        ref = np.ceil(coord - 0.5)          # 1-based index of the upper neighbour's lower cell
        delta = coord - (ref - 0.5)
        for p in range(P):
            lo = int(ref[p]) - 1            # 0-based lower neighbour
            if not (-1 <= lo <= n - 1):
                continue
            w = ow[b] * pw[p]
            if 0 <= lo < n:
                expect[lo, b] += w * (1 - delta[p])
            if 0 <= lo + 1 < n:
                expect[lo + 1, b] += w * delta[p]
    out = oracle.raster((n,), pts, R, t, bg, ow, pw)
    np.testing.assert_allclose(out, expect, rtol=1e-12, atol=1e-12)


def _appendix_a_numpy(grid, pts, R, t, bg, ow, pw):
    """SURVEY.md Appendix A restated for any (N_in, N_out) in plain numpy / python loops: the reference is
    generic in both (src/raster.jl:5-13; neighbour order src/util.jl:7-8,26-27)."""
    n_out = len(grid)
    B = R.shape[0]
    out = np.zeros(tuple(grid) + (B,))
    n = np.asarray(grid, dtype=np.float64)
    for b in range(B):
        out[..., b] = bg[b]
        coord = (pts @ R[b].T + t[b] + 1.0) * (n / 2)           # (P, n_out)
        ref = np.ceil(coord - 0.5)                               # 1-based upper-neighbour index
        delta = coord - (ref - 0.5)
        for p in range(pts.shape[0]):
            lo = ref[p].astype(np.int64) - 1                     # 0-based lower neighbour
            if np.any(lo < -1) or np.any(lo > np.asarray(grid) - 1):
                continue
            for s in range(1 << n_out):
                idx, w = [], ow[b] * pw[p]
                ok = True
                for d in range(n_out):
                    sd = (s >> d) & 1
                    i = lo[d] + sd
                    ok = ok and 0 <= i < grid[d]
                    idx.append(i)
                    w *= delta[p, d] if sd else (1.0 - delta[p, d])
                if ok:
                    out[tuple(idx) + (b,)] += w
    return out


@pytest.mark.parametrize("n_in,n_out", [(1, 2), (2, 3), (1, 3), (4, 1), (4, 2), (4, 3), (4, 4), (2, 4), (3, 4), (1, 4)])
def test_dimension_pairs_beyond_the_tested_three_follow_appendix_a(oracle, n_in, n_out):
    """(N_in, N_out) pairs the reference's tests do not use -- embeddings with N_out > N_in and 4-D points / grids
    included: the oracle's dimension-generic loops against the numpy restatement above, and its pullback against
    central differences of its forward."""
    rng = np.random.default_rng(40 + 10 * n_in + n_out)
    grid = tuple(int(x) for x in rng.integers(3, 7, size=n_out))
    P, B = 40, 2
    pts = 0.5 * rng.normal(size=(P, n_in))
    R = 0.7 * rng.normal(size=(B, n_out, n_in))
    t = 0.1 * rng.normal(size=(B, n_out))
    bg, ow, pw = rng.normal(size=B), rng.uniform(0.5, 2, size=B), rng.uniform(0.2, 1, size=P)
    out = oracle.raster(grid, pts, R, t, bg, ow, pw)
    np.testing.assert_allclose(out, _appendix_a_numpy(grid, pts, R, t, bg, ow, pw), rtol=1e-12, atol=1e-12)
    g = np.asfortranarray(rng.normal(size=grid + (B,)))
    pb = oracle.raster_pullback(g, pts, R, t, ow, pw)
    f = lambda **kw: float(np.sum(g * oracle.raster(grid, kw.get("pts", pts), kw.get("R", R), kw.get("t", t), bg,
                                                    kw.get("ow", ow), kw.get("pw", pw))))
    h = 1e-6
    for name, arr, got in (("pts", pts, pb.points), ("R", R, pb.rotation), ("t", t, pb.translation),
                           ("ow", ow, pb.out_weight), ("pw", pw, pb.point_weight)):
        it = np.nditer(arr, flags=["multi_index"])
        for k, _ in enumerate(it):
            if k % 3:  # (a third of the entries)
                continue
            idx = it.multi_index
            ap = arr.copy(); ap[idx] += h
            am = arr.copy(); am[idx] -= h
            fdv = (f(**{name: ap}) - f(**{name: am})) / (2 * h)
            assert abs(got[idx] - fdv) <= 1e-5 * max(1.0, abs(fdv)), (name, idx, got[idx], fdv)
