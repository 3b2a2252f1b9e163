"""Multi-rank (pose-sharded) path on CPU: world_size-2 `gloo` processes.

The product's local compute needs a GPU, so these tests inject the CPU oracle as the
per-rank compute (`local_raster=` / `local_pullback=`) and exercise exactly the code the
GPU ranks run around it: pose partitioning with an uneven remainder (B % world != 0, cf.
batch_size_for_test, test/data.jl:5-11), the fused [ds_dpoints | ds_dpoint_weight] buffer
and its single all-reduce(sum) -- the multi-process form of the reference's per-thread
slabs + sum (src/raster_pullback.jl:112-147)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import data as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_raster(grid_size, points, rotation, translation, background=None, out_weight=None,
                   point_weight=None):
    from oracle import oracle

    npf = lambda t: None if t is None else t.numpy()
    out = oracle.raster(tuple(grid_size), npf(points), npf(rotation), npf(translation),
                        npf(background), npf(out_weight), npf(point_weight))
    return torch.from_numpy(np.ascontiguousarray(out))


def _oracle_pullback(ds_dout, points, rotation, translation, background=None, out_weight=None,
                     point_weight=None, *, ds_dpoints=None, ds_dpoint_weight=None):
    from oracle import oracle

    import dpr_amd

    npf = lambda t: None if t is None else t.numpy()
    r = oracle.raster_pullback(ds_dout.numpy(), npf(points), npf(rotation), npf(translation),
                               npf(out_weight), npf(point_weight))
    ds_dpoints.copy_(torch.from_numpy(r.points))
    ds_dpoint_weight.copy_(torch.from_numpy(r.point_weight))
    return dpr_amd.PullbackResult(ds_dpoints, torch.from_numpy(r.rotation),
                                  torch.from_numpy(r.translation), torch.from_numpy(r.background),
                                  torch.from_numpy(r.out_weight), ds_dpoint_weight)


def _worker(rank, world, port, batch, n_out, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dpr_amd

        d = D.make(n_points=500, n_in=3, n_out=n_out, batch=batch, seed=31)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        pts, Rs, ts = t(d.points), t(d.rotations), t(d.translations)
        bgs, ows, pw = t(d.backgrounds), t(d.weights), t(d.point_weights)
        out_local, (lo, hi) = dpr_amd.raster_sharded(d.grid, pts, Rs, ts, bgs, ows, pw,
                                                     local_raster=_oracle_raster)
        assert (lo, hi) == dpr_amd.shard_range(batch, rank, world)
        g = torch.from_numpy(np.ascontiguousarray(d.ds_dout))
        res = dpr_amd.raster_pullback_sharded_(g[..., lo:hi], pts, Rs[lo:hi], ts[lo:hi],
                                               bgs[lo:hi], ows[lo:hi], pw,
                                               local_pullback=_oracle_pullback)
        torch.save(dict(lo=lo, hi=hi, out=out_local, points=res.points, pw=res.point_weight,
                        rot=res.rotation, trans=res.translation, bg=res.background,
                        ow=res.out_weight), os.path.join(tmpdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _worker_points(rank, world, port, n_out, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dpr_amd

        d = D.make(n_points=501, n_in=3, n_out=n_out, batch=2, seed=37)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        lo, hi = dpr_amd.shard_range(d.n_points, rank, world)
        out = dpr_amd.raster_point_sharded(d.grid, t(d.points[lo:hi]), t(d.rotations),
                                           t(d.translations), t(d.backgrounds), t(d.weights),
                                           t(d.point_weights[lo:hi]),
                                           local_raster=lambda *a: dpr_amd.to_grid_layout(_oracle_raster(*a)))

        def local_pb(ds_dout, points, rotation, translation, background, out_weight, pw):
            P = points.shape[0]
            return _oracle_pullback(ds_dout, points, rotation, translation, background, out_weight,
                                    pw, ds_dpoints=torch.empty(P, 3, dtype=torch.float64),
                                    ds_dpoint_weight=torch.empty(P, dtype=torch.float64))

        res = dpr_amd.raster_pullback_point_sharded_(t(d.ds_dout), t(d.points[lo:hi]),
                                                     t(d.rotations), t(d.translations),
                                                     t(d.backgrounds), t(d.weights),
                                                     t(d.point_weights[lo:hi]),
                                                     local_pullback=local_pb)
        torch.save(dict(lo=lo, hi=hi, out=out.contiguous(), points=res.points, pw=res.point_weight,
                        rot=res.rotation, trans=res.translation, bg=res.background,
                        ow=res.out_weight), os.path.join(tmpdir, f"prank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_out", [3, 2])
def test_point_sharded_two_ranks_match_single_process(oracle, tmp_path, n_out):
    """Point sharding (single-pose configs): grid all-reduce forward, local point gradients +
    all-reduced pose sums backward; uneven point split (501 points on 2 ranks)."""
    world = 2
    port = _free_port()
    mp.spawn(_worker_points, args=(world, port, n_out, str(tmp_path)), nprocs=world, join=True)
    d = D.make(n_points=501, n_in=3, n_out=n_out, batch=2, seed=37)
    ref_out = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds,
                            d.weights, d.point_weights)
    ref = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                 d.point_weights)
    for rank in range(world):
        r = torch.load(os.path.join(str(tmp_path), f"prank{rank}.pt"))
        lo, hi = r["lo"], r["hi"]
        np.testing.assert_allclose(r["out"].numpy(), ref_out, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(r["points"].numpy(), ref.points[lo:hi], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(r["pw"].numpy(), ref.point_weight[lo:hi], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(r["rot"].numpy(), ref.rotation, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["trans"].numpy(), ref.translation, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["ow"].numpy(), ref.out_weight, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["bg"].numpy(), ref.background, rtol=1e-11)


@pytest.mark.parametrize("batch,n_out", [(5, 3), (3, 2), (1, 3)])
def test_pose_sharded_two_ranks_match_single_process(oracle, tmp_path, batch, n_out):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, batch, n_out, str(tmp_path)), nprocs=world, join=True)
    d = D.make(n_points=500, n_in=3, n_out=n_out, batch=batch, seed=31)
    ref_out = oracle.raster(d.grid, d.points, d.rotations, d.translations, d.backgrounds,
                            d.weights, d.point_weights)
    ref = oracle.raster_pullback(d.ds_dout, d.points, d.rotations, d.translations, d.weights,
                                 d.point_weights)
    covered = []
    for rank in range(world):
        r = torch.load(os.path.join(str(tmp_path), f"rank{rank}.pt"))
        lo, hi = r["lo"], r["hi"]
        covered += list(range(lo, hi))
        np.testing.assert_allclose(r["out"].numpy(), ref_out[..., lo:hi], rtol=1e-12, atol=1e-12)
        # per-pose outputs: disjoint slices, no communication
        np.testing.assert_allclose(r["rot"].numpy(), ref.rotation[lo:hi], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(r["trans"].numpy(), ref.translation[lo:hi], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(r["bg"].numpy(), ref.background[lo:hi], rtol=1e-11)
        np.testing.assert_allclose(r["ow"].numpy(), ref.out_weight[lo:hi], rtol=1e-11, atol=1e-12)
        # point gradients: identical global sums on every rank after the all-reduce
        np.testing.assert_allclose(r["points"].numpy(), ref.points, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["pw"].numpy(), ref.point_weight, rtol=1e-10, atol=1e-12)
    assert covered == list(range(batch))
