"""Pose-sharded multi-GPU drivers (one process per GPU, torch.distributed over RCCL).

The reference has no multi-process path; its in-process analogue is the threaded
batched pullback, which splits the poses into chunks with private `ds_dpoints` /
`ds_dpoint_weight` slabs and sums the slabs at the end
(/root/reference/src/raster_pullback.jl:112-147, src/interface.jl:402-412).  Here a
"chunk" is a rank: every per-pose quantity (out[.., b], ds_drotation[.., b],
ds_dtranslation[:, b], ds_dbackground[b], ds_dout_weight[b]) is disjoint across ranks
and needs no communication; only the point gradients sum over poses, which is ONE
all-reduce(sum) of the fused [ds_dpoints | ds_dpoint_weight] buffer (RCCL over xGMI).
Points (and point weights) are replicated on every rank.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from .interface import PullbackResult, raster, raster_pullback_


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous pose block of `rank` (sizes differ by at most one, like
    ChunkSplitters.chunks used at src/raster_pullback.jl:117)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist

    return dist


def raster_sharded(grid_size, points, rotation, translation, background=None, out_weight=None,
                   point_weight=None, *, group=None, local_raster: Callable = raster, **kw):
    """Forward for this rank's pose block.  `rotation`, `translation`, `background`,
    `out_weight` are the GLOBAL batched arguments (B poses); returns
    (out_local[.., b_lo:b_hi], (b_lo, b_hi)).  No communication."""
    dist = _dist()
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(rotation.shape[0], rank, world)
    sl = slice(lo, hi)
    out = local_raster(grid_size, points, rotation[sl], translation[sl],
                       None if background is None else background[sl],
                       None if out_weight is None else out_weight[sl], point_weight, **kw)
    return out, (lo, hi)


def raster_pullback_sharded_(ds_dout_local, points, rotation_local, translation_local,
                             background_local=None, out_weight_local=None, point_weight=None, *,
                             group=None, local_pullback: Callable = raster_pullback_,
                             fused_buffer: Optional[torch.Tensor] = None, **kw) -> PullbackResult:
    """Pullback of this rank's pose block + all-reduce of the point gradients.

    Arguments with `_local` are this rank's slice (as produced by `shard_range`).  The
    returned per-pose fields cover the local poses only; `points` / `point_weight` are the
    global sums (identical on every rank)."""
    dist = _dist()
    P, n_in = points.shape
    dtype = ds_dout_local.dtype
    if fused_buffer is None:
        fused_buffer = torch.empty(P * (n_in + 1), dtype=dtype, device=points.device)
    d_pts = fused_buffer[: P * n_in].view(P, n_in)
    d_pw = fused_buffer[P * n_in:]
    if rotation_local.shape[0] > 0:
        res = local_pullback(ds_dout_local, points, rotation_local, translation_local,
                             background_local, out_weight_local, point_weight,
                             ds_dpoints=d_pts, ds_dpoint_weight=d_pw, **kw)
    else:  # a rank may own no pose when B < world size
        fused_buffer.zero_()
        n_out = rotation_local.shape[1]
        z = lambda *s: torch.zeros(s, dtype=dtype, device=points.device)
        res = PullbackResult(d_pts, z(0, n_out, n_in), z(0, n_out), z(0), z(0), d_pw)
    if dist.is_initialized():  # also at world size 1: the same stream hand-over as at N ranks
        dist.all_reduce(fused_buffer, op=dist.ReduceOp.SUM, group=group)
    return PullbackResult(d_pts, res.rotation, res.translation, res.background, res.out_weight,
                          d_pw)


# ---------------------------------------------------------------------------------------
# Point sharding: the alternative for single-pose (or few-pose) problems, where pose sharding
# has nothing to split.  Rank r owns a contiguous block of the POINTS; every
# rank holds all poses.  Forward: each rank splats its points into a full grid (background
# only on rank 0) and ONE all-reduce(sum) of the grid assembles `out` (67 MB at C3).
# Pullback: ds_dout is replicated, point gradients stay local to the owning rank (no
# communication), the per-pose sums (rotation, translation, out_weight: N_out*(N_in+1)+1
# scalars per pose) are all-reduced; the background gradient is the same on every rank.
def raster_point_sharded(grid_size, points_local, rotation, translation, background=None,
                         out_weight=None, point_weight_local=None, *, group=None,
                         local_raster: Callable = raster, **kw):
    """`points_local` / `point_weight_local` are this rank's block of the cloud
    (`shard_range(P, rank, world)`); the pose arguments are global.  Returns the full `out`
    (identical on every rank)."""
    dist = _dist()
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bg = background if rank == 0 else None  # the background must be counted once
    out = local_raster(grid_size, points_local, rotation, translation, bg, out_weight,
                       point_weight_local, **kw)
    if dist.is_initialized():
        flat = out.permute(*reversed(range(out.ndim)))  # the contiguous buffer behind the view
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return out


def raster_pullback_point_sharded_(ds_dout, points_local, rotation, translation, background=None,
                                   out_weight=None, point_weight_local=None, *, group=None,
                                   local_pullback: Callable = raster_pullback_,
                                   **kw) -> PullbackResult:
    """Pullback for this rank's block of the points.  `points` / `point_weight` of the result
    are LOCAL (gradients of the local block); rotation / translation / out_weight are summed
    over ranks; background is already global."""
    dist = _dist()
    res = local_pullback(ds_dout, points_local, rotation, translation, background, out_weight,
                         point_weight_local, **kw)
    if dist.is_initialized():
        rot, tr, ow = res.rotation, res.translation, res.out_weight
        fused = torch.cat([rot.reshape(-1), tr.reshape(-1), ow.reshape(-1)])
        dist.all_reduce(fused, op=dist.ReduceOp.SUM, group=group)
        n1, n2 = rot.numel(), tr.numel()
        rot = fused[:n1].reshape(rot.shape)
        tr = fused[n1:n1 + n2].reshape(tr.shape)
        ow = fused[n1 + n2:].reshape(ow.shape)
        res = PullbackResult(res.points, rot, tr, res.background, ow, res.point_weight)
    return res
