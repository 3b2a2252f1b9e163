"""Reverse-mode rule for `raster`: the host-side analogue of the reference's ChainRules
extension (/root/reference/ext/DiffPointRasterisationChainRulesCoreExt.jl:6-27 single image,
:47-74 batch of images) for torch autograd.

    out = raster_ad(grid_size, points, rotation, translation[, background, out_weight,
                    point_weight])
    loss(out).backward()      # -> points.grad, rotation.grad, translation.grad, ...

Like the rrule, the primal is `raster` and the pullback closure is `raster_pullback!` on the
same arguments; the tangents come back in the rrule's order (points, rotation, translation and
then the optional arguments that were passed).  Where the reference recomputes everything in
the closure (src/raster_pullback.jl:20-22), a call whose raster + pullback pair shares on the
device -- one pose on the tiled path, a batch on the chunk-owner path, a batch on the tiled path
where DPR_ALGO_AUTO keeps every pose's binning -- keeps the forward's binning (or sorted cloud)
in a private workspace and the first backward pass reuses it (DPR_FLAG_KEEP_BINNING /
DPR_FLAG_REUSE_BINNING); it is consumed by that pass, any further backward pass through the same
node re-bins.
"""
from __future__ import annotations

import torch

from .interface import (empty_grid, raster_, raster_pullback_, resolve_algo, sharing_effective,
                        workspace_bytes)


class _RasterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid_size, algo, points, rotation, translation, background, out_weight,
                point_weight):
        grid_size = tuple(int(g) for g in grid_size)
        single = rotation.ndim == 2
        batch = None if single else rotation.shape[0]
        dtype = torch.promote_types(points.dtype, torch.promote_types(rotation.dtype,
                                                                      translation.dtype))
        for opt in (background, out_weight, point_weight):
            if isinstance(opt, torch.Tensor):
                dtype = torch.promote_types(dtype, opt.dtype)
        out = empty_grid(grid_size, batch, dtype, points.device)
        # The device-side check of a REUSE_BINNING pullback compares the ADDRESSES of the point /
        # point_weight buffers with the forward's: canonicalise them ONCE here (promoted dtype,
        # contiguous) and hand the same tensors to both calls -- raster_ and raster_pullback_
        # would otherwise each make their own temporary for mixed dtypes or strided views, the
        # header would not match and every gradient would come back NaN.
        points_in, pw_in = points, point_weight
        points = points.detach().to(dtype).contiguous()
        if isinstance(point_weight, torch.Tensor):
            point_weight = point_weight.detach().to(dtype).contiguous()
        ws = None
        P, n_in = points.shape
        B = 1 if single else int(batch)
        if P > 0 and B > 0:
            # one algorithm for the raster + pullback pair (dpr_resolve_algo_ex with a sharing
            # flag); share where the library does: one pose, a batch on the chunk-owner path (the
            # sorted copy of the cloud), a batch on the tiled path where AUTO keeps every pose's
            # binning (large grids, bounded record volume: dpr_resolve_flags_ex)
            f = algo if algo != "auto" else resolve_algo("raster", grid_size, P, B, n_in, sharing=True)
            b = algo if algo != "auto" else resolve_algo("pullback", grid_size, P, B, n_in, sharing=True)
            shares = f == b and (
                (B == 1 and f in ("tiled", "chunked"))
                or (f == "chunked" and len(grid_size) == 2)
                or (f == "tiled" and algo == "auto" and sharing_effective(grid_size, P, B, n_in)))
            if shares:
                need = max(workspace_bytes("raster", grid_size, P, B, n_in, dtype, f, sharing=True),
                           workspace_bytes("pullback", grid_size, P, B, n_in, dtype, f, sharing=True))
                ws = torch.empty(max(need, 16), dtype=torch.uint8, device=points.device)
                algo = f
        raster_(out, points, rotation, translation, background, out_weight, point_weight,
                algo=algo, workspace=ws, keep_binning=ws is not None)
        ctx.save_for_backward(points, rotation, translation,
                              *[t for t in (background, out_weight, point_weight)
                                if isinstance(t, torch.Tensor)])
        ctx.opt = tuple(t if not isinstance(t, torch.Tensor) else None
                        for t in (background, out_weight, point_weight))
        ctx.opt_is_tensor = tuple(isinstance(t, torch.Tensor)
                                  for t in (background, out_weight, point_weight))
        ctx.algo, ctx.ws = algo, ws
        ctx.points_dtype = points_in.dtype
        ctx.pw_like = (pw_in.shape, pw_in.dtype) if isinstance(pw_in, torch.Tensor) else None
        return out

    @staticmethod
    def backward(ctx, ds_dout):
        saved = list(ctx.saved_tensors)
        points, rotation, translation = saved[:3]
        rest = saved[3:]
        opt = []
        for k in range(3):
            opt.append(rest.pop(0) if ctx.opt_is_tensor[k] else ctx.opt[k])
        ws, ctx.ws = ctx.ws, None  # the binning is consumed by the pass that reuses it
        need = ctx.needs_input_grad  # (grid_size, algo, points, rotation, translation, bg, ow, pw)
        # the rrule drops the point_weight tangent when that argument was defaulted
        # (ext/DiffPointRasterisationChainRulesCoreExt.jl:23,70): do not compute / store it then
        pb = raster_pullback_(ds_dout.detach(), points, rotation, translation, *opt,
                              algo=ctx.algo, workspace=ws, reuse_binning=ws is not None,
                              point_weight_grad=bool(ctx.opt_is_tensor[2] and need[7]))
        grads = [None, None,
                 pb.points.to(ctx.points_dtype) if need[2] else None,
                 pb.rotation.to(rotation.dtype) if need[3] else None,
                 pb.translation.to(translation.dtype) if need[4] else None]
        for k, g in enumerate((pb.background, pb.out_weight, pb.point_weight)):
            given = ctx.opt_is_tensor[k] and need[5 + k]
            if given and k == 2:  # (the saved point_weight is the canonical copy)
                grads.append(g.reshape(ctx.pw_like[0]).to(ctx.pw_like[1]))
            elif given:
                ref = saved[3 + sum(ctx.opt_is_tensor[:k])]
                grads.append(g.reshape(ref.shape).to(ref.dtype))
            else:
                grads.append(None)
        return tuple(grads)


def raster_ad(grid_size, points, rotation, translation, background=None, out_weight=None,
              point_weight=None, *, algo: str = "auto") -> torch.Tensor:
    """Differentiable `raster` (torch autograd).  Tensor arguments may require grad; Python
    scalars / None for the optional arguments are constants, as FillArrays defaults are for
    the rrule.  Same shapes and errors as `raster`."""
    return _RasterFn.apply(tuple(grid_size), algo, points, rotation, translation, background,
                           out_weight, point_weight)
