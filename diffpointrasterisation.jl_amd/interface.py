"""Host-side mirror of the reference's public operator API for the hot path.

  raster(grid_size, points, rotation, translation[, background, out_weight, point_weight])
  raster_(out, ...)              == raster!            (/root/reference/src/interface.jl:48-55)
  raster_pullback_(ds_dout, ...) == raster_pullback!   (/root/reference/src/interface.jl:164-194)

Same argument meaning, defaults and error behaviour as the reference's dispatch
funnel (src/interface.jl:62-129, 196-308), with torch ROCm tensors as the device
arrays (the CuArray role in ext/DiffPointRasterisationCUDAExt.jl) and the C ABI of
include/dpr.h as the canonical methods (src/raster.jl:5-34,
ext/DiffPointRasterisationCUDAExt.jl:231-321).

Python-facing shapes are the mathematical ones; memory is the reference's:

  points        (P, N_in) contiguous                      Vector{SVector{N_in}}
  single pose:  rotation (N_out, N_in), translation (N_out,), background / out_weight scalars
  batched:      rotation (B, N_out, N_in), translation (B, N_out), background / out_weight (B,)
  out / ds_dout indexable as [i_1, .., i_N(, b)] like the reference's column-major arrays:
                a permuted view of a contiguous (B, n_N, .., n_1) buffer (`empty_grid`)

There is no CPU path: tensors must live on a HIP device and libdpr.so must be built.
"""
from __future__ import annotations

import ctypes
from collections import namedtuple
from typing import Optional, Sequence

import torch

from . import _lib

PullbackResult = namedtuple(
    "PullbackResult",
    ["points", "rotation", "translation", "background", "out_weight", "point_weight"],
)  # field order = src/raster_pullback.jl:74-81,140-147 (ChainRules slices it positionally)


class DimensionMismatch(ValueError):
    """Counterpart of Julia's DimensionMismatch thrown by the reference's @argcheck's."""


_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}


# --------------------------------------------------------------------------- helpers
def empty_grid(grid_size: Sequence[int], batch: Optional[int], dtype, device) -> torch.Tensor:
    """Allocate an `out`/`ds_dout`-shaped array with the reference's memory order
    (`similar(points, T, (grid_size..., B))`, src/interface.jl:67-74): returns a view
    of shape grid_size (+ (B,)) whose axis 1 is the fastest in memory."""
    shape = tuple(int(n) for n in grid_size) + (() if batch is None else (int(batch),))
    buf = torch.empty(tuple(reversed(shape)), dtype=dtype, device=device)
    return buf.permute(*reversed(range(len(shape))))


def to_grid_layout(t: torch.Tensor) -> torch.Tensor:
    """Copy an arbitrary-strided [i_1..i_N(,b)] tensor into the reference memory order."""
    out = empty_grid(t.shape, None, t.dtype, t.device)
    out.copy_(t)
    return out


def _is_grid_layout(t: torch.Tensor) -> bool:
    return t.permute(*reversed(range(t.ndim))).is_contiguous()


def _promote(*tensors) -> torch.dtype:
    """promote_type over the array arguments (src/interface.jl:63-64).  Python scalars and
    lists are weakly typed and do not take part (they adopt the promoted dtype)."""
    dt = None
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            continue
        d = t.dtype
        if not d.is_floating_point:  # Bool / Int rotations such as I(2) (README.md:36)
            continue
        dt = d if dt is None else torch.promote_types(dt, d)
    if dt is None:
        dt = torch.get_default_dtype()
    if dt not in _SUFFIX:
        raise TypeError(f"libdpr supports float32/float64, got {dt}")
    return dt


def _device_of(points: torch.Tensor) -> torch.device:
    if not isinstance(points, torch.Tensor):
        raise TypeError("points must be a torch.Tensor on a HIP device")
    if points.device.type != "cuda":
        raise RuntimeError(
            "DiffPointRasterisation MI355X backend: `points` lives on "
            f"{points.device}; there is no CPU path in this package (device tensors required)."
        )
    return points.device


def _as(t, dtype, device, shape=None, name="argument") -> torch.Tensor:
    t = torch.as_tensor(t, dtype=dtype, device=device) if not isinstance(t, torch.Tensor) else t.to(
        device=device, dtype=dtype)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise DimensionMismatch(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


def _scalar(x) -> torch.Tensor:
    """Single-pose scalar argument -> 1-element vector (src/interface.jl:113-116) without
    rounding a Python float through float32."""
    if isinstance(x, torch.Tensor):
        return x.reshape(1)
    return torch.as_tensor(x, dtype=torch.float64).reshape(1)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _workspace(op, algo, suf, n_in, n_out, grid_arr, P, B, device, workspace, flags=0):
    need = getattr(_lib.lib(), f"dpr_workspace_bytes_ex_{suf}")(
        op, algo, flags, n_in, n_out, grid_arr.ctypes.data_as(ctypes.c_void_p), P, B)
    if need == ctypes.c_size_t(-1).value:
        raise _lib.DprError(_lib.ERR_INVALID_ARG, _lib.last_error())
    if need == 0:
        return None, 0
    if workspace is not None:
        if workspace.device != device or workspace.numel() * workspace.element_size() < need:
            raise ValueError(f"workspace too small: need {need} bytes")
        return workspace, workspace.numel() * workspace.element_size()
    ws = torch.empty(need, dtype=torch.uint8, device=device)
    return ws, need


def workspace_bytes(op: str, grid_size, n_points: int, batch: int, n_in: int, dtype=torch.float32,
                    algo: str = "auto", max_pose_group: int = 0,
                    coherent_points: bool = False, sharing: bool = False) -> int:
    """dpr_workspace_bytes_ex_*: device bytes `op` needs.  `max_pose_group` (1..16, 0 = default)
    bounds how many poses of a batch the tiled path bins together -- the speed / memory trade of
    DPR_FLAG_MAX_POSE_GROUP (include/dpr.h).  `sharing`: the calls will carry keep_binning /
    reuse_binning (`algo="auto"` then sizes for the algorithm the pair runs)."""
    import numpy as np

    grid_arr = np.asarray(grid_size, dtype=np.int64)
    opc = {"raster": _lib.OP_RASTER, "pullback": _lib.OP_PULLBACK,
           "residual_pullback": _lib.OP_RESIDUAL_PULLBACK}[op]
    need = getattr(_lib.lib(), f"dpr_workspace_bytes_ex_{_SUFFIX[dtype]}")(
        opc, _lib.ALGOS[algo], _lib.flag_max_pose_group(max_pose_group)
        | (_lib.FLAG_COHERENT_POINTS if coherent_points else 0)
        | (_lib.FLAG_KEEP_BINNING if sharing else 0), n_in, len(grid_size),
        grid_arr.ctypes.data_as(ctypes.c_void_p), n_points, batch)
    if need == ctypes.c_size_t(-1).value:
        raise _lib.DprError(_lib.ERR_INVALID_ARG, _lib.last_error())
    return int(need)


def resolve_algo(op: str, grid_size, n_points: int, batch: int, n_in: int, *,
                 sharing: bool = False, coherent_points: bool = False) -> str:
    """Name of the algorithm `algo="auto"` picks for this problem.  `sharing`: the call carries
    keep_binning / reuse_binning (the choice is then made for the raster + pullback pair, see
    include/dpr.h)."""
    import numpy as np

    grid_arr = np.asarray(grid_size, dtype=np.int64)
    opc = {"raster": _lib.OP_RASTER, "pullback": _lib.OP_PULLBACK,
           "residual_pullback": _lib.OP_RESIDUAL_PULLBACK}[op]
    flags = (_lib.FLAG_KEEP_BINNING if sharing else 0) | (_lib.FLAG_COHERENT_POINTS if coherent_points else 0)
    rc = _lib.lib().dpr_resolve_algo_ex(opc, flags, n_in, len(grid_size),
                                        grid_arr.ctypes.data_as(ctypes.c_void_p), n_points, batch)
    if rc < 0:
        _lib.check(rc)
    return {v: k for k, v in _lib.ALGOS.items()}[rc]


def sharing_effective(grid_size, n_points: int, batch: int, n_in: int, *,
                      coherent_points: bool = False) -> bool:
    """Will `algo="auto"` honour keep_binning / reuse_binning for this problem
    (dpr_resolve_flags_ex)?  False where the pair's algorithm has nothing to share."""
    import numpy as np

    grid_arr = np.asarray(grid_size, dtype=np.int64)
    flags = _lib.FLAG_KEEP_BINNING | (_lib.FLAG_COHERENT_POINTS if coherent_points else 0)
    rc = _lib.lib().dpr_resolve_flags_ex(_lib.OP_RASTER, flags, n_in, len(grid_size),
                                         grid_arr.ctypes.data_as(ctypes.c_void_p), n_points, batch)
    if rc < 0:
        _lib.check(rc)
    return bool(rc & _lib.FLAG_KEEP_BINNING)


class ColumnMajorRotation:
    """A rotation argument already in the memory order of the C ABI (`Vector{SMatrix}`: every pose
    column-major), made once with `column_major_rotation` and accepted wherever a rotation tensor
    is.  For a torch tensor that order is a transpose + copy, two small kernels per call (2 x 4.4 us
    inside a 0.37 ms step); a caller whose pose does not change between calls -- or who keeps its
    pose in this order anyway, as a Julia host does -- skips them.  (Caching the copy per tensor and
    `_version` is not safe: writes through `.data`, as torch.autograd.gradcheck does, do not bump
    the version.)"""

    def __init__(self, cm: torch.Tensor, single: bool):
        self.cm, self.single = cm, single  # cm: contiguous (B, N_in, N_out)

    @property
    def ndim(self):
        return 2 if self.single else 3

    @property
    def shape(self):
        B, n_in, n_out = self.cm.shape
        return (n_out, n_in) if self.single else (B, n_out, n_in)

    @property
    def dtype(self):
        return self.cm.dtype


def column_major_rotation(rotation: torch.Tensor, dtype=None) -> ColumnMajorRotation:
    """(N_out, N_in) or (B, N_out, N_in) rotation tensor -> `ColumnMajorRotation` (a snapshot: later
    changes of `rotation` are not seen)."""
    r = rotation if dtype is None else rotation.to(dtype)
    single = r.ndim == 2
    if r.ndim not in (2, 3):
        raise DimensionMismatch("rotation must be (N_out, N_in) or (B, N_out, N_in)")
    r = r[None] if single else r
    return ColumnMajorRotation(r.transpose(1, 2).contiguous(), single)


def _check_dims(n_in_pts, rot_shape, trans_shape):
    """Step 5 of the reference funnel: explicit dimension errors
    (src/interface.jl:137-162, 315-366)."""
    n_out_rot, n_in_rot = rot_shape[-2], rot_shape[-1]
    n_out_trans = trans_shape[-1]
    if n_out_trans != n_out_rot:
        raise DimensionMismatch(
            f"Row dimension of rotation (got {n_out_rot}) and translation (got {n_out_trans}) must agree!")
    if n_in_rot != n_in_pts:
        raise DimensionMismatch(
            f"Column dimension of rotation (got {n_in_rot}) and points (got {n_in_pts}) must agree!")


def _canonicalise(points, rotation, translation, background, out_weight, point_weight, extra=()):
    """Steps 2-4 of the funnel: defaults (None == FillArrays Zeros/Ones -> NULL pointer),
    single pose -> batch of one, contiguous device buffers in the reference layout."""
    device = _device_of(points)
    if points.ndim != 2:
        raise DimensionMismatch(f"points must be (P, N_in), got {tuple(points.shape)}")
    pre = rotation if isinstance(rotation, ColumnMajorRotation) else None
    if pre is not None:
        rotation_t = pre.cm.transpose(1, 2)  # (a view in the mathematical shape, for the checks below)
        rotation_t = rotation_t[0] if pre.single else rotation_t
    else:
        rotation_t = rotation if isinstance(rotation, torch.Tensor) else torch.as_tensor(rotation)
    translation_t = translation if isinstance(translation, torch.Tensor) else torch.as_tensor(translation)
    single = rotation_t.ndim == 2  # src/interface.jl:67 `rotation isa AbstractMatrix`
    if rotation_t.ndim not in (2, 3):
        raise DimensionMismatch("rotation must be (N_out, N_in) or (B, N_out, N_in)")
    dtype = _promote(points, rotation_t, translation_t, background, out_weight, point_weight, *extra)
    if single:
        rotation_t = rotation_t[None]
        if translation_t.ndim != 1:
            raise DimensionMismatch("single-pose translation must be a vector")
        translation_t = translation_t[None]
        background = None if background is None else _scalar(background)
        out_weight = None if out_weight is None else _scalar(out_weight)
    if translation_t.ndim != 2:
        raise DimensionMismatch("batched translation must be (B, N_out)")
    P, n_in = points.shape
    _check_dims(n_in, rotation_t.shape, translation_t.shape)
    B, n_out = rotation_t.shape[0], rotation_t.shape[1]
    if translation_t.shape[0] != B:
        raise DimensionMismatch(
            f"batch sizes differ: rotation {B}, translation {translation_t.shape[0]}")
    pts = _as(points, dtype, device)
    # Vector{SMatrix}: each pose column-major == row-major of the transpose
    if pre is not None and pre.cm.dtype == dtype and pre.cm.device == device:
        rot_cm = pre.cm
    else:
        rot_cm = _as(rotation_t, dtype, device).transpose(1, 2).contiguous()
    trans = _as(translation_t, dtype, device)
    bg = None if background is None else _as(background, dtype, device, (B,), "background")
    ow = None if out_weight is None else _as(out_weight, dtype, device, (B,), "out_weight")
    if point_weight is not None and tuple(torch.as_tensor(point_weight).shape) != (P,):
        raise DimensionMismatch(  # @argcheck length(point_weight) == n_points, src/raster.jl:23
            f"length(point_weight) = {tuple(torch.as_tensor(point_weight).shape)} != n_points = {P}")
    pw = None if point_weight is None else _as(point_weight, dtype, device, (P,), "point_weight")
    return dict(device=device, dtype=dtype, single=single, P=P, B=B, n_in=n_in, n_out=n_out,
                points=pts, rot=rot_cm, trans=trans, bg=bg, ow=ow, pw=pw)


# --------------------------------------------------------------------------- forward
def raster(grid_size, points, rotation, translation, background=None, out_weight=None,
           point_weight=None, *, algo: str = "auto", workspace=None,
           max_pose_group: int = 0, coherent_points: bool = False) -> torch.Tensor:
    """Allocating forward (src/interface.jl:62-77).  Returns `out[i_1..i_N]` for a single
    pose (rotation is a matrix) or `out[i_1..i_N, b]` for a batch."""
    device = _device_of(points)
    rot_like = isinstance(rotation, (torch.Tensor, ColumnMajorRotation))
    rot_nd = rotation.ndim if rot_like else torch.as_tensor(rotation).ndim
    dtype = _promote(points, rotation.cm if isinstance(rotation, ColumnMajorRotation) else rotation,
                     translation, background, out_weight, point_weight)
    batch = None if rot_nd == 2 else (rotation.shape[0] if rot_like else len(rotation))
    out = empty_grid(tuple(grid_size), batch, dtype, device)
    return raster_(out, points, rotation, translation, background, out_weight, point_weight,
                   algo=algo, workspace=workspace, max_pose_group=max_pose_group,
                   coherent_points=coherent_points)


def raster_(out, points, rotation, translation, background=None, out_weight=None,
            point_weight=None, *, algo: str = "auto", workspace=None,
            keep_binning: bool = False, max_pose_group: int = 0,
            coherent_points: bool = False) -> torch.Tensor:
    """In-place forward, the reference's `raster!`.  `out` is fully overwritten and
    returned (same object).  Enqueued on torch's current stream; not synchronised.
    `keep_binning=True` (explicit `workspace`, sized with `sharing=True`) leaves the binning of
    every pose (tiled algorithm) or the sorted copy of the cloud (chunk-owner algorithm) in
    `workspace` for `raster_pullback_(..., reuse_binning=True)` with the same arguments."""
    import numpy as np

    c = _canonicalise(points, rotation, translation, background, out_weight, point_weight)
    if not isinstance(out, torch.Tensor) or out.device != c["device"]:
        raise RuntimeError("out must be a tensor on the same HIP device as points")
    expect_ndim = c["n_out"] + (0 if c["single"] else 1)
    if out.ndim != expect_ndim:  # @argcheck N_out == N_out_p1 - 1, src/raster.jl:14
        raise DimensionMismatch(
            f"out has {out.ndim} dims, expected {expect_ndim} for N_out={c['n_out']}")
    if not c["single"] and out.shape[-1] != c["B"]:  # src/raster.jl:17-21
        raise DimensionMismatch(f"out batch dim {out.shape[-1]} != number of poses {c['B']}")
    if out.dtype != c["dtype"]:
        raise TypeError(f"out dtype {out.dtype} != promoted argument dtype {c['dtype']}")
    if not _is_grid_layout(out):
        raise ValueError("out must have the reference memory order (use empty_grid/to_grid_layout)")
    grid = tuple(out.shape[: c["n_out"]])
    grid_arr = np.asarray(grid, dtype=np.int64)
    suf = _SUFFIX[c["dtype"]]
    algo_c = _lib.ALGOS[algo]
    with torch.cuda.device(c["device"]):
        flags = _lib.flag_max_pose_group(max_pose_group)
        flags |= _lib.FLAG_COHERENT_POINTS if coherent_points else 0  # dpr_sort_points output etc.
        flags |= _lib.FLAG_KEEP_BINNING if keep_binning else 0  # (also steers DPR_ALGO_AUTO)
        ws, ws_bytes = _workspace(_lib.OP_RASTER, algo_c, suf, c["n_in"], c["n_out"], grid_arr,
                                  c["P"], c["B"], c["device"], workspace, flags)
        fn = getattr(_lib.lib(), f"dpr_raster_ex_{suf}")
        if keep_binning and workspace is None:
            raise ValueError("keep_binning needs a caller-owned workspace")
        _lib.check(fn(_stream_ptr(c["device"]), algo_c, flags, c["n_in"], c["n_out"],
                      grid_arr.ctypes.data_as(ctypes.c_void_p), c["P"], c["B"], _ptr(out),
                      _ptr(c["points"]), _ptr(c["rot"]), _ptr(c["trans"]), _ptr(c["bg"]),
                      _ptr(c["ow"]), _ptr(c["pw"]), _ptr(ws), ws_bytes))
    return out


# --------------------------------------------------------------------------- pullback
def raster_pullback_(ds_dout, points, rotation, translation, background=None, out_weight=None,
                     point_weight=None, *, ds_dpoints=None, ds_drotation=None,
                     ds_dtranslation=None, ds_dbackground=None, ds_dout_weight=None,
                     ds_dpoint_weight=None, algo: str = "auto", workspace=None,
                     reuse_binning: bool = False, max_pose_group: int = 0,
                     coherent_points: bool = False,
                     point_weight_grad: bool = True) -> PullbackResult:
    """The reference's `raster_pullback!` (src/interface.jl:196-308).  Optional keyword
    arguments are pre-allocated outputs (the reference's `points=`, `rotation=`, ... kwargs,
    src/interface.jl:278-291); they are OVERWRITTEN and returned by identity.  Unlike the
    reference's GPU path (docs/src/batch.md:4) a single pose works too: scalars/unbatched
    arrays are returned for it as on the reference's CPU path (src/raster_pullback.jl:74-81).

    Returned layouts: points (P, N_in); rotation (B, N_out, N_in) -- a transposed view of
    the column-major (N_out, N_in, B) buffer; translation (B, N_out); background,
    out_weight (B,); point_weight (P,).

    `point_weight_grad=False` (DPR_FLAG_NO_POINT_WEIGHT_GRAD): ds_dpoint_weight is neither
    allocated nor written and comes back as None -- what the reference's rrule throws away
    when `point_weight` was defaulted (ext/DiffPointRasterisationChainRulesCoreExt.jl:23,70)."""
    return _pullback(ds_dout, None, points, rotation, translation, background, out_weight,
                     point_weight, ds_dpoints, ds_drotation, ds_dtranslation, ds_dbackground,
                     ds_dout_weight, ds_dpoint_weight, algo, workspace, reuse_binning,
                     max_pose_group, coherent_points, point_weight_grad)


def raster_residual_pullback_(out, target, points, rotation, translation, background=None,
                              out_weight=None, point_weight=None, *, scale: float = 2.0,
                              loss=None, ds_dpoints=None, ds_drotation=None,
                              ds_dtranslation=None, ds_dbackground=None, ds_dout_weight=None,
                              ds_dpoint_weight=None, algo: str = "auto", workspace=None,
                              reuse_binning: bool = False, coherent_points: bool = False):
    """Pullback of a squared-error loss without materialising its sensitivity
    (dpr_raster_residual_pullback_*).  Equivalent to

        ds_dout = scale * (out - target)          # README.md:151 (there: scale = -2)
        raster_pullback_(ds_dout, points, ...)     # src/interface.jl:196-308
        loss    = ((out - target) ** 2).sum over each pose's grid

    with `out` the result of `raster` for the same arguments; ds_dout is formed inside the
    kernels, so the grid is read once (out, target) instead of written and re-read.  Returns
    (PullbackResult, loss) with loss of shape (B,) (0-d for a single pose)."""
    return _pullback(out, (target, float(scale), loss), points, rotation, translation, background,
                     out_weight, point_weight, ds_dpoints, ds_drotation, ds_dtranslation,
                     ds_dbackground, ds_dout_weight, ds_dpoint_weight, algo, workspace,
                     reuse_binning, coherent_points=coherent_points)


def _pullback(ds_dout, residual, points, rotation, translation, background, out_weight,
              point_weight, ds_dpoints, ds_drotation, ds_dtranslation, ds_dbackground,
              ds_dout_weight, ds_dpoint_weight, algo, workspace, reuse_binning,
              max_pose_group=0, coherent_points=False, point_weight_grad=True):
    import numpy as np

    c = _canonicalise(points, rotation, translation, background, out_weight, point_weight,
                      extra=(ds_dout,))
    dev, dtype, P, B, n_in, n_out = c["device"], c["dtype"], c["P"], c["B"], c["n_in"], c["n_out"]
    if not isinstance(ds_dout, torch.Tensor) or ds_dout.device != dev:
        raise RuntimeError("ds_dout must be a tensor on the same HIP device as points")
    expect_ndim = n_out + (0 if c["single"] else 1)
    if ds_dout.ndim != expect_ndim:
        raise DimensionMismatch(f"ds_dout has {ds_dout.ndim} dims, expected {expect_ndim}")
    if not c["single"] and ds_dout.shape[-1] != B:
        raise DimensionMismatch(f"ds_dout batch dim {ds_dout.shape[-1]} != number of poses {B}")
    g = ds_dout.to(dtype)
    if not _is_grid_layout(g):
        g = to_grid_layout(g)
    grid = tuple(g.shape[:n_out])
    grid_arr = np.asarray(grid, dtype=np.int64)
    tgt = None
    if residual is not None:
        target, res_scale, loss = residual
        if not isinstance(target, torch.Tensor) or target.device != dev:
            raise RuntimeError("target must be a tensor on the same HIP device as points")
        if tuple(target.shape) != tuple(ds_dout.shape):
            raise DimensionMismatch(
                f"target shape {tuple(target.shape)} != out shape {tuple(ds_dout.shape)}")
        tgt = target.to(dtype)
        if not _is_grid_layout(tgt):
            tgt = to_grid_layout(tgt)

    def out_buf(given, shape, name):
        if given is None:
            return torch.empty(shape, dtype=dtype, device=dev)
        if (not isinstance(given, torch.Tensor) or given.device != dev or given.dtype != dtype
                or tuple(given.shape) != tuple(shape) or not given.is_contiguous()):
            raise DimensionMismatch(
                f"{name}: need a contiguous {dtype} tensor of shape {tuple(shape)} on {dev}")
        return given

    d_pts = out_buf(ds_dpoints, (P, n_in), "ds_dpoints")
    # column-major (N_out, N_in, B) buffer == contiguous (B, N_in, N_out)
    if ds_drotation is not None:
        rv = ds_drotation.transpose(-1, -2) if not c["single"] else ds_drotation.t()[None]
        if rv.shape != (B, n_in, n_out) or not rv.is_contiguous() or rv.dtype != dtype:
            raise DimensionMismatch(
                "ds_drotation must be a (B, N_out, N_in) transposed view of a contiguous "
                "(B, N_in, N_out) buffer (column-major N_out x N_in per pose)")
        d_rot = rv
    else:
        d_rot = torch.empty((B, n_in, n_out), dtype=dtype, device=dev)
    d_trans = out_buf(None if ds_dtranslation is None else ds_dtranslation.reshape(B, n_out),
                      (B, n_out), "ds_dtranslation")
    d_bg = out_buf(None if ds_dbackground is None else ds_dbackground.reshape(B), (B,),
                   "ds_dbackground")
    d_ow = out_buf(None if ds_dout_weight is None else ds_dout_weight.reshape(B), (B,),
                   "ds_dout_weight")
    if not point_weight_grad and ds_dpoint_weight is not None:
        raise ValueError("point_weight_grad=False and a ds_dpoint_weight buffer contradict each other")
    d_pw = out_buf(ds_dpoint_weight, (P,), "ds_dpoint_weight") if point_weight_grad else None
    d_loss = None
    if residual is not None:
        d_loss = out_buf(None if loss is None else loss.reshape(B), (B,), "loss")

    suf = _SUFFIX[dtype]
    algo_c = _lib.ALGOS[algo]
    with torch.cuda.device(dev):
        flags = _lib.flag_max_pose_group(max_pose_group)
        flags |= _lib.FLAG_COHERENT_POINTS if coherent_points else 0
        flags |= _lib.FLAG_REUSE_BINNING if reuse_binning else 0
        flags |= 0 if point_weight_grad else _lib.FLAG_NO_POINT_WEIGHT_GRAD
        ws, ws_bytes = _workspace(_lib.OP_PULLBACK if residual is None else _lib.OP_RESIDUAL_PULLBACK,
                                  algo_c, suf, n_in, n_out, grid_arr, P, B, dev, workspace, flags)
        if reuse_binning and workspace is None:
            raise ValueError("reuse_binning needs the workspace of the preceding raster_ call")
        head = (_stream_ptr(dev), algo_c, flags, n_in, n_out,
                grid_arr.ctypes.data_as(ctypes.c_void_p), P, B)
        pose = (_ptr(c["points"]), _ptr(c["rot"]), _ptr(c["trans"]), _ptr(c["ow"]), _ptr(c["pw"]))
        outs = (_ptr(d_pts), _ptr(d_rot), _ptr(d_trans), _ptr(d_bg), _ptr(d_ow), _ptr(d_pw),
                _ptr(ws), ws_bytes)
        if residual is None:
            fn = getattr(_lib.lib(), f"dpr_raster_pullback_ex_{suf}")
            _lib.check(fn(*head, _ptr(g), *pose, *outs))
        else:
            fn = getattr(_lib.lib(), f"dpr_raster_residual_pullback_ex_{suf}")
            _lib.check(fn(*head, _ptr(g), _ptr(tgt), res_scale, *pose, _ptr(d_loss), *outs))
    rot_math = d_rot.transpose(1, 2)
    if c["single"]:
        res = PullbackResult(d_pts, rot_math[0], d_trans[0], d_bg[0], d_ow[0], d_pw)
    else:
        res = PullbackResult(d_pts, rot_math, d_trans, d_bg, d_ow, d_pw)
    if residual is None:
        return res
    return res, (d_loss[0] if c["single"] else d_loss)


def sort_points(points: torch.Tensor, point_weight: Optional[torch.Tensor] = None):
    """Pose-independent Morton pre-sort of the model-frame points (dpr_sort_points_*).
    Returns (points_sorted, perm[, point_weight_sorted]) with points_sorted[i] =
    points[perm[i]]; gradients computed on the sorted cloud go back with
    `ds_dpoints.index_copy_(0, perm.long(), ds_dpoints_sorted)`."""
    device = _device_of(points)
    if points.ndim != 2 or points.dtype not in _SUFFIX:
        raise DimensionMismatch("points must be a (P, N_in) float32/float64 tensor")
    pts = points.contiguous()
    P, n_in = pts.shape
    out = torch.empty_like(pts)
    perm = torch.empty(P, dtype=torch.int32, device=device)
    pw = None if point_weight is None else _as(point_weight, pts.dtype, device, (P,), "point_weight")
    pw_out = None if pw is None else torch.empty_like(pw)
    L = _lib.lib()
    need = L.dpr_sort_points_workspace_bytes(P)
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        fn = getattr(L, f"dpr_sort_points_{_SUFFIX[pts.dtype]}")
        _lib.check(fn(_stream_ptr(device), n_in, P, _ptr(pts), _ptr(out), _ptr(perm), _ptr(pw),
                      _ptr(pw_out), _ptr(ws), ws.numel()))
    return (out, perm) if pw is None else (out, perm, pw_out)
