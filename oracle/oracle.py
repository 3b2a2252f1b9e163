"""ORACLE -- TEST INFRASTRUCTURE ONLY (ctypes front-end of oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.

numpy conventions (mathematical layout; the wrapper converts to the reference's
memory layout):
  points      (P, n_in)
  rotation    (B, n_out, n_in)   mathematical matrices
  translation (B, n_out)
  background / out_weight (B,) or None ; point_weight (P,) or None
  out / ds_dout  (n_1, ..., n_N, B) Fortran-ordered, i.e. out[i1, i2, b] indexes
                 like the reference's column-major `out[i1, i2, b]` (0-based)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PullbackResult = namedtuple(
    "PullbackResult",
    ["points", "rotation", "translation", "background", "out_weight", "point_weight"],
)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("dpr_oracle.c", "dpr_oracle_impl.h")]
    if force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _suffix(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _prep(points, rotation, translation, dtype):
    points = np.ascontiguousarray(points, dtype=dtype)
    rotation = np.asarray(rotation, dtype=dtype)
    translation = np.ascontiguousarray(translation, dtype=dtype)
    assert points.ndim == 2 and rotation.ndim == 3 and translation.ndim == 2
    B, n_out, n_in = rotation.shape
    assert points.shape[1] == n_in and translation.shape == (B, n_out)
    # column-major N_out x N_in per pose == row-major (n_in, n_out) of the transpose
    rot_cm = np.ascontiguousarray(np.transpose(rotation, (0, 2, 1)))
    return points, rot_cm, translation, B, n_in, n_out


def _opt(a, n, dtype):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    assert a.shape == (n,)
    return a


def voxel_shifts(n: int) -> np.ndarray:
    out = np.zeros((1 << n, n), dtype=np.int64)
    lib().oracle_voxel_shifts(ctypes.c_int(n), _ptr(out))
    return out


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def raster(grid_size, points, rotation, translation, background=None, out_weight=None,
           point_weight=None, dtype=np.float64, threaded: bool = False) -> np.ndarray:
    dtype = np.dtype(dtype)
    points, rot_cm, translation, B, n_in, n_out = _prep(points, rotation, translation, dtype)
    assert len(grid_size) == n_out
    P = points.shape[0]
    background = _opt(background, B, dtype)
    out_weight = _opt(out_weight, B, dtype)
    point_weight = _opt(point_weight, P, dtype)
    grid = np.asarray(grid_size, dtype=np.int64)
    out = np.empty(tuple(grid_size) + (B,), dtype=dtype, order="F")
    name = "oracle_raster_threaded_" if threaded else "oracle_raster_"
    fn = getattr(lib(), name + _suffix(dtype))
    fn.restype = ctypes.c_int
    rc = fn(ctypes.c_int(n_in), ctypes.c_int(n_out), _ptr(grid), ctypes.c_int64(P),
            ctypes.c_int64(B), _ptr(out), _ptr(points), _ptr(rot_cm), _ptr(translation),
            _ptr(background), _ptr(out_weight), _ptr(point_weight))
    if rc != 0:
        raise RuntimeError(f"oracle raster failed rc={rc}")
    return out


def raster_pullback(ds_dout, points, rotation, translation, out_weight=None, point_weight=None,
                    dtype=np.float64, threaded: bool = False, n_threads: int = 0) -> PullbackResult:
    """ds_dout: (n_1..n_N, B) (any memory order).  Returns mathematical layouts:
    points (P, n_in), rotation (B, n_out, n_in), translation (B, n_out),
    background (B,), out_weight (B,), point_weight (P,)."""
    dtype = np.dtype(dtype)
    points, rot_cm, translation, B, n_in, n_out = _prep(points, rotation, translation, dtype)
    P = points.shape[0]
    ds_dout = np.asfortranarray(ds_dout, dtype=dtype)
    assert ds_dout.ndim == n_out + 1 and ds_dout.shape[-1] == B
    grid = np.asarray(ds_dout.shape[:-1], dtype=np.int64)
    out_weight = _opt(out_weight, B, dtype)
    point_weight = _opt(point_weight, P, dtype)
    d_points = np.empty((P, n_in), dtype=dtype)
    d_rot_cm = np.empty((B, n_in, n_out), dtype=dtype)
    d_trans = np.empty((B, n_out), dtype=dtype)
    d_bg = np.empty((B,), dtype=dtype)
    d_ow = np.empty((B,), dtype=dtype)
    d_pw = np.empty((P,), dtype=dtype)
    args = [ctypes.c_int(n_in), ctypes.c_int(n_out), _ptr(grid), ctypes.c_int64(P),
            ctypes.c_int64(B), _ptr(ds_dout), _ptr(points), _ptr(rot_cm), _ptr(translation),
            _ptr(out_weight), _ptr(point_weight), _ptr(d_points), _ptr(d_rot_cm), _ptr(d_trans),
            _ptr(d_bg), _ptr(d_ow), _ptr(d_pw)]
    if threaded:
        fn = getattr(lib(), "oracle_raster_pullback_threaded_" + _suffix(dtype))
        args.append(ctypes.c_int(n_threads if n_threads > 0 else max_threads()))
    else:
        fn = getattr(lib(), "oracle_raster_pullback_" + _suffix(dtype))
    fn.restype = ctypes.c_int
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"oracle raster_pullback failed rc={rc}")
    return PullbackResult(d_points, np.transpose(d_rot_cm, (0, 2, 1)).copy(), d_trans, d_bg, d_ow, d_pw)


def residual_pullback(out, target, points, rotation, translation, out_weight=None,
                      point_weight=None, scale: float = 2.0, dtype=np.float64, **kw):
    """The reference's explicit-interface recipe one step out of `raster_pullback!`
    (/root/reference/README.md:151-165): form the sensitivity of a squared-error loss on the
    host, `ds_dout = scale .* (out .- target)` (README.md:151 is the scale = -2 case,
    examples/logo.jl:40-44 the loss whose gradient is scale = +2), then call the pullback.
    Returns (PullbackResult, loss[B]) with loss[b] = sum((out - target)[..., b] ** 2)."""
    dtype = np.dtype(dtype)
    out = np.asarray(out, dtype=dtype)
    target = np.asarray(target, dtype=dtype)
    assert out.shape == target.shape
    resid = out - target
    ds_dout = (dtype.type(scale) * resid).astype(dtype)
    res = raster_pullback(ds_dout, points, rotation, translation, out_weight, point_weight,
                          dtype=dtype, **kw)
    loss = (resid.astype(np.float64) ** 2).reshape(-1, out.shape[-1]).sum(axis=0)
    return res, loss
