"""ctypes binding of libdpr.so (the C ABI in include/dpr.h).

There is deliberately NO fallback: if the HIP library is missing or fails to
load, importing/using the package raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPR_LIB_OVERRIDE: experiment hook to A/B differently compiled builds of the same library
LIB_PATH = os.environ.get("DPR_LIB_OVERRIDE") or os.path.join(_HERE, "libdpr.so")

# status codes / enums of include/dpr.h
OK = 0
ERR_UNSUPPORTED_DIMS = -1
ERR_INVALID_ARG = -2
ERR_WORKSPACE = -3
ERR_HIP = -4
ERR_UNSUPPORTED_ALGO = -5
OP_RASTER = 0
OP_PULLBACK = 1
OP_RESIDUAL_PULLBACK = 2
ALGO_AUTO = 0
ALGO_ATOMIC = 1
ALGO_TILED = 2
ALGO_CHUNKED = 3
FLAG_KEEP_BINNING = 1
FLAG_REUSE_BINNING = 2
FLAG_COHERENT_POINTS = 4
FLAG_NO_POINT_WEIGHT_GRAD = 8


def flag_max_pose_group(n: int) -> int:
    """DPR_FLAG_MAX_POSE_GROUP(n) of include/dpr.h (0 = library default)."""
    if not 0 <= int(n) <= 16:
        raise ValueError("max_pose_group must be in 0..16")
    return (int(n) & 0xFF) << 8
ALGOS = {"auto": ALGO_AUTO, "atomic": ALGO_ATOMIC, "tiled": ALGO_TILED, "chunked": ALGO_CHUNKED}

EXPORTS = [
    "dpr_version", "dpr_last_error", "dpr_stage_timing_begin", "dpr_stage_timing_end",
    "dpr_resolve_algo", "dpr_resolve_algo_ex", "dpr_resolve_flags_ex", "dpr_sort_points_workspace_bytes", "dpr_sort_points_f32",
    "dpr_sort_points_f64",
    "dpr_workspace_bytes_f32", "dpr_workspace_bytes_f64",
    "dpr_workspace_bytes_ex_f32", "dpr_workspace_bytes_ex_f64",
    "dpr_raster_f32", "dpr_raster_f64", "dpr_raster_ex_f32", "dpr_raster_ex_f64",
    "dpr_raster_pullback_f32", "dpr_raster_pullback_f64",
    "dpr_raster_pullback_ex_f32", "dpr_raster_pullback_ex_f64",
    "dpr_raster_residual_pullback_f32", "dpr_raster_residual_pullback_f64",
    "dpr_raster_residual_pullback_ex_f32", "dpr_raster_residual_pullback_ex_f64",
    "dpr_comm_unique_id", "dpr_comm_init", "dpr_comm_destroy", "dpr_comm_world", "dpr_comm_rank",
    "dpr_shard_range", "dpr_raster_pullback_sharded_f32", "dpr_raster_pullback_sharded_f64",
]

_lib = None


class DprError(RuntimeError):
    """A libdpr entry point returned a non-zero status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libdpr status {code}: {message}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile libdpr.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess

    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C <pkg>/csrc). "
            "There is no CPU fallback."
        )
    L = ctypes.CDLL(LIB_PATH)
    vp, i, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    L.dpr_version.restype = i
    L.dpr_version.argtypes = []
    L.dpr_last_error.restype = ctypes.c_char_p
    L.dpr_last_error.argtypes = []
    L.dpr_stage_timing_begin.restype = i
    L.dpr_stage_timing_begin.argtypes = [vp, i]
    L.dpr_stage_timing_end.restype = i
    L.dpr_stage_timing_end.argtypes = []
    L.dpr_resolve_algo.restype = i
    L.dpr_resolve_algo.argtypes = [i, i, i, vp, i64, i64]
    L.dpr_resolve_algo_ex.restype = i
    L.dpr_resolve_algo_ex.argtypes = [i, ctypes.c_uint, i, i, vp, i64, i64]
    L.dpr_resolve_flags_ex.restype = i
    L.dpr_resolve_flags_ex.argtypes = [i, ctypes.c_uint, i, i, vp, i64, i64]
    L.dpr_sort_points_workspace_bytes.restype = sz
    L.dpr_sort_points_workspace_bytes.argtypes = [i64]
    for suf in ("f32", "f64"):
        f = getattr(L, f"dpr_sort_points_{suf}")
        f.restype = i
        f.argtypes = [vp, i, i64, vp, vp, vp, vp, vp, vp, sz]
    for suf in ("f32", "f64"):
        f = getattr(L, f"dpr_workspace_bytes_{suf}")
        f.restype = sz
        f.argtypes = [i, i, i, i, vp, i64, i64]
        f = getattr(L, f"dpr_workspace_bytes_ex_{suf}")
        f.restype = sz
        f.argtypes = [i, i, ctypes.c_uint, i, i, vp, i64, i64]
        # stream, n_in, n_out, grid, P, B, out, points, rot, trans, bg, ow, pw, ws, ws_bytes
        f = getattr(L, f"dpr_raster_{suf}")
        f.restype = i
        f.argtypes = [vp, i, i, vp, i64, i64] + [vp] * 7 + [vp, sz]
        f = getattr(L, f"dpr_raster_ex_{suf}")
        f.restype = i
        f.argtypes = [vp, i, ctypes.c_uint, i, i, vp, i64, i64] + [vp] * 7 + [vp, sz]
        # stream, n_in, n_out, grid, P, B, ds_dout, points, rot, trans, ow, pw, 6 outputs, ws, ws_bytes
        f = getattr(L, f"dpr_raster_pullback_{suf}")
        f.restype = i
        f.argtypes = [vp, i, i, vp, i64, i64] + [vp] * 12 + [vp, sz]
        f = getattr(L, f"dpr_raster_pullback_ex_{suf}")
        f.restype = i
        f.argtypes = [vp, i, ctypes.c_uint, i, i, vp, i64, i64] + [vp] * 12 + [vp, sz]
        # ..., out, target, residual_scale, points, rot, trans, ow, pw, loss, 6 outputs, ws, ws_bytes
        f = getattr(L, f"dpr_raster_residual_pullback_{suf}")
        f.restype = i
        f.argtypes = [vp, i, i, vp, i64, i64, vp, vp, ctypes.c_double] + [vp] * 12 + [vp, sz]
        f = getattr(L, f"dpr_raster_residual_pullback_ex_{suf}")
        f.restype = i
        f.argtypes = ([vp, i, ctypes.c_uint, i, i, vp, i64, i64, vp, vp, ctypes.c_double]
                      + [vp] * 12 + [vp, sz])
    L.dpr_comm_unique_id.restype = i
    L.dpr_comm_unique_id.argtypes = [vp, sz]
    L.dpr_comm_init.restype = i
    L.dpr_comm_init.argtypes = [ctypes.POINTER(vp), i, i, vp]
    L.dpr_comm_destroy.restype = i
    L.dpr_comm_destroy.argtypes = [vp]
    L.dpr_comm_world.restype = i
    L.dpr_comm_world.argtypes = [vp]
    L.dpr_comm_rank.restype = i
    L.dpr_comm_rank.argtypes = [vp]
    L.dpr_shard_range.restype = None
    L.dpr_shard_range.argtypes = [i64, i, i, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    for suf in ("f32", "f64"):
        # comm, stream, n_in, n_out, grid, P, B_local, ds_dout, points, rot, trans, ow, pw, 6 outputs, ws, bytes
        f = getattr(L, f"dpr_raster_pullback_sharded_{suf}")
        f.restype = i
        f.argtypes = [vp, vp, i, i, vp, i64, i64] + [vp] * 12 + [vp, sz]
    _lib = L
    return L


def last_error() -> str:
    return lib().dpr_last_error().decode("utf-8", "replace")


def check(status: int) -> None:
    if status != OK:
        raise DprError(status, last_error())
