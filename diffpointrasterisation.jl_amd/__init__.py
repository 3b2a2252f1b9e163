"""MI355X-native `raster` / `raster_pullback!` hot path of DiffPointRasterisation.jl.

Exports mirror the reference module (`export raster, raster!, raster_pullback!`,
/root/reference/src/DiffPointRasterisation.jl:17); `!` becomes a trailing underscore.
"""
from . import _lib
from ._lib import DprError, build, lib
from .interface import (ColumnMajorRotation, DimensionMismatch, PullbackResult, column_major_rotation,
                        empty_grid, raster, raster_,
                        raster_pullback_, raster_residual_pullback_, resolve_algo,
                        sharing_effective, sort_points, to_grid_layout, workspace_bytes)
from .timing import stage_times
from .autograd import raster_ad
from .sharded import (raster_point_sharded, raster_pullback_point_sharded_,
                      raster_pullback_sharded_, raster_sharded, shard_range)

__all__ = [
    "raster", "raster_", "raster_ad", "raster_pullback_", "raster_residual_pullback_", "PullbackResult",
    "DimensionMismatch", "DprError", "ColumnMajorRotation", "column_major_rotation",
    "empty_grid", "to_grid_layout", "workspace_bytes", "resolve_algo", "sharing_effective", "sort_points", "build", "lib",
    "stage_times", "raster_sharded", "raster_pullback_sharded_", "shard_range",
    "raster_point_sharded", "raster_pullback_point_sharded_",
]
