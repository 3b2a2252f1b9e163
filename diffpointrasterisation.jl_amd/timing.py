"""Per-stage device timing of one raster / pullback call through the library's
dpr_stage_timing_begin/end hooks (include/dpr.h).  The events are raw hipEvent_t objects
recorded by libdpr on the very stream its kernels are enqueued on."""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, List

from . import _lib

STAGES = {
    ("raster", "atomic"): ["fill", "splat"],
    ("pullback", "atomic"): ["zero+grid_sum", "gather"],
    ("raster", "tiled"): ["count", "scan", "scatter", "tile_splat", "halo"],
    # DPR_ALGO_CHUNKED on 3-D grids (owner-computes tiles over a box hierarchy; one group of marks per
    # pass of <= 16 poses, folded below)
    ("raster", "chunked"): ["boxes", "plan", "own_splat", "combine"],
    # ... its sparse regime (B >= 4 poses, P <= G / 10: chunk lists over small tiles): pass algo="chunked_lists"
    ("raster", "chunked_lists"): ["boxes", "lists", "chunk_splat", "divert"],
    ("raster", "chunked_lists_sorting"): ["sort", "boxes", "lists", "chunk_splat", "divert"],
    ("pullback", "chunked"): ["direct_gather", "pose_reduce"],
    # ... of a cloud not flagged coherent over >= 8 poses (sorted inside the call): pass algo="chunked_sorting"
    ("pullback", "chunked_sorting"): ["sort", "direct_gather", "pose_reduce", "unsort"],
    ("raster", "chunked_sorting"): ["sort", "boxes", "plan", "own_splat", "combine"],
    ("pullback", "tiled"): ["count", "scan", "scatter", "tile_gather", "unpermute", "pose_reduce"],
    # DPR_ALGO_TILED with coherent_points=True (local binning): pass algo="tiled_local"
    ("raster", "tiled_local"): ["clear", "bin_local", "runscan", "tile_splat", "halo"],
    ("pullback", "tiled_local"): ["clear", "bin_local", "runscan", "tile_gather", "unpermute", "pose_reduce"],
    # DPR_ALGO_CHUNKED on 2-D grids (chunk-owned tiles): pass algo="chunked2d"
    ("raster", "chunked2d"): ["sort", "fill", "chunk_splat"],
    ("pullback", "chunked2d"): ["sort", "grid_sum", "chunk_gather", "reduce+unsort"],
}

_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        _hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p,
                                             ctypes.c_void_p]
    return _hip


def stage_times(call: Callable[[], None], op: str, algo: str, reps: int = 10,
                max_events: int = 64, prepare: Callable[[], None] = None) -> Dict[str, float]:
    """Run `call` (ONE dpr raster/pullback invocation) `reps` times with stage timing armed; returns
    {stage: mean ms} plus "total".  `prepare` (untimed) runs before every armed call, e.g. the forward
    pass whose binning a reuse-pullback consumes.  A call that walks its stages several times (the
    tiled path once per pose or pose group, the 3-D owner forward once per 16 poses, the direct
    pullback once per launch of <= 64 poses -- fp64: per pose) records one group of marks per walk:
    the groups are folded, each stage reporting the sum over the walks; one-off stages at the ends
    ("sort", "unsort") stay single.  More marks than `max_events` is an error (raise it)."""
    hip = _hiprt()
    L = _lib.lib()
    names = STAGES[(op, algo)]
    acc: List[float] = [0.0] * len(names)
    events = (ctypes.c_void_p * max_events)()
    for i in range(max_events):
        ev = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(ev)) == 0
        events[i] = ev
    try:
        for _ in range(reps):
            if prepare is not None:
                prepare()
            _lib.check(L.dpr_stage_timing_begin(events, max_events))
            try:
                call()
            finally:
                n = L.dpr_stage_timing_end()
            if n >= max_events:
                raise ValueError(f"more stage marks than max_events = {max_events}: raise it")
            # marks = 1 (call start) + [one-off leading stages] + walks x (repeating stages)
            lead = 1 if names[0] == "sort" else 0
            tail = 1 if names[-1] == "unsort" else 0
            rep = len(names) - lead - tail
            intervals = n - 1 - lead - tail
            assert intervals >= rep and intervals % rep == 0, \
                f"expected 1 + {lead} + k * {rep} + {tail} stage marks for {names}, got {n}"
            assert hip.hipEventSynchronize(events[n - 1]) == 0
            for i in range(n - 1):
                k = i if i < lead else (len(names) - 1 if i >= n - 1 - tail else lead + (i - lead) % rep)
                ms = ctypes.c_float()
                assert hip.hipEventElapsedTime(ctypes.byref(ms), events[i], events[i + 1]) == 0
                acc[k] += ms.value
    finally:
        for i in range(max_events):
            hip.hipEventDestroy(events[i])
    out = {nm: acc[k] / reps for k, nm in enumerate(names)}
    out["total"] = sum(out.values())
    return out
