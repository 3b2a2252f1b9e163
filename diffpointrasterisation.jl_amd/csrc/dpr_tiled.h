// DPR_ALGO_TILED / DPR_ALGO_CHUNKED entry points shared between dpr_api.hip, dpr_tiled.hip,
// dpr_owner.hip, dpr_chunkown.hip and dpr_sort.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dpr_device.h"

namespace dpr {

int fail(int code, const char* fmt, ...);

// A/B switches of the experiments (profiles/r0N_experiments.md) are environment variables ONLY in builds
// with -DDPR_EXPERIMENTS (`make EXPERIMENTS=1`); the shipped library compiles them out and runs the
// defaults below, so no environment variable changes its algorithm choice or its numerics.  The one
// exception, documented in include/dpr.h, is DPR_MAX_TILES (slab size of the tiled path; results unchanged).
inline int env_knob(const char* name, int dflt, int lo, int hi) {
#ifdef DPR_EXPERIMENTS
    const char* v = getenv(name);
    const int x = v ? atoi(v) : dflt;
    return x < lo ? lo : (x > hi ? hi : x);
#else
    (void)name; (void)lo; (void)hi;
    return dflt;
#endif
}

// records hipEvent k (if stage timing is armed, see dpr_stage_timing_begin) on `st`
void stage_mark(hipStream_t st);

bool tiled_supported(int n_out, const int64_t* grid);
bool tiled_preferred(int op, int n_out, const int64_t* grid, int64_t P, int64_t B, int64_t G);
size_t tiled_workspace_bytes(size_t elem, int op, unsigned flags, int n_in, int n_out,
                             const int64_t* grid, int64_t P, int64_t B);
bool tiled_batch_share_ok(int n_out, const int64_t* grid, int64_t P, int64_t B);
int tiled_tiles(int n_out, const int64_t* grid);
int tiled_slabs(int n_out, const int64_t* grid);  // 1: one piece; > 1: slabs along the last axis; 0: unsupported

template <typename T, int NI, int NO>
int raster_tiled(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B, T* out,
                 const T* points, const T* rot, const T* trans, const T* bg, const T* ow,
                 const T* pw, void* ws, size_t ws_bytes);

template <typename T, int NI, int NO>
int pullback_tiled(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                   const T* g, const T* points, const T* rot, const T* trans, const T* ow,
                   const T* pw, T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow, T* d_pw,
                   void* ws, size_t ws_bytes, Residual<T> rs);

// DPR_ALGO_CHUNKED on 2-D grids: chunk-owned LDS tiles, pose loop inside (dpr_chunkown.hip)
size_t chunkown_workspace_bytes(size_t elem, int op, unsigned flags, int n_in, int64_t P,
                                int64_t B);
template <typename T, int NI>
int raster_chunkown(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                    int64_t B, T* out, const T* points, const T* rot, const T* trans, const T* bg,
                    const T* ow, const T* pw, void* ws, size_t ws_bytes);
template <typename T, int NI>
int pullback_chunkown(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                      int64_t B, const T* g, const T* points, const T* rot, const T* trans,
                      const T* ow, const T* pw, T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow,
                      T* d_pw, void* ws, size_t ws_bytes, Residual<T> rs);

// DPR_ALGO_CHUNKED on 3-D grids, sparse clouds over several poses: chunk lists (dpr_chunked.hip)
bool chunked_supported(int n_out, const int64_t* grid);
size_t chunked_workspace_bytes(int n_out, const int64_t* grid, int64_t P, int64_t B);
template <typename T, int NI, int NO>
int raster_chunked(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                   int64_t B, T* out, const T* points, const T* rot, const T* trans, const T* bg,
                   const T* ow, const T* pw, void* ws, size_t ws_bytes);

// DPR_ALGO_CHUNKED on 3-D grids: owner-computes tiles over a box hierarchy, direct pullback (dpr_owner.hip)
size_t coarse_sort_scratch_bytes(size_t elem, int64_t P);
template <typename T>
int coarse_sort_with_perm(hipStream_t st, int n_in, int64_t P, const T* points, const T* pw, T* points_sorted,
                          T* pw_sorted, uint32_t* perm, char* scratch);
bool owner_supported(const int64_t* grid);
int64_t owner_tiles(const int64_t* grid);  // 32 x 32 x 14-cell tiles of the owner-computes forward
size_t owner_workspace_bytes(int op, const int64_t* grid, int64_t P, int64_t B);

template <typename T>
int raster_owner(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                 T* out, const T* points, const T* rot, const T* trans, const T* bg, const T* ow,
                 const T* pw, void* ws, size_t ws_bytes);

template <typename T>
int pullback_owner(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                   const T* g, const T* points, const T* rot, const T* trans, const T* ow, const T* pw,
                   T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow, T* d_pw, void* ws, size_t ws_bytes);

// dpr_sort.hip: Hilbert sort of a cloud (keys, radix sort, gather); `fine`: 30-bit keys for 3-D clouds
size_t sort_workspace_bytes(int64_t P);
template <typename T>
int sort_points_impl(void* stream, int n_in, int64_t P, const T* points, T* points_sorted, uint32_t* perm,
                     const T* pw, T* pw_sorted, void* ws_, size_t ws_bytes, uint32_t* inv_perm, bool fine);

}  // namespace dpr
