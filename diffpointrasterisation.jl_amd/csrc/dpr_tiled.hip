// DPR_ALGO_TILED: per-pose binning of the points into voxel tiles, then one workgroup per
// work item (a tile, or a part of a heavily loaded tile) that keeps the tile in LDS.
// No global float atomics anywhere on this path.  DESIGN.md 4.2 has the table of stages.
//
// Per pose b -- or per POSE GROUP of up to 16 poses when the grid has few tiles (bins are then
// (pose, tile) pairs; see pose_group()) -- sequential launches on the caller's stream,
// workspace reused:
//   count    k_count        each block histograms its slice of the points by PRIMARY tile (the
//                           tile holding max(ref,0)) in LDS -> one row of the counts table
//   scan     k_colscan      column-wise exclusive prefix of the table + tile totals
//            k_tilescan     tile offsets; work list (heavy tiles split, heaviest first)
//   scatter  k_scatter_wc   LDS cursors seeded from the table -> exact, atomic-free placement
//            (k_scatter)    of one 16/32-byte record per in-range point; sub-chunks are ordered
//                           by tile in LDS first so stores cover contiguous runs
//   forward  k_tile_splat   LDS tile (+1 upper halo) of f64 accumulators, ds_add_f64; owned
//                           voxels leave with plain stores fused with the background, the halo
//                           goes to a compact buffer (parts of split tiles: overflow slabs)
//            k_halo_gather  low-face voxels add the (<= 2^N-1) neighbour halos; split tiles are
//                           assembled from their slabs
//   pullback k_tile_gather  ds_dout tile (+halo) staged in LDS, per-point gathers from LDS, the
//                           gradient overwrites the record in place; per-item partial sums
//            k_unpermute    gradient records back to the original point order
//            k_pose_reduce  per-item partials (f64) -> ds_drotation, ds_dtranslation, ...
//
// Why f64 accumulators for fp32 data: on gfx950 ds_add_f32 retires ~1 lane per 3 cycles
// (193 cycles per wave-instruction, measured), while ds_add_f64 takes ~26 cycles per
// wave-instruction (profiles/r01_microbench_lds_atomics.txt).  It also makes fp32 results
// practically independent of the accumulation order.
//
// Experiment knobs (environment variables in -DDPR_EXPERIMENTS builds only -- `make EXPERIMENTS=1`; the
// compiled-in defaults are the measured best and all the shipped library knows):
//   DPR_SCATTER_WC=0     plain scatter instead of the write-combining one
//   DPR_SPLAT_BLOCKED=0|1  lane-adjacent (strided) / blocked record assignment in k_tile_splat
//                        (default: chosen on the device from the order of the cloud)
//   DPR_BWD_UNPERMUTE=0  owner threads store ds_dpoints directly instead of un-permuting
//   DPR_POSE_GROUP=n     at most n poses per group (1 = per-pose pipeline)
//   DPR_FIXED_POINT=0    f64 LDS accumulators instead of 64-bit fixed point in the fp32 forward
//   DPR_FUSE_TILESCAN=0  k_tilescan / k_runscan as launches of their own
//   DPR_BIN_DIRECT_STORE=0  fp64 batches stage their records in LDS like everything else
// Always read (test hook, documented in include/dpr.h; it moves slab boundaries, never results):
//   DPR_MAX_TILES=n      tiles per launch sequence (default 32768): lets a test walk slabs on small grids
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "../../include/dpr.h"
#include "dpr_device.h"
#include "dpr_tiled.h"

namespace dpr {

// ------------------------------------------------------------------ tile geometry
template <int NO> struct TileDims;
// 3-D tile shape and K4 block size (A/B measured on C3, profiles/r01_tile_shape_sweep.txt)
#ifndef DPR_TX3
#define DPR_TX3 64
#endif
#ifndef DPR_TY3
#define DPR_TY3 16
#endif
#ifndef DPR_TZ3
#define DPR_TZ3 8
#endif
template <> struct TileDims<3> {
    static constexpr int T[3] = {DPR_TX3, DPR_TY3, DPR_TZ3};
};
template <> struct TileDims<2> {
    static constexpr int T[3] = {32, 32, 1};
};
constexpr int kMaxTiles = 32768;     // LDS cursor table: 4 B per tile, <= 128 KiB
constexpr int kBinThreads = 1024;    // K1 / K3 block
#ifndef DPR_WC_THREADS
#define DPR_WC_THREADS 1024
#endif
#ifndef DPR_WC_PPT
#define DPR_WC_PPT 4
#endif
constexpr int kWcThreads = DPR_WC_THREADS;  // block of the write-combining scatter
#ifndef DPR_WC_SLICES
#define DPR_WC_SLICES 1  // 0: the slice rule of the plain scatter for the write-combining one too
#endif
#ifndef DPR_SPLAT_THREADS
#define DPR_SPLAT_THREADS 512
#endif
#ifndef DPR_GATHER_THREADS
#define DPR_GATHER_THREADS 256
#endif
constexpr int kSplatThreads = DPR_SPLAT_THREADS;    // forward tile kernel block
#ifndef DPR_SPLAT_RUNS_OCC
#define DPR_SPLAT_RUNS_OCC (DPR_SPLAT_THREADS >= 1024 ? 8 : 4)  // waves per SIMD for two blocks per CU
#endif
#ifndef DPR_GATHER_THREADS_F64
#define DPR_GATHER_THREADS_F64 512
#endif
// fp64: the ds_dout tile is 80 KB, two workgroups per CU -- 512 threads each keep 16 waves on the CU
// as the fp32 kernel's four workgroups of 256 do (with 256: 2.3 waves per SIMD, half the wave time
// spent waiting; profiles/r04_c5_sq_counters.txt)
template <typename T> __host__ __device__ constexpr int gather_threads() {
    return sizeof(T) == 8 ? DPR_GATHER_THREADS_F64 : DPR_GATHER_THREADS;
}
// waves per SIMD the pullback tile kernels are compiled for (fp64: 128 VGPRs, so that two
// workgroups of 512 fit a CU; fp32: four workgroups of 256 need no more than that either)
template <typename T> __host__ __device__ constexpr int gather_waves_per_simd() { return 4; }
#ifndef DPR_UPB
#define DPR_UPB 4  // points per thread of the single-pose un-permute (a block = one scatter sub-chunk)
#endif
#ifndef DPR_GATHER_RB
#define DPR_GATHER_RB 8  // rows of the ds_dout tile a wave requests before it stores the first
#endif
// (the pullback tile kernels run 4 workgroups = 16 waves per CU: their 40 KB ds_dout tile sets
// that, not the ~100 VGPRs)
#ifndef DPR_BIN_BLOCKS
#define DPR_BIN_BLOCKS 512
#endif
constexpr int kMaxBinBlocks = DPR_BIN_BLOCKS;  // rows of the counts table (2 per CU)
constexpr int kSplitChunks = 8;      // k_halo_gather work items per split tile (3-D: kSplitRows tile rows each;
                                     // 2-D: 256 voxels a step of the flat loop)
constexpr int kSplitRows = 16;       // rows (l1, l2) of a 64 x 16 x 8 tile per work item
constexpr int kSplitGrid = 2048;     // ... and the blocks that walk them (idle blocks cost nothing
                                     // measurable: a grid limited to the live items changed no kernel time)

// Tile geometry of one launch sequence.  Grids of up to kMaxTiles tiles are one piece; larger
// ones are processed in SLABS along the last axis: `nt[NO-1]` tile layers starting at global layer
// `tz0`, the rest of the grid is invisible to the launch (points whose primary tile lies outside
// are treated like points outside the grid).  A forward slab other than the first starts with a
// GHOST layer (ghost = 1): the top layer of the slab below, binned and accumulated once more only
// for its upper halo -- the first real layer's low faces need it, and the slab below has long
// overwritten its halo buffer; ghost tiles flush no owned voxel and receive no halo.
template <int NO> struct TileGeom {
    int nt[NO];  // tiles per axis (of this slab)
    int NT;      // tiles per pose (of this slab)
    int tz0;     // first tile layer along the last axis
    int ghost;   // 1: local layer 0 is a ghost layer
};

template <int NO> __host__ __device__ constexpr int tile_voxels() {
    int v = 1;
    for (int d = 0; d < NO; ++d) v *= TileDims<NO>::T[d];
    return v;
}
template <int NO> __host__ __device__ constexpr int tile_voxels_halo() {
    int v = 1;
    for (int d = 0; d < NO; ++d) v *= TileDims<NO>::T[d] + 1;
    return v;
}
template <int NO> __host__ __device__ constexpr int halo_count() {
    return tile_voxels_halo<NO>() - tile_voxels<NO>();
}

// How a grid is cut into slabs: `per_layer` tiles in a layer of the last axis, `layers` layers,
// at most `lps` real layers per slab (a slab's tile count incl. a ghost layer stays <= kMaxTiles).
struct SlabCut {
    int per_layer, layers, lps, nslab;
};
// tiles one launch sequence may hold: kMaxTiles, or less through DPR_MAX_TILES (read once; lets a
// test walk slabs on a small grid)
static int max_tiles() {
    static const int v = [] {
        const char* e = getenv("DPR_MAX_TILES");
        int x = e ? atoi(e) : kMaxTiles;
        return x < 16 ? 16 : (x > kMaxTiles ? kMaxTiles : x);
    }();
    return v;
}
template <int NO> static bool make_slab_cut(const int64_t* grid, SlabCut* sc) {
    const int cap = max_tiles();  // tiles one launch sequence may hold (kMaxTiles unless DPR_MAX_TILES)
    int64_t per = 1;
    for (int d = 0; d + 1 < NO; ++d) per *= (grid[d] + TileDims<NO>::T[d] - 1) / TileDims<NO>::T[d];
    const int64_t layers = (grid[NO - 1] + TileDims<NO>::T[NO - 1] - 1) / TileDims<NO>::T[NO - 1];
    if (per > cap / 2 || layers > (1 << 20)) return false;  // (a real + a ghost layer must fit)
    sc->per_layer = (int)per;
    sc->layers = (int)layers;
    if (per * layers <= cap) {
        sc->lps = (int)layers;
        sc->nslab = 1;
    } else {
        sc->lps = (int)(cap / per) - 1;
        sc->nslab = (int)((layers + sc->lps - 1) / sc->lps);
    }
    return true;
}
// geometry of slab `s` (forward: with the ghost layer for s > 0)
template <int NO>
static TileGeom<NO> slab_geom(const int64_t* grid, const SlabCut& sc, int s, bool forward) {
    TileGeom<NO> tg;
    for (int d = 0; d + 1 < NO; ++d)
        tg.nt[d] = (int)((grid[d] + TileDims<NO>::T[d] - 1) / TileDims<NO>::T[d]);
    const int first = s * sc.lps;
    const int real = (sc.layers - first < sc.lps) ? sc.layers - first : sc.lps;
    tg.ghost = (forward && s > 0) ? 1 : 0;
    tg.tz0 = first - tg.ghost;
    tg.nt[NO - 1] = real + tg.ghost;
    tg.NT = sc.per_layer * tg.nt[NO - 1];
    return tg;
}
// the largest tile count any slab of the cut has (what the workspace is planned for)
static int slab_max_tiles(const SlabCut& sc) {
    return sc.nslab == 1 ? sc.per_layer * sc.layers : sc.per_layer * (sc.lps + 1);
}

// single-piece geometry (false: the grid needs slabs, or is beyond them)
template <int NO> static bool make_geom(const int64_t* grid, TileGeom<NO>* tg) {
    SlabCut sc;
    if (!make_slab_cut<NO>(grid, &sc) || sc.nslab != 1) return false;
    *tg = slab_geom<NO>(grid, sc, 0, true);
    return true;
}

// primary tile of a point: tile of max(ref0, 0) per axis (ref0 in [-1, n-1]); -1 when that tile
// lies outside the slab of this launch
template <int NO>
__device__ __forceinline__ int primary_tile(const int (&ref0)[NO], const TileGeom<NO>& tg) {
    int t = 0, stride = 1;
    bool in = true;
#pragma unroll
    for (int d = 0; d < NO; ++d) {
        const int r = ref0[d] < 0 ? 0 : ref0[d];
        int c = r / TileDims<NO>::T[d];
        if (d == NO - 1) {
            c -= tg.tz0;
            in = (unsigned)c < (unsigned)tg.nt[d];
        }
        t += c * stride;
        stride *= tg.nt[d];
    }
    return in ? t : -1;
}

// tile -> local tile coordinates tc (neighbour bookkeeping) and GLOBAL voxel origin x0
template <int NO>
__device__ __forceinline__ void tile_origin(int tile, const TileGeom<NO>& tg, int (&x0)[NO],
                                            int (&tc)[NO]) {
#pragma unroll
    for (int d = 0; d < NO; ++d) {
        tc[d] = tile % tg.nt[d];
        tile /= tg.nt[d];
        x0[d] = (tc[d] + (d == NO - 1 ? tg.tz0 : 0)) * TileDims<NO>::T[d];
    }
}

// index inside the (T+1)^N LDS tile
template <int NO> __device__ __forceinline__ int lds_index(const int (&l)[NO]) {
    int idx = 0, stride = 1;
#pragma unroll
    for (int d = 0; d < NO; ++d) {
        idx += l[d] * stride;
        stride *= TileDims<NO>::T[d] + 1;
    }
    return idx;
}

// offset of neighbour s (bit d = +1 along axis d) inside the (T+1)^N LDS tile
template <int NO> __host__ __device__ constexpr int nbr_lds_offset(int s) {
    int off = 0, stride = 1;
    for (int d = 0; d < NO; ++d) {
        if ((s >> d) & 1) off += stride;
        stride *= TileDims<NO>::T[d] + 1;
    }
    return off;
}

// Compact halo layout of one tile.  h has at least one coordinate equal to T[d].
//   3-D: X-face (h_x == TX)            (TY+1)(TZ+1) values, index h_y + (TY+1) h_z
//        Y-face (h_y == TY, h_x < TX)  TX (TZ+1)    values, index h_x + TX h_z
//        Z-face (h_z == TZ, rest low)  TX TY        values, index h_x + TX h_y
//   2-D: X-face (TY+1) values, then Y-face TX values.
template <int NO> __device__ __forceinline__ int halo_index(const int (&h)[NO]) {
    constexpr int TX = TileDims<NO>::T[0], TY = TileDims<NO>::T[1];
    if constexpr (NO == 2) {
        return (h[0] == TX) ? h[1] : (TY + 1) + h[0];
    } else {
        constexpr int TZ = TileDims<NO>::T[2];
        if (h[0] == TX) return h[1] + (TY + 1) * h[2];
        if (h[1] == TY) return (TY + 1) * (TZ + 1) + h[0] + TX * h[2];
        return (TY + 1) * (TZ + 1) + TX * (TZ + 1) + h[0] + TX * h[1];
    }
}

// One binned record, moved as one 16-byte (fp32) / 32-byte (fp64) aligned vector:
//   v[0..2] = point coordinates (0-padded), v[3] = point weight            (HAS_PW)
//                                           v[3] = bits of the point index (!HAS_PW;
//   the default point_weight == 1 needs no storage, so the index rides for free)
// With HAS_PW the original index lives in the separate rec_idx[] array.
template <typename T> struct alignas(4 * sizeof(T)) Rec4 {
    T v[4];
};
// Compact record {x, y, z} (12 / 24 bytes) for a forward-only binning with default point
// weights on the write-combining scatter: a quarter less record traffic (C3 forward 266 -> 257
// us).  A binning that a pullback consumes keeps 4-word records: the gradient record
// overwrites them in place, and a separate gradient buffer costs the un-permute more (it falls
// out of the Infinity Cache) than the smaller records save.
template <typename T> struct Rec3 {
    T v[3];
};
template <typename T, bool W3> struct RecSel { using type = Rec4<T>; };
template <typename T> struct RecSel<T, true> { using type = Rec3<T>; };
template <typename T, bool W3> using RecT = typename RecSel<T, W3>::type;
__device__ __forceinline__ float idx_to_slot(uint32_t i, float) { return __uint_as_float(i); }
__device__ __forceinline__ double idx_to_slot(uint32_t i, double) {
    return __longlong_as_double((long long)i);
}
__device__ __forceinline__ uint32_t slot_to_idx(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t slot_to_idx(double v) {
    return (uint32_t)__double_as_longlong(v);
}

// XCD-aware work mapping (speed only, never correctness): workgroups are dealt round-robin
// over the 8 XCDs, so blocks b and b+8 share an L2.  Giving the blocks of one XCD a
// CONTIGUOUS range of slices keeps neighbouring slices -- whose record runs share cache lines
// at their ends, and whose un-permute reads hit the same 64-byte sectors -- in one L2.
__device__ __forceinline__ unsigned xcd_slice(unsigned block, unsigned nblocks) {
    const unsigned per = nblocks / 8;  // slices per XCD in the swizzled part
    if (block >= per * 8) return block;  // remainder keeps its identity
    return (block % 8) * per + block / 8;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store of the wave (s_waitcnt vmcnt(0)); in loops whose barriers
// only separate LDS phases that wait would put the latency of prefetched loads and of
// fire-and-forget scattered stores on the critical path of every phase.  Global data written
// before this barrier must not be read by other threads of the block after it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// max |point_weight| and min NON-ZERO |point_weight| of the records a binning kernel has seen (wrange_*,
// dpr_device.h: one packed register per thread): wave reduction, then at most two global atomicMax per
// wave -- dst[0] = high half (max), dst[1] = low half (the complemented min; 0 = no non-zero weight
// seen), so that ONE zero-initialisation serves both words.  The tile kernels take the scale of their
// fixed-point sums from the maximum and fall back to f64 atomics when the two are more than 2^10 apart
// (fix_guard_range).  All 64 lanes must call.
__device__ __forceinline__ void publish_one_max(uint32_t* __restrict__ dst, uint32_t bits) {
    // The running maximum is READ first (device scope, past the non-coherent caches) and the atomic
    // only issued when this wave raises it: thousands of waves doing an atomicMax on ONE address
    // serialise in a single L2 channel -- 39 000 of them took ~0.39 ms at the end of k_bin_local
    // (coherent C3 with point weights: 456 us against 66 us without).  A stale read costs an
    // unnecessary atomic, never a missed maximum.
    if ((threadIdx.x & (kWave - 1)) == 0 && bits &&
        bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(dst, bits);
}
__device__ __forceinline__ void publish_max_abs(uint32_t* __restrict__ dst, uint32_t key) {
    key = wrange_wave(key);
    publish_one_max(dst, key >> 16);
    publish_one_max(dst + 1, key & 0xffffu);
}
// |out_weight| * (upper bound of) max |point_weight| for fix_scale, or +Inf (= f64 atomics) when the
// weights of the call span more than kFixMaxWeightRange; w[0] / w[1] as published above
__device__ __forceinline__ float guarded_max_weight(float ow_abs, const uint32_t* __restrict__ w, bool has_pw) {
    if (!has_pw) return ow_abs;
    return ow_abs * wrange_guarded_max(w[0], w[1]);
}
template <typename T> __device__ __forceinline__ uint32_t abs_key(T w) {
    return wrange_key((float)w);  // (fp64 data does not use the fixed-point path)
}

// ------------------------------------------------------------------ K1: count
template <typename T, int NI, int NO, bool GROUP>
__global__ __launch_bounds__(kBinThreads) void k_count(GridDesc<NO> gd, TileGeom<NO> tg, int64_t P,
                                                       int64_t chunk, const T* __restrict__ points,
                                                       const T* __restrict__ rot,
                                                       const T* __restrict__ trans, int64_t b0,
                                                       int nb, uint32_t* __restrict__ counts,
                                                       uint32_t* __restrict__ nonzero_bins) {
    // nonzero_bins[slice] = bins this block's slice of the cloud touches: nearly all of them for
    // a cloud in random order, a few per cent for a spatially sorted one.  The tile scan turns
    // the sum into the record assignment of k_tile_splat (strided / blocked).
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t s_nz;
    const int NTe = tg.NT * nb;  // bins = (pose of the group, tile)
    for (int i = threadIdx.x; i < NTe; i += kBinThreads) hist[i] = 0;
    if (threadIdx.x == 0) s_nz = 0;
    __syncthreads();
    const unsigned slice = xcd_slice(blockIdx.x, gridDim.x);
    const int64_t lo = (int64_t)slice * chunk;
    const int64_t hi = (lo + chunk < P) ? lo + chunk : P;
    if (!GROUP) {
        const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, nullptr, b0);
        // kCU points per thread are requested before the first one is used: with one load in
        // flight per thread (768 B per wave) the 32 waves of a CU cover ~3 TB/s of the chip
#ifndef DPR_COUNT_UNROLL
#define DPR_COUNT_UNROLL 4
#endif
        constexpr int kCU = DPR_COUNT_UNROLL;
        for (int64_t base = lo + threadIdx.x; base < hi; base += (int64_t)kCU * kBinThreads) {
            T pt[kCU][NI];
            bool live[kCU];
#pragma unroll
            for (int u = 0; u < kCU; ++u) {
                const int64_t p = base + (int64_t)u * kBinThreads;
                live[u] = p < hi;
                load_point<T, NI>(points, live[u] ? p : hi - 1, pt[u]);
            }
#pragma unroll
            for (int u = 0; u < kCU; ++u) {
                int ref0[NO];
                T dlo[NO];
                if (ref_and_deltas<T, NI, NO>(pt[u], ps, gd, ref0, dlo) && live[u]) {
                    const int tl = primary_tile<NO>(ref0, tg);
                    if (tl >= 0) atomicAdd(&hist[tl], 1u);
                }
            }
        }
    } else {
        // a pose group: each point is read once and classified for every pose of the group
        // (pose parameters are wave-uniform scalar loads, amortised over kPB points)
        constexpr int kPB = 4;
        for (int64_t base = lo; base < hi; base += (int64_t)kPB * kBinThreads) {
            T pt[kPB][NI];
            bool live[kPB];
#pragma unroll
            for (int k = 0; k < kPB; ++k) {
                const int64_t p = base + threadIdx.x + (int64_t)k * kBinThreads;
                live[k] = p < hi;
                load_point<T, NI>(points, live[k] ? p : hi - 1, pt[k]);
            }
            for (int j = 0; j < nb; ++j) {
                const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, nullptr, b0 + j);
#pragma unroll
                for (int k = 0; k < kPB; ++k) {
                    int ref0[NO];
                    T dlo[NO];
                    if (ref_and_deltas<T, NI, NO>(pt[k], ps, gd, ref0, dlo) && live[k]) {
                        const int tl = primary_tile<NO>(ref0, tg);
                        if (tl >= 0) atomicAdd(&hist[j * tg.NT + tl], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    uint32_t* row = counts + (size_t)slice * NTe;
    uint32_t nz = 0;
    for (int i = threadIdx.x; i < NTe; i += kBinThreads) {
        const uint32_t v = hist[i];
        row[i] = v;
        nz += v != 0u;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) nz += __shfl_xor(nz, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && nz) atomicAdd(&s_nz, nz);
    __syncthreads();
    if (threadIdx.x == 0) nonzero_bins[slice] = s_nz;
}

// ------------------------------------------------------------------ K2: scans
// What a DPR_FLAG_KEEP_BINNING forward leaves at the start of the workspace, and what a
// DPR_FLAG_REUSE_BINNING pullback checks ON THE DEVICE before it trusts the work list, the
// records and the slot map: problem shape, element size, the identity of the point / weight
// buffers and the pose VALUES (bit patterns).  state: kBinValid after a KEEP forward, 0 after
// any other binning and after the pullback that consumed it (its gradient records overwrite
// the point records in place, so a second reuse must not pass).  A pullback that finds no
// matching header launches nothing that touches memory through the stale lists and returns
// NaN in every output (loud, not silent).
constexpr uint32_t kBinMagic = 0x44505242u, kBinValid = 1u;
struct alignas(16) BinHeader {
    uint32_t magic, state;
    uint32_t elem, n_in, n_out, has_pw;
    int64_t P;
    int32_t grid[3];
    uint32_t verdict;  // written by the consuming pullback's first kernel: 1 = header matched
    uint64_t points, pw;
    uint32_t layout, pad_[3];  // plan_layout_id() of the workspace layout the binning was written in
    unsigned char pose[96];  // rotation | translation bytes of pose 0 (<= 9 + 3 doubles)
};
template <typename T, int NI, int NO>
static BinHeader make_header(const int64_t* grid, int64_t P, const T* points, const T* pw) {
    BinHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kBinMagic;
    h.elem = (uint32_t)sizeof(T);
    h.n_in = NI;
    h.n_out = NO;
    h.has_pw = pw ? 1u : 0u;
    h.P = P;
    for (int d = 0; d < NO; ++d) h.grid[d] = (int32_t)grid[d];
    h.points = (uint64_t)(uintptr_t)points;
    h.pw = (uint64_t)(uintptr_t)pw;
    return h;
}
// device side: does the workspace header match `want` and the pose in memory?
__device__ __forceinline__ bool header_matches(const BinHeader* hdr, const BinHeader& want,
                                               const uint32_t* rot, int rot_words,
                                               const uint32_t* trans, int trans_words) {
    bool ok = hdr->magic == kBinMagic && hdr->state == kBinValid && hdr->elem == want.elem &&
              hdr->n_in == want.n_in && hdr->n_out == want.n_out && hdr->has_pw == want.has_pw &&
              hdr->P == want.P && hdr->grid[0] == want.grid[0] && hdr->grid[1] == want.grid[1] &&
              hdr->grid[2] == want.grid[2] && hdr->points == want.points && hdr->pw == want.pw &&
              hdr->layout == want.layout;
    const uint32_t* pose = (const uint32_t*)hdr->pose;
    for (int i = 0; i < rot_words; ++i) ok = ok && pose[i] == rot[i];
    for (int i = 0; i < trans_words; ++i) ok = ok && pose[rot_words + i] == trans[i];
    return ok;
}

// One unit of work of the tile kernels: a contiguous record range of one tile.  Tiles with
// more than `cap` records are split into several items (parts) so that a clustered cloud
// (few heavily loaded tiles) still fills the chip; the parts of a split tile leave their LDS
// tiles in overflow slabs that k_halo_gather sums.
struct alignas(16) WorkItem {
    uint32_t tile, begin, end;
    uint32_t part_nparts;  // part | nparts << 16
};

// Wait HERE for every outstanding global load / store of the wave (s_waitcnt vmcnt(0) through the
// builtin, which the compiler's own wait insertion sees; an inline-asm wait it does not).  Used
// in front of loops that start with a prefetched element: left pending into the loop, that
// prefetch is joined with the back edge as "outstanding behind an unknown number of accesses", and
// the compiler opens EVERY iteration with s_waitcnt vmcnt(0) -- a wait for the stores the
// previous iteration has just issued (k_scatter, the fp64 k_tile_gather: see
// profiles/r03_experiments.md).
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// ... after the prefetched values have been "used" by an empty asm, which keeps the prefetch in
// front of the wait (the builtin alone does not order a load that nothing depends on yet)
template <typename V> __device__ __forceinline__ void pin_value(V& x) { asm volatile("" : "+v"(x)); }
template <typename R> __device__ __forceinline__ void pin_record(R& r) {
#pragma unroll
    for (int k = 0; k < (int)(sizeof(r.v) / sizeof(r.v[0])); ++k) pin_value(r.v[k]);
}

// ---- LOCAL BINNING (DPR_FLAG_COHERENT_POINTS) ------------------------------------------------
// For a spatially coherent cloud the per-pose permutation can stay LOCAL: a block orders one
// sub-chunk of S consecutive points by tile in LDS and writes it out as ONE contiguous run of
// records, plus a descriptor {tile, start, count} for every tile the sub-chunk touches (a
// handful -- 13 of 2048 for a 4096-point sub-chunk of the Hilbert-sorted C3 cloud).  No count
// pass over the points, no counts table, no column scan; the descriptors (3 % of the points'
// bytes) are sorted by tile instead of the records, and the tile kernels walk the record runs
// their descriptors name.  Correct for any order -- an incoherent cloud just yields about as
// many descriptors as points and runs slowly, which is why the caller has to ask for it.
struct alignas(8) RunDesc {
    uint32_t start;       // first record of the run
    uint32_t tile_count;  // tile | count << 15   (tile < 32768, count <= S <= 4096)
    __host__ __device__ uint32_t tile() const { return tile_count & 0x7fffu; }
    __host__ __device__ uint32_t count() const { return tile_count >> 15; }
};
constexpr int kMaxLocalTiles = 16384;  // tiles per pose local binning supports (its LDS histogram)
constexpr int kMaxRuns = 256;        // descriptors per round of a work item in k_tile_splat ...
constexpr int kMaxRunsGather = 64;   // ... and in k_tile_gather (their tables live in LDS: 4
                                     // gather blocks per CU leave room for 64 runs)

// LDS tables of one work item's runs + a forward-only cursor: logical record i of the item ->
// index into the record array.  A thread asks for non-decreasing i only.
template <int N> struct RunTable {
    uint32_t start[N];
    uint32_t prefix[N + 1];
};
struct RunCursor {
    int k;
    uint32_t pos, left;  // next record of the current run, records left in it
    // position the cursor on logical record i (binary search: largest k with prefix[k] <= i)
    template <int N> __device__ __forceinline__ void seek(const RunTable<N>& rt, uint32_t i, int nruns) {
        int lo = 0, hi = nruns - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rt.prefix[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        k = lo;
        pos = rt.start[k] + (i - rt.prefix[k]);
        left = rt.prefix[k + 1] - i;
    }
    // record index of the cursor's position, then advance by `step` logical records (the table
    // is only read when a run is exhausted); `more` = there are records behind the new position
    template <int N>
    __device__ __forceinline__ uint32_t next(const RunTable<N>& rt, uint32_t step, int nruns, bool more) {
        const uint32_t p = pos;
        if (step < left) {
            pos += step;
            left -= step;
        } else if (more) {
            uint32_t skip = step - left;  // records to skip in the following runs
            ++k;
            while (k + 1 < nruns && skip >= rt.prefix[k + 1] - rt.prefix[k]) {
                skip -= rt.prefix[k + 1] - rt.prefix[k];
                ++k;
            }
            pos = rt.start[k] + skip;
            left = rt.prefix[k + 1] - rt.prefix[k] - skip;
        }
        return p;
    }
};
// all threads of the block: load the item's descriptors, build the prefix table; returns the
// number of records of the item.  `nthreads` >= 64; ends with a barrier.
template <int N>
__device__ __forceinline__ uint32_t load_runs(RunTable<N>& rt, const RunDesc* __restrict__ runs,
                                              uint32_t d0, uint32_t d1, uint32_t max_rec) {
    const int nruns = (int)(d1 - d0);
    if (threadIdx.x < kWave) {  // one wave, N / 64 descriptors per lane
        constexpr int PER = N / kWave;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = threadIdx.x * PER + q;
            c[q] = 0;
            if (j < nruns) {
                const RunDesc d = runs[d0 + j];
                // record runs never leave the record buffer, whatever a stale list says
                const uint32_t st = d.start < max_rec ? d.start : max_rec;
                c[q] = d.count() < max_rec - st ? d.count() : max_rec - st;
                rt.start[j] = st;
            }
            sum += c[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, kWave);
            if ((int)threadIdx.x >= o) incl += v;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = threadIdx.x * PER + q;
            if (j <= nruns) rt.prefix[j] = run;
            run += c[q];
        }
        if (threadIdx.x == kWave - 1) rt.prefix[N] = incl;  // == total when nruns == N
    }
    __syncthreads();
    return rt.prefix[nruns];
}

// exclusive scan of totals[NT] -> tile_start[NT + 1]   (single block, NT <= 32768), and the
// work list: items[] ordered by decreasing size (log2 buckets; the heaviest items are
// dispatched first), n_items, and per tile the number of parts and its first overflow slab.
// Everything the tile scan reads and writes (one struct: the scan kernel carries it along).
struct TileScanArgs {
    const uint32_t* totals;
    int NT;
    uint32_t cap;
    uint32_t* tile_start;
    WorkItem* items;
    uint32_t* n_items;
    uint32_t* tile_parts;
    uint32_t* tile_slab;
    uint32_t* split_list;
    uint32_t* n_split;
    BinHeader hdr;
    BinHeader* hdr_out;
    const uint32_t* rot;
    int rot_words;
    const uint32_t* trans;
    int trans_words;
    const uint32_t* nonzero_bins;  // [nblk] from k_count
    int nblk;
};
// one block of 1024 threads
__device__ __forceinline__ void tilescan_body(const uint32_t* __restrict__ totals, int NT,
                                              uint32_t cap, uint32_t* __restrict__ tile_start,
                                              WorkItem* __restrict__ items,
                                              uint32_t* __restrict__ n_items,
                                              uint32_t* __restrict__ tile_parts,
                                              uint32_t* __restrict__ tile_slab,
                                              uint32_t* __restrict__ split_list,
                                              uint32_t* __restrict__ n_split, const BinHeader& hdr,
                                              BinHeader* __restrict__ hdr_out,
                                              const uint32_t* __restrict__ rot, int rot_words,
                                              const uint32_t* __restrict__ trans,
                                              int trans_words,
                                              const uint32_t* __restrict__ nonzero_bins, int nblk) {
    // header of this binning (state = kBinValid only for a KEEP_BINNING forward): the last wave
    // copies the pose words, one lane the fixed fields
    if (threadIdx.x >= 1024 - 64) {
        const int i = threadIdx.x - (1024 - 64);
        uint32_t* pose = (uint32_t*)hdr_out->pose;
        if (i < rot_words) pose[i] = rot[i];
        else if (i < rot_words + trans_words) pose[i] = trans[i - rot_words];
        if (i == 63) {
            hdr_out->magic = hdr.magic;
            hdr_out->state = hdr.state;
            hdr_out->elem = hdr.elem;
            hdr_out->n_in = hdr.n_in;
            hdr_out->n_out = hdr.n_out;
            hdr_out->has_pw = hdr.has_pw;
            hdr_out->P = hdr.P;
            hdr_out->grid[0] = hdr.grid[0];
            hdr_out->grid[1] = hdr.grid[1];
            hdr_out->grid[2] = hdr.grid[2];
            hdr_out->verdict = 0;
            hdr_out->points = hdr.points;
            hdr_out->pw = hdr.pw;
            hdr_out->layout = hdr.layout;
        }
    }
    __shared__ uint32_t wsum[16], wslab[16];
    __shared__ uint32_t s_nsplit, s_nzsum;
    if (threadIdx.x == 0) {
        s_nsplit = 0;
        s_nzsum = 0;
    }
    __shared__ uint32_t bcount[33], bstart[33];
    if (threadIdx.x < 33) bcount[threadIdx.x] = 0;
    __syncthreads();
    {   // how coherent is the cloud's order?  (bins touched per count block, summed)
        uint32_t v = (threadIdx.x < nblk) ? nonzero_bins[threadIdx.x] : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_nzsum, v);
    }
    const int per = (NT + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    uint32_t s = 0, slabs = 0;
    // (Round 3 tried one LDS atomic per (wave, bucket) by ballot instead of one per tile -- nearly
    // all tiles share two or three buckets: the serial ballot / shuffle / returning-atomic round
    // trips cost more than the same-address atomics they replace, scan stage 18.5 -> 22.2 us.)
    // (the thread's totals are fetched four at a time in both passes: one per iteration was a
    // chain of 2 x per round trips -- 46 us per pose for the 16 384 tiles of C5)
    const int iend = (i0 + per < NT) ? i0 + per : NT;
    for (int ib = i0; ib < iend; ib += 4) {
        uint32_t cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cc[u] = (ib + u < iend) ? totals[ib + u] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ib + u >= iend) continue;
            const uint32_t c = cc[u];
            s += c;
            const uint32_t k = c > cap ? (c + cap - 1) / cap : 1u;
            const uint32_t sz = (c + k - 1) / k;  // records per part
            atomicAdd(&bcount[sz ? 32 - __clz(sz) : 0], k);
            if (k > 1) slabs += k;
        }
    }
    uint32_t incl = s, incl_slab = slabs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64), v2 = __shfl_up(incl_slab, o, 64);
        if ((threadIdx.x & 63) >= o) {
            incl += v;
            incl_slab += v2;
        }
    }
    if ((threadIdx.x & 63) == 63) {
        wsum[threadIdx.x >> 6] = incl;
        wslab[threadIdx.x >> 6] = incl_slab;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t start = 0;
        for (int k = 32; k >= 0; --k) {
            bstart[k] = start;
            start += bcount[k];
        }
        *n_items = start;
        // n_items[1]: record assignment of k_tile_splat.  A cloud in random order (a count block
        // touches more than 1/8 of the bins) takes lane-adjacent records -- coalesced loads,
        // lanes land in unrelated voxels anyway; a spatially sorted one takes a contiguous run
        // per thread (lane-adjacent records would hit the same voxel and same-address LDS
        // atomics serialise).  88 vs 94 us for the random order at C3.
        n_items[1] = ((uint64_t)s_nzsum * 8 >= (uint64_t)nblk * (uint64_t)NT) ? 0u : 1u;
    }
    uint32_t wbase = 0, sbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) {
        wbase += wsum[w];
        sbase += wslab[w];
    }
    __syncthreads();
    uint32_t run = wbase + incl - s, slab_run = sbase + incl_slab - slabs;
    for (int ib = i0; ib < iend; ib += 4) {
        uint32_t cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cc[u] = (ib + u < iend) ? totals[ib + u] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = ib + u;
            if (i >= iend) continue;
            const uint32_t c = cc[u];
            tile_start[i] = run;
            const uint32_t k = c > cap ? (c + cap - 1) / cap : 1u;
            const uint32_t sz = (c + k - 1) / k;
            const int bucket = sz ? 32 - __clz(sz) : 0;
            tile_parts[i] = k;
            tile_slab[i] = slab_run;
            for (uint32_t part = 0; part < k; ++part) {
                WorkItem it;
                it.tile = (uint32_t)i;
                it.begin = run + part * sz;
                it.end = (it.begin + sz < run + c) ? it.begin + sz : run + c;
                if (it.begin > run + c) it.begin = run + c;
                it.part_nparts = part | (k << 16);
                items[atomicAdd(&bstart[bucket], 1u)] = it;
            }
            if (k > 1) {
                slab_run += k;
                split_list[atomicAdd(&s_nsplit, 1u)] = (uint32_t)i;
            }
            run += c;
        }
    }
    if (threadIdx.x == 1023) tile_start[NT] = wbase + incl;
    __syncthreads();
    if (threadIdx.x == 0) *n_split = s_nsplit;
}

// counts[nblk][NT] -> in place exclusive prefix down each column; totals[NT].
// Block = kScanTiles tiles x kScanGroups row groups (32 x 32: NT/32 blocks keep more CUs busy
// than the 64 x 16 split; rows of 32 tiles are still 128-byte segments).
// (Round 3 tried ONE kernel whose last block -- arrival ticket, agent-scope fences -- runs the
// tile scan: the scan stage went from 18.5 to 40 us at C3, 81 us with 16-tile blocks; the
// device-wide release / acquire of 64-128 blocks costs more than a launch.  Two kernels stay.)
#ifndef DPR_SCAN_TILES
#define DPR_SCAN_TILES 32
#endif
constexpr int kScanTiles = DPR_SCAN_TILES, kScanGroups = 1024 / DPR_SCAN_TILES;
__global__ __launch_bounds__(1024) void k_colscan(uint32_t* __restrict__ counts, int nblk, int NT,
                                                  uint32_t* __restrict__ totals,
                                                  uint32_t* __restrict__ zero_word = nullptr) {
    // (max |point_weight| bits: published by the scatter that follows with atomicMax)
    if (zero_word && blockIdx.x == 0 && threadIdx.x < 2) zero_word[threadIdx.x] = 0u;  // (max, ~min)
    // A thread owns `rows` consecutive rows of one column.  Its loads are issued kRowBatch at a
    // time (one row per iteration was a chain of 2 x rows dependent round trips: 9.2 us for the
    // 512 x 2048 table of C3, 16 rows per thread); when the rows fit one batch -- up to 512 count
    // blocks -- the values stay in registers between the two passes.
    constexpr int kRowBatch = 16;
    __shared__ uint32_t part[kScanGroups][kScanTiles];
    const int j = threadIdx.x % kScanTiles, g = threadIdx.x / kScanTiles;
    const int tile = blockIdx.x * kScanTiles + j;
    const int rows = (nblk + kScanGroups - 1) / kScanGroups;
    const int r0 = g * rows, r1 = (r0 + rows < nblk) ? r0 + rows : nblk;
    const bool one = rows <= kRowBatch;  // uniform
    const bool live = tile < NT;
    uint32_t v[kRowBatch];
    uint32_t s = 0;
    for (int rb = r0; rb < r1; rb += kRowBatch) {
#pragma unroll
        for (int k = 0; k < kRowBatch; ++k)
            v[k] = (live && rb + k < r1) ? counts[(size_t)(rb + k) * NT + tile] : 0u;
#pragma unroll
        for (int k = 0; k < kRowBatch; ++k) s += v[k];
    }
    part[g][j] = s;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kScanGroups; ++k) {
        const uint32_t pv = part[k][j];
        if (k < g) base += pv;
        total += pv;
    }
    if (!live) return;
    for (int rb = r0; rb < r1; rb += kRowBatch) {
        if (!one) {
#pragma unroll
            for (int k = 0; k < kRowBatch; ++k)
                v[k] = (rb + k < r1) ? counts[(size_t)(rb + k) * NT + tile] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kRowBatch; ++k) {
            if (rb + k < r1) counts[(size_t)(rb + k) * NT + tile] = base;
            base += v[k];
        }
    }
    if (g == 0) totals[tile] = total;
}

__global__ __launch_bounds__(1024) void k_tilescan(TileScanArgs ts) {
    tilescan_body(ts.totals, ts.NT, ts.cap, ts.tile_start, ts.items, ts.n_items, ts.tile_parts,
                  ts.tile_slab, ts.split_list, ts.n_split, ts.hdr, ts.hdr_out, ts.rot,
                  ts.rot_words, ts.trans, ts.trans_words, ts.nonzero_bins, ts.nblk);
}

}  // namespace dpr
#include "dpr_coarse.h"
namespace dpr {

// ------------------------------------------------------------------ K3: scatter
// WANT_IDX: the binning will feed a pullback (original indices needed).
// The next point is fetched before the current record is stored so that the wait for
// it never includes the scattered store (vmcnt retires in issue order).
template <typename T, int NI, int NO, bool HAS_PW, bool WANT_IDX>
__global__ __launch_bounds__(kBinThreads) void k_scatter(
    GridDesc<NO> gd, TileGeom<NO> tg, int64_t P, int64_t chunk, const T* __restrict__ points,
    const T* __restrict__ pw, const T* __restrict__ rot, const T* __restrict__ trans, int64_t b,
    const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ tile_start,
    Rec4<T>* __restrict__ rec, uint32_t* __restrict__ rec_idx, uint32_t* __restrict__ slot_of,
    T* __restrict__ ds_dpoints, T* __restrict__ ds_dpw, int zero_dropped,
    uint32_t* __restrict__ maxpw) {
    extern __shared__ uint32_t cursor[];
    const unsigned slice = xcd_slice(blockIdx.x, gridDim.x);
    const uint32_t* row = prefix + (size_t)slice * tg.NT;
    for (int i = threadIdx.x; i < tg.NT; i += kBinThreads) cursor[i] = tile_start[i] + row[i];
    __syncthreads();
    const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, nullptr, b);
    const int64_t lo = (int64_t)slice * chunk;
    const int64_t hi = (lo + chunk < P) ? lo + chunk : P;
    // Fixed memory-op pattern per iteration (one clamped prefetch, one record store, the
    // latter to the spare slot P when the point has no in-range voxel) so that the wait for
    // the prefetched point is a counted vmcnt that never covers the scattered store.
    if (lo >= hi) return;
    uint32_t wkey = 0;  // max / min |point_weight| of the binned points (wrange_*: k_tile_splat's fixed point)
    int64_t p = lo + threadIdx.x;
    T nxt[NI], nxt_w = T(1);
    {
        const int64_t pl = p < hi ? p : hi - 1;
        load_point<T, NI>(points, pl, nxt);
        if (HAS_PW) nxt_w = pw[pl];
    }
    // the first point: waited for here, not at the top of every iteration
#pragma unroll
    for (int j = 0; j < NI; ++j) pin_value(nxt[j]);
    if (HAS_PW) pin_value(nxt_w);
    drain_vmem();
    while (p < hi) {
        T pt[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) pt[j] = nxt[j];
        const T w = nxt_w;
        const int64_t pc = p;
        p += kBinThreads;
        {
            const int64_t pl = p < hi ? p : hi - 1;
            load_point<T, NI>(points, pl, nxt);
            if (HAS_PW) nxt_w = pw[pl];
        }
        int ref0[NO];
        T dlo[NO];
        bool valid = ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo);
        const int tl = valid ? primary_tile<NO>(ref0, tg) : -1;
        valid = tl >= 0;  // (a point of another slab is no point of this launch)
        uint32_t pos = (uint32_t)P;  // spare slot
        if (valid) pos = atomicAdd(&cursor[tl], 1u);
        if (HAS_PW && valid) wkey = wrange_merge(wkey, abs_key(w));
        Rec4<T> r;
#pragma unroll
        for (int j = 0; j < 3; ++j) r.v[j] = (j < NI) ? pt[(j < NI) ? j : 0] : T(0);
        r.v[3] = HAS_PW ? w : idx_to_slot((uint32_t)pc, T(0));
        rec[pos] = r;
        if (HAS_PW && WANT_IDX) rec_idx[pos] = (uint32_t)pc;
        if (WANT_IDX) slot_of[pc] = pos;  // coalesced; rejected points map to the spare slot
        if (zero_dropped && !valid) {
            // no in-range voxel: empty gradient (written once, by the first pose)
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[pc * NI + j] = T(0);
            if (ds_dpw) ds_dpw[pc] = T(0);
        }
    }
    if (HAS_PW) publish_max_abs(maxpw, wkey);
}

// ------------------------------------------------------------------ K3': write-combining scatter
// Same placement as k_scatter, but the records of a sub-chunk of S points are first ordered
// by tile in LDS, so that neighbouring lanes of the write-out store neighbouring slots:
// one 64-lane store instruction then covers a few contiguous runs instead of 64 unrelated
// 16-byte pieces (measured: 10 M scattered 16-byte stores cost ~75 us more than coalesced
// ones, profiles/r01_experiments.md).
//   LDS: cursor[NT * nb] (dynamic) | lhist[NT] (dynamic) | recs[S] | dest[S]
template <typename T, int NI, int NO, bool HAS_PW, int S, bool GROUP, bool W3>
__global__ __launch_bounds__(kWcThreads) void k_scatter_wc(
    GridDesc<NO> gd, TileGeom<NO> tg, int64_t P, int64_t chunk, const T* __restrict__ points,
    const T* __restrict__ pw, const T* __restrict__ rot, const T* __restrict__ trans, int64_t b0,
    int nb, const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ tile_start,
    RecT<T, W3>* __restrict__ rec, uint32_t* __restrict__ slot_of, T* __restrict__ ds_dpoints,
    T* __restrict__ ds_dpw, int zero_dropped, uint32_t* __restrict__ maxpw, int fused_scan,
    TileScanArgs ts) {
    // fused_scan: the tile scan (tile offsets for later kernels, work list, binning header) runs
    // as ONE EXTRA workgroup of this launch instead of a launch of its own between the column scan
    // and the scatter -- nothing in the scatter depends on its results: every scatter workgroup
    // scans the 2048-4096 tile totals itself in its prologue (~1 us, under its first point
    // loads).  One dependent launch and its gap less per binning (C3: scan stage 14 -> 7 us).
    const unsigned n_slices = fused_scan ? gridDim.x - 1u : gridDim.x;
    if (fused_scan && blockIdx.x == n_slices) {
        tilescan_body(ts.totals, ts.NT, ts.cap, ts.tile_start, ts.items, ts.n_items, ts.tile_parts,
                      ts.tile_slab, ts.split_list, ts.n_split, ts.hdr, ts.hdr_out, ts.rot,
                      ts.rot_words, ts.trans, ts.trans_words, ts.nonzero_bins, ts.nblk);
        return;
    }
    static_assert(!(W3 && HAS_PW), "compact records carry no point weight");
    constexpr int PPT = S / kWcThreads;  // points per thread per sub-chunk
    uint32_t wkey = 0;  // max / min |point_weight| of the binned points (wrange_*: k_tile_splat's fixed point)
    // Pose group (nb > 1): the S points of a sub-chunk stay in registers while the poses of the
    // group are binned one after the other, each into its own NT bins -- the runs that are
    // written out stay as long as in the single-pose case, the points are read once per group.
    if (!GROUP) nb = 1;  // compile-time trip count: the single-pose kernel keeps its registers
    const int NT = tg.NT, NTe = NT * nb;
    const uint32_t Pe = (uint32_t)(P * nb);  // spare slot (rejected point-poses)
    extern __shared__ uint32_t dyn[];
    uint32_t* cursor = dyn;        // [NTe]
    uint32_t* lhist = dyn + NTe;   // [NT]
    __shared__ RecT<T, W3> recs[S];
    __shared__ uint32_t dest[S];
    // the scan's per-wave sums live in dest[]: they are dead before phase c writes it (two barriers
    // in between), and dest[] is dead after the write-out of the previous round (barrier)
    uint32_t* const wsum = dest;
    static_assert(S >= kWcThreads / kWave, "wave sums fit");
    const unsigned slice = xcd_slice(blockIdx.x, n_slices);
    const uint32_t* row = prefix + (size_t)slice * NTe;
    constexpr int kMaxBpt = 4096 / kWcThreads;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (fused_scan) {
        // exclusive scan of the tile totals (= tile_start, which the extra workgroup is writing for
        // the kernels downstream at this very moment) + this slice's column prefix
        const int bpe = (NTe + kWcThreads - 1) / kWcThreads;  // <= 4
        const int e0 = threadIdx.x * bpe;
        uint32_t c[kMaxBpt], rw[kMaxBpt], sum = 0;
#pragma unroll
        for (int q = 0; q < kMaxBpt; ++q) {
            const int i = e0 + q;
            const bool in = q < bpe && i < NTe;
            c[q] = in ? ts.totals[i] : 0u;
            rw[q] = in ? row[i] : 0u;
            sum += c[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int wv = 0; wv < wave; ++wv) run += wsum[wv];
#pragma unroll
        for (int q = 0; q < kMaxBpt; ++q) {
            const int i = e0 + q;
            if (q < bpe && i < NTe) cursor[i] = run + rw[q];
            run += c[q];
        }
    } else {
        for (int i = threadIdx.x; i < NTe; i += kWcThreads) cursor[i] = tile_start[i] + row[i];
    }
    for (int i = threadIdx.x; i < NT; i += kWcThreads) lhist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)slice * chunk;
    const int64_t hi = (lo + chunk < P) ? lo + chunk : P;
    // bins owned by this thread in the scan / cursor update (NT <= 4096: at most 4)
    const int bpt = (NT + kWcThreads - 1) / kWcThreads;
    const int bin0 = threadIdx.x * bpt;
    if (lo >= hi) return;
    // Single pose: the next sub-chunk's points are requested while the current one goes through
    // its LDS phases.  Pose groups load at the top of the round instead: the second register set
    // pushed that kernel over 128 VGPRs (it spilled the prefetched points straight to scratch,
    // i.e. waited for them at once), and a group pays the latency once per nb poses anyway.
#ifndef DPR_GROUP_PREFETCH
#define DPR_GROUP_PREFETCH 0
#endif
#ifndef DPR_WC_PREFETCH
#define DPR_WC_PREFETCH 1
#endif
    constexpr bool kPrefetch = DPR_WC_PREFETCH && (!GROUP || DPR_GROUP_PREFETCH);
    // Everything per point is addressed as (sub-chunk base: uniform, 64-bit, in scalar registers) +
    // (index inside the sub-chunk: 32-bit): with 64-bit per-point indices the pose-group variants kept
    // a dozen hoisted addresses alive across the pose loop and spilled 10-24 VGPRs.
    auto load_sub = [&](int64_t sbase, T (&dp)[PPT][NI], T (&dw)[PPT]) {
        // points of the sub-chunk at `sbase` (clamped to the slice: past its end the last point again)
        const int64_t left = hi - sbase;  // (may be <= 0 for the prefetch past the last round)
        const uint32_t n = left >= S ? (uint32_t)S : (left > 0 ? (uint32_t)left : 0u);
        const T* const pb = points + (n ? sbase : hi - 1) * NI;
        const T* const wb = HAS_PW ? pw + (n ? sbase : hi - 1) : nullptr;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const uint32_t lp = threadIdx.x + (uint32_t)k * kWcThreads;
            const uint32_t ll = n ? (lp < n ? lp : n - 1) : 0u;
            load_point<T, NI>(pb, (int64_t)ll, dp[k]);
            dw[k] = HAS_PW ? wb[ll] : T(1);
        }
    };
    T nxt_pt[kPrefetch ? PPT : 1][NI], nxt_w[kPrefetch ? PPT : 1];
    if constexpr (kPrefetch) load_sub(lo, nxt_pt, nxt_w);
    for (int64_t base = lo; base < hi; base += S) {
        const uint32_t nloc = (uint32_t)((hi - base < S) ? hi - base : S);  // points of this sub-chunk
        T pt[PPT][NI], w[PPT];
        if constexpr (kPrefetch) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
#pragma unroll
                for (int j = 0; j < NI; ++j) pt[k][j] = nxt_pt[k][j];
                w[k] = nxt_w[k];
            }
            load_sub(base + S, nxt_pt, nxt_w);
        } else {
            load_sub(base, pt, w);
        }
        for (int jp = 0; jp < (GROUP ? nb : 1); ++jp) {
            const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, nullptr, b0 + jp);
            uint32_t* cur = cursor + jp * NT;
            uint32_t* const slot_b = slot_of ? slot_of + (size_t)jp * P + base : (uint32_t*)nullptr;
            // a. classify, rank inside (sub-chunk, tile)
            int tile[PPT];
            uint32_t lrank[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const uint32_t lp = threadIdx.x + (uint32_t)k * kWcThreads;
                int ref0[NO];
                T dlo[NO];
                bool valid = ref_and_deltas<T, NI, NO>(pt[k], ps, gd, ref0, dlo) && lp < nloc;
                tile[k] = valid ? primary_tile<NO>(ref0, tg) : -1;
                valid = tile[k] >= 0;
                lrank[k] = 0;
                if (valid) lrank[k] = atomicAdd(&lhist[tile[k]], 1u);
                if (HAS_PW && valid) wkey = wrange_merge(wkey, abs_key(w[k]));
            }
            lds_barrier();
            // b. exclusive scan of lhist (in place)
            uint32_t cnt_sum = 0;
            uint32_t cnt[kMaxBpt];
#pragma unroll
            for (int q = 0; q < kMaxBpt; ++q) {
                const int i = bin0 + q;
                cnt[q] = (q < bpt && i < NT) ? lhist[i] : 0u;
                cnt_sum += cnt[q];
            }
            uint32_t incl = cnt_sum;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o, kWave);
                if (lane >= o) incl += v;
            }
            if (lane == kWave - 1) wsum[wave] = incl;
            lds_barrier();
            uint32_t run = incl - cnt_sum;
            for (int wv = 0; wv < wave; ++wv) run += wsum[wv];
            uint32_t n_valid = 0;
#pragma unroll
            for (int wv = 0; wv < kWcThreads / kWave; ++wv) n_valid += wsum[wv];
#pragma unroll
            for (int q = 0; q < kMaxBpt; ++q) {
                const int i = bin0 + q;
                if (q < bpt && i < NT) lhist[i] = run;  // exclusive offset inside the sub-chunk
                run += cnt[q];
            }
            lds_barrier();
            // c. place into LDS in tile order; remember the global destination
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const uint32_t lp = threadIdx.x + (uint32_t)k * kWcThreads;
                if (tile[k] >= 0) {
                    const uint32_t sidx = lhist[tile[k]] + lrank[k];
                    const uint32_t d = cur[tile[k]] + lrank[k];
                    RecT<T, W3> r;
#pragma unroll
                    for (int j = 0; j < 3; ++j) r.v[j] = (j < NI) ? pt[k][(j < NI) ? j : 0] : T(0);
                    if constexpr (!W3) r.v[3] = HAS_PW ? w[k] : idx_to_slot((uint32_t)base + lp, T(0));
                    recs[sidx] = r;
                    dest[sidx] = d;
                    if (slot_b) __builtin_nontemporal_store(d, &slot_b[lp]);
                } else if (lp < nloc) {
                    if (slot_b) __builtin_nontemporal_store(Pe, &slot_b[lp]);  // spare slot
                    if (zero_dropped) {
                        T* const dp_b = ds_dpoints + base * NI;
#pragma unroll
                        for (int j = 0; j < NI; ++j) dp_b[lp * NI + j] = T(0);
                        if (ds_dpw) (ds_dpw + base)[lp] = T(0);
                    }
                }
            }
            lds_barrier();
            // d. write-out in LDS (= tile) order; e. advance cursors, clear the histogram
            for (uint32_t i = threadIdx.x; i < n_valid; i += kWcThreads) rec[dest[i]] = recs[i];
            // advance the cursors by the owned bins' counts (kept in registers since the scan)
            // and clear the histogram; the next round's atomics start after the barrier
#pragma unroll
            for (int q = 0; q < kMaxBpt; ++q) {
                const int i = bin0 + q;
                if (q < bpt && i < NT) {
                    cur[i] += cnt[q];
                    lhist[i] = 0;
                }
            }
            lds_barrier();
        }
    }
    if (HAS_PW) publish_max_abs(maxpw, wkey);
}

// ------------------------------------------------------------------ local binning: K1
// One block per sub-chunk of S consecutive points (see "LOCAL BINNING" above), ALL poses of a
// batch in one launch: the sub-chunk's points are read once and stay in registers while they are
// binned pose after pose, each pose into its own copy of the per-pose workspace (copies are
// `pose_stride` bytes apart).
//   LDS: lhist[NT] (dynamic) | recs[S] | touched[kTouchCap]
// A coherent sub-chunk touches a handful of tiles (13 of 2048 at C3, ~50 of 16 384 at C5 after the
// coarse cell sort), so nothing here is proportional to NT: the wave that first adds to a bin of
// the histogram appends the tile to `touched`, and the offsets, the descriptors and the clean-up of
// the round walk that list.  A sub-chunk that touches more than kTouchCap tiles (incoherent
// input) takes the full scan over all bins instead -- slow, still correct.
// Outputs per pose: records rec[sub * S ...] in tile order (rejected points leave holes at the
// end of the sub-chunk's slab), slot_of[p] for the pullback, the sub-chunk's descriptors in ITS
// OWN slots desc[sub * S ...] with their number in blk_ndesc[sub] (no global cursor), and the
// per-tile totals tile_ndesc / tile_npts (fire-and-forget global atomics, two per descriptor).
struct LocalBinArgs {
    char* ws;            // pose copy 0 of the per-pose workspace
    size_t pose_stride;  // distance between the copies
    size_t off_rec, off_slot, off_desc, off_bdesc, off_ltot;  // ltot: ndesc[NT] | npts[NT] | max|pw| | ~min|pw|
};
constexpr int kTouchCap = 1024;
// (fp32: 74 KB of LDS, two workgroups per CU when the kernel stays within 64 VGPRs; fp64 grids
// with their larger histograms run one workgroup per CU anyway)
// ONE: a single pose (nb == 1) -- the points die after the placement instead of living across a pose
// loop, which is what lets the fp32 kernel fit 64 VGPRs.
// TH / STAGE: 1024 threads that order the records in LDS and write them out as one coalesced run
// (fp32: two 16-byte records share a 32-byte sector), or -- fp64 batches -- 512 threads that store
// their 32-byte records (whole sectors) straight to their place in the sub-chunk's 64 KB slab:
// without the 64 KB staging buffer next to the 64 KB histogram of a 16 384-tile grid TWO
// workgroups fit a CU, and the write bursts of one overlap the ranking phases of the other.
template <typename T, int NI, int NO, bool HAS_PW, int S, bool W3, bool ONE, int TH, bool STAGE>
__global__ __launch_bounds__(TH, ((sizeof(T) == 4 && ONE) ? 8 : 4)) void k_bin_local(
    GridDesc<NO> gd, TileGeom<NO> tg, int64_t P, const T* __restrict__ points,
    const T* __restrict__ pw, const T* __restrict__ rot, const T* __restrict__ trans, int64_t b0,
    int nb, LocalBinArgs la, int want_slot, uint32_t spare_slot, T* __restrict__ ds_dpoints,
    T* __restrict__ ds_dpw, int zero_dropped) {
    static_assert(!(W3 && HAS_PW), "compact records carry no point weight");
    constexpr int PPT = S / TH;
    if (ONE) nb = 1;
    const int NT = tg.NT;
    extern __shared__ uint32_t lhist[];  // [NT]
    __shared__ RecT<T, W3> recs[STAGE ? S : 1];
    __shared__ uint16_t touched[kTouchCap];
    __shared__ uint32_t wsum[TH / kWave];
    __shared__ uint32_t s_ntouch, s_nvalid;
    for (int i = threadIdx.x; i < NT; i += TH) lhist[i] = 0;
    if (threadIdx.x == 0) s_ntouch = 0;
    // Everything per point is addressed as (block base: uniform, 64-bit) + (index inside the
    // sub-chunk: 32-bit): with 64-bit per-point addresses the compiler hoists a dozen of them out
    // of the pose loop and spills them (128 VGPRs + scratch instead of < 64).
    const int64_t base = (int64_t)blockIdx.x * S;
    const uint32_t nloc = (uint32_t)((P - base < S) ? P - base : S);  // points of this sub-chunk
    const T* const pts_blk = points + base * NI;
    const T* const pw_blk = HAS_PW ? pw + base : nullptr;
    T pt[PPT][NI], w[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t lp = threadIdx.x + (uint32_t)k * TH;
        const uint32_t ll = lp < nloc ? lp : nloc - 1;
        if (nloc) {  // (uniform; an empty cloud still runs the pipeline: the tiles get their background)
            load_point<T, NI>(pts_blk, (int64_t)ll, pt[k]);
            w[k] = HAS_PW ? pw_blk[ll] : T(1);
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) pt[k][j] = T(0);
            w[k] = T(1);
        }
    }
    if constexpr (HAS_PW) {
        // max / ~min |point_weight| for the fixed-point splat, over every LOADED point, valid under a
        // given pose or not: published once for all poses of the local batch (a point outside the grid
        // under pose 0 may be inside under pose j > 0) -- and at once, so that nothing of it lives
        // across the pose loop
        uint32_t wkey = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (threadIdx.x + (uint32_t)k * TH < nloc) wkey = wrange_merge(wkey, abs_key(w[k]));
        publish_max_abs((uint32_t*)(la.ws + la.off_ltot) + 2 * NT, wkey);
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll 1
    for (int jp = 0; jp < nb; ++jp) {
        const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, nullptr, b0 + jp);
        char* const wsp = la.ws + (size_t)jp * la.pose_stride;
        RecT<T, W3>* const rec_blk = (RecT<T, W3>*)(wsp + la.off_rec) + base;
        uint32_t* const slot_blk = want_slot ? (uint32_t*)(wsp + la.off_slot) + base : (uint32_t*)nullptr;
        RunDesc* const desc = (RunDesc*)(wsp + la.off_desc) + (size_t)blockIdx.x * S;
        uint32_t* const ltot = (uint32_t*)(wsp + la.off_ltot);
        // a. classify; rank inside (sub-chunk, tile).  Neighbouring lanes of a coherent cloud fall
        // into the same tile, and same-address returning LDS atomics serialise: each wave first
        // ranks the lanes that share the tile of its first unranked lane (ballot), one atomic per
        // distinct tile and wave.
        int tile[PPT];
        uint32_t lrank[PPT];
        uint32_t left = 0;  // bit k: this lane's point k still has to be ranked
        // (a fully sorted cloud: a wave's 64 points share one or two tiles -- a few ballot rounds
        // rank them all.  A cell-sorted cloud, random inside its cell: nearly every lane has its own
        // tile, and each ballot round is a dependent LDS round trip for one lane's worth of
        // progress -- those lanes go to the per-lane atomics at once.)
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const uint32_t lp = threadIdx.x + (uint32_t)k * TH;
            int ref0[NO];
            T dlo[NO];
            bool valid = ref_and_deltas<T, NI, NO>(pt[k], ps, gd, ref0, dlo) && lp < nloc;
            tile[k] = valid ? primary_tile<NO>(ref0, tg) : -1;
            valid = tile[k] >= 0;
            lrank[k] = 0;
            unsigned long long todo = __ballot(valid);
            int rounds = 0;
            while (todo && rounds < 8) {
                const int leader = __ffsll((long long)todo) - 1;
                const int t = __shfl(tile[k], leader, kWave);
                const unsigned long long same = __ballot(tile[k] == t) & todo;
                if (__popcll(same) < 4) break;  // (uniform) not worth a round of its own
                uint32_t r0 = 0;
                if (lane == leader) {
                    r0 = atomicAdd(&lhist[t], (uint32_t)__popcll(same));
                    if (r0 == 0) {  // first into this bin: the tile joins the list
                        const uint32_t pos = atomicAdd(&s_ntouch, 1u);
                        if (pos < (uint32_t)kTouchCap) touched[pos] = (uint16_t)t;
                    }
                }
                r0 = __shfl(r0, leader, kWave);
                if ((same >> lane) & 1ull) lrank[k] = r0 + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                todo &= ~same;
                ++rounds;
            }
            if ((todo >> lane) & 1ull) left |= 1u << k;
        }
        // the rest one by one: all returning atomics of a thread are in flight together
        if (__ballot(left != 0)) {  // (uniform)
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if ((left >> k) & 1u) lrank[k] = atomicAdd(&lhist[tile[k]], 1u);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (((left >> k) & 1u) && lrank[k] == 0) {
                    const uint32_t pos = atomicAdd(&s_ntouch, 1u);
                    if (pos < (uint32_t)kTouchCap) touched[pos] = (uint16_t)tile[k];
                }
            }
        }
        lds_barrier();
        // b. counts -> exclusive offsets inside the sub-chunk (in place), descriptors, tile totals
        const uint32_t n_t = s_ntouch;  // distinct tiles of this sub-chunk under this pose
        const bool listed = n_t <= (uint32_t)kTouchCap;  // (uniform)
        if (listed) {
            if (wave == 0) {
                uint32_t carry = 0;
                for (uint32_t i0 = 0; i0 < n_t; i0 += kWave) {
                    const uint32_t i = i0 + lane;
                    const uint32_t t = i < n_t ? touched[i] : 0u;
                    const uint32_t c = i < n_t ? lhist[t] : 0u;
                    uint32_t incl = c;
#pragma unroll
                    for (int o = 1; o < kWave; o <<= 1) {
                        const uint32_t v = __shfl_up(incl, o, kWave);
                        if (lane >= o) incl += v;
                    }
                    const uint32_t off = carry + incl - c;
                    if (i < n_t) {
                        lhist[t] = off;
                        RunDesc d;
                        d.start = (uint32_t)base + off;
                        d.tile_count = t | (c << 15);
                        desc[i] = d;
                        atomicAdd(&ltot[t], 1u);
                        atomicAdd(&ltot[NT + t], c);
                    }
                    carry += __shfl(incl, kWave - 1, kWave);
                }
                if (lane == 0) {
                    s_nvalid = carry;
                    ((uint32_t*)(wsp + la.off_bdesc))[blockIdx.x] = n_t;
                }
            }
        } else {
            // every bin: (count, non-empty) packed as count + (1 << 16) per non-empty bin (<= 4096 each)
            const int bpt = (NT + TH - 1) / TH;
            const int bin0 = threadIdx.x * bpt;
            uint32_t packed = 0;
            for (int q = 0; q < bpt; ++q) {
                const int i = bin0 + q;
                const uint32_t c = i < NT ? lhist[i] : 0u;
                packed += c + (c ? (1u << 16) : 0u);
            }
            uint32_t incl = packed;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o, kWave);
                if (lane >= o) incl += v;
            }
            if (lane == kWave - 1) wsum[wave] = incl;
            lds_barrier();
            uint32_t run = incl - packed;
            for (int wv = 0; wv < wave; ++wv) run += wsum[wv];
            uint32_t total = 0;
#pragma unroll
            for (int wv = 0; wv < TH / kWave; ++wv) total += wsum[wv];
            for (int q = 0; q < bpt; ++q) {
                const int i = bin0 + q;
                if (i >= NT) break;
                const uint32_t c = lhist[i];
                if (c) {
                    RunDesc d;
                    d.start = (uint32_t)base + (run & 0xffffu);
                    d.tile_count = (uint32_t)i | (c << 15);
                    desc[run >> 16] = d;
                    atomicAdd(&ltot[i], 1u);
                    atomicAdd(&ltot[NT + i], c);
                }
                lhist[i] = run & 0xffffu;
                run += c + (c ? (1u << 16) : 0u);
            }
            if (threadIdx.x == 0) {
                s_nvalid = total & 0xffffu;
                ((uint32_t*)(wsp + la.off_bdesc))[blockIdx.x] = total >> 16;
            }
        }
        lds_barrier();
        // c. place into LDS in tile order
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const uint32_t lp = threadIdx.x + (uint32_t)k * TH;
            if (tile[k] >= 0) {
                const uint32_t sidx = lhist[tile[k]] + lrank[k];
                RecT<T, W3> r;
#pragma unroll
                for (int j = 0; j < 3; ++j) r.v[j] = (j < NI) ? pt[k][(j < NI) ? j : 0] : T(0);
                if constexpr (!W3) r.v[3] = HAS_PW ? w[k] : idx_to_slot((uint32_t)base + lp, T(0));
                if constexpr (STAGE) recs[sidx] = r;
                else rec_blk[sidx] = r;  // (a whole 32-byte sector; the slab's lines fill up within the round)
                if (slot_blk) __builtin_nontemporal_store((uint32_t)base + sidx, &slot_blk[lp]);
            } else if (lp < nloc) {
                if (slot_blk) __builtin_nontemporal_store(spare_slot, &slot_blk[lp]);
                if (zero_dropped && jp == 0) {
                    T* const dp_blk = ds_dpoints + base * NI;
#pragma unroll
                    for (int j = 0; j < NI; ++j) dp_blk[lp * NI + j] = T(0);
                    if (ds_dpw) (ds_dpw + base)[lp] = T(0);
                }
            }
        }
        lds_barrier();
        // d. write-out: one contiguous, coalesced run; clean the histogram for the next pose
        const uint32_t n_valid = s_nvalid;
        if constexpr (STAGE)
            for (uint32_t i = threadIdx.x; i < n_valid; i += TH) rec_blk[i] = recs[i];
        (void)n_valid;
        if (listed) {
            for (uint32_t i = threadIdx.x; i < n_t; i += TH) lhist[touched[i]] = 0;
        } else {
            for (int i = threadIdx.x; i < NT; i += TH) lhist[i] = 0;
        }
        if (threadIdx.x == 0) s_ntouch = 0;
        lds_barrier();
    }
}

// local binning: K2 -- tile totals -> descriptor offsets, work list (a work item is a range of
// a tile's descriptors holding about `cap` records at most), heaviest first.  One block per pose
// of the batch (blockIdx.x = pose copy).  Also clears the cursors K3 uses.
struct RunScanArgs {
    char* ws;
    size_t pose_stride;
    size_t off_ltot, off_dstart, off_dcursor, off_items, off_nitems, off_tparts, off_tslab, off_split,
        off_hdr;
    int NT;
    uint32_t cap;
    int max_items;
    int clear_cursors;  // 0: the cursors were cleared with the totals (fused with the placement)
    BinHeader hdr;
    const uint32_t* rot;  // pose b0 (words)
    int rot_words;
    const uint32_t* trans;
    int trans_words;
};
__device__ __forceinline__ void runscan_body(const RunScanArgs& a, unsigned pose) {
    char* const wsp = a.ws + (size_t)pose * a.pose_stride;
    const uint32_t* __restrict__ tile_ndesc = (const uint32_t*)(wsp + a.off_ltot);
    const uint32_t* __restrict__ tile_npts = tile_ndesc + a.NT;
    uint32_t* __restrict__ tile_dstart = (uint32_t*)(wsp + a.off_dstart);
    uint32_t* __restrict__ tile_cursor = (uint32_t*)(wsp + a.off_dcursor);
    WorkItem* __restrict__ items = (WorkItem*)(wsp + a.off_items);
    uint32_t* __restrict__ n_items = (uint32_t*)(wsp + a.off_nitems);
    uint32_t* __restrict__ tile_parts = (uint32_t*)(wsp + a.off_tparts);
    uint32_t* __restrict__ tile_slab = (uint32_t*)(wsp + a.off_tslab);
    uint32_t* __restrict__ n_split = (uint32_t*)(wsp + a.off_split);
    uint32_t* __restrict__ split_list = n_split + 1;
    BinHeader* __restrict__ hdr_out = (BinHeader*)(wsp + a.off_hdr);
    const BinHeader& hdr = a.hdr;
    const int NT = a.NT, max_items = a.max_items;
    const uint32_t cap = a.cap;
    const uint32_t* rot = a.rot + (size_t)pose * a.rot_words;
    const uint32_t* trans = a.trans + (size_t)pose * a.trans_words;
    const int rot_words = a.rot_words, trans_words = a.trans_words;
    if (threadIdx.x >= 1024 - 64) {  // binning header, as in the tile scan
        const int i = threadIdx.x - (1024 - 64);
        uint32_t* pose = (uint32_t*)hdr_out->pose;
        if (i < rot_words) pose[i] = rot[i];
        else if (i < rot_words + trans_words) pose[i] = trans[i - rot_words];
        if (i == 63) {
            hdr_out->magic = hdr.magic;
            hdr_out->state = hdr.state;
            hdr_out->elem = hdr.elem;
            hdr_out->n_in = hdr.n_in;
            hdr_out->n_out = hdr.n_out;
            hdr_out->has_pw = hdr.has_pw;
            hdr_out->P = hdr.P;
            hdr_out->grid[0] = hdr.grid[0];
            hdr_out->grid[1] = hdr.grid[1];
            hdr_out->grid[2] = hdr.grid[2];
            hdr_out->verdict = 0;
            hdr_out->points = hdr.points;
            hdr_out->pw = hdr.pw;
            hdr_out->layout = hdr.layout;
        }
    }
    __shared__ uint32_t wsum[16], wslab[16];
    __shared__ uint32_t s_nsplit, s_nitems;
    __shared__ uint32_t bcount[33], bstart[33];
    if (threadIdx.x == 0) s_nsplit = 0;
    if (threadIdx.x < 33) bcount[threadIdx.x] = 0;
    __syncthreads();
    const int per = (NT + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    // parts of a heavy tile are ranges of its descriptors (a run is never cut)
    auto parts_of = [&](uint32_t c, uint32_t nd) {
        uint32_t k = c > cap ? (c + cap - 1) / cap : 1u;
        if (k > nd && nd > 0) k = nd;
        return k ? k : 1u;
    };
    uint32_t s = 0, slabs = 0;
    const int iend = (i0 + per < NT) ? i0 + per : NT;
    for (int ib = i0; ib < iend; ib += 4) {  // four tiles' totals in flight (see tilescan_body)
        uint32_t cc[4], dd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cc[u] = (ib + u < iend) ? tile_npts[ib + u] : 0u;
            dd[u] = (ib + u < iend) ? tile_ndesc[ib + u] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ib + u >= iend) continue;
            const uint32_t c = cc[u], nd = dd[u];
            s += nd;
            const uint32_t k = parts_of(c, nd);
            const uint32_t sz = (c + k - 1) / k;  // records per part (estimate)
            atomicAdd(&bcount[sz ? 32 - __clz(sz) : 0], k);
            if (k > 1) slabs += k;
        }
    }
    uint32_t incl = s, incl_slab = slabs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64), v2 = __shfl_up(incl_slab, o, 64);
        if ((threadIdx.x & 63) >= o) {
            incl += v;
            incl_slab += v2;
        }
    }
    if ((threadIdx.x & 63) == 63) {
        wsum[threadIdx.x >> 6] = incl;
        wslab[threadIdx.x >> 6] = incl_slab;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t start = 0;
        for (int k = 32; k >= 0; --k) {
            bstart[k] = start;
            start += bcount[k];
        }
        s_nitems = start < (uint32_t)max_items ? start : (uint32_t)max_items;
        *n_items = s_nitems;
    }
    uint32_t wbase = 0, sbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) {
        wbase += wsum[w];
        sbase += wslab[w];
    }
    __syncthreads();
    uint32_t run = wbase + incl - s, slab_run = sbase + incl_slab - slabs;
    for (int ib = i0; ib < iend; ib += 4) {
        uint32_t cc[4], dd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cc[u] = (ib + u < iend) ? tile_npts[ib + u] : 0u;
            dd[u] = (ib + u < iend) ? tile_ndesc[ib + u] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = ib + u;
            if (i >= iend) continue;
            const uint32_t c = cc[u], nd = dd[u];
            tile_dstart[i] = run;
            if (a.clear_cursors) tile_cursor[i] = 0;
            const uint32_t k = parts_of(c, nd);
            const uint32_t sz = (c + k - 1) / k;
            const uint32_t dsz = (nd + k - 1) / k;  // descriptors per part
            const int bucket = sz ? 32 - __clz(sz) : 0;
            tile_parts[i] = k;
            tile_slab[i] = slab_run;
            for (uint32_t part = 0; part < k; ++part) {
                WorkItem it;
                it.tile = (uint32_t)i;
                it.begin = run + part * dsz;
                it.end = (it.begin + dsz < run + nd) ? it.begin + dsz : run + nd;
                if (it.begin > run + nd) it.begin = run + nd;
                it.part_nparts = part | (k << 16);
                const uint32_t pos = atomicAdd(&bstart[bucket], 1u);
                if (pos < (uint32_t)max_items) items[pos] = it;
            }
            if (k > 1) {
                slab_run += k;
                split_list[atomicAdd(&s_nsplit, 1u)] = (uint32_t)i;
            }
            run += nd;
        }
    }
    if (threadIdx.x == 1023) tile_dstart[NT] = wbase + incl;
    __syncthreads();
    if (threadIdx.x == 0) *n_split = s_nsplit;
}

__global__ __launch_bounds__(1024) void k_runscan(RunScanArgs a) { runscan_body(a, blockIdx.x); }

// local binning: K3 -- descriptors into tile order (any order inside a tile).  A wave per
// sub-chunk slot; blockIdx.y = pose copy.
// FUSED (grids of up to 4096 tiles): the run scan rides in this launch as ONE EXTRA workgroup per
// pose (blockIdx.x == gridDim.x - 1) -- what the placement needs from it, the exclusive scan of
// the per-tile descriptor counts, every workgroup computes itself in its prologue (as the
// write-combining scatter does with the tile scan): one dependent launch and its gap less.
// The cursors are then the caller's to clear (they sit behind the totals in `ltot`).
__global__ __launch_bounds__(1024) void k_place_desc(char* ws, size_t pose_stride, size_t off_desc,
                                                     size_t off_bdesc, size_t off_dstart,
                                                     size_t off_dcursor, size_t off_sdesc,
                                                     int64_t nsub, int S, int fused, RunScanArgs ra) {
    __shared__ uint32_t s_dstart[4096];
    __shared__ uint32_t wsum[16];
    if (fused && blockIdx.x == gridDim.x - 1) {
        runscan_body(ra, blockIdx.y);
        return;
    }
    char* const wsp = ws + (size_t)blockIdx.y * pose_stride;
    const RunDesc* __restrict__ desc = (const RunDesc*)(wsp + off_desc);
    const uint32_t* __restrict__ blk_ndesc = (const uint32_t*)(wsp + off_bdesc);
    const uint32_t* __restrict__ tile_dstart = (const uint32_t*)(wsp + off_dstart);
    uint32_t* __restrict__ tile_cursor = (uint32_t*)(wsp + off_dcursor);
    RunDesc* __restrict__ sorted = (RunDesc*)(wsp + off_sdesc);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (fused) {
        // exclusive scan of tile_ndesc[NT] (NT <= 4096) into LDS
        const uint32_t* __restrict__ tile_ndesc = (const uint32_t*)(wsp + ra.off_ltot);
        const int NT = ra.NT;
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x * 4 + q;
            c[q] = i < NT ? tile_ndesc[i] : 0u;
            sum += c[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int wv = 0; wv < wave; ++wv) run += wsum[wv];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x * 4 + q;
            if (i < NT) s_dstart[i] = run;
            run += c[q];
        }
        __syncthreads();
    }
    const unsigned nblocks = fused ? gridDim.x - 1 : gridDim.x;
    const int64_t wave0 = (int64_t)blockIdx.x * (1024 / kWave) + wave;
    const int64_t nwaves = (int64_t)nblocks * (1024 / kWave);
    for (int64_t sub = wave0; sub < nsub; sub += nwaves) {
        const uint32_t n = blk_ndesc[sub] < (uint32_t)S ? blk_ndesc[sub] : (uint32_t)S;
        for (uint32_t i = lane; i < n; i += kWave) {
            const RunDesc d = desc[(size_t)sub * S + i];
            const uint32_t base = fused ? s_dstart[d.tile()] : tile_dstart[d.tile()];
            sorted[base + atomicAdd(&tile_cursor[d.tile()], 1u)] = d;
        }
    }
}

// One record into the LDS tile of a forward tile kernel (k_tile_splat, k_tile_splat_runs).
template <bool FIX, typename T, int NI, int NO, bool HAS_PW, typename R>
__device__ __forceinline__ void splat_record(const R& rc, bool active, const Pose<T, NI, NO>& ps,
                                             const GridDesc<NO>& gd, const int (&x0)[NO],
                                             double* __restrict__ acc, const FixScale& fs) {
    T pt[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) pt[j] = rc.v[j];
    const T w = HAS_PW ? ps.ow * rc.v[HAS_PW ? 3 : 0] : ps.ow * T(1);  // src/raster.jl:52
    int ref0[NO];
    T dlo[NO];
    ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo);  // in range by construction
    // Individual drop of out-of-range neighbours (src/raster.jl:62) without branches:
    // an upper neighbour beyond the grid lands in an LDS cell that is never flushed;
    // a lower neighbour at -1 (ref0 == -1) is redirected to cell 0 with weight 0.
    int lb[NO];
    bool low_ok[NO];
#pragma unroll
    for (int d = 0; d < NO; ++d) {
        lb[d] = ref0[d] - x0[d];
        // records of this tile have lb in [-1, T-1]; the clamp only matters if the
        // caller breaks the REUSE_BINNING contract (stale workspace): LDS indices
        // stay legal
        lb[d] = lb[d] < -1 ? -1 : (lb[d] > TileDims<NO>::T[d] - 1 ? TileDims<NO>::T[d] - 1 : lb[d]);
        low_ok[d] = lb[d] >= 0;
    }
    bool interior = true;
#pragma unroll
    for (int d = 0; d < NO; ++d) interior = interior && low_ok[d];
#if defined(DPR_ABL) && DPR_ABL == 1  // ablation: all the arithmetic, no LDS atomics (timing only, wrong results)
    if (active) {
        unsigned long long sink = 0;
#pragma unroll
        for (int s = 0; s < (1 << NO); ++s) sink ^= fix_bits((float)voxel_weight<T, NO>(dlo, s, w), fs) + lds_index<NO>(lb);
        if (sink == 0x123456789abcull) acc[threadIdx.x] = 1.0;
    }
    return;
#elif defined(DPR_ABL) && DPR_ABL == 7  // ablation: records loaded and looked at, NO arithmetic, NO LDS atomics
    if (active && __float_as_uint((float)pt[0]) == 0x12345678u) acc[threadIdx.x] = 1.0;
    return;
#elif defined(DPR_ABL) && DPR_ABL == 2  // ablation: the LDS atomics only, at pseudo-random cells
    if (active) {
        uint32_t h = __float_as_uint((float)pt[0]) * 2654435761u;
        h ^= h >> 15;
        double* base = &acc[h % (uint32_t)(tile_voxels_halo<NO>() - nbr_lds_offset<NO>((1 << NO) - 1) - 1)];
#pragma unroll
        for (int s = 0; s < (1 << NO); ++s) atomicAdd((unsigned long long*)(base + nbr_lds_offset<NO>(s)), 12345ull);
    }
    return;
#endif
    if (active && interior) {
        // common case: one base address, the 2^N neighbours are compile-time offsets
        // (they fold into the ds_add offset field)
        double* base = &acc[lds_index<NO>(lb)];
#pragma unroll
        for (int s = 0; s < (1 << NO); ++s)
            cell_add<FIX, T>(base + nbr_lds_offset<NO>(s), voxel_weight<T, NO>(dlo, s, w), fs);
    } else if (active) {  // a lower neighbour at -1: only at the low faces of the grid
#pragma unroll
        for (int s = 0; s < (1 << NO); ++s) {
            int l[NO];
            bool ok = true;
#pragma unroll
            for (int d = 0; d < NO; ++d) {
                const int sd = (s >> d) & 1;
                ok = ok && (sd || low_ok[d]);
                l[d] = (sd || low_ok[d]) ? lb[d] + sd : 0;
            }
            const T v = voxel_weight<T, NO>(dlo, s, w);
            cell_add<FIX, T>(&acc[lds_index<NO>(l)], ok ? v : T(0), fs);
        }
    }
}

// Flush of a forward tile kernel's LDS tile, one row (TX + 1 cells along x) per wave pass: the row's y / z
// coordinates, bounds and base offsets are wave-uniform, a lane only adds its x.  Owned rows leave as
// out = background + acc (plain non-temporal stores of TX contiguous values); the rows of the upper y / z
// halo and the x == TX column go to the compact per-tile halo buffer (always fully written, zeros
// included).  kFB rows per wave are in flight together (round 6): one row at a time was a chain of ~19
// dependent LDS round trips per wave, and the tile kernel WITHOUT its record loop took 38 of its 71 us at C3
// (ablation builds, profiles/r06_experiments.md).
template <typename T, int NO>
__device__ __forceinline__ void flush_tile(const double* __restrict__ acc, const FixScale& fs, double bgv,
                                           const GridDesc<NO>& gd, const int (&x0)[NO], bool ghost_tile,
                                           T* __restrict__ o, T* __restrict__ hb) {
    constexpr int NVH = tile_voxels_halo<NO>();
    constexpr int TX = TileDims<NO>::T[0], TY = TileDims<NO>::T[1];
    constexpr int TZ = (NO == 3) ? TileDims<NO>::T[NO - 1] : 0;
    constexpr int ROWS = NVH / (TX + 1);   // (TY + 1) [* (TZ + 1)]
    constexpr int RPW = kWave / TX;        // rows per wave pass (1 for TX = 64, 2 for TX = 32)
    static_assert(kWave % TX == 0, "tile rows must divide the wavefront");
    constexpr int NW = kSplatThreads / kWave;
#ifndef DPR_FLUSH_BATCH
#define DPR_FLUSH_BATCH 4
#endif
    constexpr int kFB = DPR_FLUSH_BATCH;
    const int lane = threadIdx.x & (kWave - 1);
    const int x = lane % TX;
    const bool x_ok = x0[0] + x < gd.n[0];
    for (int row0 = (threadIdx.x / kWave) * RPW; row0 < ROWS; row0 += NW * RPW * kFB) {
        double raw[kFB];
        int rows[kFB];
#pragma unroll
        for (int u = 0; u < kFB; ++u) {
            int row = row0 + u * NW * RPW + lane / TX;
            if (RPW == 1) row = __builtin_amdgcn_readfirstlane(row);
            rows[u] = row;
            raw[u] = acc[(row < ROWS ? row : 0) * (TX + 1) + x];
        }
#pragma unroll
        for (int u = 0; u < kFB; ++u) {
            const int row = rows[u];
            if (row >= ROWS) continue;
            const int l1 = row % (TY + 1), l2 = (NO == 3) ? row / (TY + 1) : 0;
            const double a = fix_value(raw[u], fs);
            const bool owned = l1 < TY && (NO == 2 || l2 < TZ);
            if (owned) {
                const int g1 = x0[1] + l1, g2 = (NO == 3) ? x0[NO - 1] + l2 : 0;
                const bool in = g1 < gd.n[1] && (NO == 2 || g2 < gd.n[NO - 1]);
                if (in && x_ok && !ghost_tile)
                    __builtin_nontemporal_store(
                        (T)(bgv + a),
                        &o[((NO == 3) ? (size_t)g2 * gd.n[1] + g1 : (size_t)g1) * gd.n[0] + x0[0] + x]);
            } else {
                int h[NO];
                h[0] = x;
                h[1] = l1;
                if (NO == 3) h[NO - 1] = l2;
                hb[halo_index<NO>(h)] = (T)a;
            }
        }
    }
    // x == TX column: the X-face of the halo buffer is indexed by the row number
    for (int row = threadIdx.x; row < ROWS; row += kSplatThreads)
        hb[row] = (T)fix_value(acc[row * (TX + 1) + TX], fs);
}

// ------------------------------------------------------------------ forward K4
template <typename T, int NI, int NO, bool HAS_PW, bool W3>
__global__ __launch_bounds__(kSplatThreads) void k_tile_splat(
    GridDesc<NO> gd, TileGeom<NO> tg, const RecT<T, W3>* __restrict__ rec,
    const WorkItem* __restrict__ items, const uint32_t* __restrict__ n_items,
    const uint32_t* __restrict__ tile_slab, const T* __restrict__ rot,
    const T* __restrict__ trans, const T* __restrict__ ow, const T* __restrict__ bg, int64_t b0,
    T* __restrict__ out, T* __restrict__ halo, T* __restrict__ ovf, int blocked,
    const uint32_t* __restrict__ maxpw, int fixed) {
    constexpr int NVH = tile_voxels_halo<NO>();
    __shared__ double acc[NVH];
    // Everything the block needs from memory before it can touch its records is requested at
    // once -- the item, the item count (the list is allocated for the whole grid, so reading
    // past the count is safe) and the pose of the group's first image -- instead of one after
    // the other (count -> item -> pose were three dependent round trips of 1-2 us each on a
    // busy chip).  Only a pose group's later images need a second pose fetch.
    const uint32_t n_it = *n_items;
    const uint32_t order_flag = n_items[1];  // the tile scan's verdict on the cloud's order
    const float maxw_call = sizeof(T) == 4 ? guarded_max_weight(1.f, maxpw, HAS_PW) : __builtin_inff();
    const WorkItem item = items[blockIdx.x];
    Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b0);
    for (int i = threadIdx.x; i < NVH; i += kSplatThreads) acc[i] = 0.0;
    if (blockIdx.x >= n_it) return;  // the grid is sized for the worst case
    // (uniform)  A projection (N_in > N_out) of a sorted cloud stays strided: neighbours in 3-D land
    // on the same pixels whichever share of the list a thread takes, and the blocked assignment only
    // adds its uncoalesced loads (10 M points x 64 poses -> 512^2, Hilbert-sorted: 13.9 against 11.3 ms).
    if (blocked == 2) blocked = (NI == NO) ? (int)order_flag : 0;
    // item.tile = (pose within the group) * NT + tile
    const int tile = (int)(item.tile % (uint32_t)tg.NT);
    const int64_t b = b0 + (int64_t)(item.tile / (uint32_t)tg.NT);
    int x0[NO], tc[NO];
    tile_origin<NO>(tile, tg, x0, tc);
    if (b != b0) ps = load_pose<T, NI, NO>(rot, trans, ow, b);
    // Record assignment: strided (lane-adjacent records, coalesced) or blocked (each thread
    // owns a contiguous run, so lanes are far apart in the list: with spatially sorted input
    // lane-adjacent records hit the same voxel and same-address LDS atomics serialise).
    // blocked == 2 on entry: whichever the tile scan chose from the order of the cloud.
    uint32_t r1 = item.end;
    uint32_t r = item.begin;
    uint32_t step = kSplatThreads;
    if (blocked) {
        const uint32_t per = (r1 - r + kSplatThreads - 1) / kSplatThreads;
        r += threadIdx.x * per;
        r1 = (r + per < r1) ? r + per : r1;
        step = 1;
    } else {
        r += threadIdx.x;
    }
    // kPF records per thread are kept in flight: with one load per iteration the heaviest
    // item's per-thread chain (records / threads iterations x memory latency) sets the
    // kernel time, whatever the LDS atomic rate.
#ifndef DPR_PF
#define DPR_PF 2
#endif
    constexpr int kPF = DPR_PF;
    RecT<T, W3> nxt[kPF];
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
        const uint32_t ru = r + u * step;
        nxt[u] = rec[ru < r1 ? ru : (r1 > item.begin ? r1 - 1 : item.begin)];
    }
    lds_barrier();  // LDS phases only: prefetched records stay in flight
    // fp32 data: exact 64-bit fixed-point sums (see FixScale); fp64 data and non-finite weights: f64
    const FixScale fs = fix_scale(sizeof(T) == 4 ? fabsf((float)ps.ow) * maxw_call : __builtin_inff(),
                                  item.end - item.begin, fixed);
    auto record_loop = [&](auto fix_tag) {
        constexpr bool FIX = decltype(fix_tag)::value;
        while (r < r1) {
            RecT<T, W3> cur[kPF];
#pragma unroll
            for (int u = 0; u < kPF; ++u) cur[u] = nxt[u];
            const uint32_t r_cur = r;
            r += kPF * step;
#pragma unroll
            for (int u = 0; u < kPF; ++u) {
                const uint32_t ru = r + u * step;
                nxt[u] = rec[ru < r1 ? ru : r1 - 1];  // clamped prefetch (branch-free loop body)
            }
#pragma unroll
            for (int u = 0; u < kPF; ++u)
                splat_record<FIX, T, NI, NO, HAS_PW>(cur[u], r_cur + u * step < r1, ps, gd, x0, acc, fs);
        }
    };
#if defined(DPR_ABL) && DPR_ABL == 3  // ablation: no record loop (clear + flush + dispatch only)
    (void)record_loop;
#else
    if (fs.mul != 0.0) record_loop(std::true_type{});  // (uniform)
    else record_loop(std::false_type{});
#endif
    lds_barrier();  // LDS phases only: prefetched records stay in flight
    if ((item.part_nparts >> 16) > 1) {
        // part of a split tile: the whole LDS tile goes to this part's overflow slab;
        // k_halo_gather sums the parts
        T* slab = ovf + (size_t)(tile_slab[item.tile] + (item.part_nparts & 0xffffu)) * NVH;
        constexpr int kSB = 4;  // cells per thread in flight
        for (int i0 = threadIdx.x; i0 < NVH; i0 += kSplatThreads * kSB) {
            double raw[kSB];
#pragma unroll
            for (int u = 0; u < kSB; ++u) {
                const int i = i0 + u * kSplatThreads;
                raw[u] = acc[i < NVH ? i : NVH - 1];
            }
#pragma unroll
            for (int u = 0; u < kSB; ++u) {
                const int i = i0 + u * kSplatThreads;
                if (i < NVH) slab[i] = (T)fix_value(raw[u], fs);
            }
        }
        return;
    }
    const double bgv = bg ? (double)bg[b] : 0.0;
    flush_tile<T, NO>(acc, fs, bgv, gd, x0, tg.ghost && tc[NO - 1] == 0 /* (only its upper halo is wanted) */,
                      out + b * gd.G, halo + (size_t)item.tile * halo_count<NO>());
}

// ------------------------------------------------------------------ forward K4, local binning
// (k_tile_splat with the record loop walking run descriptors; instantiated with RUNS = true)
template <typename T, int NI, int NO, bool HAS_PW, bool W3, bool RUNS>
__global__ __launch_bounds__(kSplatThreads, DPR_SPLAT_RUNS_OCC) void k_tile_splat_runs(  // (2 blocks / CU)
    GridDesc<NO> gd, TileGeom<NO> tg, const RecT<T, W3>* __restrict__ rec,
    const RunDesc* __restrict__ runs, uint32_t max_rec,
    const WorkItem* __restrict__ items, const uint32_t* __restrict__ n_items,
    const uint32_t* __restrict__ tile_slab, const T* __restrict__ rot,
    const T* __restrict__ trans, const T* __restrict__ ow, const T* __restrict__ bg, int64_t b0,
    T* __restrict__ out, T* __restrict__ halo, T* __restrict__ ovf, int blocked,
    const uint32_t* __restrict__ maxpw, int fixed) {
    constexpr int NVH = tile_voxels_halo<NO>();
    __shared__ double acc[NVH];
    // Everything the block needs from memory before it can touch its records is requested at
    // once -- the item, the item count (the list is allocated for the whole grid, so reading
    // past the count is safe) and the pose of the group's first image -- instead of one after
    // the other (count -> item -> pose were three dependent round trips of 1-2 us each on a
    // busy chip).  Only a pose group's later images need a second pose fetch.
    const uint32_t n_it = *n_items;
    const WorkItem item = items[blockIdx.x];
    const float maxw_call = sizeof(T) == 4 ? guarded_max_weight(1.f, maxpw, HAS_PW) : __builtin_inff();
    Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b0);
    for (int i = threadIdx.x; i < NVH; i += kSplatThreads) acc[i] = 0.0;
    if (blockIdx.x >= n_it) return;  // the grid is sized for the worst case
    // item.tile = (pose within the group) * NT + tile
    const int tile = (int)(item.tile % (uint32_t)tg.NT);
    const int64_t b = b0 + (int64_t)(item.tile / (uint32_t)tg.NT);
    int x0[NO], tc[NO];
    tile_origin<NO>(tile, tg, x0, tc);
    if (b != b0) ps = load_pose<T, NI, NO>(rot, trans, ow, b);
#ifndef DPR_PF
#define DPR_PF 2
#endif
    constexpr int kPF = DPR_PF;
    // fp32 data: exact 64-bit fixed-point sums (see FixScale).  Bound on the records of the item:
    // a run holds at most one sub-chunk (<= 4096 records).
    uint32_t n_bound = (item.end - item.begin) < (1u << 19) ? (item.end - item.begin) * 4096u : max_rec;
    n_bound = n_bound < max_rec ? n_bound : max_rec;
    const FixScale fs = fix_scale(sizeof(T) == 4 ? fabsf((float)ps.ow) * maxw_call : __builtin_inff(), n_bound, fixed);
    {
        // LOCAL BINNING: the item is a range of run descriptors; the records of a run are
        // contiguous.  Each thread takes a contiguous share of the item's records (as in the
        // blocked assignment below: neighbouring records of a coherent cloud hit the same
        // voxels) and walks it through the run table with a forward-only cursor.
        __shared__ RunTable<kMaxRuns> rt;
        for (uint32_t d0 = item.begin; d0 < item.end; d0 += kMaxRuns) {
            const uint32_t d1 = (d0 + kMaxRuns < item.end) ? d0 + kMaxRuns : item.end;
            const int nruns = (int)(d1 - d0);
            const uint32_t n = load_runs(rt, runs, d0, d1, max_rec);  // barrier inside
            // Record assignment.  blocked: a contiguous share per thread (a fully sorted cloud:
            // neighbouring records hit the same voxels, lanes far apart in the list do not
            // collide); strided: lane-adjacent records, coalesced loads (a cloud that is only
            // cell-sorted, random inside its cell: no collisions to avoid).
            const uint32_t per = (n + kSplatThreads - 1) / kSplatThreads;
            const uint32_t step = blocked ? 1u : (uint32_t)kSplatThreads;
            uint32_t i = blocked ? threadIdx.x * per : threadIdx.x;
            const uint32_t i1 = blocked ? ((i + per < n) ? i + per : n) : n;
            if (i < i1) {
                RunCursor cu;
                cu.seek(rt, i, nruns);
                uint32_t fetched = i;  // logical index of the next record to request
                RecT<T, W3> nxt[kPF];
#pragma unroll
                for (int u = 0; u < kPF; ++u) {
                    // past the end the last record is requested again (branch-free loop body)
                    const bool adv = fetched + step < i1;
                    nxt[u] = rec[cu.next(rt, adv ? step : 0u, nruns, adv)];
                    fetched += adv ? step : 0u;
                }
                while (i < i1) {
                    RecT<T, W3> cur[kPF];
#pragma unroll
                    for (int u = 0; u < kPF; ++u) cur[u] = nxt[u];
                    const uint32_t i_cur = i;
                    i += kPF * step;
#pragma unroll
                    for (int u = 0; u < kPF; ++u) {
                        const bool adv = fetched + step < i1;
                        nxt[u] = rec[cu.next(rt, adv ? step : 0u, nruns, adv)];
                        fetched += adv ? step : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < kPF; ++u) {
                        if (fs.mul != 0.0)  // (uniform)
                            splat_record<true, T, NI, NO, HAS_PW>(cur[u], i_cur + u * step < i1, ps, gd, x0, acc, fs);
                        else
                            splat_record<false, T, NI, NO, HAS_PW>(cur[u], i_cur + u * step < i1, ps, gd, x0, acc, fs);
                    }
                }
            }
            __syncthreads();  // the table is rebuilt by the next round
        }
    }
    lds_barrier();  // LDS phases only: prefetched records stay in flight
    if ((item.part_nparts >> 16) > 1) {
        // part of a split tile: the whole LDS tile goes to this part's overflow slab;
        // k_halo_gather sums the parts
        T* slab = ovf + (size_t)(tile_slab[item.tile] + (item.part_nparts & 0xffffu)) * NVH;
        constexpr int kSB = 4;  // cells per thread in flight
        for (int i0 = threadIdx.x; i0 < NVH; i0 += kSplatThreads * kSB) {
            double raw[kSB];
#pragma unroll
            for (int u = 0; u < kSB; ++u) {
                const int i = i0 + u * kSplatThreads;
                raw[u] = acc[i < NVH ? i : NVH - 1];
            }
#pragma unroll
            for (int u = 0; u < kSB; ++u) {
                const int i = i0 + u * kSplatThreads;
                if (i < NVH) slab[i] = (T)fix_value(raw[u], fs);
            }
        }
        return;
    }
    const double bgv = bg ? (double)bg[b] : 0.0;
    flush_tile<T, NO>(acc, fs, bgv, gd, x0, tg.ghost && tc[NO - 1] == 0 /* (only its upper halo is wanted) */,
                      out + b * gd.G, halo + (size_t)item.tile * halo_count<NO>());
}

// ------------------------------------------------------------------ forward K5
// Low-face voxels of a tile, enumerated compactly:
//   3-D: [0, TX*TY) the z == 0 face; then y == 0 (z > 0); then x == 0 (y > 0, z > 0)
//   2-D: [0, TX) the y == 0 row; then x == 0 (y > 0)
template <int NO> __host__ __device__ constexpr int low_face_count() {
    if (NO == 2) return TileDims<2>::T[0] + TileDims<2>::T[1] - 1;
    return TileDims<3>::T[0] * TileDims<3>::T[1] + TileDims<3>::T[0] * (TileDims<3>::T[2] - 1) +
           (TileDims<3>::T[1] - 1) * (TileDims<3>::T[2] - 1);
}
template <int NO> __device__ __forceinline__ void low_face_coords(int i, int (&l)[NO]) {
    constexpr int TX = TileDims<NO>::T[0], TY = TileDims<NO>::T[1];
    if constexpr (NO == 2) {
        if (i < TX) {
            l[0] = i;
            l[1] = 0;
        } else {
            l[0] = 0;
            l[1] = 1 + (i - TX);
        }
    } else {
        constexpr int TZ = TileDims<NO>::T[2];
        if (i < TX * TY) {
            l[0] = i % TX;
            l[1] = i / TX;
            l[2] = 0;
        } else if (i < TX * TY + TX * (TZ - 1)) {
            const int j = i - TX * TY;
            l[0] = j % TX;
            l[1] = 0;
            l[2] = 1 + j / TX;
        } else {
            const int j = i - TX * TY - TX * (TZ - 1);
            l[0] = 0;
            l[1] = 1 + j % (TY - 1);
            l[2] = 1 + j / (TY - 1);
        }
    }
}

// One block per tile; threads walk the tile's low-face voxels and add what the lower
// neighbours accumulated for them.  Gather form: each voxel has exactly one writer.
// Split tiles (tile_parts > 1): their parts left whole LDS tiles in overflow slabs; the block
// then walks ALL owned voxels (out = background + sum of parts) and neighbours read a split
// tile's halo as the sum over its slabs.
// One (pose, tile) = `ptile`: the low faces of an unsplit tile (split == false), or chunk `c` of
// kSplitChunks of ALL owned voxels of a split tile.
template <typename T, int NO>
__device__ __forceinline__ void halo_gather_tile(const GridDesc<NO>& gd, const TileGeom<NO>& tg,
                                                 const T* __restrict__ halo,
                                                 const T* __restrict__ ovf,
                                                 const uint32_t* __restrict__ tile_parts,
                                                 const uint32_t* __restrict__ tile_slab,
                                                 const T* __restrict__ bg, int64_t b0,
                                                 T* __restrict__ out, bool any_split, int ptile,
                                                 bool split, int c) {
    constexpr int NV = tile_voxels<NO>();
    constexpr int NVH = tile_voxels_halo<NO>();
    constexpr int CH = kSplitChunks;
    const int i_begin = split ? c * 256 : 0;
    const int i_end = split ? NV : low_face_count<NO>();
    const int tile = ptile % tg.NT;
    const int pbase = ptile - tile;  // first bin of this pose
    const int64_t b = b0 + ptile / tg.NT;
    const int i_step = split ? 256 * CH : 256;
    int x0[NO], tc[NO];
    tile_origin<NO>(tile, tg, x0, tc);
    if (tg.ghost && tc[NO - 1] == 0) return;  // a ghost tile owns nothing in this launch
    T* o = out + b * gd.G;
    // The 2^N - 1 lower neighbours (combination m: bit d = one tile down along axis d): which exist,
    // which are split tiles and where their slabs start is block-uniform and fetched ONCE, up
    // front -- looked up per voxel and combination it was a chain of dependent loads.
    constexpr int NC = 1 << NO;
    uint32_t nparts[NC], nslab[NC];
    bool nvalid[NC], any_nbr_split = false;
    int nsrc[NC];
#pragma unroll
    for (int m = 1; m < NC; ++m) {
        bool v = true;
        int src = ptile, tstride = 1;
#pragma unroll
        for (int d = 0; d < NO; ++d) {
            if ((m >> d) & 1) {
                v = v && tc[d] > 0;
                src -= tstride;
            }
            tstride *= tg.nt[d];
        }
        nvalid[m] = v;
        nsrc[m] = v ? src : ptile;
        nparts[m] = any_split ? tile_parts[nsrc[m]] : 1u;
        nslab[m] = any_split ? tile_slab[nsrc[m]] : 0u;
        any_nbr_split = any_nbr_split || (v && nparts[m] > 1);
    }
    if constexpr (NO == 3 && TileDims<NO>::T[0] == kWave) {
        if (split) {
            // A SPLIT 3-D tile, ROW form (round 6): chunk `c` = kSplitRows rows (l1, l2) of the tile's
            // TY * TZ owned rows, a wave takes kSplitRows / 4 of them at once, lane = l0.  Every owned
            // voxel = background + the parts' slab values in part order (coalesced 256-byte rows, the
            // loads of the wave's rows in flight together); rows on a low face add the lower
            // neighbours' halos, lane 0 also the four combinations that reach through x.  Same terms
            // in the same order as the flat loop at the end of this function, which took ~25 ns per
            // voxel (div / mod per voxel, one dependent slab chain at a time) and made finer splits
            // of heavy tiles a net loss -- with the heaviest item of C3 at 39 000 records (8 x the
            // mean) setting the time of BOTH tile kernels.
            constexpr int TX = TileDims<NO>::T[0], TY = TileDims<NO>::T[1], TZ = TileDims<NO>::T[2];
            constexpr int YF = (TY + 1) * (TZ + 1), ZF = YF + TX * (TZ + 1);  // face bases (halo_index)
            constexpr int NW = 256 / kWave, RIT = kSplitRows / NW;
            static_assert(kSplitRows % NW == 0 && (TY * TZ) % kSplitRows == 0, "whole rows per wave");
            static_assert(kSplitChunks * kSplitRows == TY * TZ, "chunks cover the tile");
            const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
            const uint32_t my_parts = tile_parts[ptile], my_slab = tile_slab[ptile];
            const double bgv = bg ? (double)bg[b] : 0.0;
            const bool x_ok = x0[0] + lane < gd.n[0];
            const size_t HC = halo_count<NO>();
            double own[RIT];
            bool act[RIT];
            int offs[RIT], cell[RIT], l1s[RIT], l2s[RIT];
#pragma unroll
            for (int k = 0; k < RIT; ++k) {
                const int row = __builtin_amdgcn_readfirstlane(c * kSplitRows + wave * RIT + k);
                l1s[k] = row % TY;
                l2s[k] = row / TY;
                const int g1 = x0[1] + l1s[k], g2 = x0[2] + l2s[k];
                act[k] = x_ok && g1 < gd.n[1] && g2 < gd.n[2];
                offs[k] = act[k] ? (g2 * gd.n[1] + g1) * gd.n[0] + x0[0] + lane : 0;
                cell[k] = lane + (TX + 1) * (l1s[k] + (TY + 1) * l2s[k]);
                own[k] = bgv;
            }
            const T* sp = ovf + (size_t)my_slab * NVH;
            for (uint32_t q = 0; q < my_parts; ++q) {  // (uniform)
                T a[RIT];
#pragma unroll
                for (int k = 0; k < RIT; ++k) a[k] = sp[(size_t)q * NVH + cell[k]];
#pragma unroll
                for (int k = 0; k < RIT; ++k) own[k] += (double)a[k];
            }
            // a neighbour's contribution to one cell: its halo value, or -- a split neighbour -- the sum
            // over its parts' slabs (cell index in the (T + 1)^3 tile)
            auto nbr = [&](int m, bool on, int hidx, int ncell) -> double {
                if (nparts[m] > 1) {  // (uniform)
                    const T* np_ = ovf + (size_t)nslab[m] * NVH + (on ? ncell : 0);
                    double sum = 0.0;
                    for (uint32_t q = 0; q < nparts[m]; ++q) sum += (double)np_[(size_t)q * NVH];
                    return on ? sum : 0.0;
                }
                const T v = halo[on ? (size_t)nsrc[m] * HC + hidx : (size_t)0];
                return on ? (double)v : 0.0;
            };
#pragma unroll
            for (int k = 0; k < RIT; ++k) {
                const int l1 = l1s[k], l2 = l2s[k];
                const bool lowy = l1 == 0, lowz = l2 == 0, l0z = lane == 0;  // (lowy, lowz: uniform)
                if (!(lowy || lowz || nvalid[1])) {  // an interior row of a tile without an x neighbour
                    if (act[k]) o[offs[k]] = (T)own[k];
                    continue;
                }
                double add = 0.0;  // m = 1 .. 7 in order, as the flat loop adds them
                if (nvalid[1]) add += nbr(1, act[k] && l0z, l1 + (TY + 1) * l2, TX + (TX + 1) * (l1 + (TY + 1) * l2));
                if (lowy && nvalid[2]) add += nbr(2, act[k], YF + lane + TX * l2, lane + (TX + 1) * (TY + (TY + 1) * l2));
                if (lowy && nvalid[3]) add += nbr(3, act[k] && l0z, TY + (TY + 1) * l2, TX + (TX + 1) * (TY + (TY + 1) * l2));
                if (lowz && nvalid[4]) add += nbr(4, act[k], ZF + lane + TX * l1, lane + (TX + 1) * (l1 + (TY + 1) * TZ));
                if (lowz && nvalid[5]) add += nbr(5, act[k] && l0z, l1 + (TY + 1) * TZ, TX + (TX + 1) * (l1 + (TY + 1) * TZ));
                if (lowy && lowz && nvalid[6])
                    add += nbr(6, act[k], YF + TX * TZ + lane, lane + (TX + 1) * (TY + (TY + 1) * TZ));
                if (lowy && lowz && nvalid[7])
                    add += nbr(7, act[k] && l0z, TY + (TY + 1) * TZ, TX + (TX + 1) * (TY + (TY + 1) * TZ));
                if (act[k]) o[offs[k]] = (T)(own[k] + add);
            }
            return;
        }
        if (!split) {
            // An unsplit 3-D tile, ROW form.  The flat form below spends ~2000 VALU instructions
            // per thread on index arithmetic (face coordinates by div / mod, seven neighbour
            // combinations each with its own tile and halo index) for 7 voxels: the kernel was
            // bound by that, not by its 40 MB of traffic (20 us at C3), and with a split tile
            // anywhere every block walked ~100 dependent loads (22 us for the 256 tiles of C2).  Here
            //   (A) a wave takes a row (l1, l2) with l1 == 0 or l2 == 0 -- TY + TZ - 1 rows -- its
            //       lanes are l0 = 1..63: everything but "+ lane" is wave-uniform, and only the
            //       y / z / yz neighbours can contribute;
            //   (B) the 128 voxels with l0 == 0 (the only ones the x neighbours reach) take the
            //       general seven-combination form, one voxel per thread.
            // Every voxel still has exactly one writer and all loads are issued before the first
            // store.  Which of the seven lower neighbours are split tiles is block-uniform and
            // fetched once; their contribution (a sum over the parts' slabs) is added in a second,
            // rare pass.
            constexpr int TX = TileDims<NO>::T[0], TY = TileDims<NO>::T[1], TZ = TileDims<NO>::T[2];
            constexpr int NROW = TY + TZ - 1, NW = 256 / kWave, RIT = (NROW + NW - 1) / NW;
            constexpr int YF = (TY + 1) * (TZ + 1), ZF = YF + TX * (TZ + 1);  // face bases (halo_index)
            static_assert(TY * TZ <= 256, "one thread per l0 == 0 voxel");
            const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
            const bool x_ok = x0[0] + lane < gd.n[0] && lane >= 1;
            const T* hy_t = halo + (size_t)nsrc[2] * halo_count<NO>() + YF + lane;            // y neighbour
            const T* hz_t = halo + (size_t)nsrc[4] * halo_count<NO>() + ZF + lane;            // z neighbour
            const T* hyz_t = halo + (size_t)nsrc[6] * halo_count<NO>() + YF + TX * TZ + lane;  // yz
            const bool uy = nvalid[2] && nparts[2] == 1, uz = nvalid[4] && nparts[4] == 1,
                       uyz = nvalid[6] && nparts[6] == 1;  // unsplit sources: one halo value each
            T cur[RIT], vy[RIT], vz[RIT], vyz[RIT];
            bool act[RIT], by[RIT], bz[RIT];
            int offs[RIT];
#pragma unroll
            for (int k = 0; k < RIT; ++k) {
                const int row = __builtin_amdgcn_readfirstlane(wave + k * NW);
                const bool live = row < NROW;
                const int l1 = row < TY ? row : 0, l2 = row < TY ? 0 : row - (TY - 1);
                const int g1 = x0[1] + l1, g2 = x0[2] + l2;
                by[k] = live && l1 == 0 && tc[1] > 0;
                bz[k] = live && l2 == 0 && tc[2] > 0;
                act[k] = live && x_ok && g1 < gd.n[1] && g2 < gd.n[2] && (by[k] || bz[k]);
                offs[k] = act[k] ? (g2 * gd.n[1] + g1) * gd.n[0] + x0[0] + lane : 0;
                cur[k] = o[offs[k]];
                // (selects on the ADDRESS: an invalid combination reads halo[0] and is dropped)
                vy[k] = *((act[k] && by[k] && uy) ? hy_t + TX * l2 : halo);
                vz[k] = *((act[k] && bz[k] && uz) ? hz_t + TX * l1 : halo);
                vyz[k] = *((act[k] && by[k] && bz[k] && uyz) ? hyz_t : halo);
            }
            // (B)
            const int j = threadIdx.x < TY * TZ ? threadIdx.x : 0;
            const int lB[NO] = {0, j % TY, j / TY};
            int offB = 0, strideB = 1;
            bool okB = threadIdx.x < TY * TZ, lowB = false;
#pragma unroll
            for (int d = 0; d < NO; ++d) {
                const int gcoord = x0[d] + lB[d];
                okB = okB && gcoord < gd.n[d];
                lowB = lowB || (lB[d] == 0 && tc[d] > 0);
                offB += gcoord * strideB;
                strideB *= gd.n[d];
            }
            const bool actB = okB && lowB;
            offB = actB ? offB : 0;
            const T curB = o[offB];
            T vB[8];
            bool bB[8];
            int cellB[8];  // index of the source cell in a (T + 1)^3 tile (for the slab pass)
#pragma unroll
            for (int m = 1; m < 8; ++m) {
                bool valid = actB && nvalid[m];
                int h[NO];
#pragma unroll
                for (int d = 0; d < NO; ++d) {
                    const bool in_m = (m >> d) & 1;
                    valid = valid && (!in_m || lB[d] == 0);
                    h[d] = in_m ? TileDims<NO>::T[d] : lB[d];
                }
                bB[m] = valid;
                cellB[m] = lds_index<NO>(h);
                const size_t hi = (valid && nparts[m] == 1)
                                      ? (size_t)nsrc[m] * halo_count<NO>() + halo_index<NO>(h)
                                      : (size_t)0;
                vB[m] = halo[hi];
            }
            double addA[RIT][3], addB[8];
#pragma unroll
            for (int k = 0; k < RIT; ++k) {
                addA[k][0] = (act[k] && by[k] && uy) ? (double)vy[k] : 0.0;
                addA[k][1] = (act[k] && bz[k] && uz) ? (double)vz[k] : 0.0;
                addA[k][2] = (act[k] && by[k] && bz[k] && uyz) ? (double)vyz[k] : 0.0;
            }
#pragma unroll
            for (int m = 1; m < 8; ++m) addB[m] = (bB[m] && nparts[m] == 1) ? (double)vB[m] : 0.0;
            if (any_nbr_split) {  // uniform, rare: sources that are split tiles = sums over slabs
                auto slab_sum = [&](int m, int cell) {
                    const T* sp = ovf + (size_t)nslab[m] * NVH + cell;
                    double sum = 0.0;
                    uint32_t q = 0;
                    for (; q + 4 <= nparts[m]; q += 4) {  // four loads in flight
                        const T a0 = sp[(size_t)q * NVH], a1 = sp[(size_t)(q + 1) * NVH],
                                a2 = sp[(size_t)(q + 2) * NVH], a3 = sp[(size_t)(q + 3) * NVH];
                        sum += (double)a0;
                        sum += (double)a1;
                        sum += (double)a2;
                        sum += (double)a3;
                    }
                    for (; q < nparts[m]; ++q) sum += (double)sp[(size_t)q * NVH];
                    return sum;
                };
#pragma unroll
                for (int k = 0; k < RIT; ++k) {
                    const int row = __builtin_amdgcn_readfirstlane(wave + k * NW);
                    const int l1 = row < TY ? row : 0, l2 = row < TY ? 0 : row - (TY - 1);
                    if (nvalid[2] && nparts[2] > 1 && act[k] && by[k])
                        addA[k][0] = slab_sum(2, lane + (TX + 1) * (TY + (TY + 1) * l2));
                    if (nvalid[4] && nparts[4] > 1 && act[k] && bz[k])
                        addA[k][1] = slab_sum(4, lane + (TX + 1) * (l1 + (TY + 1) * TZ));
                    if (nvalid[6] && nparts[6] > 1 && act[k] && by[k] && bz[k])
                        addA[k][2] = slab_sum(6, lane + (TX + 1) * (TY + (TY + 1) * TZ));
                }
#pragma unroll
                for (int m = 1; m < 8; ++m)
                    if (bB[m] && nparts[m] > 1) addB[m] = slab_sum(m, cellB[m]);
            }
#pragma unroll
            for (int k = 0; k < RIT; ++k) {
                const double add = (addA[k][0] + addA[k][1]) + addA[k][2];  // m = 2, 4, 6
                if (act[k]) o[offs[k]] = (T)((double)cur[k] + add);
            }
            double sumB = 0.0;
#pragma unroll
            for (int m = 1; m < 8; ++m) sumB += addB[m];
            if (actB) o[offB] = (T)((double)curB + sumB);
            return;
        }
    }
    if (!any_split) {
        // Common case (no split tile anywhere), flat form (2-D grids): the thread's voxels are
        // handled with every load issued before the first store -- branch-free, invalid neighbour
        // combinations read halo[0] and are multiplied away -- so the ~IT dependent round trips of
        // the general loop below collapse into one.
        constexpr int IT = (low_face_count<NO>() + 255) / 256;
        T cur[IT];
        double add[IT];
        int offs[IT];
        bool act[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = threadIdx.x + k * 256;
            int l[NO];
            low_face_coords<NO>(i < low_face_count<NO>() ? i : 0, l);
            int off = 0, stride = 1;
            bool ok = i < low_face_count<NO>(), low = false;
#pragma unroll
            for (int d = 0; d < NO; ++d) {
                const int gcoord = x0[d] + l[d];
                ok = ok && gcoord < gd.n[d];
                low = low || (l[d] == 0 && tc[d] > 0);
                off += gcoord * stride;
                stride *= gd.n[d];
            }
            act[k] = ok && low;
            offs[k] = act[k] ? off : 0;
            cur[k] = o[offs[k]];
            add[k] = 0.0;
#pragma unroll
            for (int m = 1; m < (1 << NO); ++m) {
                bool valid = act[k];
                int h[NO];
                int src = 0, tstride = 1;
#pragma unroll
                for (int d = 0; d < NO; ++d) {
                    const bool in_m = (m >> d) & 1;
                    valid = valid && (!in_m || (l[d] == 0 && tc[d] > 0));
                    h[d] = in_m ? TileDims<NO>::T[d] : l[d];
                    src += (tc[d] - (in_m ? 1 : 0)) * tstride;
                    tstride *= tg.nt[d];
                }
                const size_t hi = valid ? (size_t)(src + pbase) * halo_count<NO>() + halo_index<NO>(h)
                                        : (size_t)0;
                const T hv = halo[hi];
                add[k] += valid ? (double)hv : 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < IT; ++k)
            if (act[k]) o[offs[k]] = (T)((double)cur[k] + add[k]);
        return;
    }
    const uint32_t my_parts = split ? tile_parts[ptile] : 1u;
    const uint32_t my_slab = split ? tile_slab[ptile] : 0u;
    const double bgv = bg ? (double)bg[b] : 0.0;
    // acc + the n parts' values of one cell, in part order; four loads in flight (one at a time
    // it was a chain of n round trips per voxel: 22 us for the 256 tiles of C2, 8 parts at the centre)
    auto add_parts = [&](double acc, const T* sp, uint32_t n) {
        uint32_t q = 0;
        for (; q + 4 <= n; q += 4) {
            const T a0 = sp[(size_t)q * NVH], a1 = sp[(size_t)(q + 1) * NVH],
                    a2 = sp[(size_t)(q + 2) * NVH], a3 = sp[(size_t)(q + 3) * NVH];
            acc += (double)a0;
            acc += (double)a1;
            acc += (double)a2;
            acc += (double)a3;
        }
        for (; q < n; ++q) acc += (double)sp[(size_t)q * NVH];
        return acc;
    };
    for (int i = i_begin + threadIdx.x; i < i_end; i += i_step) {
        int l[NO];
        if (split) {
            int rem = i;
#pragma unroll
            for (int d = 0; d < NO; ++d) {
                l[d] = rem % TileDims<NO>::T[d];
                rem /= TileDims<NO>::T[d];
            }
        } else {
            low_face_coords<NO>(i, l);
        }
        int off = 0, stride = 1;
        bool ok = true, low = false;
#pragma unroll
        for (int d = 0; d < NO; ++d) {
            const int gcoord = x0[d] + l[d];
            ok = ok && gcoord < gd.n[d];
            low = low || (l[d] == 0 && tc[d] > 0);
            off += gcoord * stride;
            stride *= gd.n[d];
        }
        if (!ok || (!low && !split)) continue;
        double add = 0.0;
        if (low) {
#pragma unroll
            for (int m = 1; m < (1 << NO); ++m) {
                bool valid = nvalid[m];
                int h[NO];
#pragma unroll
                for (int d = 0; d < NO; ++d) {
                    const bool in_m = (m >> d) & 1;
                    valid = valid && (!in_m || l[d] == 0);
                    h[d] = in_m ? TileDims<NO>::T[d] : l[d];
                }
                if (!valid) continue;
                if (nparts[m] > 1)
                    add = add_parts(add, ovf + (size_t)nslab[m] * NVH + lds_index<NO>(h), nparts[m]);
                else
                    add += (double)halo[(size_t)nsrc[m] * halo_count<NO>() + halo_index<NO>(h)];
            }
        }
        if (split) {
            const double own = add_parts(bgv, ovf + (size_t)my_slab * NVH + lds_index<NO>(l), my_parts);
            o[off] = (T)(own + add);
        } else {
            o[off] = (T)((double)o[off] + add);
        }
    }
}

// blocks [0, NT*nb): one (pose, tile) each (a split tile's block returns: its voxels belong to
// the chunk blocks); the kSplitGrid blocks behind them walk the (split tile, chunk) work items --
// a fixed number, whatever the worst case of split tiles (a grid sized for that case was ~20 000
// blocks at C3 that did nothing but read n_split and exit).
template <typename T, int NO>
__global__ __launch_bounds__(256) void k_halo_gather(GridDesc<NO> gd, TileGeom<NO> tg,
                                                     const T* __restrict__ halo,
                                                     const T* __restrict__ ovf,
                                                     const uint32_t* __restrict__ tile_parts,
                                                     const uint32_t* __restrict__ tile_slab,
                                                     const uint32_t* __restrict__ split_list,
                                                     const uint32_t* __restrict__ n_split,
                                                     const T* __restrict__ bg, int64_t b0,
                                                     int nb, T* __restrict__ out) {
    const int NTe = tg.NT * nb;
    const uint32_t ns = *n_split;
    const bool any_split = ns != 0;  // uniform; the common case has no split tile
    if ((int)blockIdx.x < NTe) {
        const int ptile = blockIdx.x;
        if (any_split && tile_parts[ptile] > 1) return;  // handled by the chunk blocks
        halo_gather_tile<T, NO>(gd, tg, halo, ovf, tile_parts, tile_slab, bg, b0, out, any_split,
                                ptile, false, 0);
        return;
    }
    const int n_items = (int)ns * kSplitChunks;
    for (int w = (int)blockIdx.x - NTe; w < n_items; w += (int)gridDim.x - NTe)
        halo_gather_tile<T, NO>(gd, tg, halo, ovf, tile_parts, tile_slab, bg, b0, out, true,
                                (int)split_list[w / kSplitChunks], true, w % kSplitChunks);
}

// ------------------------------------------------------------------ pullback K4
// UNPERM: the per-point gradient {d point, d point_weight} overwrites the point's record in
// place (coalesced 16/32-byte stores in binned order); k_unpermute then brings it back to the
// original order with one random read per point.  !UNPERM: the owner thread stores straight
// to ds_dpoints[idx] / ds_dpoint_weight[idx] (good when the input order is spatially coherent).
template <typename T, int NI, int NO, bool HAS_PW, bool FIRST_POSE, bool UNPERM>
__global__ __launch_bounds__(gather_threads<T>(), gather_waves_per_simd<T>()) void k_tile_gather(
    GridDesc<NO> gd, TileGeom<NO> tg, Rec4<T>* rec, int64_t P,
    const uint32_t* __restrict__ rec_idx, const WorkItem* __restrict__ items,
    const uint32_t* __restrict__ n_items, int max_items, const T* __restrict__ g,
    const T* __restrict__ rot, const T* __restrict__ trans, const T* __restrict__ ow, int64_t b0,
    T* __restrict__ ds_dpoints, T* __restrict__ ds_dpw, double* __restrict__ partials,
    Residual<T> rs, BinHeader want, BinHeader* hdr) {
    constexpr int GT = gather_threads<T>();
    constexpr int NVH = tile_voxels_halo<NO>();
    constexpr int NVAL = NO * NI + NO + 3;  // dR | dt | d out_weight | d background | loss
    constexpr int NW = GT / kWave;
    __shared__ T tile_g[NVH];
    __shared__ double red[NW][NVAL];
    if (want.magic) {
        // DPR_FLAG_REUSE_BINNING: trust the lists in the workspace only if a KEEP_BINNING forward
        // with the same problem, buffers and pose wrote them and nobody has consumed them since
        const bool ok = header_matches(hdr, want, (const uint32_t*)(rot + b0 * (NO * NI)),
                                       NO * NI * (int)(sizeof(T) / 4),
                                       (const uint32_t*)(trans + b0 * NO), NO * (int)(sizeof(T) / 4));
        if (blockIdx.x == 0 && threadIdx.x == 0) hdr->verdict = ok ? 1u : 0u;
        if (!ok) return;  // k_unpermute / k_pose_reduce turn the verdict into NaN outputs
    }
    // item, item count and the first image's pose are requested together (see k_tile_splat)
    const uint32_t n_it = *n_items;
    WorkItem item = items[blockIdx.x];
    Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b0);
    if (blockIdx.x >= n_it) return;  // the grid is sized for the worst case
    // record ranges never leave the record buffer, whatever the lists say
    if (item.end > (uint32_t)P) item.end = (uint32_t)P;
    if (item.begin > item.end) item.begin = item.end;
    const int tile = (int)(item.tile % (uint32_t)tg.NT);
    const int64_t b = b0 + (int64_t)(item.tile / (uint32_t)tg.NT);
    int x0[NO], tc[NO];
    tile_origin<NO>(tile, tg, x0, tc);
    const T* gb = g + b * gd.G;
    const T* tb = rs.target ? rs.target + b * gd.G : nullptr;
    const uint32_t r1 = item.end;
    uint32_t r = item.begin + threadIdx.x;
    Rec4<T> nxt;
    uint32_t nxt_idx = 0;
    if (r < r1) {
        nxt = rec[r];
        if (HAS_PW && !UNPERM) nxt_idx = rec_idx[r];
    }
    if (UNPERM && blockIdx.x == 0 && threadIdx.x == 0) {
        Rec4<T> z;
        z.v[0] = z.v[1] = z.v[2] = z.v[3] = T(0);
        rec[P] = z;  // spare slot: the gradient of every rejected point
    }
    // Stage the ds_dout tile + upper halo in LDS, one row (TX + 1 cells along x) at a time: the
    // row's y / z coordinates, bounds and base offset are computed once per row, a lane only
    // adds its x; kRB rows are in flight per wave.  Cells beyond the grid are staged as 0.
    // Owned voxels are summed for ds_dbackground (residual mode: ds_dout = scale * (out -
    // target) formed here, squared residuals summed for the loss).
    double bg_sum = 0.0, sq_sum = 0.0;
    {
        constexpr int TX = TileDims<NO>::T[0], TY = TileDims<NO>::T[1];
        constexpr int TZ = (NO == 3) ? TileDims<NO>::T[NO - 1] : 0;
        constexpr int ROWS = NVH / (TX + 1);
        constexpr int RPW = kWave / TX;  // rows per wave pass (1 for TX = 64, 2 for TX = 32)
        constexpr int kRB = DPR_GATHER_RB;  // row passes in flight
        static_assert(kWave % TX == 0, "tile rows must divide the wavefront");
        const bool first_part = (item.part_nparts & 0xffffu) == 0;
        const int lane = threadIdx.x & (kWave - 1);
        const int x = lane % TX;
        const bool x_ok = x0[0] + x < gd.n[0];
        constexpr int STEP = (GT / kWave) * RPW;
        // an unsplit tile's ds_dout cells are read by this block only (plus the neighbours' halo
        // rows): streamed with non-temporal loads; the parts of a split tile re-read them
        auto stage_rows = [&](auto nt_tag) {
        constexpr bool NT = decltype(nt_tag)::value;
        for (int row0 = (threadIdx.x / kWave) * RPW; row0 < ROWS; row0 += STEP * kRB) {
            T v[kRB], tv[kRB];
            bool in[kRB], owned[kRB];
            int lrow[kRB];
#pragma unroll
            for (int k = 0; k < kRB; ++k) {
                int row = row0 + k * STEP + lane / TX;
                if (RPW == 1) row = __builtin_amdgcn_readfirstlane(row);
                const bool live = row < ROWS;
                const int l1 = row % (TY + 1), l2 = (NO == 3) ? row / (TY + 1) : 0;
                const int g1 = x0[1] + l1, g2 = (NO == 3) ? x0[NO - 1] + l2 : 0;
                in[k] = live && x_ok && g1 < gd.n[1] && (NO == 2 || g2 < gd.n[NO - 1]);
                owned[k] = in[k] && l1 < TY && (NO == 2 || l2 < TZ);
                lrow[k] = live ? row * (TX + 1) + x : -1;
                const size_t off =
                    ((NO == 3) ? (size_t)g2 * gd.n[1] + g1 : (size_t)g1) * gd.n[0] + x0[0] + x;
                const size_t oc = in[k] ? off : 0;
                v[k] = NT ? __builtin_nontemporal_load(&gb[oc]) : gb[oc];
                tv[k] = tb ? (NT ? __builtin_nontemporal_load(&tb[oc]) : tb[oc]) : T(0);
            }
#pragma unroll
            for (int k = 0; k < kRB; ++k) {
                T val = in[k] ? v[k] : T(0);
                if (tb) {
                    const T d = val - (in[k] ? tv[k] : T(0));
                    if (owned[k] && first_part) sq_sum += (double)d * (double)d;
                    val = rs.scale * d;
                }
                if (lrow[k] >= 0) tile_g[lrow[k]] = val;
                if (owned[k] && first_part) bg_sum += (double)val;
            }
        }
        };
#if defined(DPR_ABL) && DPR_ABL == 6  // ablation: the ds_dout tile is not staged (timing only, wrong results)
        (void)stage_rows;
#else
        if ((item.part_nparts >> 16) > 1) stage_rows(std::false_type{});
        else stage_rows(std::true_type{});
#endif
        // the x == TX column (halo cells only): one cell per row
        for (int row = threadIdx.x; row < ROWS; row += GT) {
            const int l1 = row % (TY + 1), l2 = (NO == 3) ? row / (TY + 1) : 0;
            const int g0 = x0[0] + TX, g1 = x0[1] + l1, g2 = (NO == 3) ? x0[NO - 1] + l2 : 0;
            const bool in = g0 < gd.n[0] && g1 < gd.n[1] && (NO == 2 || g2 < gd.n[NO - 1]);
            const size_t off = ((NO == 3) ? (size_t)g2 * gd.n[1] + g1 : (size_t)g1) * gd.n[0] + g0;
            T val = in ? gb[off] : T(0);
            if (tb) val = in ? rs.scale * (val - tb[off]) : T(0);
            tile_g[row * (TX + 1) + TX] = val;
        }
    }
    if (b != b0) ps = load_pose<T, NI, NO>(rot, trans, ow, b);  // later image of a pose group
    lds_barrier();  // LDS phases only: prefetched records stay in flight
    // per-thread sums of the per-pose scalars: T within the thread (few records each),
    // f64 across threads / tiles
    T vals[NVAL - 2];
#pragma unroll
    for (int k = 0; k < NVAL - 2; ++k) vals[k] = T(0);
    // the first record was requested before the staging: it has arrived
    pin_record(nxt);
    if (HAS_PW && !UNPERM) pin_value(nxt_idx);
    drain_vmem();
#if defined(DPR_ABL) && DPR_ABL == 5  // ablation: no record loop (dispatch, ds_dout staging, reductions only)
    r = r1;
#endif
    while (r < r1) {
        const Rec4<T> rc = nxt;
        const uint32_t p = HAS_PW ? nxt_idx : slot_to_idx(rc.v[3]);
        const uint32_t rcur = r;
        r += GT;
        {
            const uint32_t rl = r < r1 ? r : r1 - 1;  // clamped prefetch
            nxt = rec[rl];
            if (HAS_PW && !UNPERM) nxt_idx = rec_idx[rl];
        }
        T pt[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) pt[j] = rc.v[j];
        const T pwi = HAS_PW ? rc.v[3] : T(1);
        int ref0[NO];
        T dlo[NO];
        ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo);
        // Branch-free neighbour loop: cells beyond the grid were staged as 0 (they add
        // nothing, which equals dropping them, src/raster_pullback.jl:51); a lower neighbour
        // at -1 reads cell 0 and is zeroed.
        int lb[NO];
        bool low_ok[NO];
#pragma unroll
        for (int d = 0; d < NO; ++d) {
            lb[d] = ref0[d] - x0[d];
            // records of this tile have lb in [-1, T-1]; the clamp only matters if the caller
            // breaks the REUSE_BINNING contract (stale workspace) and keeps LDS indices legal
            lb[d] = lb[d] < -1 ? -1 : (lb[d] > TileDims<NO>::T[d] - 1 ? TileDims<NO>::T[d] - 1 : lb[d]);
            low_ok[d] = lb[d] >= 0;
        }
        T gv[1 << NO];
        bool interior = true;
#pragma unroll
        for (int d = 0; d < NO; ++d) interior = interior && low_ok[d];
#if defined(DPR_ABL) && DPR_ABL == 4  // ablation: the record loop WITHOUT its 2^N LDS reads (timing only, wrong results)
        if (interior) {
#pragma unroll
            for (int s = 0; s < (1 << NO); ++s) gv[s] = (T)(lds_index<NO>(lb) + s);
        } else {
#else
        if (interior) {
            // common case: one base address, neighbours at compile-time offsets
            const T* base = &tile_g[lds_index<NO>(lb)];
#pragma unroll
            for (int s = 0; s < (1 << NO); ++s) gv[s] = base[nbr_lds_offset<NO>(s)];
        } else {
#endif
#pragma unroll
            for (int s = 0; s < (1 << NO); ++s) {
                int l[NO];
                bool ok = true;
#pragma unroll
                for (int d = 0; d < NO; ++d) {
                    const int sd = (s >> d) & 1;
                    ok = ok && (sd || low_ok[d]);
                    l[d] = (sd || low_ok[d]) ? lb[d] + sd : 0;
                }
                const T gi = tile_g[lds_index<NO>(l)];
                gv[s] = ok ? gi : T(0);
            }
        }
        T scaled[NO], dow_part = T(0), dpw_part = T(0);
        {
            T dcoord[NO];
#pragma unroll
            for (int n = 0; n < NO; ++n) dcoord[n] = T(0);
#pragma unroll
            for (int s = 0; s < (1 << NO); ++s) {
                const T gi = gv[s];
                const T dweight = voxel_weight<T, NO>(dlo, s, gi);  // :55
                dow_part += dweight * pwi;                          // :57
                dpw_part += dweight * ps.ow;                        // :58
                const T factor = gi * ps.ow * pwi;                  // :60
#pragma unroll
                for (int n = 0; n < NO; ++n) dcoord[n] += factor * interp_weight<T, NO>(n, dlo, s);
            }
#pragma unroll
            for (int n = 0; n < NO; ++n) scaled[n] = dcoord[n] * (T(gd.n[n]) / T(2));  // :67
        }
#pragma unroll
        for (int n = 0; n < NO; ++n) {
#pragma unroll
            for (int j = 0; j < NI; ++j) vals[n + j * NO] += scaled[n] * pt[j];  // :69
            vals[NO * NI + n] += scaled[n];                                     // :68
        }
        vals[NO * NI + NO] += dow_part;
        T dp[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {  // rotation' * scaled (:70)
            T v = ps.R[0 + j * NO] * scaled[0];
#pragma unroll
            for (int n = 1; n < NO; ++n) v = v + ps.R[n + j * NO] * scaled[n];
            dp[j] = v;
        }
        if (UNPERM) {
            Rec4<T> gr;
#pragma unroll
            for (int j = 0; j < 3; ++j) gr.v[j] = (j < NI) ? dp[(j < NI) ? j : 0] : T(0);
            gr.v[3] = dpw_part;
            rec[rcur] = gr;
        } else if (FIRST_POSE) {  // this thread is the only writer of point p for this pose
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[(size_t)p * NI + j] = dp[j];
            if (ds_dpw) ds_dpw[p] = dpw_part;
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[(size_t)p * NI + j] += dp[j];
            if (ds_dpw) ds_dpw[p] += dpw_part;
        }
    }
    // per-tile partial sums of the per-pose scalars (f64), reduced later by k_pose_reduce
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < NVAL; ++k) {
        const double v = (k < NVAL - 2) ? (double)vals[k < NVAL - 2 ? k : 0]
                                        : (k == NVAL - 2 ? bg_sum : sq_sum);
        const double s = wave_sum<double>(v);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < (rs.target ? NVAL : NVAL - 1)) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w][threadIdx.x];
        partials[(size_t)threadIdx.x * max_items + blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------ pullback K4, local binning
// (k_tile_gather with the record loop walking run descriptors; instantiated with RUNS = true)
// UNPERM: the per-point gradient {d point, d point_weight} overwrites the point's record in
// place (coalesced 16/32-byte stores in binned order); k_unpermute then brings it back to the
// original order with one random read per point.  !UNPERM: the owner thread stores straight
// to ds_dpoints[idx] / ds_dpoint_weight[idx] (good when the input order is spatially coherent).
template <typename T, int NI, int NO, bool HAS_PW, bool FIRST_POSE, bool UNPERM, bool RUNS>
__global__ __launch_bounds__(gather_threads<T>(), gather_waves_per_simd<T>()) void k_tile_gather_runs(
    GridDesc<NO> gd, TileGeom<NO> tg, Rec4<T>* rec, const RunDesc* __restrict__ runs, int64_t P,
    const uint32_t* __restrict__ rec_idx, const WorkItem* __restrict__ items,
    const uint32_t* __restrict__ n_items, int max_items, const T* __restrict__ g,
    const T* __restrict__ rot, const T* __restrict__ trans, const T* __restrict__ ow, int64_t b0,
    T* __restrict__ ds_dpoints, T* __restrict__ ds_dpw, double* __restrict__ partials,
    Residual<T> rs, BinHeader want, BinHeader* hdr) {
    constexpr int GT = gather_threads<T>();
    constexpr int NVH = tile_voxels_halo<NO>();
    constexpr int NVAL = NO * NI + NO + 3;  // dR | dt | d out_weight | d background | loss
    constexpr int NW = GT / kWave;
    __shared__ T tile_g[NVH];
    __shared__ double red[NW][NVAL];
    if (want.magic) {
        // DPR_FLAG_REUSE_BINNING: trust the lists in the workspace only if a KEEP_BINNING forward
        // with the same problem, buffers and pose wrote them and nobody has consumed them since
        const bool ok = header_matches(hdr, want, (const uint32_t*)(rot + b0 * (NO * NI)),
                                       NO * NI * (int)(sizeof(T) / 4),
                                       (const uint32_t*)(trans + b0 * NO), NO * (int)(sizeof(T) / 4));
        if (blockIdx.x == 0 && threadIdx.x == 0) hdr->verdict = ok ? 1u : 0u;
        if (!ok) return;  // k_unpermute / k_pose_reduce turn the verdict into NaN outputs
    }
    // item, item count and the first image's pose are requested together (see k_tile_splat)
    const uint32_t n_it = *n_items;
    WorkItem item = items[blockIdx.x];
    Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b0);
    if (blockIdx.x >= n_it) return;  // the grid is sized for the worst case
    // record ranges never leave the record buffer, whatever the lists say (RUNS: load_runs)
    if (!RUNS && item.end > (uint32_t)P) item.end = (uint32_t)P;
    if (item.begin > item.end) item.begin = item.end;
    const int tile = (int)(item.tile % (uint32_t)tg.NT);
    const int64_t b = b0 + (int64_t)(item.tile / (uint32_t)tg.NT);
    int x0[NO], tc[NO];
    tile_origin<NO>(tile, tg, x0, tc);
    const T* gb = g + b * gd.G;
    const T* tb = rs.target ? rs.target + b * gd.G : nullptr;
    // RUNS (local binning): the item is a range of run descriptors (see k_tile_splat); the first
    // round's table is built here so that the first record is in flight during the staging.
    __shared__ RunTable<RUNS ? kMaxRunsGather : 1> rt;
    uint32_t d0 = RUNS ? item.begin : 0u, n_round = 0;
    int nruns = 0;
    RunCursor cu{0, 0, 0};
    if constexpr (RUNS) {
        const uint32_t d1 = (d0 + kMaxRunsGather < item.end) ? d0 + kMaxRunsGather : item.end;
        nruns = (int)(d1 - d0);
        n_round = load_runs(rt, runs, d0, d1, (uint32_t)P);
    }
    // threads stride over the records (RUNS: over the round's logical records, mapped through
    // the run table by a forward-only cursor)
    const uint32_t r1 = RUNS ? n_round : item.end;
    uint32_t r = (RUNS ? 0u : item.begin) + threadIdx.x;
    Rec4<T> nxt;
    uint32_t nxt_idx = 0, nxt_phys = 0;
    if (r < r1) {
        if constexpr (RUNS) {
            cu.seek(rt, r, nruns);
            nxt_phys = cu.next(rt, GT, nruns, r + GT < r1);
        } else {
            nxt_phys = r;
        }
        nxt = rec[nxt_phys];
        if (HAS_PW && !UNPERM) nxt_idx = rec_idx[nxt_phys];
    }
    if (UNPERM && blockIdx.x == 0 && threadIdx.x == 0) {
        Rec4<T> z;
        z.v[0] = z.v[1] = z.v[2] = z.v[3] = T(0);
        rec[P] = z;  // spare slot: the gradient of every rejected point
    }
    // Stage the ds_dout tile + upper halo in LDS, one row (TX + 1 cells along x) at a time: the
    // row's y / z coordinates, bounds and base offset are computed once per row, a lane only
    // adds its x; kRB rows are in flight per wave.  Cells beyond the grid are staged as 0.
    // Owned voxels are summed for ds_dbackground (residual mode: ds_dout = scale * (out -
    // target) formed here, squared residuals summed for the loss).
    double bg_sum = 0.0, sq_sum = 0.0;
    {
        constexpr int TX = TileDims<NO>::T[0], TY = TileDims<NO>::T[1];
        constexpr int TZ = (NO == 3) ? TileDims<NO>::T[NO - 1] : 0;
        constexpr int ROWS = NVH / (TX + 1);
        constexpr int RPW = kWave / TX;  // rows per wave pass (1 for TX = 64, 2 for TX = 32)
        constexpr int kRB = DPR_GATHER_RB;  // row passes in flight
        static_assert(kWave % TX == 0, "tile rows must divide the wavefront");
        const bool first_part = (item.part_nparts & 0xffffu) == 0;
        const int lane = threadIdx.x & (kWave - 1);
        const int x = lane % TX;
        const bool x_ok = x0[0] + x < gd.n[0];
        constexpr int STEP = (GT / kWave) * RPW;
        // an unsplit tile's ds_dout cells are read by this block only (plus the neighbours' halo
        // rows): streamed with non-temporal loads; the parts of a split tile re-read them
        auto stage_rows = [&](auto nt_tag) {
        constexpr bool NT = decltype(nt_tag)::value;
        for (int row0 = (threadIdx.x / kWave) * RPW; row0 < ROWS; row0 += STEP * kRB) {
            T v[kRB], tv[kRB];
            bool in[kRB], owned[kRB];
            int lrow[kRB];
#pragma unroll
            for (int k = 0; k < kRB; ++k) {
                int row = row0 + k * STEP + lane / TX;
                if (RPW == 1) row = __builtin_amdgcn_readfirstlane(row);
                const bool live = row < ROWS;
                const int l1 = row % (TY + 1), l2 = (NO == 3) ? row / (TY + 1) : 0;
                const int g1 = x0[1] + l1, g2 = (NO == 3) ? x0[NO - 1] + l2 : 0;
                in[k] = live && x_ok && g1 < gd.n[1] && (NO == 2 || g2 < gd.n[NO - 1]);
                owned[k] = in[k] && l1 < TY && (NO == 2 || l2 < TZ);
                lrow[k] = live ? row * (TX + 1) + x : -1;
                const size_t off =
                    ((NO == 3) ? (size_t)g2 * gd.n[1] + g1 : (size_t)g1) * gd.n[0] + x0[0] + x;
                const size_t oc = in[k] ? off : 0;
                v[k] = NT ? __builtin_nontemporal_load(&gb[oc]) : gb[oc];
                tv[k] = tb ? (NT ? __builtin_nontemporal_load(&tb[oc]) : tb[oc]) : T(0);
            }
#pragma unroll
            for (int k = 0; k < kRB; ++k) {
                T val = in[k] ? v[k] : T(0);
                if (tb) {
                    const T d = val - (in[k] ? tv[k] : T(0));
                    if (owned[k] && first_part) sq_sum += (double)d * (double)d;
                    val = rs.scale * d;
                }
                if (lrow[k] >= 0) tile_g[lrow[k]] = val;
                if (owned[k] && first_part) bg_sum += (double)val;
            }
        }
        };
        if ((item.part_nparts >> 16) > 1) stage_rows(std::false_type{});
        else stage_rows(std::true_type{});
        // the x == TX column (halo cells only): one cell per row
        for (int row = threadIdx.x; row < ROWS; row += GT) {
            const int l1 = row % (TY + 1), l2 = (NO == 3) ? row / (TY + 1) : 0;
            const int g0 = x0[0] + TX, g1 = x0[1] + l1, g2 = (NO == 3) ? x0[NO - 1] + l2 : 0;
            const bool in = g0 < gd.n[0] && g1 < gd.n[1] && (NO == 2 || g2 < gd.n[NO - 1]);
            const size_t off = ((NO == 3) ? (size_t)g2 * gd.n[1] + g1 : (size_t)g1) * gd.n[0] + g0;
            T val = in ? gb[off] : T(0);
            if (tb) val = in ? rs.scale * (val - tb[off]) : T(0);
            tile_g[row * (TX + 1) + TX] = val;
        }
    }
    if (b != b0) ps = load_pose<T, NI, NO>(rot, trans, ow, b);  // later image of a pose group
    lds_barrier();  // LDS phases only: prefetched records stay in flight
    // per-thread sums of the per-pose scalars: T within the thread (few records each),
    // f64 across threads / tiles
    T vals[NVAL - 2];
#pragma unroll
    for (int k = 0; k < NVAL - 2; ++k) vals[k] = T(0);
    {
        uint32_t r1r = r1;
        for (;;) {
            // the round's first record: waited for before the loop
            pin_record(nxt);
            if (HAS_PW && !UNPERM) pin_value(nxt_idx);
            drain_vmem();
            while (r < r1r) {
                const Rec4<T> rc = nxt;
                const uint32_t p = HAS_PW ? nxt_idx : slot_to_idx(rc.v[3]);
                const uint32_t rcur = nxt_phys;
                r += GT;
                // The prefetch is issued unconditionally (past the end: this record again): with a
                // branch around it the compiler cannot count the memory operations of an iteration
                // and waits for EVERYTHING outstanding (s_waitcnt vmcnt(0)), i.e. also for the
                // in-place gradient store of the previous iteration -- 135 instead of ~100 us.
                if (r < r1r) nxt_phys = cu.next(rt, GT, nruns, r + GT < r1r);
                nxt = rec[nxt_phys];
                if (HAS_PW && !UNPERM) nxt_idx = rec_idx[nxt_phys];
                T pt[NI];
        #pragma unroll
                for (int j = 0; j < NI; ++j) pt[j] = rc.v[j];
                const T pwi = HAS_PW ? rc.v[3] : T(1);
                int ref0[NO];
                T dlo[NO];
                ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo);
                // Branch-free neighbour loop: cells beyond the grid were staged as 0 (they add
                // nothing, which equals dropping them, src/raster_pullback.jl:51); a lower neighbour
                // at -1 reads cell 0 and is zeroed.
                int lb[NO];
                bool low_ok[NO];
        #pragma unroll
                for (int d = 0; d < NO; ++d) {
                    lb[d] = ref0[d] - x0[d];
                    // records of this tile have lb in [-1, T-1]; the clamp only matters if the caller
                    // breaks the REUSE_BINNING contract (stale workspace) and keeps LDS indices legal
                    lb[d] = lb[d] < -1 ? -1 : (lb[d] > TileDims<NO>::T[d] - 1 ? TileDims<NO>::T[d] - 1 : lb[d]);
                    low_ok[d] = lb[d] >= 0;
                }
                T gv[1 << NO];
                bool interior = true;
        #pragma unroll
                for (int d = 0; d < NO; ++d) interior = interior && low_ok[d];
                if (interior) {
                    // common case: one base address, neighbours at compile-time offsets
                    const T* base = &tile_g[lds_index<NO>(lb)];
        #pragma unroll
                    for (int s = 0; s < (1 << NO); ++s) gv[s] = base[nbr_lds_offset<NO>(s)];
                } else {
        #pragma unroll
                    for (int s = 0; s < (1 << NO); ++s) {
                        int l[NO];
                        bool ok = true;
        #pragma unroll
                        for (int d = 0; d < NO; ++d) {
                            const int sd = (s >> d) & 1;
                            ok = ok && (sd || low_ok[d]);
                            l[d] = (sd || low_ok[d]) ? lb[d] + sd : 0;
                        }
                        const T gi = tile_g[lds_index<NO>(l)];
                        gv[s] = ok ? gi : T(0);
                    }
                }
                T scaled[NO], dow_part = T(0), dpw_part = T(0);
                {
                    T dcoord[NO];
        #pragma unroll
                    for (int n = 0; n < NO; ++n) dcoord[n] = T(0);
        #pragma unroll
                    for (int s = 0; s < (1 << NO); ++s) {
                        const T gi = gv[s];
                        const T dweight = voxel_weight<T, NO>(dlo, s, gi);  // :55
                        dow_part += dweight * pwi;                          // :57
                        dpw_part += dweight * ps.ow;                        // :58
                        const T factor = gi * ps.ow * pwi;                  // :60
        #pragma unroll
                        for (int n = 0; n < NO; ++n) dcoord[n] += factor * interp_weight<T, NO>(n, dlo, s);
                    }
        #pragma unroll
                    for (int n = 0; n < NO; ++n) scaled[n] = dcoord[n] * (T(gd.n[n]) / T(2));  // :67
                }
        #pragma unroll
                for (int n = 0; n < NO; ++n) {
        #pragma unroll
                    for (int j = 0; j < NI; ++j) vals[n + j * NO] += scaled[n] * pt[j];  // :69
                    vals[NO * NI + n] += scaled[n];                                     // :68
                }
                vals[NO * NI + NO] += dow_part;
                T dp[NI];
        #pragma unroll
                for (int j = 0; j < NI; ++j) {  // rotation' * scaled (:70)
                    T v = ps.R[0 + j * NO] * scaled[0];
        #pragma unroll
                    for (int n = 1; n < NO; ++n) v = v + ps.R[n + j * NO] * scaled[n];
                    dp[j] = v;
                }
                if (UNPERM) {
                    Rec4<T> gr;
        #pragma unroll
                    for (int j = 0; j < 3; ++j) gr.v[j] = (j < NI) ? dp[(j < NI) ? j : 0] : T(0);
                    gr.v[3] = dpw_part;
                    rec[rcur] = gr;
                } else if (FIRST_POSE) {  // this thread is the only writer of point p for this pose
        #pragma unroll
                    for (int j = 0; j < NI; ++j) ds_dpoints[(size_t)p * NI + j] = dp[j];
                    if (ds_dpw) ds_dpw[p] = dpw_part;
                } else {
        #pragma unroll
                    for (int j = 0; j < NI; ++j) ds_dpoints[(size_t)p * NI + j] += dp[j];
                    if (ds_dpw) ds_dpw[p] += dpw_part;
                }
            }
            d0 += kMaxRunsGather;
            if (d0 >= item.end) break;  // uniform
            __syncthreads();  // next round of descriptors: rebuild the table
            const uint32_t d1 = (d0 + kMaxRunsGather < item.end) ? d0 + kMaxRunsGather : item.end;
            nruns = (int)(d1 - d0);
            r1r = load_runs(rt, runs, d0, d1, (uint32_t)P);
            r = threadIdx.x;
            if (r < r1r) {
                cu.seek(rt, r, nruns);
                nxt_phys = cu.next(rt, GT, nruns, r + GT < r1r);
                nxt = rec[nxt_phys];
                if (HAS_PW && !UNPERM) nxt_idx = rec_idx[nxt_phys];
            }
        }
    }
    // per-tile partial sums of the per-pose scalars (f64), reduced later by k_pose_reduce
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < NVAL; ++k) {
        const double v = (k < NVAL - 2) ? (double)vals[k < NVAL - 2 ? k : 0]
                                        : (k == NVAL - 2 ? bg_sum : sq_sum);
        const double s = wave_sum<double>(v);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < (rs.target ? NVAL : NVAL - 1)) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w][threadIdx.x];
        partials[(size_t)threadIdx.x * max_items + blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------ pullback K5 (body)
// partials[NVAL][items] (f64) -> the per-pose outputs: one block of 1024 threads per (scalar k,
// pose j of the group); an item belongs to pose item.tile / NT.  Runs as k_pose_reduce or, for a
// single pose, as the first blocks of k_unpermute (one launch and its gap less per pullback).
template <typename T> struct PoseReduceArgs {
    const double* partials;
    const WorkItem* items;
    const uint32_t* n_items;
    int max_items, NT, n_in, n_out;
    int64_t b0;
    T* ds_drotation;
    T* ds_dtranslation;
    T* ds_dbackground;
    T* ds_dout_weight;
    T* loss;
    BinHeader* hdr;
    int accumulate;  // 1: add to the outputs (a later slab of the same pose)
};
template <typename T>
__device__ __forceinline__ void pose_reduce_body(int k, uint32_t j, bool grouped,
                                                 const PoseReduceArgs<T>& a) {
    __shared__ double wsum[16];
    const int64_t b = a.b0 + j;
    double s = 0.0;
    const bool stale = a.hdr && a.hdr->verdict != 1u;  // see k_tile_gather
    // the binning is consumed: the gradient records have overwritten the point records
    if (a.hdr && k == 0 && j == 0 && threadIdx.x == 0) a.hdr->state = 0u;
    const int n = stale ? 0 : (int)*a.n_items;
    if (!grouped) {
        for (int t = threadIdx.x; t < n; t += 1024) s += a.partials[(size_t)k * a.max_items + t];
    } else {
        for (int t = threadIdx.x; t < n; t += 1024)
            if (a.items[t].tile / (uint32_t)a.NT == j) s += a.partials[(size_t)k * a.max_items + t];
    }
    s = wave_sum<double>(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        if (stale) tot = __builtin_nan("");
        const int nr = a.n_out * a.n_in;
        T* dst = nullptr;
        if (k < nr)
            dst = &a.ds_drotation[b * nr + k];
        else if (k < nr + a.n_out)
            dst = &a.ds_dtranslation[b * a.n_out + (k - nr)];
        else if (k == nr + a.n_out)
            dst = &a.ds_dout_weight[b];
        else if (k == nr + a.n_out + 1)
            dst = &a.ds_dbackground[b];
        else if (a.loss)
            dst = &a.loss[b];
        if (dst) *dst = a.accumulate ? (T)((double)*dst + tot) : (T)tot;
    }
}

// ------------------------------------------------------------------ pullback un-permute
// thread per ORIGINAL point: gradient record(s) of its slot(s) -> ds_dpoints /
// ds_dpoint_weight (coalesced stores; accumulating over poses when !FIRST_POSE).  A pose
// group contributes nb records per point, summed here in registers.
template <typename T, int NI, bool FIRST_POSE, int kUPB>
__global__ __launch_bounds__(1024) void k_unpermute(int64_t P, int nb,
                                                    const Rec4<T>* __restrict__ grad,
                                                    const uint32_t* __restrict__ slot_of,
                                                    T* __restrict__ ds_dpoints,
                                                    T* __restrict__ ds_dpw,
                                                    const BinHeader* __restrict__ hdr,
                                                    int reduce_blocks, PoseReduceArgs<T> pr,
                                                    size_t pose_stride) {
    // pose_stride == 0: the nb records of a point are the poses of a GROUP -- one record buffer,
    //   slot maps P entries apart.  pose_stride > 0: the nb poses of a batch whose binning was
    //   KEPT per pose (Plan::pose_stride): record buffers, slot maps and headers pose_stride
    //   bytes apart; all poses are summed here in one pass instead of one read-modify-write of
    //   the gradient buffer per pose (C5's share: 0.62 + 7 x 0.89 ms of un-permutes per pullback).
    // the first `reduce_blocks` blocks are the per-pose reduction (single pose only)
    if ((int)blockIdx.x < reduce_blocks) {
        pose_reduce_body<T>((int)blockIdx.x, 0u, false, pr);
        return;
    }
    const unsigned ublock = blockIdx.x - (unsigned)reduce_blocks;
    const unsigned ublocks = gridDim.x - (unsigned)reduce_blocks;
    bool stale = hdr && hdr->verdict != 1u;
    if (hdr && pose_stride)  // a batch: every pose's header must have matched
        for (int j = 1; j < nb; ++j)
            stale = stale || ((const BinHeader*)((const char*)hdr + (size_t)j * pose_stride))->verdict != 1u;
    if (stale) {
        // REUSE_BINNING without a matching KEEP_BINNING forward: no gradient was computed
        const T nan = T(__builtin_nanf(""));
        const int64_t b0 = (int64_t)ublock * (kUPB * 1024) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < kUPB; ++k) {
            const int64_t p = b0 + k * 1024;
            if (p >= P) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[p * NI + j] = nan;
            if (ds_dpw) ds_dpw[p] = nan;
        }
        return;
    }
    // One block covers kUPB * 1024 consecutive points = one sub-chunk of the scatter: points of
    // a sub-chunk that fell into the same tile sit next to each other in that tile's record
    // run, so the 64-byte sectors this block fetches are shared among its own threads.
    const int64_t base = (int64_t)xcd_slice(ublock, ublocks) * (kUPB * 1024) + threadIdx.x;
    uint32_t slot[kUPB];
#pragma unroll
    for (int k = 0; k < kUPB; ++k) {
        const int64_t p = base + k * 1024;
        slot[k] = __builtin_nontemporal_load(&slot_of[p < P ? p : P - 1]);
    }
    Rec4<T> g[kUPB];
#pragma unroll
    for (int k = 0; k < kUPB; ++k) g[k] = grad[slot[k]];
    for (int j = 1; j < nb; ++j) {
        const Rec4<T>* gradj =
            pose_stride ? (const Rec4<T>*)((const char*)grad + (size_t)j * pose_stride) : grad;
        const uint32_t* slotj = pose_stride
                                    ? (const uint32_t*)((const char*)slot_of + (size_t)j * pose_stride)
                                    : slot_of + (size_t)j * P;
#pragma unroll
        for (int k = 0; k < kUPB; ++k) {
            const int64_t p = base + k * 1024;
            const Rec4<T> gj = gradj[slotj[p < P ? p : P - 1]];
#pragma unroll
            for (int c = 0; c < 4; ++c) g[k].v[c] += gj.v[c];
        }
    }
#pragma unroll
    for (int k = 0; k < kUPB; ++k) {
        const int64_t p = base + k * 1024;
        if (p >= P) continue;
        if (FIRST_POSE) {
#pragma unroll
            for (int j = 0; j < NI; ++j) __builtin_nontemporal_store(g[k].v[j], &ds_dpoints[p * NI + j]);
            if (ds_dpw) __builtin_nontemporal_store(g[k].v[3], &ds_dpw[p]);
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[p * NI + j] += g[k].v[j];
            if (ds_dpw) ds_dpw[p] += g[k].v[3];
        }
    }
}

// ------------------------------------------------------------------ pullback K5
// (pose groups and the direct-store mode: the reduction as a launch of its own)
template <typename T>
__global__ __launch_bounds__(1024) void k_pose_reduce(PoseReduceArgs<T> pr) {
    pose_reduce_body<T>((int)blockIdx.x, blockIdx.y, gridDim.y > 1, pr);
}

// gradients of an internally sorted cloud back to the caller's order, GATHER form: thread q of
// the caller's order reads sorted element inv_perm[q] (random reads, coalesced stores; the
// scatter form dst[perm[i]] = src[i] took 3.8 ms for the 50 M fp64 points of C5's share)
template <typename T, int NI>
__global__ __launch_bounds__(256) void k_unsort(int64_t P, const uint32_t* __restrict__ inv_perm,
                                                const T* __restrict__ dp_sorted,
                                                const T* __restrict__ dpw_sorted,
                                                T* __restrict__ ds_dpoints, T* __restrict__ ds_dpw,
                                                const BinHeader* __restrict__ hdr) {
    constexpr int PER = 4;
    const int64_t q0 = (int64_t)blockIdx.x * (256 * PER) + threadIdx.x;
    if (hdr && hdr->verdict != 1u) {
        // REUSE_BINNING without the matching KEEP_BINNING forward: the permutation in the
        // workspace is not this call pair's -- nothing is read through it, NaN gradients
        const T nan = T(__builtin_nanf(""));
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t q = q0 + k * 256;
            if (q >= P) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[q * NI + j] = nan;
            if (ds_dpw) ds_dpw[q] = nan;
        }
        return;
    }
    size_t i[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t q = q0 + k * 256;
        const size_t v = inv_perm[q < P ? q : P - 1];
        i[k] = v < (size_t)P ? v : 0;  // (never out of range for an inverse this library wrote)
    }
    T v[PER][NI], w[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
#pragma unroll
        for (int j = 0; j < NI; ++j) v[k][j] = dp_sorted[i[k] * NI + j];
        w[k] = ds_dpw ? dpw_sorted[i[k]] : T(0);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t q = q0 + k * 256;
        if (q >= P) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) ds_dpoints[q * NI + j] = v[k][j];
        if (ds_dpw) ds_dpw[q] = w[k];
    }
}

// ------------------------------------------------------------------ host side
static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Experiment knobs: in builds with -DDPR_EXPERIMENTS environment overrides of the compiled-in defaults, read
// ONCE per process (function-local static: thread-safe, no getenv on the call path) and clamped to valid
// ranges; in the shipped library the defaults (env_knob, dpr_tiled.h).
struct Knobs {
    int cap3d_div, cap2d_div, cap_min, pose_group, scatter_wc, bwd_unpermute, compact_records,
        splat_blocked, fixed_point, fuse_tilescan, bin_direct_store;
};
static const Knobs& knobs() {
    static const Knobs k = [] {
        auto env_int = [](const char* name, int dflt, int lo, int hi) { return env_knob(name, dflt, lo, hi); };
        Knobs q;
        q.cap3d_div = env_int("DPR_CAP3D_DIV", 256, 1, 1 << 20);
        q.cap2d_div = env_int("DPR_CAP2D_DIV", 2048, 0, 1 << 20);  // 0: use the 3-D rule
        q.cap_min = env_int("DPR_CAP_MIN", 4096, 256, 1 << 24);
        q.pose_group = env_int("DPR_POSE_GROUP", 16, 1, 16);
        q.scatter_wc = env_int("DPR_SCATTER_WC", 1, 0, 1);
        q.bwd_unpermute = env_int("DPR_BWD_UNPERMUTE", 1, 0, 1);
        q.compact_records = env_int("DPR_COMPACT_RECORDS", 1, 0, 1);
        q.splat_blocked = env_int("DPR_SPLAT_BLOCKED", 2, 0, 2);  // 2: decided on the device
        q.fixed_point = env_int("DPR_FIXED_POINT", 1, 0, 1);  // 0: f64 LDS accumulators for fp32 data too
        q.fuse_tilescan = env_int("DPR_FUSE_TILESCAN", 1, 0, 1);  // 0: k_tilescan as a launch of its own
        q.bin_direct_store = env_int("DPR_BIN_DIRECT_STORE", 1, 0, 1);  // 0: fp64 batches stage records in LDS too
        return q;
    }();
    return k;
}

// compute units of the current device (cached)
static int cu_count() {
    static std::mutex mu;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// Workspace layout (identical for raster and pullback so that a pullback can reuse the
// binning a raster call left behind, DPR_FLAG_KEEP_BINNING / DPR_FLAG_REUSE_BINNING):
//   counts table | totals | tile_start | work items, n_items, tile_parts, tile_slab | records | indices | slot_of | aux (halo / partials)
struct Plan {
    int bg;          // poses binned together (pose group, a power of two; 1 = per-pose pipeline)
    int nblk;
    int64_t chunk;
    uint32_t cap;    // records per work item above which a tile is split
    int max_items;   // NT + worst-case number of extra parts
    int max_slabs;   // overflow slabs (parts of split tiles)
    size_t off_hdr, off_counts, off_totals, off_tile_start, off_items, off_nitems, off_nzbins, off_tparts, off_tslab,
        off_split, off_rec, off_idx, off_slot, off_aux, total;
    // Cell sort of the cloud inside the call (dpr_coarse.h; batched poses on grids with more than
    // 4096 tiles): what local binning of all poses of a batch needs; up to 16384 tiles per pose --
    // beyond that the plain count / scatter pipeline runs on the cell-sorted copy
    bool sort_inside;
    size_t off_spts, off_spw, off_perm, off_iperm, off_sgrad, off_sgradw, off_sorttmp;
    // KEEP_BINNING / REUSE_BINNING with B > 1: every pose owns a copy of the per-pose part of the
    // layout (header ... slot map), pose_stride bytes apart, so that the pullback finds the binning
    // of EVERY pose of the forward call; 0 when the poses share one copy (nothing is kept)
    size_t pose_stride;
    // local binning (DPR_FLAG_COHERENT_POINTS, or the cloud cell-sorted inside the call; NT <= 16384)
    bool local;
    int lb;                // poses binned per k_bin_local launch (each into its own copy of the
                           // per-pose workspace: `copies` of them, pose_stride bytes apart)
    int64_t copies;
    int sub;               // points per sub-chunk
    int64_t nsub;          // sub-chunks = blocks of k_bin_local
    int64_t max_desc;      // descriptor slots: `sub` per sub-chunk
    size_t off_ltot, off_dstart, off_dcursor, off_bdesc, off_desc, off_sdesc;  // ltot: ndesc[NT] | npts[NT] | max|pw| | ~min|pw|
};

// Pose groups: with few tiles per pose (2-D projections, small 3-D grids) the bins become
// (pose, tile) pairs of up to kMaxGroup poses, as long as they fit the write-combining
// scatter's 4096 LDS cursors: the points are read once per group instead of once per pose and
// the fixed per-launch costs (scans, halo pass, reductions, launch gaps) are shared.  Measured
// (tools/pose_group_probe.py): 10 M points -> 512^2, 485 -> 383 us per pose (fwd + bwd);
// 1 M points -> 128^3, 149 -> 65 us per pose.
// Memory: records and slot map are sized P * g (20 / 36 bytes per point-pose), so a group of g
// poses multiplies that part of the workspace by g -- bounded by P * g <= 2^27 (2.7 GB fp32,
// 4.8 GB fp64) and by the caller through DPR_FLAG_MAX_POSE_GROUP(n) (include/dpr.h).
constexpr int kMaxGroup = 16;
static int pose_group(int NT, int64_t P, int64_t B, int max_group) {
    int limit = knobs().pose_group;
    if (max_group > 0 && max_group < limit) limit = max_group;
    int bg = 1;
    while (bg * 2 <= limit && bg * 2 <= kMaxGroup && bg * 2 <= B && NT * bg * 2 <= 4096 &&
           P * bg * 2 <= ((int64_t)1 << 27))  // records of a group: <= 2 GiB (fp32)
        bg *= 2;
    return bg;
}

static Plan make_plan(size_t elem, int n_out, int NT1, int64_t P1, int64_t B, int max_group,
                      bool coherent = false, int n_in = 3, bool share_batch = false,
                      bool slabbed = false, bool fwd_only = false) {
    Plan pl;
    if (slabbed) coherent = false;  // local binning keeps a batch's bins: one slab at a time cannot
    share_batch = share_batch && B > 1;
    if (share_batch) max_group = 1;  // a kept binning is per pose
    pl.pose_stride = 0;
    pl.sort_inside = !coherent && NT1 > 4096 && B >= 4 && P1 >= 200000;
    // (the direct-store pullback mode, an experiment knob, needs the index array only the plain
    // scatter writes)
    // ... and local binning is per pose: a batch that forms pose groups (few tiles) keeps the
    // grouped pipeline, which reads the points once per group (10 M points -> 512^2, 4 poses:
    // 0.56 ms grouped, 0.63 ms pose by pose on local bins)
    pl.bg = pose_group(NT1, P1, B, max_group);
    pl.local = (coherent || pl.sort_inside) && !slabbed && NT1 <= kMaxLocalTiles &&
               knobs().bwd_unpermute && pl.bg == 1;
    // poses binned by one k_bin_local launch (the points are read once for all of them): every
    // pose of a kept batch, else up to 8 -- each needs its own records, P * lb <= 2^29
    pl.lb = 1;
    if (pl.local && B > 1) {
        if (share_batch) {
            pl.lb = B < 16 ? (int)B : 16;  // (the B copies exist anyway)
        } else {
            pl.lb = B < 8 ? (int)B : 8;
            while (pl.lb > 1 && P1 * pl.lb > ((int64_t)1 << 29)) --pl.lb;
        }
    }
    pl.copies = share_batch ? B : pl.lb;
    const int NT = NT1 * pl.bg;          // bins
    const int64_t P = P1 * pl.bg;        // records
    // Slices of the cloud = blocks of k_count / the scatter = rows of the counts table.
    int64_t nblk, chunk;
    if (DPR_WC_SLICES && NT <= 4096) {
        // write-combining scatter (one workgroup per CU: its LDS): at most one slice per CU, so
        // that all of them run at once, and whole sub-chunks per slice (a partly filled round costs
        // as much as a full one; 3e6 points: 489 slices of 1.5 rounds -> 245 of 3: 0.131 -> 0.124 ms)
        const int64_t sub = (elem == 4) ? DPR_WC_PPT * kWcThreads : DPR_WC_PPT * kWcThreads / 2;
        chunk = ((P1 + 255) / 256 + sub - 1) / sub * sub;
        if (chunk < sub) chunk = sub;
    } else {
        nblk = (P1 + 8191) / 8192;
        if (nblk < 1) nblk = 1;
        if (nblk > kMaxBinBlocks) nblk = kMaxBinBlocks;
        chunk = (P1 + nblk - 1) / nblk;
        chunk = (chunk + kBinThreads - 1) / kBinThreads * kBinThreads;
        if (chunk < kBinThreads) chunk = kBinThreads;
    }
    nblk = (P1 + chunk - 1) / chunk;
    if (nblk < 1) nblk = 1;
    pl.nblk = (int)nblk;
    pl.chunk = chunk;
    size_t o = 0;
    pl.off_hdr = o;  // BinHeader: what a KEEP_BINNING forward left, checked by a REUSE pullback
    o += align_up(sizeof(BinHeader));
    pl.off_counts = o;
    o += align_up((size_t)nblk * NT * 4);
    pl.off_totals = o;
    o += align_up((size_t)NT * 4);
    pl.off_tile_start = o;
    o += align_up((size_t)(NT + 1) * 4);
    // split threshold: ~P/256 records (even a fully clustered cloud then yields >= 256 items,
    // one per CU, while the headline Gaussian cloud has no tile above it), at least 4096; a
    // split tile's parts hold more than cap/2 records each
    // (a forward call that keeps nothing for a pullback splits later: the parts of a split tile
    // cost the halo pass more than a 2x longer item costs the fixed-point tile kernel -- 1 M points
    // -> 128^3: forward 0.065 -> 0.059 ms; the pullback's gather prefers the finer split)
    int64_t cap = P / knobs().cap3d_div;
    const int64_t cap_min = fwd_only ? 2 * (int64_t)knobs().cap_min : knobs().cap_min;
    if (cap < cap_min) cap = cap_min;
    if (n_out == 2) {
        // 2-D grids have few tiles (256 at 512^2) with cheap LDS tiles (8.7 KB): split
        // earlier so that a dense projection still gives the chip ~2048 items
        cap = knobs().cap2d_div > 0 ? P / knobs().cap2d_div : cap;
        if (cap < 2048) cap = 2048;
    }
    pl.cap = (uint32_t)cap;
    pl.max_slabs = (int)(2 * ((P + cap - 1) / cap) + 1);
    pl.max_items = NT + pl.max_slabs;
    pl.off_items = o;
    o += align_up((size_t)pl.max_items * sizeof(WorkItem));
    pl.off_nitems = o;  // [0] = items, [1] = record assignment of k_tile_splat, [2] / [3] = max / ~min |point_weight| bits
    o += align_up(4);
    pl.off_nzbins = o;  // bins touched per count block (k_count -> k_tilescan)
    o += align_up((size_t)kMaxBinBlocks * 4);
    pl.off_tparts = o;
    o += align_up((size_t)NT * 4);
    pl.off_tslab = o;
    o += align_up((size_t)NT * 4);
    pl.off_split = o;  // [0] = n_split, [1..] = split tile ids (at most max_slabs / 2)
    o += align_up((size_t)(pl.max_slabs / 2 + 2) * 4);
    pl.sub = elem == 4 ? 4096 : 2048;
    pl.nsub = (P1 + pl.sub - 1) / pl.sub;
    if (pl.nsub < 1) pl.nsub = 1;
    pl.max_desc = 0;
    int64_t nrec = P;  // records (+ spare slot for rejected points)
    if (pl.local) {
        pl.max_desc = pl.nsub * pl.sub;  // every sub-chunk owns `sub` descriptor slots ...
        nrec = pl.nsub * pl.sub;         // ... and a slab of `sub` records
        pl.off_ltot = o;  // (the cursors follow the totals directly: one clear covers both)
        o += align_up((size_t)(2 * NT1 + 2) * 4);
        pl.off_dcursor = o;
        o += align_up((size_t)NT1 * 4);
        pl.off_dstart = o;
        o += align_up((size_t)(NT1 + 1) * 4);
        pl.off_bdesc = o;
        o += align_up((size_t)pl.nsub * 4);
        pl.off_desc = o;
        o += align_up((size_t)pl.max_desc * sizeof(RunDesc));
        pl.off_sdesc = o;
        o += align_up((size_t)pl.max_desc * sizeof(RunDesc));
    }
    pl.off_rec = o;
    o += align_up((size_t)(nrec + 1) * 4 * elem);  // + spare slot for rejected points
    pl.off_idx = o;
    o += align_up((size_t)(P1 + 1) * 4);
    pl.off_slot = o;
    o += align_up((size_t)(P + 1) * 4);
    (void)nrec;
    if (pl.copies > 1) {  // everything up to here exists once per pose (of a kept batch / a local batch)
        pl.pose_stride = o;
        o += (size_t)(pl.copies - 1) * pl.pose_stride;
    }
    pl.off_spts = pl.off_spw = pl.off_perm = pl.off_iperm = pl.off_sgrad = pl.off_sgradw = pl.off_sorttmp = o;
    if (pl.sort_inside) {
        pl.off_spts = o;
        o += align_up((size_t)P1 * n_in * elem);
        pl.off_spw = o;
        o += align_up((size_t)P1 * elem);
        pl.off_perm = o;  // (unused since the coarse cell sort: only the inverse is needed)
        pl.off_iperm = o;  // inverse permutation: the un-sort of the gradients gathers through it
        o += align_up((size_t)P1 * 4);
        pl.off_sgrad = o;
        o += align_up((size_t)P1 * n_in * elem);
        pl.off_sgradw = o;
        o += align_up((size_t)P1 * elem);
        pl.off_sorttmp = o;
        o += align_up(coarse_workspace_bytes(elem, P1));
    }
    pl.off_aux = o;
    // aux: forward = halo buffer | overflow slabs ; pullback = per-item partials
    const size_t nvh = (n_out == 3) ? tile_voxels_halo<3>() : tile_voxels_halo<2>();
    const size_t halo = align_up((size_t)NT * ((n_out == 3) ? halo_count<3>() : halo_count<2>()) * elem) +
                        align_up((size_t)pl.max_slabs * nvh * elem);
    const size_t partials = (size_t)pl.max_items * 16 * 8;
    o += align_up(halo > partials ? halo : partials);
    pl.total = o;
    return pl;
}

// Identity of a workspace layout: a REUSE_BINNING pullback must read the lists where -- and in
// the form in which -- the KEEP_BINNING forward wrote them.  The two calls compute their plans
// independently (DPR_FLAG_COHERENT_POINTS, DPR_FLAG_MAX_POSE_GROUP and the environment knobs all
// move regions), so the forward stores this id in the header and the pullback's kernels compare
// it on the device like the rest of the header.
static uint32_t plan_layout_id(const Plan& pl) {
    uint64_t h = 1469598103934665603ull;  // FNV-1a over the fields that place or shape the lists
    auto mix = [&](uint64_t v) {
        for (int i = 0; i < 8; ++i) {
            h ^= (v >> (8 * i)) & 0xffu;
            h *= 1099511628211ull;
        }
    };
    mix(pl.local ? 1 : 0);
    mix((uint64_t)pl.bg);
    mix((uint64_t)pl.sub);
    mix((uint64_t)pl.cap);
    mix(pl.off_items);
    mix(pl.off_rec);
    mix(pl.off_idx);
    mix(pl.off_slot);
    mix(pl.off_aux);
    mix(pl.local ? pl.off_sdesc : 0);
    mix(pl.pose_stride);
    mix(pl.off_iperm);
    mix((uint64_t)pl.lb);
    const uint32_t id = (uint32_t)(h ^ (h >> 32));
    return id ? id : 1u;
}

static bool grid_cut(int n_out, const int64_t* grid, SlabCut* sc) {
    return n_out == 3 ? make_slab_cut<3>(grid, sc) : make_slab_cut<2>(grid, sc);
}

bool tiled_supported(int n_out, const int64_t* grid) {
    SlabCut sc;
    return grid_cut(n_out, grid, &sc);
}

// slabs the tiled path cuts the grid into (1: one piece; 0: not supported)
int tiled_slabs(int n_out, const int64_t* grid) {
    SlabCut sc;
    return grid_cut(n_out, grid, &sc) ? sc.nslab : 0;
}

bool tiled_preferred(int op, int n_out, const int64_t* grid, int64_t P, int64_t B, int64_t G) {
    (void)G;
    if (P >= (int64_t)1 << 32) return false;
    SlabCut sc;
    if (!grid_cut(n_out, grid, &sc)) return false;
    // A cloud that is SPARSE on the grid: the tiled path pays per tile (zeroing and flushing an LDS
    // tile, staging a ds_dout tile: ~25 ns each) whether points fall into it or not, the direct
    // kernels pay per point only.  Crossovers measured on 256^3 ... 768^3 and 2048^2 / 4096^2 with
    // 1e5 ... 1e7 points, 1 and 4 poses (tools/sparse_grid_probe.py, profiles/r04_sparse_grids.txt):
    // forward ~60 points per tile (3-D) / ~48 (2-D), pullback ~320 (3-D) / 150-430 (2-D) -- the
    // direct pullback only READS the cells its points touch.  Below that AUTO regretted up to 2.8x
    // (3e5 points -> 4096^2, pullback) with the thresholds that follow, which were fitted on grids
    // of up to 2048 tiles.
    const int64_t NT_all = (int64_t)sc.per_layer * sc.layers;
    const int64_t per_tile = op == DPR_OP_RASTER ? (n_out == 3 ? 60 : 48)
                                                 : (n_out == 3 ? 320 : (NT_all <= 4096 ? 150 : 430));
    if (P < per_tile * NT_all) return false;
    if (sc.nslab > 1) {
        // More than 32768 tiles (e.g. 1024^3): every slab re-reads the cloud, and the tile kernels
        // write the whole grid -- which the direct path's background fill does as well.  Forward:
        // the LDS tiles beat scattered global atomics from ~1e6 points on (1e7 points -> 1024^3:
        // measured in profiles/r04_experiments.md); the pullback's gathers are reads, the direct
        // kernel keeps them.
        return op == DPR_OP_RASTER && P >= 1000000;
    }
    const int NT = sc.per_layer * sc.layers;
    // Measured crossovers (profiles/r01_algo_sweep.txt: one pose; r01_algo_sweep_batched.txt:
    // 4-64 poses; 64^3 ... 256^3 and 128^2 / 512^2 grids).  One pose: the tiled pipeline's fixed
    // cost (6-7 launches) is repaid from ~2-3e5 points on, forward and backward alike.  Batched
    // poses on a grid that forms pose groups: the fixed cost is shared, the forward pays from
    // ~6e4 points; the direct pullback kernel, which keeps a point in registers across the poses
    // of a slice, stays ahead up to ~3e5 points (~6e5 when the grid is too large for groups).
    const bool grouped = B >= 4 && pose_group(NT, P, B, 0) >= 4;
    // (one pose on a small grid -- up to 256^2 or 128^3: from 1e5 points -- the direct kernel's atomics are
    // at most 1.25x ahead there on a cloud that fills the grid and 2x behind on a clustered one,
    // profiles/r03_auto_regret.txt)
    // (two or three poses: the same per pose -- 1e5 points x 2 poses -> 128^3: tiled 0.058 ms, direct 0.099)
    if (op == DPR_OP_RASTER && B < 4 && (NT <= 64 || (n_out == 3 && NT <= 256))) return P >= 100000;
    if (op == DPR_OP_RASTER) return P >= (grouped ? 60000 : 250000);
    if (B >= 4) return P >= (grouped ? 300000 : 600000);
    return P >= 250000;
}

// tiles per pose of the tiled path's geometry (of its largest slab; -1: not supported)
int tiled_tiles(int n_out, const int64_t* grid) {
    SlabCut sc;
    return grid_cut(n_out, grid, &sc) ? slab_max_tiles(sc) : -1;
}

// May a KEEP_BINNING / REUSE_BINNING pair with B > 1 poses share on the tiled path when
// DPR_ALGO_AUTO decides?  Every pose then keeps its own records (Plan::pose_stride): only where
// pose groups are not an option anyway (more than 2048 tiles per pose) and the kept records stay
// below ~21 GB at fp64, ~13 GB at fp32 (P * B <= 2^29: a 4-word record, a slot and an index per
// point and pose; independent of the element type, so that dpr_resolve_algo_ex needs none).  An
// explicit DPR_ALGO_TILED shares for any B.  Never on a grid that is processed in slabs (a slab's
// binning is overwritten by the next one).
bool tiled_batch_share_ok(int n_out, const int64_t* grid, int64_t P, int64_t B) {
    if (B < 2 || P < 1 || P * B > ((int64_t)1 << 29)) return false;
    SlabCut sc;
    if (!grid_cut(n_out, grid, &sc) || sc.nslab != 1) return false;
    return sc.per_layer * sc.layers * 2 > 4096;
}

size_t tiled_workspace_bytes(size_t elem, int op, unsigned flags, int n_in, int n_out,
                             const int64_t* grid, int64_t P, int64_t B) {
    (void)op;
    if (P >= (int64_t)1 << 32) return (size_t)-1;  // refused by raster_tiled / pullback_tiled
    SlabCut sc;
    if (!grid_cut(n_out, grid, &sc)) return (size_t)-1;
    if (sc.nslab > 1 && (flags & 3u)) return (size_t)-1;
    // (sized for the plan of a sharing pair / a pullback: a forward call that keeps nothing splits
    // heavy tiles later and needs no more than this -- so a workspace sized for `raster` serves every
    // call of the same problem, as before)
    return make_plan(elem, n_out, slab_max_tiles(sc), P, B, (int)((flags >> 8) & 0xffu),
                     (flags & DPR_FLAG_COHERENT_POINTS) != 0, n_in, (flags & 3u) != 0, sc.nslab > 1)
        .total;
}

#define DPR_HIP(expr)                                                                \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            return fail(DPR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Dynamic LDS above 48 KiB has to be allowed per kernel (and per device): done once for the
// maximum the kernel can ask for (128 KiB of cursors at kMaxTiles), not per call.
template <typename K> static int allow_big_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return DPR_OK;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;  // (kernel, device) already raised
    int dev = 0;
    DPR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({(const void*)kernel, dev})) return DPR_OK;
    DPR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kMaxTiles * 4));
    done.insert({(const void*)kernel, dev});
    return DPR_OK;
}

// Which scatter a binning uses: the write-combining one whenever its LDS tables fit and the
// original indices are not needed as a separate array.
static bool scatter_is_wc(int NT, int nb, bool has_pw, bool want_idx) {
    const int wc = knobs().scatter_wc || nb > 1;
    const bool needs_idx = has_pw && want_idx && knobs().bwd_unpermute == 0 && nb == 1;
    return wc && NT * nb <= 4096 && !needs_idx;
}

// Compact 3-word records: default point weights, write-combining scatter, forward-only binning
// (nothing downstream needs the original index or room for a gradient record).
static bool records_are_compact(int NT, int nb, bool has_pw, bool want_idx) {
    if (has_pw || want_idx || !knobs().compact_records) return false;
    return scatter_is_wc(NT, nb, has_pw, want_idx);
}

template <typename T, int NI, int NO, bool HAS_PW, bool WANT_IDX>
static int launch_scatter(hipStream_t st, const GridDesc<NO>& gd, const TileGeom<NO>& tg,
                          const Plan& pl, char* ws, int64_t P, const T* points, const T* pw,
                          const T* rot, const T* trans, int64_t b, int nb, T* d_pts, T* d_pw,
                          int zero_dropped, const TileScanArgs& ts, bool fused) {
    // write-combining variant: needs 2 NT counters + the sub-chunk in LDS, and does not
    // produce rec_idx (only the direct-store pullback mode with point weights reads that)
    if (scatter_is_wc(tg.NT, nb, HAS_PW, WANT_IDX)) {
        constexpr int S = (sizeof(T) == 4) ? DPR_WC_PPT * kWcThreads : DPR_WC_PPT * kWcThreads / 2;
        const size_t lds2 = (size_t)tg.NT * (nb + 1) * 4;
#define DPR_LAUNCH_WC(GROUP, W3)                                                                  \
    hipLaunchKernelGGL((k_scatter_wc<T, NI, NO, HAS_PW, S, GROUP, W3>),                          \
                       dim3(pl.nblk + (fused ? 1 : 0)),                                          \
                       dim3(kWcThreads), lds2, st, gd, tg, P, pl.chunk, points, pw, rot, trans, \
                       b, nb, (const uint32_t*)(ws + pl.off_counts),                             \
                       (const uint32_t*)(ws + pl.off_tile_start),                                \
                       (RecT<T, W3>*)(ws + pl.off_rec),                     \
                       WANT_IDX ? (uint32_t*)(ws + pl.off_slot) : (uint32_t*)nullptr, d_pts,     \
                       d_pw, zero_dropped, (uint32_t*)(ws + pl.off_nitems) + 2, fused ? 1 : 0, ts)
        if constexpr (!HAS_PW) {
            if (records_are_compact(tg.NT, nb, false, WANT_IDX)) {
                if (nb > 1) DPR_LAUNCH_WC(true, true);
                else DPR_LAUNCH_WC(false, true);
                return DPR_OK;
            }
        }
        if (nb > 1) DPR_LAUNCH_WC(true, false);
        else DPR_LAUNCH_WC(false, false);
#undef DPR_LAUNCH_WC
        return DPR_OK;
    }
    const size_t lds = (size_t)tg.NT * 4;
    if (int rc = allow_big_lds(k_scatter<T, NI, NO, HAS_PW, WANT_IDX>, lds)) return rc;
    hipLaunchKernelGGL((k_scatter<T, NI, NO, HAS_PW, WANT_IDX>), dim3(pl.nblk), dim3(kBinThreads),
                       lds, st, gd, tg, P, pl.chunk, points, pw, rot, trans, b,
                       (const uint32_t*)(ws + pl.off_counts),
                       (const uint32_t*)(ws + pl.off_tile_start), (Rec4<T>*)(ws + pl.off_rec),
                       (uint32_t*)(ws + pl.off_idx), (uint32_t*)(ws + pl.off_slot), d_pts, d_pw,
                       zero_dropped, (uint32_t*)(ws + pl.off_nitems) + 2);
    return DPR_OK;
}

// K1-K3 for the poses [b, b + nb).  want_idx: a pullback will consume the binning.
template <typename T, int NI, int NO>
static int bin_points(hipStream_t st, const GridDesc<NO>& gd, const TileGeom<NO>& tg,
                      const Plan& pl, char* ws, int64_t P, const T* points, const T* pw,
                      const T* rot, const T* trans, int64_t b, int nb, bool want_idx, T* d_pts,
                      T* d_pw, int zero_dropped, bool keep_valid = false,
                      const T* hdr_points = nullptr, const T* hdr_pw = nullptr) {
    // (the header names the caller's buffers; `points` may be the library's sorted copy)
    if (!hdr_points) {
        hdr_points = points;
        hdr_pw = pw;
    }
    uint32_t* counts = (uint32_t*)(ws + pl.off_counts);
    uint32_t* totals = (uint32_t*)(ws + pl.off_totals);
    uint32_t* tile_start = (uint32_t*)(ws + pl.off_tile_start);
    const int NTe = tg.NT * nb;
    const size_t lds = (size_t)NTe * 4;
    if (nb > 1) {
        if (int rc = allow_big_lds(k_count<T, NI, NO, true>, lds)) return rc;
        hipLaunchKernelGGL((k_count<T, NI, NO, true>), dim3(pl.nblk), dim3(kBinThreads), lds, st,
                           gd, tg, P, pl.chunk, points, rot, trans, b, nb, counts,
                           (uint32_t*)(ws + pl.off_nzbins));
    } else {
        if (int rc = allow_big_lds(k_count<T, NI, NO, false>, lds)) return rc;
        hipLaunchKernelGGL((k_count<T, NI, NO, false>), dim3(pl.nblk), dim3(kBinThreads), lds, st,
                           gd, tg, P, pl.chunk, points, rot, trans, b, nb, counts,
                           (uint32_t*)(ws + pl.off_nzbins));
    }
    stage_mark(st);
    int64_t grid64[3] = {1, 1, 1};
    for (int d = 0; d < NO; ++d) grid64[d] = gd.n[d];
    TileScanArgs ts;
    ts.totals = totals;
    ts.NT = NTe;
    ts.cap = pl.cap;
    ts.tile_start = tile_start;
    ts.items = (WorkItem*)(ws + pl.off_items);
    ts.n_items = (uint32_t*)(ws + pl.off_nitems);
    ts.tile_parts = (uint32_t*)(ws + pl.off_tparts);
    ts.tile_slab = (uint32_t*)(ws + pl.off_tslab);
    ts.split_list = (uint32_t*)(ws + pl.off_split) + 1;
    ts.n_split = (uint32_t*)(ws + pl.off_split);
    ts.hdr = make_header<T, NI, NO>(grid64, P, hdr_points, hdr_pw);
    ts.hdr.state = keep_valid ? kBinValid : 0u;  // only a KEEP_BINNING forward may be reused
    ts.hdr.layout = plan_layout_id(pl);
    ts.hdr_out = (BinHeader*)(ws + pl.off_hdr);
    ts.rot = (const uint32_t*)(rot + b * (NO * NI));
    ts.rot_words = (int)(NO * NI * sizeof(T) / 4);
    ts.trans = (const uint32_t*)(trans + b * NO);
    ts.trans_words = (int)(NO * sizeof(T) / 4);
    ts.nonzero_bins = (const uint32_t*)(ws + pl.off_nzbins);
    ts.nblk = pl.nblk;
    hipLaunchKernelGGL(k_colscan, dim3((NTe + kScanTiles - 1) / kScanTiles), dim3(1024), 0, st,
                       counts, pl.nblk, NTe, totals, (uint32_t*)(ws + pl.off_nitems) + 2);
    // The tile scan rides in the scatter launch as one extra workgroup when the write-combining
    // scatter runs (its workgroups scan the totals themselves) and a CU is left for it.
    const bool fused = knobs().fuse_tilescan && scatter_is_wc(tg.NT, nb, pw != nullptr, want_idx) &&
                       pl.nblk < cu_count();
    if (!fused) hipLaunchKernelGGL(k_tilescan, dim3(1), dim3(1024), 0, st, ts);
    stage_mark(st);
    int rc;
    if (pw) {
        rc = want_idx ? launch_scatter<T, NI, NO, true, true>(st, gd, tg, pl, ws, P, points, pw,
                                                              rot, trans, b, nb, d_pts,
                                                              d_pw, zero_dropped, ts, fused)
                      : launch_scatter<T, NI, NO, true, false>(st, gd, tg, pl, ws, P, points, pw,
                                                               rot, trans, b, nb, d_pts,
                                                               d_pw, zero_dropped, ts, fused);
    } else {
        // without point weights the original index rides in the record for free; slot_of is
        // only written when a pullback will consume the binning
        rc = want_idx ? launch_scatter<T, NI, NO, false, true>(st, gd, tg, pl, ws, P, points, pw,
                                                               rot, trans, b, nb, d_pts,
                                                               d_pw, zero_dropped, ts, fused)
                      : launch_scatter<T, NI, NO, false, false>(st, gd, tg, pl, ws, P, points, pw,
                                                                rot, trans, b, nb, d_pts,
                                                                d_pw, zero_dropped, ts, fused);
    }
    stage_mark(st);
    return rc;
}

// Local binning of the poses [b, b + nb) (coherent cloud): clear the tile totals of the nb pose
// copies, k_bin_local (all nb poses, the points read once), k_runscan, k_place_desc.  `ws` is pose
// copy 0 of the batch.  Same stage marks as bin_points (count | scan | scatter become
// clear | bin_local | runscan + place).
template <typename K> static int allow_lds_bytes(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return DPR_OK;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    DPR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({(const void*)kernel, dev})) return DPR_OK;
    DPR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kMaxLocalTiles * 4));
    done.insert({(const void*)kernel, dev});
    return DPR_OK;
}
template <typename T, int NI, int NO>
static int bin_points_local(hipStream_t st, const GridDesc<NO>& gd, const TileGeom<NO>& tg,
                            const Plan& pl, char* ws, int64_t P, const T* points, const T* pw,
                            const T* rot, const T* trans, int64_t b, int nb, bool want_idx, T* d_pts,
                            T* d_pw, int zero_dropped, bool keep_valid,
                            const T* hdr_points = nullptr, const T* hdr_pw = nullptr) {
    if (!hdr_points) {  // the header names the caller's buffers
        hdr_points = points;
        hdr_pw = pw;
    }
    // ndesc[NT] | npts[NT] | - | max|pw| bits, and behind them the placement's cursors
    const size_t ltot_bytes = (pl.off_dcursor - pl.off_ltot) + (size_t)tg.NT * 4;
    if (nb > 1)
        DPR_HIP(hipMemset2DAsync(ws + pl.off_ltot, pl.pose_stride, 0, ltot_bytes, (size_t)nb, st));
    else
        DPR_HIP(hipMemsetAsync(ws + pl.off_ltot, 0, ltot_bytes, st));
    stage_mark(st);
    const uint32_t spare = (uint32_t)(pl.nsub * pl.sub);
    const size_t lds = (size_t)tg.NT * 4;
    LocalBinArgs la;
    la.ws = ws;
    la.pose_stride = pl.pose_stride;
    la.off_rec = pl.off_rec;
    la.off_slot = pl.off_slot;
    la.off_desc = pl.off_desc;
    la.off_bdesc = pl.off_bdesc;
    la.off_ltot = pl.off_ltot;
#define DPR_LAUNCH_LOCAL2(HAS_PW, W3, ONE, TH, STAGE)                                              \
    do {                                                                                          \
        auto kern = k_bin_local<T, NI, NO, HAS_PW, (sizeof(T) == 4 ? 4096 : 2048), W3, ONE, TH,  \
                                STAGE>;                                                           \
        if (int rc = allow_lds_bytes(kern, lds)) return rc;                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)pl.nsub), dim3(TH), lds, st, gd, tg, P, points,   \
                           pw, rot, trans, b, nb, la, want_idx ? 1 : 0, spare, d_pts, d_pw,       \
                           zero_dropped);                                                         \
    } while (0)
    // fp64 batches: two 512-thread workgroups per CU without LDS staging (see k_bin_local)
    const bool direct_store = sizeof(T) == 8 && nb > 1 && knobs().bin_direct_store;
    // (the 512-thread variant exists for fp64 only: the branch is not even instantiated for fp32 data)
#define DPR_LAUNCH_LOCAL(HAS_PW, W3)                                        \
    do {                                                                    \
        if (nb == 1) DPR_LAUNCH_LOCAL2(HAS_PW, W3, true, 1024, true);       \
        else if (direct_store) {                                            \
            if constexpr (sizeof(T) == 8) DPR_LAUNCH_LOCAL2(HAS_PW, W3, false, 512, false); \
        } else DPR_LAUNCH_LOCAL2(HAS_PW, W3, false, 1024, true);            \
    } while (0)
    if (pw) DPR_LAUNCH_LOCAL(true, false);
    else if (!want_idx && knobs().compact_records) DPR_LAUNCH_LOCAL(false, true);
    else DPR_LAUNCH_LOCAL(false, false);
#undef DPR_LAUNCH_LOCAL
#undef DPR_LAUNCH_LOCAL2
    stage_mark(st);
    int64_t grid64[3] = {1, 1, 1};
    for (int d = 0; d < NO; ++d) grid64[d] = gd.n[d];
    RunScanArgs ra;
    ra.ws = ws;
    ra.pose_stride = pl.pose_stride;
    ra.off_ltot = pl.off_ltot;
    ra.off_dstart = pl.off_dstart;
    ra.off_dcursor = pl.off_dcursor;
    ra.off_items = pl.off_items;
    ra.off_nitems = pl.off_nitems;
    ra.off_tparts = pl.off_tparts;
    ra.off_tslab = pl.off_tslab;
    ra.off_split = pl.off_split;
    ra.off_hdr = pl.off_hdr;
    ra.NT = tg.NT;
    ra.cap = pl.cap;
    ra.max_items = pl.max_items;
    ra.hdr = make_header<T, NI, NO>(grid64, P, hdr_points, hdr_pw);
    ra.hdr.state = keep_valid ? kBinValid : 0u;
    ra.hdr.layout = plan_layout_id(pl);
    ra.rot = (const uint32_t*)(rot + b * (NO * NI));
    ra.rot_words = (int)(NO * NI * sizeof(T) / 4);
    ra.trans = (const uint32_t*)(trans + b * NO);
    ra.trans_words = (int)(NO * sizeof(T) / 4);
    // the run scan rides in the placement launch when every placement workgroup can scan the
    // descriptor counts itself (up to 4096 tiles)
    const bool fused = knobs().fuse_tilescan && tg.NT <= 4096;
    ra.clear_cursors = 0;  // (cleared with the totals above)
    if (!fused) hipLaunchKernelGGL(k_runscan, dim3((unsigned)nb), dim3(1024), 0, st, ra);
    int64_t pblocks = (pl.nsub + 15) / 16;  // a wave per sub-chunk slot, 16 waves per workgroup
    if (pblocks > 1024) pblocks = 1024;
    hipLaunchKernelGGL(k_place_desc, dim3((unsigned)pblocks + (fused ? 1u : 0u), (unsigned)nb), dim3(1024),
                       0, st, ws, pl.pose_stride, pl.off_desc, pl.off_bdesc, pl.off_dstart,
                       pl.off_dcursor, pl.off_sdesc, pl.nsub, pl.sub, fused ? 1 : 0, ra);
    stage_mark(st);
    return DPR_OK;
}

template <int NO> static GridDesc<NO> make_grid_desc(const int64_t* grid, int64_t G) {
    GridDesc<NO> gd;
    for (int d = 0; d < NO; ++d) gd.n[d] = (int)grid[d];
    gd.G = G;
    return gd;
}

template <typename T, int NI, int NO>
int raster_tiled(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                 int64_t B, T* out, const T* points, const T* rot, const T* trans, const T* bg,
                 const T* ow, const T* pw, void* ws_, size_t ws_bytes) {
    SlabCut sc;
    if (!make_slab_cut<NO>(grid, &sc))
        return fail(DPR_ERR_UNSUPPORTED_ALGO,
                    "DPR_ALGO_TILED: a tile layer of the grid has more than %d tiles", kMaxTiles / 2);
    if (P >= (int64_t)1 << 32)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_TILED: P must be < 2^32");
    if (sc.nslab > 1 && (flags & 3u))
        return fail(DPR_ERR_UNSUPPORTED_ALGO,
                    "DPR_ALGO_TILED: a grid of more than %d tiles is processed in slabs, whose "
                    "binning cannot be kept (DPR_FLAG_KEEP_BINNING)", kMaxTiles);
    const int NTmax = slab_max_tiles(sc);
    const bool keep = flags & DPR_FLAG_KEEP_BINNING;
    // KEEP_BINNING with B > 1: every pose keeps its own binning (Plan::pose_stride)
    const Plan pl = make_plan(sizeof(T), NO, NTmax, P, B, (int)((flags >> 8) & 0xffu),
                              (flags & DPR_FLAG_COHERENT_POINTS) != 0, NI, (flags & 3u) != 0,
                              sc.nslab > 1, !(flags & 3u));
    if (!ws_ || ws_bytes < pl.total)
        return fail(DPR_ERR_WORKSPACE, "DPR_ALGO_TILED raster needs %zu workspace bytes, got %zu",
                    pl.total, ws_ ? ws_bytes : (size_t)0);
    char* ws = (char*)ws_;
    const GridDesc<NO> gd = make_grid_desc<NO>(grid, G);
    const T* const user_points = points;  // what the binning header names
    const T* const user_pw = pw;
    if (pl.sort_inside) {
        // coarse cell sort of the cloud (dpr_coarse.h): what local binning needs, a third of the
        // cost of the full Hilbert sort; the inverse permutation brings gradients back
        T* spw = pw ? (T*)(ws + pl.off_spw) : (T*)nullptr;
        if (int rc = coarse_sort_points<T, NI>(st, P, points, pw, (T*)(ws + pl.off_spts), spw,
                                               (uint32_t*)(ws + pl.off_iperm), ws + pl.off_sorttmp))
            return rc;
        points = (const T*)(ws + pl.off_spts);
        pw = spw;
    }
    T* halo = (T*)(ws + pl.off_aux);
    T* ovf = (T*)(ws + pl.off_aux +
                  align_up((size_t)NTmax * pl.bg * halo_count<NO>() * sizeof(T)));
    const int blocked = knobs().splat_blocked;
    // copy of the per-pose workspace pose b lives in: its own when the binning is kept, else its
    // place in the local batch
    auto copy_of = [&](int64_t b) { return (flags & 3u) && B > 1 ? b : (pl.local ? b % pl.lb : 0); };
    for (int64_t b = 0, nb = 1; b < B; b += nb) {
        for (nb = 1; nb * 2 <= pl.bg && b + nb * 2 <= B;) nb *= 2;  // poses of this group
        // per-pose part of the workspace (one copy, or one per pose when the binning is kept)
        char* const wsb = ws + (size_t)copy_of(b) * pl.pose_stride;
        char* const ws0 = pl.local ? ws + (size_t)copy_of(b - b % pl.lb) * pl.pose_stride : wsb;  // copy 0 of the local batch
        // slabs of a grid beyond kMaxTiles tiles, bottom to top (one iteration otherwise): slab s + 1
        // re-bins the top layer of slab s as its ghost layer
        for (int slab = 0; slab < sc.nslab; ++slab) {
        const TileGeom<NO> tg = slab_geom<NO>(grid, sc, slab, true);
        if (pl.local) {
            if (b % pl.lb == 0) {  // first pose of a local batch: bin all its poses
                const int nlb = (int)((B - b < pl.lb) ? B - b : pl.lb);
                if (int rc = bin_points_local<T, NI, NO>(st, gd, tg, pl, wsb, P, points, pw, rot, trans,
                                                         b, nlb, keep, (T*)nullptr, (T*)nullptr, 0,
                                                         keep, user_points, user_pw))
                    return rc;
            }
        } else if (int rc = bin_points<T, NI, NO>(st, gd, tg, pl, wsb, P, points, pw, rot, trans, b,
                                                  (int)nb, keep, (T*)nullptr, (T*)nullptr, 0, keep,
                                                  user_points, user_pw))
            return rc;
#define DPR_LAUNCH_SPLAT_RUNS(HAS_PW, W3)                                                        \
    hipLaunchKernelGGL((k_tile_splat_runs<T, NI, NO, HAS_PW, W3, true>), dim3(pl.max_items),    \
                       dim3(kSplatThreads), 0, st, gd, tg,                                      \
                       (const RecT<T, W3>*)(wsb + pl.off_rec),                                  \
                       (const RunDesc*)(wsb + pl.off_sdesc), (uint32_t)(pl.nsub * pl.sub),      \
                       (const WorkItem*)(wsb + pl.off_items),                                   \
                       (const uint32_t*)(wsb + pl.off_nitems),                                  \
                       (const uint32_t*)(wsb + pl.off_tslab), rot, trans, ow, bg, b, out, halo, \
                       ovf, (pl.sort_inside || knobs().splat_blocked == 0) ? 0 : 1,                 \
                       (const uint32_t*)(ws0 + pl.off_ltot) + 2 * tg.NT,                        \
                       knobs().fixed_point)
#define DPR_LAUNCH_SPLAT(HAS_PW, W3)                                                             \
    hipLaunchKernelGGL((k_tile_splat<T, NI, NO, HAS_PW, W3>), dim3(pl.max_items),               \
                       dim3(kSplatThreads), 0, st, gd, tg,                                      \
                       (const RecT<T, W3>*)(wsb + pl.off_rec),                                  \
                       (const WorkItem*)(wsb + pl.off_items),                                   \
                       (const uint32_t*)(wsb + pl.off_nitems),                                  \
                       (const uint32_t*)(wsb + pl.off_tslab), rot, trans, ow, bg, b, out, halo, \
                       ovf, blocked, (const uint32_t*)(wsb + pl.off_nitems) + 2,                \
                       knobs().fixed_point)
        if (pl.local) {
            if (pw) DPR_LAUNCH_SPLAT_RUNS(true, false);
            else if (!keep && knobs().compact_records) DPR_LAUNCH_SPLAT_RUNS(false, true);
            else DPR_LAUNCH_SPLAT_RUNS(false, false);
        } else if (pw) DPR_LAUNCH_SPLAT(true, false);
        else if (records_are_compact(tg.NT, (int)nb, false, keep)) DPR_LAUNCH_SPLAT(false, true);
        else DPR_LAUNCH_SPLAT(false, false);
#undef DPR_LAUNCH_SPLAT
#undef DPR_LAUNCH_SPLAT_RUNS
        stage_mark(st);
        hipLaunchKernelGGL((k_halo_gather<T, NO>),
                           dim3(tg.NT * (int)nb + kSplitGrid),
                           dim3(256), 0, st, gd, tg, (const T*)halo, (const T*)ovf,
                           (const uint32_t*)(wsb + pl.off_tparts),
                           (const uint32_t*)(wsb + pl.off_tslab),
                           (const uint32_t*)(wsb + pl.off_split) + 1,
                           (const uint32_t*)(wsb + pl.off_split), bg, b, (int)nb, out);
        stage_mark(st);
        }  // slabs
    }
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

template <typename T, int NI, int NO>
int pullback_tiled(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                   int64_t B, const T* g, const T* points, const T* rot, const T* trans,
                   const T* ow, const T* pw, T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow,
                   T* d_pw, void* ws_, size_t ws_bytes, Residual<T> rs) {
    SlabCut sc;
    if (!make_slab_cut<NO>(grid, &sc))
        return fail(DPR_ERR_UNSUPPORTED_ALGO,
                    "DPR_ALGO_TILED: a tile layer of the grid has more than %d tiles", kMaxTiles / 2);
    if (P >= (int64_t)1 << 32)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_TILED: P must be < 2^32");
    if (sc.nslab > 1 && (flags & 3u))
        return fail(DPR_ERR_UNSUPPORTED_ALGO,
                    "DPR_ALGO_TILED: a grid of more than %d tiles is processed in slabs, whose "
                    "binning cannot be kept (DPR_FLAG_REUSE_BINNING)", kMaxTiles);
    const int NTmax = slab_max_tiles(sc);
    const bool reuse = flags & DPR_FLAG_REUSE_BINNING;
    const Plan pl = make_plan(sizeof(T), NO, NTmax, P, B, (int)((flags >> 8) & 0xffu),
                              (flags & DPR_FLAG_COHERENT_POINTS) != 0, NI, (flags & 3u) != 0,
                              sc.nslab > 1);
    if (!ws_ || ws_bytes < pl.total)
        return fail(DPR_ERR_WORKSPACE,
                    "DPR_ALGO_TILED pullback needs %zu workspace bytes, got %zu", pl.total,
                    ws_ ? ws_bytes : (size_t)0);
    char* ws = (char*)ws_;
    const GridDesc<NO> gd = make_grid_desc<NO>(grid, G);
    T* d_pts_user = d_pts;
    T* d_pw_user = d_pw;
    const T* const user_points = points;  // what the binning header names
    const T* const user_pw = pw;
    if (pl.sort_inside) {
        T* spw = pw ? (T*)(ws + pl.off_spw) : (T*)nullptr;
        // REUSE_BINNING: the sorted copy and its permutation are the KEEP forward's (the per-pose
        // headers, checked on the device, vouch for the call pair)
        if (!reuse)
            if (int rc = coarse_sort_points<T, NI>(st, P, points, pw, (T*)(ws + pl.off_spts), spw,
                                                   (uint32_t*)(ws + pl.off_iperm),
                                                   ws + pl.off_sorttmp))
                return rc;
        points = (const T*)(ws + pl.off_spts);
        pw = spw;
        d_pts = (T*)(ws + pl.off_sgrad);   // gradients in sorted order, scattered back at the end
        if (d_pw) d_pw = (T*)(ws + pl.off_sgradw);  // (NULL: the caller declined this gradient)
    }
    double* partials = (double*)(ws + pl.off_aux);
    constexpr int NVAL = NO * NI + NO + 2;  // + 1 loss column in residual mode
    const bool unperm1 = knobs().bwd_unpermute != 0;
    BinHeader want = make_header<T, NI, NO>(grid, P, user_points, user_pw);
    want.layout = plan_layout_id(pl);
    if (!reuse) want.magic = 0;  // own binning: nothing to validate
    // every pose of the batch has its own gradient records (kept binning): one un-permute pass
    // over all of them at the end instead of a read-modify-write of the gradients per pose
    // (the same for the poses of a local batch, which were binned together into their own copies:
    // one pass per batch, the first one overwriting, the later ones accumulating)
    const bool batch_unperm = pl.pose_stride > 0 && B > 1 && unperm1 && pl.bg == 1 && P > 0 &&
                              (reuse || pl.local);
    // poses whose gradient records are summed by one un-permute pass
    const int64_t ub = !batch_unperm ? 1 : ((flags & 3u) ? B : pl.lb);
    auto copy_of = [&](int64_t b) { return (flags & 3u) && B > 1 ? b : (pl.local ? b % pl.lb : 0); };
    for (int64_t b = 0, nb = 1; b < B; b += nb) {
        for (nb = 1; nb * 2 <= pl.bg && b + nb * 2 <= B;) nb *= 2;  // poses of this group
        // per-pose part of the workspace (one copy, or one per pose when the binning was kept)
        char* const wsb = ws + (size_t)copy_of(b) * pl.pose_stride;
        // slabs of a grid beyond kMaxTiles tiles (one iteration otherwise): every slab bins the
        // cloud again, gathers its tiles, and ADDS its share of the point gradients and per-pose sums
        for (int slab = 0; slab < sc.nslab; ++slab) {
        const TileGeom<NO> tg = slab_geom<NO>(grid, sc, slab, false);
        const bool first_acc = b == 0 && slab == 0;  // the pass that overwrites ds_dpoints / ds_dpoint_weight
        BinHeader* hdr = reuse ? (BinHeader*)(wsb + pl.off_hdr) : (BinHeader*)nullptr;
        PoseReduceArgs<T> pr;
        pr.partials = partials;
        pr.items = (const WorkItem*)(wsb + pl.off_items);
        pr.n_items = (const uint32_t*)(wsb + pl.off_nitems);
        pr.max_items = pl.max_items;
        pr.NT = tg.NT;
        pr.n_in = NI;
        pr.n_out = NO;
        pr.b0 = b;
        pr.ds_drotation = d_rot;
        pr.ds_dtranslation = d_trans;
        pr.ds_dbackground = d_bg;
        pr.ds_dout_weight = d_ow;
        pr.loss = rs.target ? rs.loss : nullptr;
        pr.hdr = hdr;
        pr.accumulate = slab > 0 ? 1 : 0;
        const unsigned n_reduce = (unsigned)(rs.target ? NVAL + 1 : NVAL);
        bool reduced = false;
        // a pose group always goes through the gradient records (several (pose, tile) blocks
        // own the same point, so they cannot store to ds_dpoints directly)
        const bool unperm = unperm1 || nb > 1;
        if (reuse) {
            // The binning of the preceding raster call (same points / pose / grid) is in the
            // workspace.  Direct-store mode: points without an in-range voxel are in no tile,
            // clear the outputs first (the un-permute mode reads zeros from the spare slot).
            // (once, before the first pose: the later poses of a kept batch accumulate)
            if (P > 0 && !unperm && b == 0) {
                DPR_HIP(hipMemsetAsync(d_pts, 0, sizeof(T) * (size_t)(P * NI), st));
                if (d_pw) DPR_HIP(hipMemsetAsync(d_pw, 0, sizeof(T) * (size_t)P, st));
            }
            stage_mark(st);
            stage_mark(st);
            stage_mark(st);
        } else if (pl.local) {
            if (b % pl.lb == 0) {  // first pose of a local batch: bin all its poses
                const int nlb = (int)((B - b < pl.lb) ? B - b : pl.lb);
                if (int rc = bin_points_local<T, NI, NO>(st, gd, tg, pl, wsb, P, points, pw, rot, trans,
                                                         b, nlb, true, d_pts, d_pw,
                                                         (first_acc && !unperm) ? 1 : 0, false,
                                                         user_points, user_pw))
                    return rc;
            } else {
                stage_mark(st);
                stage_mark(st);
                stage_mark(st);
            }
        } else if (int rc = bin_points<T, NI, NO>(st, gd, tg, pl, wsb, P, points, pw, rot, trans, b,
                                                  (int)nb, true, d_pts, d_pw,
                                                  (first_acc && !unperm) ? 1 : 0, false, user_points,
                                                  user_pw))
            return rc;
#define DPR_LAUNCH_GATHER_RUNS(HAS_PW, FIRST, UNP)                                               \
    hipLaunchKernelGGL((k_tile_gather_runs<T, NI, NO, HAS_PW, FIRST, UNP, true>),                \
                       dim3(pl.max_items), dim3(gather_threads<T>()), 0, st, gd, tg,                  \
                       (Rec4<T>*)(wsb + pl.off_rec), (const RunDesc*)(wsb + pl.off_sdesc),       \
                       pl.nsub * pl.sub, (const uint32_t*)(wsb + pl.off_idx),                    \
                       (const WorkItem*)(wsb + pl.off_items),                                    \
                       (const uint32_t*)(wsb + pl.off_nitems), pl.max_items, g, rot, trans, ow,  \
                       b, d_pts, d_pw, partials, rs, want, hdr)
#define DPR_LAUNCH_GATHER_PLAIN(HAS_PW, FIRST, UNP)                                              \
    hipLaunchKernelGGL((k_tile_gather<T, NI, NO, HAS_PW, FIRST, UNP>), dim3(pl.max_items),       \
                       dim3(gather_threads<T>()), 0, st, gd, tg, (Rec4<T>*)(wsb + pl.off_rec), P * nb, \
                       (const uint32_t*)(wsb + pl.off_idx), (const WorkItem*)(wsb + pl.off_items), \
                       (const uint32_t*)(wsb + pl.off_nitems), pl.max_items, g, rot, trans, ow,   \
                       b, d_pts, d_pw, partials, rs, want, hdr)
#define DPR_LAUNCH_GATHER(HAS_PW, FIRST, UNP)                       \
    do {                                                            \
        if (pl.local) DPR_LAUNCH_GATHER_RUNS(HAS_PW, FIRST, UNP);   \
        else DPR_LAUNCH_GATHER_PLAIN(HAS_PW, FIRST, UNP);           \
    } while (0)
        if (unperm) {
            if (pw) DPR_LAUNCH_GATHER(true, true, true);
            else DPR_LAUNCH_GATHER(false, true, true);
            stage_mark(st);
            if (P > 0) {
                // one pose: a block covers a whole scatter sub-chunk (4 points per thread); a
                // pose group already has nb gradient records in flight per point
#define DPR_LAUNCH_UNPERM(FIRST, UPB, RB)                                                        \
    hipLaunchKernelGGL((k_unpermute<T, NI, FIRST, UPB>),                                         \
                       dim3((unsigned)((P + UPB * 1024 - 1) / (UPB * 1024)) + (RB)), dim3(1024), \
                       0, st, P, (int)nb, (const Rec4<T>*)(wsb + pl.off_rec),                    \
                       (const uint32_t*)(wsb + pl.off_slot), d_pts, d_pw, (const BinHeader*)hdr, \
                       (int)(RB), pr, (size_t)0)
                if (batch_unperm) {
                    // a kept batch: the records of all poses are summed once, after the loop
                } else if (nb > 1) {
                    if (first_acc) DPR_LAUNCH_UNPERM(true, 1, 0);
                    else DPR_LAUNCH_UNPERM(false, 1, 0);
                } else {
                    // one pose: the per-pose reduction rides in the same launch
                    if (first_acc) DPR_LAUNCH_UNPERM(true, DPR_UPB, n_reduce);
                    else DPR_LAUNCH_UNPERM(false, DPR_UPB, n_reduce);
                    reduced = true;
                }
#undef DPR_LAUNCH_UNPERM
            }
        } else {
            if (pw) {
                if (b == 0) DPR_LAUNCH_GATHER(true, true, false);
                else DPR_LAUNCH_GATHER(true, false, false);
            } else {
                if (b == 0) DPR_LAUNCH_GATHER(false, true, false);
                else DPR_LAUNCH_GATHER(false, false, false);
            }
            stage_mark(st);
        }
#undef DPR_LAUNCH_GATHER
#undef DPR_LAUNCH_GATHER_RUNS
#undef DPR_LAUNCH_GATHER_PLAIN
        stage_mark(st);
        if (!reduced)
            hipLaunchKernelGGL((k_pose_reduce<T>), dim3(n_reduce, (unsigned)nb), dim3(1024), 0, st,
                               pr);
        stage_mark(st);
        }  // slabs
        if (batch_unperm && ((b + 1) % ub == 0 || b + 1 == B)) {
            // the gradient records of the batch that ends with pose b: summed per point in one pass
            const int64_t bs = b - b % ub;  // first pose of the batch
            char* const wsu = ws + (size_t)copy_of(bs) * pl.pose_stride;
            const BinHeader* uh = reuse ? (const BinHeader*)(wsu + pl.off_hdr) : (const BinHeader*)nullptr;
            PoseReduceArgs<T> none{};
            if (bs == 0)
                hipLaunchKernelGGL((k_unpermute<T, NI, true, 1>), dim3((unsigned)((P + 1023) / 1024)),
                                   dim3(1024), 0, st, P, (int)(b + 1 - bs),
                                   (const Rec4<T>*)(wsu + pl.off_rec), (const uint32_t*)(wsu + pl.off_slot),
                                   d_pts, d_pw, uh, 0, none, pl.pose_stride);
            else
                hipLaunchKernelGGL((k_unpermute<T, NI, false, 1>), dim3((unsigned)((P + 1023) / 1024)),
                                   dim3(1024), 0, st, P, (int)(b + 1 - bs),
                                   (const Rec4<T>*)(wsu + pl.off_rec), (const uint32_t*)(wsu + pl.off_slot),
                                   d_pts, d_pw, uh, 0, none, pl.pose_stride);
        }
    }
    if (pl.sort_inside && P > 0)
        hipLaunchKernelGGL((k_unsort<T, NI>), dim3((unsigned)((P + 1023) / 1024)), dim3(256), 0, st, P,
                           (const uint32_t*)(ws + pl.off_iperm), (const T*)d_pts, (const T*)d_pw,
                           d_pts_user, d_pw_user,
                           reuse ? (const BinHeader*)(ws + pl.off_hdr) : (const BinHeader*)nullptr);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

#define DPR_INST(T, NI, NO)                                                                       \
    template int raster_tiled<T, NI, NO>(hipStream_t, unsigned, const int64_t*, int64_t, int64_t, \
                                         int64_t, T*, const T*, const T*, const T*, const T*,     \
                                         const T*, const T*, void*, size_t);                      \
    template int pullback_tiled<T, NI, NO>(hipStream_t, unsigned, const int64_t*, int64_t,        \
                                           int64_t, int64_t, const T*, const T*, const T*,        \
                                           const T*, const T*, const T*, T*, T*, T*, T*, T*, T*,  \
                                           void*, size_t, Residual<T>);
DPR_INST(float, 2, 2)
DPR_INST(float, 3, 3)
DPR_INST(float, 3, 2)
DPR_INST(double, 2, 2)
DPR_INST(double, 3, 3)
DPR_INST(double, 3, 2)
// Coarse cell sort for the chunk-owner kernels (dpr_chunkown.hip): the sorted copy, the sorted weights and the
// permutation sorted position -> original index.  `scratch`: coarse_sort_scratch_bytes(sizeof(T), P) bytes
// (slice histograms | inverse permutation, which this caller does not use).
size_t coarse_sort_scratch_bytes(size_t elem, int64_t P) {
    return (coarse_workspace_bytes(elem, P) + 255) / 256 * 256 + (size_t)(P < 1 ? 1 : P) * 4;
}
template <typename T>
int coarse_sort_with_perm(hipStream_t st, int n_in, int64_t P, const T* points, const T* pw, T* points_sorted,
                          T* pw_sorted, uint32_t* perm, char* scratch) {
    uint32_t* inv = (uint32_t*)(scratch + (coarse_workspace_bytes(sizeof(T), P) + 255) / 256 * 256);
    if (n_in == 3) return coarse_sort_points<T, 3>(st, P, points, pw, points_sorted, pw_sorted, inv, scratch, perm);
    if (n_in == 2) return coarse_sort_points<T, 2>(st, P, points, pw, points_sorted, pw_sorted, inv, scratch, perm);
    return fail(DPR_ERR_UNSUPPORTED_DIMS, "coarse cell sort: n_in = %d", n_in);
}
template int coarse_sort_with_perm<float>(hipStream_t, int, int64_t, const float*, const float*, float*, float*,
                                          uint32_t*, char*);
template int coarse_sort_with_perm<double>(hipStream_t, int, int64_t, const double*, const double*, double*,
                                           double*, uint32_t*, char*);

}  // namespace dpr
