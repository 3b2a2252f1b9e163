// DPR_ALGO_CHUNKED on 3-D grids: owner-computes voxel tiles over a box hierarchy of the cloud
// (forward) and a thread-per-point gather in cloud order (pullback).  No per-point record is ever
// written: both read the points in place.
//
// For a spatially coherent cloud (dpr_sort_points_*: Hilbert order) 16 consecutive points are a
// blob of a few voxels.  One streaming pass per call gives the cloud a three-level hierarchy of
// boxes IN THE GRID FRAME of each pose -- the exact range of cells a group of points contributes
// to, as 16-bit integers:
//
//   level 0   sub-chunk = 16 consecutive points      {lo[3], hi[3]}  12 B
//   level 1   chunk     = 64 sub-chunks = 1024 pts   same (+ max |point_weight|, pose independent)
//   level 2   64 chunks = 65536 points               same
//
// (Model-frame boxes, rotated into the grid as centre +- sum_j |R_dj| h_j, would be pose independent
// and could be kept across calls -- measured: a tile then looks at 1.98 points per point of the
// cloud against 1.09 it needs, the rotation inflates every box by up to sqrt(3); the pass over the
// points that builds the boxes costs the same either way and transforms the points for free in
// the shadow of its loads.)  A voxel tile of 32 x 32 x 14 cells (with a one-cell pad 148 KB of
// 64-bit LDS accumulators: one workgroup of 1024 threads per CU) finds the points it needs by
// integer interval tests:
//
//   boxes    k_own_boxes    one streaming pass over the points (the only forward kernel that reads
//                           all of them): level 0 and 1 for every pose of the group;
//            k_own_boxes2   level 2 from level 1
//   plan     k_own_plan     block per (tile, pose): walks the hierarchy top down, writes the tile's
//                           candidate chunks, estimates its load from the box overlaps, splits
//                           heavy tiles into parts (every n-th candidate) and files the work items
//                           in buckets by size (heaviest first)
//   forward  k_own_splat    block per work item.  A WAVE takes a candidate chunk, tests its 64
//                           level-0 boxes (one per lane), queues the hits; whenever 64 sub-chunks
//                           are queued every lane walks ONE of them (16 points, 4 per 48-byte
//                           load) -- lanes are >= 16 points apart in the cloud, so LDS atomics of
//                           one instruction rarely share an address.  Only contributions to cells
//                           the tile OWNS count (a point near a tile face is visited by both
//                           tiles; the pad cells take the rest and are dropped): no halo exchange,
//                           no global atomics; out = background + tile with plain stores.  fp32
//                           data: exact 64-bit fixed-point sums (dpr_device.h FixScale), fp64 /
//                           non-finite weights: f64 atomics.
//            k_own_combine  split tiles only: the parts left their raw 64-bit tiles in slabs,
//                           summed here (integer sums: exact, whatever the split)
//   pullback k_own_pullback a thread per point in cloud order, gathers straight from ds_dout (the
//                           cloud is coherent: a wave's gathers share cache lines), 13 per-pose sums
//                           in registers per block, grid sum for ds_dbackground alongside; no tiles
//            k_own_reduce   per-block partial sums (f64) -> ds_drotation, ds_dtranslation, ...
//
// Correct for ANY point order: a box that covers half the grid is listed by every tile it
// overlaps (slow, never wrong); candidate lists that outgrow their buffer make the tile take every
// chunk as a candidate.  Reference semantics: /root/reference/src/raster.jl:36-66 (forward kernel),
// src/raster_pullback.jl:39-72 (per-point pullback), :85-148 (batch).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/dpr.h"
#include "dpr_device.h"
#include "dpr_tiled.h"

namespace dpr {

constexpr int kSC = 16;              // points per sub-chunk (level-0 box, one lane's share)
constexpr int kFan = 64;             // sub-chunks per chunk (level-1 box, one wave's box tests)
constexpr int kL1 = kSC * kFan;      // 1024 points
constexpr int kL2 = 64;              // chunks per level-2 box (65536 points)
// (experiment builds override the tile kernels' block and tile: -DDPR_OWN_THREADS=512 -DDPR_OWN_TY=16 -DDPR_OWN_TZ=13
// is the half tile that fits two workgroups per CU, profiles/r06_experiments.md)
#ifndef DPR_OWN_THREADS
#define DPR_OWN_THREADS 1024
#endif
#ifndef DPR_OWN_TY
#define DPR_OWN_TY 32
#endif
#ifndef DPR_OWN_TZ
#define DPR_OWN_TZ 14
#endif
constexpr int kOT = DPR_OWN_THREADS; // threads of the tile kernels
constexpr int kOW = kOT / kWave;     // 16 waves
constexpr int kTX = 32, kTY = DPR_OWN_TY, kTZ = DPR_OWN_TZ;
constexpr int kCells = kTX * kTY * kTZ;                     // 14336 owned cells
// forward: the LDS tile is PADDED by one cell on every side (34 x 34 x 16 cells of 8 bytes = 148 KB):
// all eight neighbours of every point the tile looks at have a cell, no ownership tests; the pad
// cells are never flushed (the neighbouring tile computes them itself)
constexpr int kPX = kTX + 2, kPY = kTY + 2, kPZ = kTZ + 2;
constexpr int kPCells = kPX * kPY * kPZ;                    // 18496
constexpr int kBuckets = 16;
constexpr int kMaxParts = 32;
constexpr int kOwnBw = 16;           // poses planned at once (one copy of the per-pose arrays each)
constexpr int kQueue = 128;          // queued sub-chunks per wave (64 + up to 63 left over)
constexpr int kCtlWords = 32;        // per pose: [0] list cursor, [1] slab cursor (pose 0's), [2] split
                                     // tiles, [3] list overflow seen, [16..31] bucket counts
constexpr int kMaxOwnTiles = 1 << 20;
constexpr int kPlanHits = 1024;      // level-2 boxes a plan block can descend into (more: scan-all tile)

// cells [lo, hi] (per axis, existing cells only) a group of points contributes to under one pose;
// lo > hi: none (no point of the group has a cell in the grid)
struct alignas(4) IBox {
    int16_t lo[3], hi[3];
};
struct alignas(16) TileRec {
    uint32_t begin;   // first entry of the candidate list, 0xffffffff: overflow (every chunk)
    uint32_t count;   // candidate chunks
    int sexp;         // fixed-point exponent of the tile (all parts share it), kFixNone: f64 sums
    uint32_t parts;   // nparts | first_slab << 8
};
struct alignas(16) OwnItem {
    uint32_t tile, part_nparts, begin, count;
    int sexp;
    uint32_t first_slab, pad0, pad1;
};
struct OGeom {
    int nt[3];
    int NT;
};

static bool make_ogeom(const int64_t* grid, OGeom* tg) {
    const int T[3] = {kTX, kTY, kTZ};
    int64_t NT = 1;
    for (int d = 0; d < 3; ++d) {
        tg->nt[d] = (int)((grid[d] + T[d] - 1) / T[d]);
        NT *= tg->nt[d];
    }
    if (NT > kMaxOwnTiles) return false;
    tg->NT = (int)NT;
    return true;
}

__device__ __forceinline__ void tile_coords(int tile, const OGeom& tg, int (&tc)[3], int (&x0)[3]) {
    const int T[3] = {kTX, kTY, kTZ};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        tc[d] = tile % tg.nt[d];
        tile /= tg.nt[d];
        x0[d] = tc[d] * T[d];
    }
}

// does a box reach cells of the tile at x0?
__device__ __forceinline__ bool box_hits(const IBox& b, const int (&x0)[3]) {
    return b.lo[0] <= x0[0] + kTX - 1 && b.hi[0] >= x0[0] && b.lo[1] <= x0[1] + kTY - 1 && b.hi[1] >= x0[1] &&
           b.lo[2] <= x0[2] + kTZ - 1 && b.hi[2] >= x0[2];
}

// A wave-uniform floating-point value the compiler computed with vector instructions (this chip has
// no scalar float ALU) sits in a vector register for the whole kernel; through readfirstlane it
// lives in scalar registers instead -- the tile kernels hoist a dozen of these (scales, origins).
__device__ __forceinline__ float uniform(float x) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ double uniform(double x) {
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// src/raster.jl:88-99 with the loop invariants (origin = -1 - t, scale = n / 2, n as T) held in
// scalar registers; the same operations in the same order as ref_and_deltas (dpr_device.h)
template <typename T> struct OwnXform {
    T origin[3], scale[3], nf[3];
};
template <typename T>
__device__ __forceinline__ OwnXform<T> own_xform(const Pose<T, 3, 3>& ps, const GridDesc<3>& gd) {
    OwnXform<T> xf;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        xf.origin[d] = uniform(T(-1) - ps.t[d]);
        xf.scale[d] = uniform(T(gd.n[d]) / T(2));
        xf.nf[d] = uniform(T(gd.n[d]));
    }
    return xf;
}
// lower neighbour of a point (0-based, may be -1) and its deltas; false: no cell in the grid
template <typename T>
__device__ __forceinline__ bool own_ref(const T (&pt)[3], const Pose<T, 3, 3>& ps, const OwnXform<T>& xf,
                                        int (&ref0)[3], T (&dlo)[3]) {
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        T proj = ps.R[d] * pt[0];
        proj = proj + ps.R[d + 3] * pt[1];
        proj = proj + ps.R[d + 6] * pt[2];
        const T coord = (proj - xf.origin[d]) * xf.scale[d];
        const T c = coord - T(0.5);
        ok = ok && (c > T(-1)) && (c <= xf.nf[d]);
        const T r = ceil_t<T>(c);
        ref0[d] = (int)r - 1;  // (meaningless unless ok: the conversion saturates)
        dlo[d] = coord - (r - T(0.5));
    }
    return ok;
}

// The same for a tile kernel: PADDED tile coordinates l = ref0 - x0 + 1 of the lower neighbour and the deltas,
// without the range tests of own_ref -- a point whose l is within 0 .. T on every axis has all eight
// neighbours in the padded tile, and a neighbour outside the GRID then lies in a pad cell or in an owned cell
// beyond the grid's edge, neither of which is ever flushed (the individual drop of src/raster.jl:62; a point
// with no neighbour in the grid at all cannot reach an owned cell inside it).  Returns false for NaN / Inf
// coordinates (a float -> int conversion of those is 0 / saturated and must not be trusted).
template <typename T>
__device__ __forceinline__ bool own_ref_local(const T (&pt)[3], const Pose<T, 3, 3>& ps, const OwnXform<T>& xf,
                                              const int (&x0)[3], uint32_t (&l)[3], T (&dlo)[3]) {
    T csum = T(0);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        T proj = ps.R[d] * pt[0];
        proj = proj + ps.R[d + 3] * pt[1];
        proj = proj + ps.R[d + 6] * pt[2];
        const T coord = (proj - xf.origin[d]) * xf.scale[d];
        const T c = coord - T(0.5);
        const T r = ceil_t<T>(c);
        l[d] = (uint32_t)(int)r - (uint32_t)x0[d];  // = ref0 - x0 + 1 (wraps far out of range for saturated r)
        dlo[d] = coord - (r - T(0.5));
        csum += c;
    }
    return (csum - csum) == T(0);  // finite
}

// 16 bytes at a time where the caller's buffer allows it (`vec`: wave-uniform)
template <typename T, int N> __device__ __forceinline__ void load_run(const T* __restrict__ src, bool vec, T (&v)[N]) {
    constexpr int PER = 16 / sizeof(T);
    static_assert(N % PER == 0, "whole 16-byte vectors");
    if (vec) {
        typedef T VecT __attribute__((ext_vector_type(PER)));
#pragma unroll
        for (int k = 0; k < N / PER; ++k) {
            const VecT x = ((const VecT*)src)[k];
#pragma unroll
            for (int e = 0; e < PER; ++e) v[k * PER + e] = x[e];
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = src[k];
    }
}

// ---------------------------------------------------------------- boxes
// One block per chunk of 1024 points; thread t owns points 4t .. 4t + 3 of it, four threads a
// sub-chunk.  The points stay in registers while the poses of the group are walked.  Also clears
// the control words of the plan (first block).
struct OwnBoxArgs {
    IBox *b0, *b1;      // [pose copy][nSC], [pose copy][nL1]
    float* mw1;         // [nL1] max |point_weight| per chunk (inf: a NaN weight); at [nL1 + 1 + c] the min non-zero one
    int64_t nSC, nL1;
    uint32_t* ctl;
    int ctl_words;
};
template <typename T>
__global__ __launch_bounds__(256) void k_own_boxes(GridDesc<3> gd, int64_t P, const T* __restrict__ points,
                                                   const T* __restrict__ pw, int vec_ok,
                                                   const T* __restrict__ rot, const T* __restrict__ trans,
                                                   int64_t bfirst, int nb, OwnBoxArgs ba) {
    __shared__ int red[4][6];
    __shared__ float redw[4], redn[4];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < ba.ctl_words; i += 256) ba.ctl[i] = 0u;
    const int64_t c = blockIdx.x;
    const int64_t p0 = c * kL1 + (int64_t)threadIdx.x * 4;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    T v[12];
    int npts = 0;
    if (p0 < P) {
        npts = p0 + 4 <= P ? 4 : (int)(P - p0);
        if (npts == 4) {
            load_run<T, 12>(points + p0 * 3, vec_ok != 0, v);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int64_t p = p0 + k / 3;
                v[k] = points[(p < P ? p : P - 1) * 3 + k % 3];
            }
        }
    }
    if (pw) {  // (pose independent)
        float mw = 0.f, mn = __builtin_inff();  // max |w|, min non-zero |w| (fix_guard_range, dpr_device.h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < npts) {
                const T w = pw[p0 + q];
                const float a = (w == w) ? fabsf((float)w) : __builtin_inff();  // NaN weights: f64 sums (IEEE)
                mw = fmaxf(mw, a);
                mn = a > 0.f ? fminf(mn, a) : mn;
            }
        }
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            mw = fmaxf(mw, __shfl_xor(mw, o, kWave));
            mn = fminf(mn, __shfl_xor(mn, o, kWave));
        }
        if (lane == 0) {
            redw[wave] = mw;
            redn[wave] = mn;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            ba.mw1[c] = fmaxf(fmaxf(redw[0], redw[1]), fmaxf(redw[2], redw[3]));
            ba.mw1[ba.nL1 + 1 + c] = fminf(fminf(redn[0], redn[1]), fminf(redn[2], redn[3]));
        }
    }
#pragma unroll 1
    for (int bl = 0; bl < nb; ++bl) {
        const Pose<T, 3, 3> ps = load_pose<T, 3, 3>(rot, trans, nullptr, bfirst + bl);
        const OwnXform<T> xf = own_xform<T>(ps, gd);
        int lo[3] = {32767, 32767, 32767}, hi[3] = {-32768, -32768, -32768};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const T pt[3] = {v[q * 3], v[q * 3 + 1], v[q * 3 + 2]};
            int ref0[3];
            T dlo[3];
            if (own_ref<T>(pt, ps, xf, ref0, dlo) && q < npts) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int a = ref0[d] < 0 ? 0 : ref0[d];
                    const int e = ref0[d] + 1 < gd.n[d] ? ref0[d] + 1 : gd.n[d] - 1;
                    lo[d] = a < lo[d] ? a : lo[d];
                    hi[d] = e > hi[d] ? e : hi[d];
                }
            }
        }
        // sub-chunk = 4 neighbouring lanes
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int a = __shfl_xor(lo[d], o, kWave), e = __shfl_xor(hi[d], o, kWave);
                lo[d] = a < lo[d] ? a : lo[d];
                hi[d] = e > hi[d] ? e : hi[d];
            }
        }
        if ((threadIdx.x & 3) == 0 && p0 < P) {
            IBox b;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                b.lo[d] = (int16_t)lo[d];
                b.hi[d] = (int16_t)hi[d];
            }
            ba.b0[(size_t)bl * ba.nSC + c * kFan + threadIdx.x / 4] = b;
        }
        // chunk = the block
#pragma unroll
        for (int o = 4; o < kWave; o <<= 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int a = __shfl_xor(lo[d], o, kWave), e = __shfl_xor(hi[d], o, kWave);
                lo[d] = a < lo[d] ? a : lo[d];
                hi[d] = e > hi[d] ? e : hi[d];
            }
        }
        __syncthreads();  // (red of the previous pose has been read)
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                red[wave][d] = lo[d];
                red[wave][3 + d] = hi[d];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            IBox b;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int a = red[0][d], e = red[0][3 + d];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    a = red[w][d] < a ? red[w][d] : a;
                    e = red[w][3 + d] > e ? red[w][3 + d] : e;
                }
                b.lo[d] = (int16_t)a;
                b.hi[d] = (int16_t)e;
            }
            ba.b1[(size_t)bl * ba.nL1 + c] = b;
        }
    }
}

// level-2 boxes: one wave per (64 chunks, pose)
__global__ __launch_bounds__(256) void k_own_boxes2(int64_t nL1, int64_t nL2, const IBox* __restrict__ b1,
                                                    const float* __restrict__ mw1, IBox* __restrict__ b2,
                                                    float* __restrict__ mw2) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t c2 = (int64_t)blockIdx.x * 4 + threadIdx.x / kWave;
    const int bl = blockIdx.y;
    if (c2 >= nL2) return;  // (uniform)
    const int64_t c = c2 * kL2 + lane;
    int lo[3] = {32767, 32767, 32767}, hi[3] = {-32768, -32768, -32768};
    float mw = 0.f;
    if (c < nL1) {
        const IBox b = b1[(size_t)bl * nL1 + c];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = b.lo[d];
            hi[d] = b.hi[d];
        }
        if (mw1) mw = mw1[c];
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int a = __shfl_xor(lo[d], o, kWave), e = __shfl_xor(hi[d], o, kWave);
            lo[d] = a < lo[d] ? a : lo[d];
            hi[d] = e > hi[d] ? e : hi[d];
        }
        mw = fmaxf(mw, __shfl_xor(mw, o, kWave));
    }
    if (lane == 0) {
        IBox b;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            b.lo[d] = (int16_t)lo[d];
            b.hi[d] = (int16_t)hi[d];
        }
        b2[(size_t)bl * nL2 + c2] = b;
        if (mw1 && bl == 0) mw2[c2] = mw;
    }
}

// ---------------------------------------------------------------- plan
struct OwnPlanArgs {
    char* ws;
    size_t off_ctl, off_rec, off_list, off_items, off_split;
    size_t rec_stride, list_stride, items_stride, split_stride;  // per pose copy (bytes)
    const IBox *b1, *b2;  // [pose copy][nL1], [pose copy][nL2]
    const float* mw1;     // [nL1] max, then at [nL1 + 1 + c] min non-zero |point_weight| per chunk; or nullptr (no point weights)
    int64_t nL1, nL2;
    uint32_t list_cap;   // entries per tile
    int max_items;       // per bucket
    int max_slabs;       // per pose group
    int max_split;       // per pose
    uint32_t cap;        // visits per part above which a tile is split
    int fixed;           // fixed-point forward wanted
};

// block per (tile, pose copy): level-2 boxes that reach the tile are collected first (wave w scans
// the w-th quarter: a deterministic order), then the waves test the 64 chunks under each of them
// (independent loads, two in flight per wave) and keep the hit masks in LDS; the candidate list is
// written from the masks at the offset the block reserved.
template <typename T>
__global__ __launch_bounds__(256) void k_own_plan(OGeom tg, GridDesc<3> gd, int64_t P, const T* __restrict__ ow,
                                                  int has_pw, int64_t bfirst, OwnPlanArgs pa) {
    __shared__ uint32_t s_hit[kPlanHits];
    __shared__ unsigned long long s_mask[kPlanHits];
    __shared__ uint32_t s_off[kPlanHits];
    __shared__ uint32_t s_cnt2[4], s_begin;
    __shared__ float s_est[4], s_mw[4], s_mn[4];
    const int tile = blockIdx.x, bl = blockIdx.y;
    const int64_t b = bfirst + bl;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    int tc[3], x0[3];
    tile_coords(tile, tg, tc, x0);
    const IBox* b1 = pa.b1 + (size_t)bl * pa.nL1;
    const IBox* b2 = pa.b2 + (size_t)bl * pa.nL2;
    uint32_t* ctl = (uint32_t*)(pa.ws + pa.off_ctl) + (size_t)bl * kCtlWords;
    uint32_t* ctl0 = (uint32_t*)(pa.ws + pa.off_ctl);
    // 1. level 2: wave w takes the w-th quarter, hits into its own segment of s_hit
    const int64_t per = ((pa.nL2 + 3) / 4 + kWave - 1) / kWave * kWave;
    const int64_t i0 = wave * per, i1 = (i0 + per < pa.nL2) ? i0 + per : pa.nL2;
    uint32_t n2 = 0;
    bool too_many = false;
    for (int64_t i = i0; i < i1; i += kWave) {  // (uniform)
        bool hit = false;
        if (i + lane < i1) hit = box_hits(b2[i + lane], x0);
        const unsigned long long m = __ballot(hit);
        const uint32_t pos = n2 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (hit && pos < kPlanHits / 4) s_hit[wave * (kPlanHits / 4) + pos] = (uint32_t)(i + lane);
        n2 += (uint32_t)__popcll(m);
    }
    if (n2 > kPlanHits / 4) {
        too_many = true;
        n2 = kPlanHits / 4;
    }
    if (lane == 0) s_cnt2[wave] = n2 | (too_many ? 0x80000000u : 0u);
    __syncthreads();
    const uint32_t c0 = s_cnt2[0] & 0x7fffffffu, c1 = s_cnt2[1] & 0x7fffffffu, c2n = s_cnt2[2] & 0x7fffffffu,
                   c3 = s_cnt2[3] & 0x7fffffffu;
    const bool scan_all = ((s_cnt2[0] | s_cnt2[1] | s_cnt2[2] | s_cnt2[3]) & 0x80000000u) != 0;
    const uint32_t H = c0 + c1 + c2n + c3;
    auto hit_id = [&](uint32_t h) -> uint32_t {  // h-th level-2 hit, waves' segments concatenated
        if (h < c0) return s_hit[h];
        h -= c0;
        if (h < c1) return s_hit[kPlanHits / 4 + h];
        h -= c1;
        if (h < c2n) return s_hit[2 * (kPlanHits / 4) + h];
        return s_hit[3 * (kPlanHits / 4) + (h - c2n)];
    };
    // 2. level 1 under every hit: masks, load estimate, largest weight
    float est = 0.f, mw = 0.f, mn = __builtin_inff();
    for (uint32_t h = wave; h < H; h += 8) {  // two hits per step: their loads overlap
        const uint32_t ha = h, hb = h + 4;
        const int64_t ca = (int64_t)hit_id(ha) * kL2 + lane;
        const int64_t cb = hb < H ? (int64_t)hit_id(hb) * kL2 + lane : -1;
        IBox xa, xb;
        if (ca < pa.nL1) xa = b1[ca];
        if (cb >= 0 && cb < pa.nL1) xb = b1[cb];
        float wa = 0.f, wb = 0.f, na = __builtin_inff(), nb_ = __builtin_inff();
        if (pa.mw1) {
            if (ca < pa.nL1) {
                wa = pa.mw1[ca];
                na = pa.mw1[pa.nL1 + 1 + ca];
            }
            if (cb >= 0 && cb < pa.nL1) {
                wb = pa.mw1[cb];
                nb_ = pa.mw1[pa.nL1 + 1 + cb];
            }
        }
        auto one = [&](int64_t c, const IBox& bx, float w, float wn, uint32_t hh) {
            const bool hit = c >= 0 && c < pa.nL1 && box_hits(bx, x0);
            if (hit) {
                float f = 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int Td = d == 0 ? kTX : (d == 1 ? kTY : kTZ);
                    const int a = bx.lo[d] > x0[d] ? bx.lo[d] : x0[d];
                    const int e = bx.hi[d] < x0[d] + Td - 1 ? bx.hi[d] : x0[d] + Td - 1;
                    f *= (float)(e - a + 1) / (float)(bx.hi[d] - bx.lo[d] + 1);
                }
                const int64_t npts = (c + 1) * kL1 <= P ? kL1 : P - c * kL1;
                est += f * (float)npts;
                mw = fmaxf(mw, w);
                mn = fminf(mn, wn);
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) s_mask[hh] = m;
        };
        one(ca, xa, wa, na, ha);
        if (hb < H) one(cb, xb, wb, nb_, hb);  // (uniform)
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        est += __shfl_xor(est, o, kWave);
        mw = fmaxf(mw, __shfl_xor(mw, o, kWave));
        mn = fminf(mn, __shfl_xor(mn, o, kWave));
    }
    if (lane == 0) {
        s_est[wave] = est;
        s_mw[wave] = mw;
        s_mn[wave] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t h = 0; h < H; ++h) {
            s_off[h] = total;
            total += (uint32_t)__popcll(s_mask[h]);
        }
        const float e = s_est[0] + s_est[1] + s_est[2] + s_est[3];
        // (largest weight of the candidates, or +Inf = f64 atomics when their non-zero weights span > 2^10)
        const float m = fix_guard_range(fmaxf(fmaxf(s_mw[0], s_mw[1]), fmaxf(s_mw[2], s_mw[3])),
                                        fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3])));
        // every tile has its own fixed slot of the list buffer (a cursor shared by all the blocks
        // of a launch made them queue up on one address: 1216 returning atomics = 20 us)
        uint32_t begin = (uint32_t)tile * pa.list_cap;
        if (scan_all || total > pa.list_cap) {
            begin = 0xffffffffu;  // more candidates than the slot holds: the tile takes every chunk
            total = (uint32_t)pa.nL1;
            ctl[3] = 1u;
        }
        // parts: every n-th candidate each
        uint32_t ev = e < 4.0e9f ? (uint32_t)e : 4000000000u;
        // (a tile whose candidates outgrow its slot still has the estimate of the masks; only a tile that could
        // not scan its level-2 hits knows nothing)
        if (scan_all) ev = P < 4000000000ll ? (uint32_t)P : 4000000000u;
        uint32_t np = 1;
        if (ev > pa.cap) {
            np = (ev + pa.cap - 1) / pa.cap;
            if (np > (uint32_t)kMaxParts) np = kMaxParts;
            if (np > total) np = total ? total : 1;
        }
        uint32_t first_slab = 0;
        if (np > 1) {
            first_slab = atomicAdd(&ctl0[1], np);
            uint32_t sidx = 0;
            // (the slabs [first_slab, first_slab + np) are this tile's whatever happens: when they straddle the
            // end of the buffer the tile takes the ones that exist)
            if (first_slab + 2 <= (uint32_t)pa.max_slabs && first_slab + np > (uint32_t)pa.max_slabs)
                np = (uint32_t)pa.max_slabs - first_slab;
            if (first_slab + np > (uint32_t)pa.max_slabs ||
                (sidx = atomicAdd(&ctl[2], 1u)) >= (uint32_t)pa.max_split) {
                np = 1;  // no slab left: one long item (slower, still right)
                first_slab = 0;
            } else {
                ((uint32_t*)(pa.ws + pa.off_split + (size_t)bl * pa.split_stride))[sidx] = (uint32_t)tile;
            }
        }
        // fixed-point exponent of the tile: |contribution| <= |out_weight| * max|pw| of the
        // candidates, at most 1024 contributions per candidate to one cell
        const float owv = ow ? fabsf((float)ow[b]) : 1.f;
        float maxw = __builtin_inff();
        if (sizeof(T) == 4) maxw = owv * (has_pw ? (begin == 0xffffffffu ? __builtin_inff() : m) : 1.f);
        const uint64_t nmax = (uint64_t)(total ? total : 1) * kL1;
        const int sexp = fix_exponent(maxw, nmax < 0x7fffffffu ? (uint32_t)nmax : 0x7fffffffu, pa.fixed);
        TileRec rec;
        rec.begin = begin;
        rec.count = total;
        rec.sexp = sexp;
        rec.parts = np | (first_slab << 8);
        ((TileRec*)(pa.ws + pa.off_rec + (size_t)bl * pa.rec_stride))[tile] = rec;
        // work items, bucketed by size
        const uint32_t epart = ev / np;
        int bucket = epart < 512 ? 0 : (32 - __clz((int)epart)) - 9;
        if (bucket > kBuckets - 1) bucket = kBuckets - 1;
        const uint32_t slot = atomicAdd(&ctl[16 + bucket], np);
        OwnItem* items = (OwnItem*)(pa.ws + pa.off_items + (size_t)bl * pa.items_stride) + (size_t)bucket * pa.max_items;
        for (uint32_t k = 0; k < np; ++k) {
            if (slot + k >= (uint32_t)pa.max_items) break;
            OwnItem it;
            it.tile = (uint32_t)tile;
            it.part_nparts = k | (np << 16);
            it.begin = begin;
            it.count = total;
            it.sexp = sexp;
            it.first_slab = first_slab;
            it.pad0 = it.pad1 = 0u;
            items[slot + k] = it;
        }
        s_begin = begin;
    }
    __syncthreads();
    const uint32_t begin = s_begin;
    if (begin == 0xffffffffu) return;
    // 3. the candidate list from the masks
    uint32_t* list = (uint32_t*)(pa.ws + pa.off_list + (size_t)bl * pa.list_stride) + begin;
    for (uint32_t h = wave; h < H; h += 4) {
        const unsigned long long m = s_mask[h];
        if ((m >> lane) & 1ull)
            list[s_off[h] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = hit_id(h) * kL2 + (uint32_t)lane;
    }
}

// ---------------------------------------------------------------- shared by the tile kernels
struct OwnTileArgs {
    const char* ws;
    size_t off_ctl, off_rec, off_list, off_items;
    size_t rec_stride, list_stride, items_stride;
    int max_items;
    int64_t nL1, nSC;
    const IBox* b0;    // [pose copy][nSC]
    size_t dbg_words;  // (stats build: 32-bit words of the slab area, per-item records at its end)
};

// work item of this block: buckets from the heaviest down
__device__ __forceinline__ bool own_item(const OwnTileArgs& ta, int bl, uint32_t idx, OwnItem& it) {
    const uint32_t* ctl = (const uint32_t*)(ta.ws + ta.off_ctl) + (size_t)bl * kCtlWords;
    uint32_t cnt[kBuckets];
#pragma unroll
    for (int k = 0; k < kBuckets; ++k) cnt[k] = ctl[16 + k];
    int bucket = -1;
#pragma unroll
    for (int k = kBuckets - 1; k >= 0; --k) {
        const uint32_t c = cnt[k] < (uint32_t)ta.max_items ? cnt[k] : (uint32_t)ta.max_items;
        if (bucket < 0) {
            if (idx < c) bucket = k;
            else idx -= c;
        }
    }
    if (bucket < 0) return false;
    it = ((const OwnItem*)(ta.ws + ta.off_items + (size_t)bl * ta.items_stride))[(size_t)bucket * ta.max_items + idx];
    return true;
}

// LDS the discovery loop needs (the tile kernels put it in front of their tile)
constexpr int kLanesPerSC = 4;                   // lanes that share a sub-chunk (4 points each)
constexpr int kBatchSC = kWave / kLanesPerSC;    // 16 sub-chunks per batch of a wave
constexpr int kPtsPerLane = kSC / kLanesPerSC;   // 4
struct OwnWalkLds {
    uint32_t queue[kOW][kQueue];  // per wave: sub-chunks waiting for a full batch
    uint32_t pool[kOT];           // what the waves had left over (< 16 each), shared out again
    uint32_t next, pool_n;
    uint32_t pad[62];
};

// The discovery loop of a tile kernel.  Waves draw candidate chunks of the item from a counter in
// LDS, test the 64 level-0 boxes of a chunk (one per lane; the boxes of the NEXT candidate are
// requested before the current batch is worked on), queue the hits and call `visit(sc, quarter,
// have)` with 16 queued sub-chunks at a time: four neighbouring lanes share a sub-chunk, four
// points each (one 48-byte load; the four lanes read 192 contiguous bytes).  A batch is four points
// deep -- with one lane per sub-chunk it was sixteen, 16-32 us during which the waves that had run
// out of candidates waited, and the leftovers (up to 63 sub-chunks per wave, re-batched from a
// pool) cost another sixteen on half the waves: half of an item's time (r05 experiments).  What a
// wave has left when the candidates run out (< 16 sub-chunks) goes to a pool of the block, which
// the waves share out again.  One block barrier inside, reached exactly once by every wave.
template <typename Visit>
__device__ __forceinline__ void own_walk(const OwnTileArgs& ta, const OwnItem& item, int bl, const int (&x0)[3],
                                         OwnWalkLds* wl, Visit visit) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t* queue = wl->queue[wave];
    const uint32_t part = item.part_nparts & 0xffffu, nparts = item.part_nparts >> 16;
    // a tile whose candidate list did not fit its buffer (an incoherent cloud) takes EVERY chunk
    // as a candidate: slow, never wrong
    const bool overflow = item.begin == 0xffffffffu;
    const uint32_t* list = (const uint32_t*)(ta.ws + ta.off_list + (size_t)bl * ta.list_stride) + (overflow ? 0u : item.begin);
    const IBox* b0 = ta.b0 + (size_t)bl * ta.nSC;
    const uint32_t units = item.count;
    uint32_t qn = 0;
    int64_t sc_mine = 0;
    IBox bx;
    // next candidate chunk of this wave: its level-0 boxes are requested here, tested later
    auto acquire = [&]() -> bool {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(&wl->next, 1u);
        k = __builtin_amdgcn_readfirstlane(k);
        const uint64_t u = (uint64_t)part + (uint64_t)k * nparts;
        if (u >= units) return false;
        const uint32_t c = overflow ? (uint32_t)u : list[u];
        sc_mine = (int64_t)c * kFan + lane;
        if (sc_mine < ta.nSC) bx = b0[sc_mine];
        return true;
    };
    bool have_c = acquire();
    int stage = 0;  // 0: candidates, 2: pooled batches
    uint32_t pool_pos = 0, pool_total = 0;
#ifdef DPR_OWN_STATS
    uint32_t st_batches = 0, st_take = 0, st_tests = 0;
#endif
    for (;;) {
        if (qn >= (uint32_t)kBatchSC || (stage == 2 && qn > 0)) {
            const uint32_t take = qn >= (uint32_t)kBatchSC ? kBatchSC : qn;
            const uint32_t base = qn - take;
            const uint32_t e = (uint32_t)lane / kLanesPerSC;
            const bool have = e < take;
            const uint32_t sc = queue[base + (have ? e : 0u)];  // (idle lanes: a valid sub-chunk, not worked on)
#ifdef DPR_OWN_STATS
            ++st_batches;
            st_take += take * kLanesPerSC;
#endif
            visit(sc, lane % kLanesPerSC, have);
            qn = base;
            continue;
        }
        if (stage == 0) {
            if (have_c) {
                const bool hit = sc_mine < ta.nSC && box_hits(bx, x0);
                const unsigned long long mask = __ballot(hit);
                if (hit) queue[qn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)sc_mine;
                qn += (uint32_t)__popcll(mask);
#ifdef DPR_OWN_STATS
                ++st_tests;
#endif
                have_c = acquire();
                continue;
            }
            // the candidates are gone: what is left (< 16) goes to the block's pool
            uint32_t pos = 0;
            if (lane == 0 && qn) pos = atomicAdd(&wl->pool_n, qn);
            pos = __builtin_amdgcn_readfirstlane(pos);
            if ((uint32_t)lane < qn) wl->pool[pos + lane] = queue[lane];
            qn = 0;
#ifdef DPR_OWN_STATS
            if (threadIdx.x == 0) wl->pad[5] = (uint32_t)wall_clock64();
#endif
            __syncthreads();
#ifdef DPR_OWN_STATS
            if (threadIdx.x == 0) wl->pad[6] = (uint32_t)wall_clock64();
#endif
            pool_total = wl->pool_n;
            pool_pos = (uint32_t)wave * kBatchSC;
            stage = 2;
        }
        if (pool_pos >= pool_total) break;
        const uint32_t n = pool_total - pool_pos < (uint32_t)kBatchSC ? pool_total - pool_pos : (uint32_t)kBatchSC;
        if ((uint32_t)lane < n) queue[lane] = wl->pool[pool_pos + lane];
        qn = n;
        pool_pos += kOW * kBatchSC;
    }
#ifdef DPR_OWN_STATS
    if (lane == 0) {
        atomicAdd(&wl->pad[2], st_batches);
        atomicAdd(&wl->pad[3], st_take);
        atomicAdd(&wl->pad[4], st_tests);
    }
#endif
}

// the four points of a lane's quarter of a sub-chunk: body(point index in the sub-chunk, live, pt[3], w)
template <typename T, bool HAS_PW, typename Body>
__device__ __forceinline__ void own_points(uint32_t sc, int quarter, bool have, int64_t P,
                                           const T* __restrict__ points, const T* __restrict__ pw, bool vec,
                                           Body body) {
    constexpr int N = kPtsPerLane;
    // (a lane without a sub-chunk was handed a valid one by own_walk: it loads like the others and
    // treats every point as dead)
    const int64_t p0 = (int64_t)sc * kSC + quarter * N;
    const bool full = p0 + N <= P;  // false only in the last sub-chunk of the cloud
    const int npts = have ? (full ? N : (p0 < P ? (int)(P - p0) : 0)) : 0;
    T v[N * 3], w[N];
    if (full) {
        load_run<T, N * 3>(points + p0 * 3, vec, v);
        if constexpr (HAS_PW) load_run<T, N>(pw + p0, vec, w);
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const int64_t p = p0 + q;
            const int64_t pc = p < P ? p : P - 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) v[q * 3 + j] = points[pc * 3 + j];
            if constexpr (HAS_PW) w[q] = pw[pc];
        }
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const T pt[3] = {v[q * 3], v[q * 3 + 1], v[q * 3 + 2]};
        body(quarter * N + q, q < npts, pt, HAS_PW ? w[q] : T(1));
    }
}

// ---------------------------------------------------------------- forward
template <typename T, bool HAS_PW>
__global__ __launch_bounds__(kOT) void k_own_splat(OGeom tg, GridDesc<3> gd, int64_t P,
                                                   const T* __restrict__ points, const T* __restrict__ pw,
                                                   int vec_ok, const T* __restrict__ rot,
                                                   const T* __restrict__ trans, const T* __restrict__ ow,
                                                   const T* __restrict__ bg, int64_t bfirst,
                                                   OwnTileArgs ta, unsigned long long* __restrict__ slabs,
                                                   T* __restrict__ out) {
    extern __shared__ unsigned char smem[];
    OwnWalkLds* wl = (OwnWalkLds*)smem;
    double* acc = (double*)(smem + sizeof(OwnWalkLds));  // kPCells cells
    const int bl = blockIdx.y;
    const int64_t b = bfirst + bl;
    OwnItem rec;
    if (!own_item(ta, bl, blockIdx.x, rec)) return;
    const uint32_t tile = rec.tile, part = rec.part_nparts & 0xffffu, nparts = rec.part_nparts >> 16;
#ifdef DPR_OWN_STATS
    const uint64_t st_t0 = wall_clock64();
#define DPR_STAMP(k) do { if (threadIdx.x == 0) wl->pad[8 + (k)] = (uint32_t)(wall_clock64() - st_t0); } while (0)
#else
#define DPR_STAMP(k) do { } while (0)
#endif
    int tc[3], x0[3];
    tile_coords((int)tile, tg, tc, x0);
    const double bgv = bg ? (double)bg[b] : 0.0;
    T* o = out + b * gd.G;
    const bool vec_out = (gd.n[0] & 3) == 0 && (((uintptr_t)o) & 15) == 0;
    if (rec.count == 0 && rec.begin != 0xffffffffu) {
        // no candidate: the tile is background
        for (int i = threadIdx.x; i < kCells; i += kOT) {
            const int x = i % kTX, y = (i / kTX) % kTY, z = i / (kTX * kTY);
            const int g0 = x0[0] + x, g1 = x0[1] + y, g2 = x0[2] + z;
            if (g0 < gd.n[0] && g1 < gd.n[1] && g2 < gd.n[2])
                __builtin_nontemporal_store((T)bgv, &o[((size_t)g2 * gd.n[1] + g1) * gd.n[0] + g0]);
        }
        return;
    }
    for (int i = threadIdx.x; i < kPCells; i += kOT) acc[i] = 0.0;
    if (threadIdx.x == 0) {
        wl->next = 0u;
        wl->pool_n = 0u;
        for (int k = 0; k < 8; ++k) wl->pad[k] = 0u;
    }
    const Pose<T, 3, 3> ps = load_pose<T, 3, 3>(rot, trans, ow, b);
    const FixScale fs = fix_scale_from_exponent(rec.sexp);
    const OwnXform<T> xf = own_xform<T>(ps, gd);
    DPR_STAMP(0);
    __syncthreads();
    DPR_STAMP(1);
#ifdef DPR_OWN_STATS
    uint32_t st_vis = 0, st_touch = 0;
#endif
#ifdef DPR_OWN_EXP
    double exp_sink = 0.0;
#endif
    auto run = [&](auto fix_tag) {
        constexpr bool FIX = decltype(fix_tag)::value;
        auto visit = [&](uint32_t sc, int quarter, bool have) {
            own_points<T, HAS_PW>(sc, quarter, have, P, points, pw, vec_ok != 0, [&](int, bool live, const T (&pt)[3], T pwi) {
                uint32_t l[3];
                T dlo[3];
                const bool ok = own_ref_local<T>(pt, ps, xf, x0, l, dlo) && live;
                // padded tile coordinates of the lower neighbour: 0 .. T
                const uint32_t l0 = l[0], l1 = l[1], l2 = l[2];
                const bool touches = ok && l0 <= (uint32_t)kTX && l1 <= (uint32_t)kTY && l2 <= (uint32_t)kTZ;
#if defined(DPR_OWN_STATS) && DPR_OWN_STATS >= 2  /* (per-point counters slow the kernel down 2x) */
                st_vis += live ? 1u : 0u;
                st_touch += touches ? 1u : 0u;
#endif
#if defined(DPR_OWN_EXP) && DPR_OWN_EXP == 2  /* transform + test only */
                exp_sink += touches ? (double)dlo[0] : 0.0;
                if (false) {
#else
                if (touches) {
#endif
                    const T w = HAS_PW ? ps.ow * pwi : ps.ow;  // src/raster.jl:52
                    double* cell = acc + (l0 + kPX * l1 + kPX * kPY * l2);
                    // all eight neighbours have a cell (pad cells, and owned cells beyond the grid
                    // edge, are never flushed: the individual drop of src/raster.jl:62)
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
#if defined(DPR_OWN_EXP) && DPR_OWN_EXP == 1  /* no LDS atomics: where does the time go? */
                        exp_sink += (double)voxel_weight<T, 3>(dlo, s, w) * fs.mul + (double)(size_t)cell;
#else
                        cell_add<FIX, T>(cell + ((s & 1) + kPX * ((s >> 1) & 1) + kPX * kPY * (s >> 2)),
                                         voxel_weight<T, 3>(dlo, s, w), fs);
#endif
                    }
                }
            });
        };
        own_walk(ta, rec, bl, x0, wl, visit);
    };
    if (fs.mul != 0.0) run(std::true_type{});  // (uniform)
    else run(std::false_type{});
#ifdef DPR_OWN_STATS
    {
        const uint32_t a = (uint32_t)wave_sum<int>((int)st_vis), c = (uint32_t)wave_sum<int>((int)st_touch);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&wl->pad[0], a);
            atomicAdd(&wl->pad[1], c);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t* ctl = (uint32_t*)(ta.ws + ta.off_ctl);
            atomicAdd(&ctl[4], wl->pad[0]);
            atomicAdd(&ctl[5], wl->pad[1]);
            atomicAdd(&ctl[6], wl->pad[2]);
            atomicAdd(&ctl[7], wl->pad[3]);
            atomicAdd(&ctl[9], wl->pad[4]);
            atomicAdd(&ctl[8], 1u);
            const uint32_t dt = (uint32_t)(wall_clock64() - st_t0);
            atomicAdd(&ctl[10], dt);
            atomicMax(&ctl[11], dt);
            // per-item record at the end of the slab area: {ticks, visits, tile, part | nparts << 16, start tick}
            uint32_t* dbg = (uint32_t*)(slabs) + (size_t)ta.dbg_words - 8 * ((size_t)blockIdx.x + 1);
            dbg[0] = dt;
            dbg[1] = wl->pad[0];
            dbg[2] = tile;
            dbg[3] = part | (nparts << 16);
            dbg[4] = (uint32_t)st_t0;
            dbg[5] = rec.count;
            // phases (ticks of 10 ns): setup | zero+barrier | wave 0 to the pool barrier | its wait there | pooled batches + final barrier
            const uint32_t t_end_walk = (uint32_t)(wall_clock64() - st_t0);
            atomicAdd(&ctl[12], wl->pad[8]);
            atomicAdd(&ctl[13], wl->pad[9] - wl->pad[8]);
            atomicAdd(&ctl[14], (wl->pad[5] - (uint32_t)st_t0) - wl->pad[9]);
            atomicAdd(&ctl[15], wl->pad[6] - wl->pad[5]);
            atomicAdd(&ctl[1 + 2], 0u);
            dbg[6] = t_end_walk - (wl->pad[6] - (uint32_t)st_t0);
            dbg[7] = wl->pad[8];
        }
    }
#endif
#ifdef DPR_OWN_EXP
    if (exp_sink == 1.2345) acc[threadIdx.x] = exp_sink;
#endif
    __syncthreads();
    // flush the owned cells: thread -> 4 cells along x
    const bool to_slab = nparts > 1;
    // (part of a split tile: the raw 64-bit cells go to this part's slab; k_own_combine sums)
    unsigned long long* slab = slabs + (size_t)(rec.first_slab + part) * kCells;
    for (int q = threadIdx.x; q < kCells / 4; q += kOT) {
        const int x = (q % (kTX / 4)) * 4, y = (q / (kTX / 4)) % kTY, z = q / (kTX / 4 * kTY);
        const double* src = acc + ((x + 1) + kPX * (y + 1) + kPX * kPY * (z + 1));
        if (to_slab) {
#pragma unroll
            for (int e = 0; e < 4; ++e) slab[q * 4 + e] = (unsigned long long)__double_as_longlong(src[e]);
            continue;
        }
        const int g0 = x0[0] + x, g1 = x0[1] + y, g2 = x0[2] + z;
        if (g1 >= gd.n[1] || g2 >= gd.n[2] || g0 >= gd.n[0]) continue;
        T* dst = &o[((size_t)g2 * gd.n[1] + g1) * gd.n[0] + g0];
        T v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (T)(bgv + fix_value(src[e], fs));
        if (vec_out && sizeof(T) == 4) {  // (g0 + 3 < n0 since n0 % 4 == 0)
            typedef T V4 __attribute__((ext_vector_type(4)));
            V4 vv;
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = v[e];
            __builtin_nontemporal_store(vv, (V4*)dst);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (g0 + e < gd.n[0]) __builtin_nontemporal_store(v[e], dst + e);
        }
    }
}

// split tiles: out = background + sum of the parts' raw tiles (fixed point: integer sums, exact;
// f64: a fixed order).  Four blocks per split tile.
template <typename T>
__global__ __launch_bounds__(kOT) void k_own_combine(OGeom tg, GridDesc<3> gd, const T* __restrict__ bg,
                                                     int64_t bfirst, OwnTileArgs ta, size_t off_split,
                                                     size_t split_stride, int max_split,
                                                     const unsigned long long* __restrict__ slabs,
                                                     T* __restrict__ out) {
    const int bl = blockIdx.y;
    const int64_t b = bfirst + bl;
    const uint32_t* ctl = (const uint32_t*)(ta.ws + ta.off_ctl) + (size_t)bl * kCtlWords;
    uint32_t nsplit = ctl[2];
    if (nsplit > (uint32_t)max_split) nsplit = max_split;
    const uint32_t* split = (const uint32_t*)(ta.ws + off_split + (size_t)bl * split_stride);
    const double bgv = bg ? (double)bg[b] : 0.0;
    T* o = out + b * gd.G;
    constexpr int kQuarter = (kCells + 3) / 4;
    for (uint32_t s = blockIdx.x; s < nsplit * 4u; s += gridDim.x) {
        const uint32_t tile = split[s >> 2];
        const int i_lo = (int)(s & 3u) * kQuarter, i_hi = i_lo + kQuarter < kCells ? i_lo + kQuarter : kCells;
        const TileRec rec = ((const TileRec*)(ta.ws + ta.off_rec + (size_t)bl * ta.rec_stride))[tile];
        const uint32_t np = rec.parts & 0xffu, first = rec.parts >> 8;
        const FixScale fs = fix_scale_from_exponent(rec.sexp);
        int tc[3], x0[3];
        tile_coords((int)tile, tg, tc, x0);
        for (int i = i_lo + threadIdx.x; i < i_hi; i += kOT) {
            const int x = i % kTX, y = (i / kTX) % kTY, z = i / (kTX * kTY);
            const int g0 = x0[0] + x, g1 = x0[1] + y, g2 = x0[2] + z;
            if (g0 >= gd.n[0] || g1 >= gd.n[1] || g2 >= gd.n[2]) continue;
            const unsigned long long* sp = slabs + (size_t)first * kCells + i;
            double v;
            if (fs.mul != 0.0) {
                unsigned long long a = 0;
                uint32_t k = 0;
                for (; k + 4 <= np; k += 4) {
                    const unsigned long long a0 = sp[(size_t)k * kCells], a1 = sp[(size_t)(k + 1) * kCells],
                                             a2 = sp[(size_t)(k + 2) * kCells], a3 = sp[(size_t)(k + 3) * kCells];
                    a += a0 + a1 + a2 + a3;
                }
                for (; k < np; ++k) a += sp[(size_t)k * kCells];
                v = fix_value(__longlong_as_double((long long)a), fs);
            } else {
                v = 0.0;
                for (uint32_t k = 0; k < np; ++k) v += __longlong_as_double((long long)sp[(size_t)k * kCells]);
            }
            __builtin_nontemporal_store((T)(bgv + v), &o[((size_t)g2 * gd.n[1] + g1) * gd.n[0] + g0]);
        }
    }
}

// ---------------------------------------------------------------- pullback
// On a coherent cloud the pullback needs no tile at all: a thread per point IN CLOUD ORDER reads its
// eight ds_dout cells straight from memory -- neighbouring lanes are neighbouring points, so a wave's
// gathers fall into a handful of cache lines and every cell is fetched from HBM about once --,
// differentiates the point exactly once and stores its 16 bytes coalesced: no ownership tests, no
// repeated visits, no divergence, nothing staged.  (The LDS-staged owner-computes gather this file
// had first ran 0.53 ms at 10 M points -> 256^3: twice the visits of a per-point kernel, and the
// heavy gradient arithmetic only on the half of the lanes that own their point.)  What the direct
// kernel of DPR_ALGO_ATOMIC pays for -- 13 wave reductions and 13 same-address global atomics per 256
// points -- is gone: a block walks a contiguous slice of the cloud, keeps the 13 per-pose sums in
// registers and reduces them once; the grid sum for ds_dbackground rides along (every block sums a
// slice of ds_dout with 16-byte loads).  Correct for any point order; an incoherent cloud pays
// cache misses on its gathers.  src/raster_pullback.jl:39-72 (per point), :78 (background).
constexpr int kNVal = 3 * 3 + 3 + 2;  // d rotation, d translation, d out_weight, sum(ds_dout)
constexpr int kDT = 256;             // threads of the direct pullback kernel
constexpr int kDBlocksPerCU = 8;

template <typename T, bool HAS_PW, bool FIRST>
__global__ __launch_bounds__(kDT) void k_own_pullback(GridDesc<3> gd, int64_t P, int64_t per_block,
                                                      int64_t cells_per_block,
                                                      const T* __restrict__ points, const T* __restrict__ pw,
                                                      const T* __restrict__ g, const T* __restrict__ rot,
                                                      const T* __restrict__ trans, const T* __restrict__ ow,
                                                      int64_t b, T* __restrict__ ds_dpoints,
                                                      T* __restrict__ ds_dpw, double* __restrict__ partials) {
    __shared__ double red[kDT / kWave][kNVal];
    const Pose<T, 3, 3> ps = load_pose<T, 3, 3>(rot, trans, ow, b);
    const OwnXform<T> xf = own_xform<T>(ps, gd);
    const T* gb = g + b * gd.G;
    const int64_t p_lo = (int64_t)blockIdx.x * per_block;
    const int64_t p_hi = p_lo + per_block < P ? p_lo + per_block : P;
    T vals[kNVal - 1];
#pragma unroll
    for (int k = 0; k < kNVal - 1; ++k) vals[k] = T(0);
    const int n0 = gd.n[0], n1 = gd.n[1], n2 = gd.n[2];
    // Software pipeline, one point deep: the transform of point i + 1 and its eight gathers are issued
    // BEFORE the arithmetic of point i (and the coordinates of point i + 2 before that), so a wave
    // covers its own gather latency with ~250 instructions of work instead of leaving it to the
    // other waves of the SIMD (4-5 of them at ~90 registers: 0.18 -> ? ms at 10 M points, r05).
    struct Stage {
        T pt[3], pwi, dlo[3], gv[8];
        uint32_t in;  // bit r: row r = (s1, s2) of the neighbourhood is inside the grid
        int xsel;     // which halves of the loaded x-pairs are the lower / upper neighbour
        bool ok, live, edge;
        int64_t p;
    };
    const int64_t n_it = (p_hi - p_lo + kDT - 1) / kDT;  // (uniform)
    const uint32_t row1 = (uint32_t)n0, row2 = (uint32_t)n0 * (uint32_t)n1;  // (G < 2^31: 32-bit cell offsets)
    T nxt[3] = {T(0), T(0), T(0)}, wn = T(1);
    auto fetch_point = [&](int64_t p) {  // (clamped: the loop body stays branch-free)
        const int64_t q = p < P ? p : P - 1;
#pragma unroll
        for (int j = 0; j < 3; ++j) nxt[j] = points[q * 3 + j];
        if constexpr (HAS_PW) wn = pw[q];
    };
    auto front = [&](int64_t p, Stage& st) {  // transform + gathers of the point in `nxt`
        st.p = p;
        st.live = p < p_hi;
#pragma unroll
        for (int j = 0; j < 3; ++j) st.pt[j] = nxt[j];
        st.pwi = wn;
        int ref0[3];
        st.ok = own_ref<T>(st.pt, ps, xf, ref0, st.dlo) && st.live;
        if (!st.ok) ref0[0] = ref0[1] = ref0[2] = 0;
        // the eight cells as FOUR loads of an x-pair (the two x-neighbours are adjacent in memory:
        // half the gather instructions, 0.164 -> 0.139 ms), all requested before the first is used.
        // The pair starts at xb = clamp(ref0.x, 0, n0 - 2), so it never leaves its row; a neighbour
        // outside the grid (individual drop, src/raster_pullback.jl:51) counts as 0: in y / z the row
        // is redirected to cell 0 and masked, in x `xsel` says which half of the pair is which
        const bool lo1 = ref0[1] >= 0, hi1 = ref0[1] + 1 < n1, lo2 = ref0[2] >= 0, hi2 = ref0[2] + 1 < n2;
        const int xb = ref0[0] < 0 ? 0 : (ref0[0] > n0 - 2 ? n0 - 2 : ref0[0]);
        st.xsel = ref0[0] - xb;  // -1: pair = (x+1, x+2) -> only hi = pair[0]; 0: (lo, hi); 1: only lo = pair[1]
        const uint32_t base = ((uint32_t)ref0[2] * (uint32_t)n1 + (uint32_t)ref0[1]) * (uint32_t)n0 + (uint32_t)xb;
        st.in = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s1 = r & 1, s2 = r >> 1;
            const bool in = st.ok && (s1 ? hi1 : lo1) && (s2 ? hi2 : lo2);
            st.in |= in ? (1u << r) : 0u;
            const uint32_t off = base + (s1 ? row1 : 0u) + (s2 ? row2 : 0u);  // (wraps harmlessly when !in)
            typedef T Pair __attribute__((ext_vector_type(2), aligned(sizeof(T))));
            const Pair pr = *(const Pair*)(gb + (in ? off : 0u));
            st.gv[2 * r] = pr[0];
            st.gv[2 * r + 1] = pr[1];
        }
        st.edge = st.in != 0xfu || st.xsel != 0;
    };
    auto back = [&](const Stage& st) {  // the point's arithmetic and its stores
        // nearly every wave holds interior points only: the loaded pairs ARE the eight values; a wave
        // with a point on the border of the grid (or outside) sorts them out with selects
        T gi[8];
        if (__ballot(st.edge) == 0ull) {  // (uniform)
#pragma unroll
            for (int s = 0; s < 8; ++s) gi[s] = st.gv[s];
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int r = s >> 1;
                const T gx = (s & 1) ? (st.xsel == 0 ? st.gv[2 * r + 1] : (st.xsel < 0 ? st.gv[2 * r] : T(0)))
                                     : (st.xsel == 0 ? st.gv[2 * r] : (st.xsel > 0 ? st.gv[2 * r + 1] : T(0)));
                gi[s] = ((st.in >> r) & 1u) ? gx : T(0);
            }
        }
        T gout[3] = {T(0), T(0), T(0)}, dpw_part = T(0);
        T dcoord[3] = {T(0), T(0), T(0)}, dow_part = T(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const T dweight = voxel_weight<T, 3>(st.dlo, s, gi[s]);  // raster_pullback.jl:55
            dow_part += dweight * st.pwi;                            // :57
            dpw_part += dweight * ps.ow;                             // :58
            const T factor = gi[s] * ps.ow * st.pwi;                 // :60
#pragma unroll
            for (int n = 0; n < 3; ++n) dcoord[n] += factor * interp_weight<T, 3>(n, st.dlo, s);
        }
        T scaled[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) scaled[n] = st.ok ? dcoord[n] * xf.scale[n] : T(0);  // :67
        dpw_part = st.ok ? dpw_part : T(0);
        // (the per-pose sums have no summation order in common with the reference's serial loop:
        // explicit FMAs, the library is built with -ffp-contract=off)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
#pragma unroll
            for (int j = 0; j < 3; ++j)  // (a rejected point may be NaN / Inf: 0 * NaN must not reach the sums)
                vals[n + j * 3] = fma_t(scaled[n], st.ok ? st.pt[j] : T(0), vals[n + j * 3]);  // :69
            vals[9 + n] += scaled[n];                                                                  // :68
        }
        vals[12] += st.ok ? dow_part : T(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // rotation' * scaled (:70)
            T v = ps.R[0 + j * 3] * scaled[0];
            v = v + ps.R[1 + j * 3] * scaled[1];
            v = v + ps.R[2 + j * 3] * scaled[2];
            gout[j] = v;
        }
        if (st.live) {
            if (FIRST) {
#pragma unroll
                for (int j = 0; j < 3; ++j) __builtin_nontemporal_store(gout[j], &ds_dpoints[st.p * 3 + j]);
                if (ds_dpw) __builtin_nontemporal_store(dpw_part, &ds_dpw[st.p]);
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) ds_dpoints[st.p * 3 + j] += gout[j];
                if (ds_dpw) ds_dpw[st.p] += dpw_part;
            }
        }
    };
    if (n_it > 0 && P > 0) {
        // (two stages used alternately would save the copy below and cost 35 registers: 5 -> 3 waves per SIMD)
        Stage cur, nx;
        const int64_t pa = p_lo + threadIdx.x;
        fetch_point(pa);
        front(pa, cur);
        fetch_point(pa + kDT);
#pragma unroll 1
        for (int64_t i = 0; i < n_it; ++i) {
            const int64_t p1 = pa + (i + 1) * kDT;
            front(p1, nx);  // (beyond the slice: dead, its gathers read cell 0)
            fetch_point(p1 + kDT);
            back(cur);
            cur = nx;
        }
    }
    // ds_dbackground: this block's slice of the grid (16 bytes per load where the slice allows)
    double bg_sum = 0.0;
    {
        const int64_t c_lo = (int64_t)blockIdx.x * cells_per_block;
        const int64_t c_hi = c_lo + cells_per_block < gd.G ? c_lo + cells_per_block : gd.G;
        constexpr int PER = 16 / (int)sizeof(T);
        typedef T VecT __attribute__((ext_vector_type(PER)));
        int64_t i = c_lo;
        if ((((uintptr_t)(gb + c_lo)) & 15) == 0) {  // (uniform)
            T a[PER];
#pragma unroll
            for (int e = 0; e < PER; ++e) a[e] = T(0);
            for (i = c_lo + (int64_t)threadIdx.x * PER; i + PER <= c_hi; i += (int64_t)kDT * PER) {
                const VecT v = __builtin_nontemporal_load((const VecT*)(gb + i));
#pragma unroll
                for (int e = 0; e < PER; ++e) a[e] += v[e];
            }
#pragma unroll
            for (int e = 0; e < PER; ++e) bg_sum += (double)a[e];
            // the tail of the slice (fewer than PER cells): first lanes
            i = c_lo + (c_hi - c_lo) / PER * PER + threadIdx.x;
            if (i < c_hi) bg_sum += (double)gb[i];
        } else {
            for (i = c_lo + threadIdx.x; i < c_hi; i += kDT) bg_sum += (double)gb[i];
        }
    }
    // per-block partial sums: T per thread, f64 across the block
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kNVal; ++k) {
        const double v = (k < kNVal - 1) ? (double)vals[k < kNVal - 1 ? k : 0] : bg_sum;
        const double sm = wave_sum<double>(v);
        if (lane == 0) red[wave][k] = sm;
    }
    __syncthreads();
    if (threadIdx.x < kNVal) {
        double sm = 0.0;
#pragma unroll
        for (int w = 0; w < kDT / kWave; ++w) sm += red[w][threadIdx.x];
        partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sm;
    }
}

// The same for a BATCH of poses, pose loop inside: a thread keeps K points and their gradient sums
// in registers while the poses go by, so the cloud is read once and the point gradients are stored
// once -- launched pose by pose (k_own_pullback with FIRST = false) every pose re-reads the cloud and
// read-modify-writes 16 bytes per point: 6.3 GB per pose against 1.3 GB of gathers at 50 M fp64
// points.  The per-pose sums cannot stay in registers across the poses (13 per pose): each pose's 13
// sums over the thread's K points are reduced across the wave (DPP row shifts for fp32) and added to
// the block's accumulators in LDS by one lane -- 13 reductions per K points and pose.
constexpr int kDBatch = 64;  // poses per launch (their block accumulators: 64 x 14 doubles of LDS)
// (K = 4 / 2 would halve the share of the reductions but needs ~230 / ~175 registers: 2 waves per SIMD)
template <typename T> __host__ __device__ constexpr int own_batch_k() { return sizeof(T) == 4 ? 2 : 1; }

template <typename T, bool HAS_PW>
__global__ __launch_bounds__(kDT) void k_own_pullback_batch(GridDesc<3> gd, int64_t P, int64_t per_block,
                                                            int64_t cells_per_block,
                                                            const T* __restrict__ points, const T* __restrict__ pw,
                                                            const T* __restrict__ g, const T* __restrict__ rot,
                                                            const T* __restrict__ trans, const T* __restrict__ ow,
                                                            int64_t b0, int nb, int accumulate,
                                                            T* __restrict__ ds_dpoints, T* __restrict__ ds_dpw,
                                                            double* __restrict__ partials) {
    constexpr int K = own_batch_k<T>();
    // fp64 (PARK): a pose's 13 per-point values are parked per thread in LDS and summed by one wave per value --
    // 4 reads + one wave reduction per value and BLOCK -- instead of 13 wave reductions (156 ds_bpermute + 13 LDS
    // atomics) per WAVE and pose, which made the in-kernel pose loop slower than a launch per pose (r05: 14.0 vs
    // 12.8 ms at C5's share).  Two barriers per pose; the order of every sum is fixed.
    constexpr bool PARK = sizeof(T) == 8;
    static_assert(!PARK || K == 1, "one point per thread parks its values");
    __shared__ double acc[kDBatch][kNVal];
    __shared__ double park[PARK ? (kNVal - 1) * kDT : 1];
    for (int i = threadIdx.x; i < nb * kNVal; i += kDT) (&acc[0][0])[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t p_lo = (int64_t)blockIdx.x * per_block;
    const int64_t p_hi = p_lo + per_block < P ? p_lo + per_block : P;
    const int n0 = gd.n[0], n1 = gd.n[1], n2 = gd.n[2];
    const uint32_t row1 = (uint32_t)n0, row2 = (uint32_t)n0 * (uint32_t)n1;
    struct Stage {
        T dlo[3], gv[8];
        uint32_t in;
        int xsel;
        bool ok, edge;
    };
#pragma unroll 1
    for (int64_t base = p_lo; base < p_hi; base += (int64_t)kDT * K) {  // (uniform)
        T pt[K][3], pwi[K], gacc[K][4];
        bool live[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int64_t p = base + threadIdx.x + (int64_t)j * kDT;
            live[j] = p < p_hi;
            const int64_t q = p < P ? p : P - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) pt[j][c] = points[q * 3 + c];
            pwi[j] = HAS_PW ? pw[q] : T(1);
#pragma unroll
            for (int c = 0; c < 4; ++c) gacc[j][c] = T(0);
        }
#pragma unroll 1
        for (int bl = 0; bl < nb; ++bl) {
            const int64_t b = b0 + bl;
            const Pose<T, 3, 3> ps = load_pose<T, 3, 3>(rot, trans, ow, b);
            const OwnXform<T> xf = own_xform<T>(ps, gd);
            const T* gb = g + b * gd.G;
            T vals[kNVal - 1];
#pragma unroll
            for (int k = 0; k < kNVal - 1; ++k) vals[k] = T(0);
            auto front = [&](int j, Stage& st) {
                int ref0[3];
                st.ok = own_ref<T>(pt[j], ps, xf, ref0, st.dlo) && live[j];
                if (!st.ok) ref0[0] = ref0[1] = ref0[2] = 0;
                const bool lo1 = ref0[1] >= 0, hi1 = ref0[1] + 1 < n1, lo2 = ref0[2] >= 0, hi2 = ref0[2] + 1 < n2;
                const int xb = ref0[0] < 0 ? 0 : (ref0[0] > n0 - 2 ? n0 - 2 : ref0[0]);
                st.xsel = ref0[0] - xb;
                const uint32_t cbase = ((uint32_t)ref0[2] * (uint32_t)n1 + (uint32_t)ref0[1]) * (uint32_t)n0 + (uint32_t)xb;
                st.in = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s1 = r & 1, s2 = r >> 1;
                    const bool in = st.ok && (s1 ? hi1 : lo1) && (s2 ? hi2 : lo2);
                    st.in |= in ? (1u << r) : 0u;
                    const uint32_t off = cbase + (s1 ? row1 : 0u) + (s2 ? row2 : 0u);
                    typedef T Pair __attribute__((ext_vector_type(2), aligned(sizeof(T))));
                    const Pair pr = *(const Pair*)(gb + (in ? off : 0u));
                    st.gv[2 * r] = pr[0];
                    st.gv[2 * r + 1] = pr[1];
                }
                st.edge = st.in != 0xfu || st.xsel != 0;
            };
            auto back = [&](int j, const Stage& st) {
                T gi[8];
                if (__ballot(st.edge) == 0ull) {  // (uniform)
#pragma unroll
                    for (int s = 0; s < 8; ++s) gi[s] = st.gv[s];
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int r = s >> 1;
                        const T gx = (s & 1) ? (st.xsel == 0 ? st.gv[2 * r + 1] : (st.xsel < 0 ? st.gv[2 * r] : T(0)))
                                             : (st.xsel == 0 ? st.gv[2 * r] : (st.xsel > 0 ? st.gv[2 * r + 1] : T(0)));
                        gi[s] = ((st.in >> r) & 1u) ? gx : T(0);
                    }
                }
                T dpw_part = T(0), dcoord[3] = {T(0), T(0), T(0)}, dow_part = T(0);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const T dweight = voxel_weight<T, 3>(st.dlo, s, gi[s]);  // raster_pullback.jl:55
                    dow_part += dweight * pwi[j];                            // :57
                    dpw_part += dweight * ps.ow;                             // :58
                    const T factor = gi[s] * ps.ow * pwi[j];                 // :60
#pragma unroll
                    for (int n = 0; n < 3; ++n) dcoord[n] += factor * interp_weight<T, 3>(n, st.dlo, s);
                }
                T scaled[3];
#pragma unroll
                for (int n = 0; n < 3; ++n) scaled[n] = st.ok ? dcoord[n] * xf.scale[n] : T(0);  // :67
#pragma unroll
                for (int n = 0; n < 3; ++n) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)  // (0 * NaN of a rejected point must not reach the sums)
                        vals[n + c * 3] = fma_t(scaled[n], st.ok ? pt[j][c] : T(0), vals[n + c * 3]);  // :69
                    vals[9 + n] += scaled[n];                                                                 // :68
                }
                vals[12] += st.ok ? dow_part : T(0);
#pragma unroll
                for (int c = 0; c < 3; ++c) {  // rotation' * scaled (:70), summed over the poses in index order (:141)
                    T v = ps.R[0 + c * 3] * scaled[0];
                    v = v + ps.R[1 + c * 3] * scaled[1];
                    v = v + ps.R[2 + c * 3] * scaled[2];
                    gacc[j][c] += v;
                }
                gacc[j][3] += st.ok ? dpw_part : T(0);
            };
            Stage sa, sb;
            front(0, sa);
#pragma unroll
            for (int j = 0; j < K; ++j) {  // (gathers of point j + 1 in flight while point j is worked on)
                Stage& cur = (j & 1) ? sb : sa;
                Stage& nx = (j & 1) ? sa : sb;
                if (j + 1 < K) front(j + 1, nx);
                back(j, cur);
            }
            // the pose's 13 sums over this wave's K x 64 points -> the block's accumulators
            if constexpr (PARK) {
#pragma unroll
                for (int k = 0; k < kNVal - 1; ++k) park[k * kDT + threadIdx.x] = (double)vals[k];
                __syncthreads();
                for (int k = threadIdx.x / kWave; k < kNVal - 1; k += kDT / kWave) {  // (uniform per wave)
                    const double* pk = park + k * kDT + lane;
                    double sm = (pk[0] + pk[kWave]) + (pk[2 * kWave] + pk[3 * kWave]);
                    static_assert(kDT == 4 * kWave, "four waves park");
                    sm = wave_sum<double>(sm);
                    if (lane == 0) acc[bl][k] += sm;  // (the one writer of this value in the block)
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int k = 0; k < kNVal - 1; ++k) {
                    const float sm = wave_sum_lane63((float)vals[k]);
                    if (lane == kWave - 1) atomicAdd(&acc[bl][k], (double)sm);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int64_t p = base + threadIdx.x + (int64_t)j * kDT;
            if (!live[j]) continue;
            if (!accumulate) {
#pragma unroll
                for (int c = 0; c < 3; ++c) __builtin_nontemporal_store(gacc[j][c], &ds_dpoints[p * 3 + c]);
                if (ds_dpw) __builtin_nontemporal_store(gacc[j][3], &ds_dpw[p]);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) ds_dpoints[p * 3 + c] += gacc[j][c];
                if (ds_dpw) ds_dpw[p] += gacc[j][3];
            }
        }
    }
    // ds_dbackground: this block's slice of every pose's grid
    for (int bl = 0; bl < nb; ++bl) {
        const T* gb = g + (b0 + bl) * gd.G;
        const int64_t c_lo = (int64_t)blockIdx.x * cells_per_block;
        const int64_t c_hi = c_lo + cells_per_block < gd.G ? c_lo + cells_per_block : gd.G;
        double bg_sum = 0.0;
        for (int64_t i = c_lo + threadIdx.x; i < c_hi; i += kDT) bg_sum += (double)__builtin_nontemporal_load(gb + i);
        bg_sum = wave_sum<double>(bg_sum);
        if (lane == 0) atomicAdd(&acc[bl][kNVal - 1], bg_sum);
    }
    __syncthreads();
    // partials[(pose, value)][block]
    for (int i = threadIdx.x; i < nb * kNVal; i += kDT)
        partials[(size_t)i * gridDim.x + blockIdx.x] = (&acc[0][0])[i];
}

// per-pose sums of a batch launch: block per (value, pose)
template <typename T>
__global__ __launch_bounds__(1024) void k_own_reduce_batch(const double* __restrict__ partials, int nblocks,
                                                           int64_t b0, T* __restrict__ ds_drotation,
                                                           T* __restrict__ ds_dtranslation,
                                                           T* __restrict__ ds_dbackground,
                                                           T* __restrict__ ds_dout_weight) {
    __shared__ double wsum[16];
    const int k = blockIdx.x, bl = blockIdx.y;
    const int64_t b = b0 + bl;
    double s = 0.0;
    for (int t = threadIdx.x; t < nblocks; t += 1024) s += partials[((size_t)bl * kNVal + k) * nblocks + t];
    s = wave_sum<double>(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        if (k < 9) ds_drotation[b * 9 + k] = (T)tot;
        else if (k < 12) ds_dtranslation[b * 3 + (k - 9)] = (T)tot;
        else if (k == 12) ds_dout_weight[b] = (T)tot;
        else ds_dbackground[b] = (T)tot;
    }
}

// per-pose sums from the per-block partials, in block order
template <typename T>
__global__ __launch_bounds__(1024) void k_own_reduce(const double* __restrict__ partials, int nblocks, int64_t b,
                                                     T* __restrict__ ds_drotation, T* __restrict__ ds_dtranslation,
                                                     T* __restrict__ ds_dbackground, T* __restrict__ ds_dout_weight) {
    __shared__ double wsum[16];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int t = threadIdx.x; t < nblocks; t += 1024) s += partials[(size_t)k * nblocks + t];
    s = wave_sum<double>(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        if (k < 9) ds_drotation[b * 9 + k] = (T)tot;
        else if (k < 12) ds_dtranslation[b * 3 + (k - 9)] = (T)tot;
        else if (k == 12) ds_dout_weight[b] = (T)tot;
        else ds_dbackground[b] = (T)tot;
    }
}

// ---------------------------------------------------------------- host side
static size_t oalign(size_t x) { return (x + 255) & ~(size_t)255; }

struct OwnPlan {
    int64_t nSC, nL1, nL2, Bw;
    uint32_t list_cap, cap;
    int max_items, max_slabs, max_split;
    size_t off_ctl, off_b0, off_b1, off_b2, off_mw1, off_mw2, off_rec, off_list, off_items, off_split, off_slabs, total;
    size_t rec_stride, list_stride, items_stride, split_stride;
};

struct OwnKnobs {
    int cap_div, cap_min, max_slabs, fixed;
};
static const OwnKnobs& oknobs() {
    static const OwnKnobs k = [] {
        auto env_int = [](const char* name, int dflt, int lo, int hi) { return env_knob(name, dflt, lo, hi); };
        OwnKnobs q;
        q.cap_div = env_int("DPR_OWN_CAP_DIV", 512, 1, 1 << 20);     // a part: ~1.6 P / 512 visits
        q.cap_min = env_int("DPR_OWN_CAP_MIN", 8192, 64, 1 << 24);
        q.max_slabs = env_int("DPR_OWN_MAX_SLABS", 0, 0, 1 << 16);   // 0: 1024 for one or two poses, 2048 for batches
        q.fixed = env_int("DPR_FIXED_POINT", 1, 0, 1);
        return q;
    }();
    return k;
}

static OwnPlan make_oplan(int op, const OGeom& tg, int64_t P, int64_t B) {
    OwnPlan pl;
    pl.nSC = (P + kSC - 1) / kSC;
    pl.nL1 = (P + kL1 - 1) / kL1;
    pl.nL2 = (pl.nL1 + kL2 - 1) / kL2;
    pl.Bw = B < 1 ? 1 : (B < kOwnBw ? B : kOwnBw);
    // a part holds ~1/512 of the visits of the whole pose group: the number of parts -- and of the slabs they
    // leave their tiles in -- stays bounded whatever the batch (with a per-pose cap a clustered cloud ran out of
    // slabs at 4 poses, its heaviest tiles stayed whole: 10.5 ms instead of 1.5)
    const int64_t cap = (P + P / 2) * pl.Bw / oknobs().cap_div;
    pl.cap = (uint32_t)(cap < oknobs().cap_min ? oknobs().cap_min : (cap > 0x3fffffff ? 0x3fffffff : cap));
    pl.max_slabs = oknobs().max_slabs ? oknobs().max_slabs : (pl.Bw <= 2 ? 1024 : 2048);
    pl.max_split = pl.max_slabs / 2 + 1;
    pl.max_items = tg.NT + pl.max_slabs;
    // candidate chunks per tile: 64 times the average of a cloud that fills the grid, 256 .. 8192;
    // a tile with more takes every chunk as a candidate (its level-0 tests still cull exactly)
    int64_t lc = (64 * pl.nL1 / tg.NT + 63) / 64 * 64;
    lc = lc < 256 ? 256 : (lc > 8192 ? 8192 : lc);
    pl.list_cap = (uint32_t)lc;
    size_t o = 0;
    pl.off_ctl = o;
    o += oalign((size_t)pl.Bw * kCtlWords * 4);
    pl.off_b0 = o;
    o += oalign((size_t)(pl.nSC + 1) * sizeof(IBox) * pl.Bw);
    pl.off_b1 = o;
    o += oalign((size_t)(pl.nL1 + 1) * sizeof(IBox) * pl.Bw);
    pl.off_b2 = o;
    o += oalign((size_t)(pl.nL2 + 1) * sizeof(IBox) * pl.Bw);
    pl.off_mw1 = o;  // max | min non-zero, per chunk
    o += oalign((size_t)(pl.nL1 + 1) * 4 * 2);
    pl.off_mw2 = o;
    o += oalign((size_t)(pl.nL2 + 1) * 4);
    pl.rec_stride = oalign((size_t)tg.NT * sizeof(TileRec));
    pl.off_rec = o;
    o += pl.rec_stride * pl.Bw;
    pl.list_stride = oalign((size_t)pl.list_cap * 4 * tg.NT);
    pl.off_list = o;
    o += pl.list_stride * pl.Bw;
    pl.items_stride = oalign((size_t)kBuckets * pl.max_items * sizeof(OwnItem));
    pl.off_items = o;
    o += pl.items_stride * pl.Bw;
    pl.split_stride = oalign((size_t)pl.max_split * 4);
    pl.off_split = o;
    o += pl.split_stride * pl.Bw;
    pl.off_slabs = o;
    o += oalign((size_t)pl.max_slabs * kCells * 8);
    pl.total = o;
    return pl;
}

bool owner_supported(const int64_t* grid) {
    OGeom tg;
    return make_ogeom(grid, &tg);
}

int64_t owner_tiles(const int64_t* grid) {
    OGeom tg;
    return make_ogeom(grid, &tg) ? tg.NT : 0;
}

size_t owner_workspace_bytes(int op, const int64_t* grid, int64_t P, int64_t B) {
    OGeom tg;
    if (!make_ogeom(grid, &tg) || P >= ((int64_t)1 << 32)) return (size_t)-1;
    // pullback: per-block partial sums only (at most 8192 blocks, up to 64 poses per launch)
    if (op == DPR_OP_PULLBACK) return oalign((size_t)8192 * kNVal * 8 * (size_t)(B < 1 ? 1 : (B < 64 ? B : 64)));
    return make_oplan(DPR_OP_RASTER, tg, P, B).total;
}

#define DPR_HIP(expr)                                                                \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            return fail(DPR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static GridDesc<3> ogrid_desc(const int64_t* grid, int64_t G) {
    GridDesc<3> gd;
    for (int d = 0; d < 3; ++d) gd.n[d] = (int)grid[d];
    gd.G = G;
    return gd;
}

static OwnPlanArgs plan_args(const OwnPlan& pl, char* ws) {
    OwnPlanArgs pa;
    pa.ws = ws;
    pa.off_ctl = pl.off_ctl;
    pa.off_rec = pl.off_rec;
    pa.off_list = pl.off_list;
    pa.off_items = pl.off_items;
    pa.off_split = pl.off_split;
    pa.rec_stride = pl.rec_stride;
    pa.list_stride = pl.list_stride;
    pa.items_stride = pl.items_stride;
    pa.split_stride = pl.split_stride;
    pa.list_cap = pl.list_cap;
    pa.max_items = pl.max_items;
    pa.max_slabs = pl.max_slabs;
    pa.max_split = pl.max_split;
    pa.cap = pl.cap;
    pa.fixed = oknobs().fixed;
    pa.b1 = (const IBox*)(ws + pl.off_b1);
    pa.b2 = (const IBox*)(ws + pl.off_b2);
    pa.mw1 = nullptr;
    pa.nL1 = pl.nL1;
    pa.nL2 = pl.nL2;
    return pa;
}
static OwnTileArgs tile_args(const OwnPlan& pl, const char* ws) {
    OwnTileArgs ta;
    ta.ws = ws;
    ta.off_ctl = pl.off_ctl;
    ta.off_rec = pl.off_rec;
    ta.off_list = pl.off_list;
    ta.off_items = pl.off_items;
    ta.rec_stride = pl.rec_stride;
    ta.list_stride = pl.list_stride;
    ta.items_stride = pl.items_stride;
    ta.max_items = pl.max_items;
    ta.nL1 = pl.nL1;
    ta.nSC = pl.nSC;
    ta.b0 = (const IBox*)(ws + pl.off_b0);
    ta.dbg_words = (size_t)pl.max_slabs * kCells * 2;
    return ta;
}

template <typename T> static bool vec_ok(const T* points, const T* pw) {
    return (((uintptr_t)points) & 15) == 0 && (((uintptr_t)pw) & 15) == 0;
}

// boxes + plan of poses [b0, b0 + nb)
template <typename T>
static int own_prepare(hipStream_t st, const OGeom& tg, const GridDesc<3>& gd, const OwnPlan& pl, char* ws,
                       int64_t P, const T* points, const T* pw, const T* rot, const T* trans, const T* ow,
                       int64_t b0, int64_t nb) {
    uint32_t* ctl = (uint32_t*)(ws + pl.off_ctl);
    const int ctl_words = (int)(pl.Bw * kCtlWords);
    float* mw1 = pw ? (float*)(ws + pl.off_mw1) : (float*)nullptr;
    if (P > 0) {
        OwnBoxArgs ba;
        ba.b0 = (IBox*)(ws + pl.off_b0);
        ba.b1 = (IBox*)(ws + pl.off_b1);
        ba.mw1 = mw1;
        ba.nSC = pl.nSC;
        ba.nL1 = pl.nL1;
        ba.ctl = ctl;
        ba.ctl_words = ctl_words;
        hipLaunchKernelGGL((k_own_boxes<T>), dim3((unsigned)pl.nL1), dim3(256), 0, st, gd, P, points, pw,
                           vec_ok(points, pw) ? 1 : 0, rot, trans, b0, (int)nb, ba);
        hipLaunchKernelGGL(k_own_boxes2, dim3((unsigned)((pl.nL2 + 3) / 4), (unsigned)nb), dim3(256), 0, st,
                           pl.nL1, pl.nL2, (const IBox*)(ws + pl.off_b1), (const float*)mw1,
                           (IBox*)(ws + pl.off_b2), (float*)(ws + pl.off_mw2));
    } else {
        DPR_HIP(hipMemsetAsync(ctl, 0, (size_t)ctl_words * 4, st));
    }
    stage_mark(st);
    OwnPlanArgs pa = plan_args(pl, ws);
    pa.mw1 = mw1;
    hipLaunchKernelGGL((k_own_plan<T>), dim3((unsigned)tg.NT, (unsigned)nb), dim3(256), 0, st, tg, gd, P, ow,
                       pw ? 1 : 0, b0, pa);
    stage_mark(st);
    return DPR_OK;
}

template <typename K> static int own_lds(K kernel, size_t bytes) {
    DPR_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DPR_OK;
}

template <typename T>
int raster_owner(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                 T* out, const T* points, const T* rot, const T* trans, const T* bg, const T* ow,
                 const T* pw, void* ws_, size_t ws_bytes) {
    OGeom tg;
    if (!make_ogeom(grid, &tg))
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: grid needs more than %d tiles", kMaxOwnTiles);
    if (P >= ((int64_t)1 << 32)) return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: P must be < 2^32");
    // (DPR_FLAG_KEEP_BINNING: the pullback of this path reads nothing a forward could leave -- accepted)
    (void)flags;
    const size_t need = owner_workspace_bytes(DPR_OP_RASTER, grid, P, B);
    if (!ws_ || ws_bytes < need)
        return fail(DPR_ERR_WORKSPACE, "DPR_ALGO_CHUNKED raster needs %zu workspace bytes, got %zu", need,
                    ws_ ? ws_bytes : (size_t)0);
    char* ws = (char*)ws_;
    const GridDesc<3> gd = ogrid_desc(grid, G);
    const OwnPlan pl = make_oplan(DPR_OP_RASTER, tg, P, B);
    const size_t lds = sizeof(OwnWalkLds) + (size_t)kPCells * 8;
    if (pw) {
        if (int rc = own_lds(k_own_splat<T, true>, lds)) return rc;
    } else {
        if (int rc = own_lds(k_own_splat<T, false>, lds)) return rc;
    }
    for (int64_t b0 = 0; b0 < B; b0 += pl.Bw) {
        const int64_t nb = (B - b0 < pl.Bw) ? B - b0 : pl.Bw;
        if (int rc = own_prepare<T>(st, tg, gd, pl, ws, P, points, pw, rot, trans, ow, b0, nb))
            return rc;
        const OwnTileArgs ta = tile_args(pl, ws);
        unsigned long long* slabs = (unsigned long long*)(ws + pl.off_slabs);
        const dim3 tgrid((unsigned)pl.max_items, (unsigned)nb);
        if (pw)
            hipLaunchKernelGGL((k_own_splat<T, true>), tgrid, dim3(kOT), lds, st, tg, gd, P, points, pw,
                               vec_ok(points, pw) ? 1 : 0, rot, trans, ow, bg, b0, ta, slabs, out);
        else
            hipLaunchKernelGGL((k_own_splat<T, false>), tgrid, dim3(kOT), lds, st, tg, gd, P, points, pw,
                               vec_ok(points, pw) ? 1 : 0, rot, trans, ow, bg, b0, ta, slabs, out);
        stage_mark(st);
        const int ncomb = pl.max_split * 4 < 1024 ? pl.max_split * 4 : 1024;
        hipLaunchKernelGGL((k_own_combine<T>), dim3((unsigned)ncomb, (unsigned)nb), dim3(kOT), 0, st, tg, gd, bg,
                           b0, ta, pl.off_split, pl.split_stride, pl.max_split,
                           (const unsigned long long*)slabs, out);
        stage_mark(st);
    }
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// compute units of the current device
static int own_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
        n = 256;
    return n;
}
static int own_pullback_blocks(int64_t P, int64_t G) {
    int64_t nb = (int64_t)own_cu_count() * kDBlocksPerCU;
    if (nb > 8192) nb = 8192;
    const int64_t by_points = (P + kDT - 1) / kDT, by_cells = (G + 4 * kDT - 1) / (4 * kDT);
    const int64_t want = by_points > by_cells ? by_points : by_cells;
    if (nb > want) nb = want;
    return (int)(nb < 1 ? 1 : nb);
}

// (experiments: DPR_OWN_BATCH_F64=0 -- a launch per pose for fp64 batches, round 5's choice)
static bool own_batch_f64() {
    static const bool v = env_knob("DPR_OWN_BATCH_F64", 1, 0, 1) != 0;
    return v;
}

template <typename T>
int pullback_owner(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                   const T* g, const T* points, const T* rot, const T* trans, const T* ow, const T* pw,
                   T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow, T* d_pw, void* ws_, size_t ws_bytes) {
    // (DPR_FLAG_REUSE_BINNING: this pullback reads nothing the forward left -- accepted, nothing to validate)
    (void)flags;
    const size_t need = owner_workspace_bytes(DPR_OP_PULLBACK, grid, P, B);
    if (!ws_ || ws_bytes < need)
        return fail(DPR_ERR_WORKSPACE, "DPR_ALGO_CHUNKED pullback needs %zu workspace bytes, got %zu", need,
                    ws_ ? ws_bytes : (size_t)0);
    const GridDesc<3> gd = ogrid_desc(grid, G);
    const int nblocks = own_pullback_blocks(P, G);
    int64_t per_block = ((P + nblocks - 1) / nblocks + kDT - 1) / kDT * kDT;
    if (per_block < kDT) per_block = kDT;
    const int64_t cells_per_block = ((G + nblocks - 1) / nblocks + 3) / 4 * 4;
    // (the partial sums sit at the start of the workspace: a forward's boxes and plan, further up in
    // a KEEP / REUSE pair's shared buffer, are not this call's business -- it overwrites the header)
    double* partials = (double*)ws_;
    if (B > 1 && (sizeof(T) == 4 || own_batch_f64())) {
        // pose loop inside the kernel, kDBatch poses per launch (further launches add to the point
        // gradients): fp32 1e7 points x 16 poses -> 256^3 2.87 -> 2.36 ms, x 4 poses 0.68 -> 0.56.  fp64 (round 6):
        // one point per thread, the per-pose sums parked in LDS and reduced per block -- with 13 wave reductions
        // per point and pose (round 5) the loop lost to a launch per pose, 14.0 vs 12.8 ms at C5's share
        constexpr int K = own_batch_k<T>();
        int64_t pb = ((P + nblocks - 1) / nblocks + (int64_t)kDT * K - 1) / ((int64_t)kDT * K) * ((int64_t)kDT * K);
        if (pb < (int64_t)kDT * K) pb = (int64_t)kDT * K;
        for (int64_t b0 = 0; b0 < B; b0 += kDBatch) {
            const int nb = (int)(B - b0 < kDBatch ? B - b0 : kDBatch);
            if (pw)
                hipLaunchKernelGGL((k_own_pullback_batch<T, true>), dim3((unsigned)nblocks), dim3(kDT), 0, st, gd, P,
                                   pb, cells_per_block, points, pw, g, rot, trans, ow, b0, nb, b0 > 0 ? 1 : 0, d_pts,
                                   d_pw, partials);
            else
                hipLaunchKernelGGL((k_own_pullback_batch<T, false>), dim3((unsigned)nblocks), dim3(kDT), 0, st, gd, P,
                                   pb, cells_per_block, points, pw, g, rot, trans, ow, b0, nb, b0 > 0 ? 1 : 0, d_pts,
                                   d_pw, partials);
            stage_mark(st);  // (every launch group marks its two stages: timing.py folds the repeats)
            hipLaunchKernelGGL((k_own_reduce_batch<T>), dim3(kNVal, (unsigned)nb), dim3(1024), 0, st,
                               (const double*)partials, nblocks, b0, d_rot, d_trans, d_bg, d_ow);
            stage_mark(st);
        }
        DPR_HIP(hipGetLastError());
        return DPR_OK;
    }
    for (int64_t b = 0; b < B; ++b) {
        // the point gradients accumulate over poses: one pose per launch (stream order = race-free
        // read-modify-write)
#define DPR_OWN_PB(HAS_PW, FIRST)                                                                            \
    hipLaunchKernelGGL((k_own_pullback<T, HAS_PW, FIRST>), dim3((unsigned)nblocks), dim3(kDT), 0, st, gd, P, \
                       per_block, cells_per_block, points, pw, g, rot, trans, ow, b, d_pts, d_pw, partials)
        if (pw) {
            if (b == 0) DPR_OWN_PB(true, true);
            else DPR_OWN_PB(true, false);
        } else {
            if (b == 0) DPR_OWN_PB(false, true);
            else DPR_OWN_PB(false, false);
        }
#undef DPR_OWN_PB
        stage_mark(st);
        hipLaunchKernelGGL((k_own_reduce<T>), dim3(kNVal), dim3(1024), 0, st, (const double*)partials, nblocks, b,
                           d_rot, d_trans, d_bg, d_ow);
        stage_mark(st);
    }
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

#define DPR_INST(T)                                                                                        \
    template int raster_owner<T>(hipStream_t, unsigned, const int64_t*, int64_t, int64_t, int64_t, T*,     \
                                 const T*, const T*, const T*, const T*, const T*, const T*, void*, size_t); \
    template int pullback_owner<T>(hipStream_t, unsigned, const int64_t*, int64_t, int64_t, int64_t,      \
                                   const T*, const T*, const T*, const T*, const T*, const T*, T*, T*, T*, \
                                   T*, T*, T*, void*, size_t);
DPR_INST(float)
DPR_INST(double)
}  // namespace dpr
