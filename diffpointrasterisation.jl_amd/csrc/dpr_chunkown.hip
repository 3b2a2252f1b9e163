// DPR_ALGO_CHUNKED for 2-D grids (projections 3 -> 2 and 2 -> 2): CHUNK-OWNED LDS tiles, pose
// loop inside the block.  This is the many-poses shape of the reference's README timings
// (/root/reference/README.md:189-192, ext/DiffPointRasterisationCUDAExt.jl:19-210 keeps one point
// and a block of poses per thread block; here a block keeps a CHUNK of points and walks the poses).
//
// A chunk is kChunk consecutive points of a spatially coherent (Hilbert-sorted) cloud: a compact
// blob in the model frame, so under any pose its projection covers a small pixel rectangle
// (~35 x 35 pixels for a 4096-point chunk of the 10 M-point cloud on 512^2), bounded WITHOUT
// looking at the points again: footprint = projected centre +- sum_j |R[d,j]| h_j of the chunk's
// 3-D bounding box.  Per block, with the chunk's points held in registers for ALL poses:
//
//   forward   (k_co_splat; footprints larger than the tile: work list + k_co_splat_wide)
//             per pose: 4 ds_add_f64 per point into the footprint tile in LDS, then the tile is
//             flushed with global atomic adds, one image row segment (contiguous x) per wave
//             instruction -- the 256-byte shape float atomics run at full rate in; `out` was
//             pre-filled with the background.  A pixel of a projection collects ~40 points, so
//             the LDS tile turns 4 global atomics per (point, pose) into ~0.3.
//   pullback  per pose: the ds_dout footprint is staged in LDS (coalesced rows), every point
//             gathers its 4 values from LDS; ds_dpoints / ds_dpoint_weight accumulate in
//             registers across the poses and are stored once per point; the per-pose sums go
//             wave -> LDS (f64) -> one partial per (block, pose), reduced by k_co_reduce.
//             No records, no permutation, no global atomics.
//
// Lanes of a wave take points that are kChunk / 64 apart in the chunk (not neighbours): sorted
// neighbours project onto the same pixels, and same-address LDS atomics serialise (26 -> 190
// cycles per wave instruction, profiles/r02_microbench_lds_conflicts.txt).
//
// Input that is not known to be coherent (no DPR_FLAG_COHERENT_POINTS) is cell-sorted into the
// workspace first (a counting sort into 4096 Hilbert-numbered cells of the model frame, 0.2 ms for
// 10 M points -- compact chunks are what counts, not sorted neighbours: amortised over the poses of the
// call) and the point gradients are scattered back through the permutation.  Whatever the order,
// the result is correct: a (chunk, pose) whose footprint does not fit the LDS tile, and any
// neighbour outside the footprint bound, goes to / comes from global memory directly.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/dpr.h"
#include "dpr_device.h"
#include "dpr_kernels_atomic.h"
#include "dpr_tiled.h"

namespace dpr {

constexpr int kCOThreads = 1024;
#ifndef DPR_CO_SPLAT_OCC
#define DPR_CO_SPLAT_OCC 8  // waves per SIMD the fp32 forward kernel is compiled for
#endif
#ifndef DPR_CO_PPT
#define DPR_CO_PPT 4
#endif
#ifndef DPR_CO_WIDE_BLOCKS
#define DPR_CO_WIDE_BLOCKS 1024
#endif
constexpr int kCOWideBlocks = DPR_CO_WIDE_BLOCKS; // grid of k_co_splat_wide (walks a work list)
#ifndef DPR_CO_WIDE_GROUP
#define DPR_CO_WIDE_GROUP 8
#endif
constexpr int kCOWideGroup = DPR_CO_WIDE_GROUP;   // poses per work item of k_co_splat_wide
constexpr int kCOPPT = DPR_CO_PPT;               // points per thread
constexpr int kCOChunk = kCOThreads * kCOPPT;    // 4096 points per block
constexpr int kCOWaves = kCOThreads / kWave;
#ifndef DPR_CO_CAP
#define DPR_CO_CAP 9984
#endif
constexpr int kCOCap = DPR_CO_CAP;               // LDS tile cells (8 bytes each): 78 KiB, 2 blocks / CU
                                                 // (round 2: 9216; fewer wide pairs, C4 forward -2 %)
constexpr int kCOWideCap = 2 * kCOCap;            // tile of k_co_splat_wide (one workgroup per CU)
#ifndef DPR_CO_GATHER_CAP
#define DPR_CO_GATHER_CAP 8192
#endif
constexpr int kCOGatherCap = DPR_CO_GATHER_CAP;   // cells (of T) of the pullback's ds_dout tile
constexpr int kCOMaxSlice = 64;                  // poses per block (per-pose sums live in LDS)
static_assert(kCOChunk / kWave == kCOWaves * kCOPPT, "spread assignment covers the chunk");

// index of the k-th point of thread (lane, wave) inside the chunk: lanes kChunk/64 apart
// workgroup barrier that orders LDS traffic only (global atomics / loads stay in flight)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int co_point(int lane, int wave, int k) {
    return lane * (kCOChunk / kWave) + wave * kCOPPT + k;
}

template <typename T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const T u = __shfl_xor(v, o, kWave);
        v = u < v ? u : v;
    }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const T u = __shfl_xor(v, o, kWave);
        v = u > v ? u : v;
    }
    return v;
}

// Chunk points into registers + the chunk's bounding box (centre c, half extent h), identical
// in every thread.  Non-finite coordinates take no part in the box (such points are rejected by
// ref_and_deltas for every pose).  Returns false when the chunk has no finite point.
template <typename T, int NI>
__device__ __forceinline__ bool co_load_chunk(const T* __restrict__ points,
                                              const T* __restrict__ pw, int64_t P, int64_t base,
                                              T (&pt)[kCOPPT][NI], T (&w)[kCOPPT],
                                              bool (&live)[kCOPPT], T (&c)[NI], T (&h)[NI],
                                              T (*sbox)[2 * 3]) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    T lo[NI], hi[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        lo[j] = T(INFINITY);
        hi[j] = T(-INFINITY);
    }
#pragma unroll
    for (int k = 0; k < kCOPPT; ++k) {
        const int64_t p = base + co_point(lane, wave, k);
        live[k] = p < P;
        const int64_t pl = live[k] ? p : P - 1;
        load_point<T, NI>(points, pl, pt[k]);
        // a slot past the end of the cloud holds a NaN point: ref_and_deltas rejects it for every
        // pose, so the kernels need no per-point `live` flag in their pose loops
        if (!live[k]) pt[k][0] = T(__builtin_nanf(""));
        w[k] = pw ? pw[pl] : T(1);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const T x = pt[k][j];
            if (live[k] && x - x == T(0)) {  // finite
                lo[j] = x < lo[j] ? x : lo[j];
                hi[j] = x > hi[j] ? x : hi[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        lo[j] = wave_min<T>(lo[j]);
        hi[j] = wave_max<T>(hi[j]);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            sbox[wave][j] = lo[j];
            sbox[wave][3 + j] = hi[j];
        }
    }
    __syncthreads();
    bool any = true;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        T l = sbox[0][j], u = sbox[0][3 + j];
        // (not unrolled all the way: the compiler requests the 16 x 6 values at once -- 96 VGPRs in
        // fp32, 192 in fp64 -- and spills what lives across this point; with point weights the fp64
        // forward then needed 300 bytes of scratch per lane, which costs every dispatch ~0.14 ms)
#pragma unroll 2
        for (int q = 1; q < kCOWaves; ++q) {
            l = sbox[q][j] < l ? sbox[q][j] : l;
            u = sbox[q][3 + j] > u ? sbox[q][3 + j] : u;
        }
        any = any && (l <= u);
        c[j] = T(0.5) * l + T(0.5) * u;
        h[j] = T(0.5) * u - T(0.5) * l;
    }
    return any;
}

// (an upper bound within 2^-7 of) max |point_weight| over the chunk (1 without point weights), identical
// in every thread; NaN and Inf come out non-finite -- and so does a chunk whose non-zero weights span
// more than 2^10 (wrange_*, fix_guard_range in dpr_device.h: its fp32 sums then run on f64 atomics
// instead of fixed point).  `smax`: one word per wave; ends with a barrier.
template <typename T, bool HAS_PW>
__device__ __forceinline__ float co_max_abs_weight(const T (&w)[kCOPPT], const bool (&live)[kCOPPT],
                                                   float* smax) {
    if (!HAS_PW) return 1.f;
    uint32_t key = 0;
#pragma unroll
    for (int k = 0; k < kCOPPT; ++k) key = live[k] ? wrange_merge(key, wrange_key((float)w[k])) : key;
    key = wrange_wave(key);
    uint32_t* const sk = (uint32_t*)smax;
    if ((threadIdx.x & (kWave - 1)) == 0) sk[threadIdx.x / kWave] = key;
    __syncthreads();
    uint32_t r = 0;
#pragma unroll
    for (int q = 0; q < kCOWaves; ++q) r = wrange_merge(r, sk[q]);
    return wrange_guarded_max(r >> 16, r & 0xffffu);
}

// Pixel rectangle [lo, hi] (inclusive, clipped to the grid) that bounds every in-grid neighbour
// of every point of the chunk under pose `ps`, with one cell of slack for rounding.  Returns the
// number of cells (0: nothing lands in the grid; > kCOCap: does not fit the LDS tile).
template <typename T, int NI>
__device__ __forceinline__ int64_t co_footprint(const T (&c)[NI], const T (&h)[NI],
                                                const Pose<T, NI, 2>& ps, const GridDesc<2>& gd,
                                                int (&lo)[2], int (&hi)[2]) {
    int64_t cells = 1;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        T pc = ps.R[d] * c[0], ph = (ps.R[d] < T(0) ? -ps.R[d] : ps.R[d]) * h[0];
#pragma unroll
        for (int j = 1; j < NI; ++j) {
            const T r = ps.R[d + j * 2];
            pc = pc + r * c[j];
            ph = ph + (r < T(0) ? -r : r) * h[j];
        }
        ph = ph * T(1.0001) + T(1e-6);
        const T origin = T(-1) - ps.t[d];
        const T scale = T(gd.n[d]) / T(2);
        T a = ((pc - ph) - origin) * scale - T(2.5);  // ref0 >= coord - 1.5, one cell of slack
        T b = ((pc + ph) - origin) * scale + T(1.5);  // ref0 + 1 <= coord + 0.5, one cell of slack
        // clamp before float -> int (also maps NaN to an empty range)
        a = a > T(-4) ? a : T(-4);
        a = a < T(gd.n[d] + 4) ? a : T(gd.n[d] + 4);
        b = b > T(-4) ? b : T(-4);
        b = b < T(gd.n[d] + 4) ? b : T(gd.n[d] + 4);
        int l = (int)a, u = (int)b + 1;
        if (!(a == a) || !(b == b)) {
            l = 1;
            u = 0;
        }
        lo[d] = l < 0 ? 0 : l;
        hi[d] = u > gd.n[d] - 1 ? gd.n[d] - 1 : u;
        cells *= (hi[d] >= lo[d]) ? (int64_t)(hi[d] - lo[d] + 1) : 0;
    }
    return cells;
}

#ifndef DPR_CO_NO_FOOT_TABLE
// Footprints of the block's poses, computed ONCE per pose by one thread each and parked in LDS
// (every thread recomputing them cost ~50 VALU per pose: a tenth of the pose loop's instructions).
// Ends with a barrier.  foot[j] = {lo0, lo1, hi0, hi1} of pose b_lo + j.
// fexp (forward, fp32 data): exponent of the pose's fixed-point scale, from |out_weight| times the
// chunk's max |point_weight|; kFixOff = f64 atomics for this pose (non-finite weights, fp64 data).
constexpr int kFixOff = -(1 << 30);
template <typename T, int NI>
__device__ __forceinline__ void co_fill_footprints(int (*foot)[4], const T (&c)[NI], const T (&h)[NI],
                                                   const GridDesc<2>& gd, const T* __restrict__ rot,
                                                   const T* __restrict__ trans, int64_t b_lo, int nbs,
                                                   int* fexp = nullptr, const T* __restrict__ ow = nullptr,
                                                   float maxpw = 1.f, int fixed = 0) {
    if ((int)threadIdx.x < nbs) {
        const Pose<T, NI, 2> ps = load_pose<T, NI, 2>(rot, trans, ow, b_lo + threadIdx.x);
        int lo[2], hi[2];
        (void)co_footprint<T, NI>(c, h, ps, gd, lo, hi);
        foot[threadIdx.x][0] = lo[0];
        foot[threadIdx.x][1] = lo[1];
        foot[threadIdx.x][2] = hi[0];
        foot[threadIdx.x][3] = hi[1];
        if (fexp) {
            // a cell collects at most one contribution per point of the chunk
            const int e = fix_exponent(sizeof(T) == 4 ? fabsf((float)ps.ow) * maxpw : __builtin_inff(),
                                       (uint32_t)kCOChunk, fixed);
            fexp[threadIdx.x] = e == kFixNone ? kFixOff : e;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ int64_t co_read_footprint(const int (*foot)[4], int j, int (&lo)[2],
                                                     int (&hi)[2]) {
    lo[0] = foot[j][0];
    lo[1] = foot[j][1];
    hi[0] = foot[j][2];
    hi[1] = foot[j][3];
    const int W = hi[0] - lo[0] + 1, H = hi[1] - lo[1] + 1;
    return (W > 0 && H > 0) ? (int64_t)W * H : 0;
}
#endif

// What a DPR_FLAG_KEEP_BINNING forward leaves at the start of the workspace: the identity of the
// cloud whose sorted copy (+ permutation) follows.  A DPR_FLAG_REUSE_BINNING pullback skips its
// own sort and checks this ON THE DEVICE; on a mismatch it reads nothing through the stale
// permutation and returns NaN gradients (the host cannot see the mismatch without synchronising).
// The CONTENT of `points` is not checked: like the tiled path's binning, the sorted copy belongs
// to the raster call of the same call pair.
constexpr uint32_t kSortMagic = 0x44505253u;
struct alignas(16) SortHeader {
    uint32_t magic, elem, n_in, has_pw;
    int64_t P;
    uint64_t points, pw;
};
__global__ void k_co_write_header(SortHeader h, SortHeader* dst, uint32_t* wide_count) {
    if (threadIdx.x == 0) {
        *dst = h;
        *wide_count = 0;
    }
}
__device__ __forceinline__ bool sort_header_ok(const SortHeader* hdr, const SortHeader& want) {
    return hdr->magic == kSortMagic && hdr->elem == want.elem && hdr->n_in == want.n_in &&
           hdr->has_pw == want.has_pw && hdr->P == want.P && hdr->points == want.points &&
           hdr->pw == want.pw;
}

// DPR_FIXED_POINT=0 (experiment knob, read once): f64 LDS accumulators for fp32 data too
static int co_fixed_point() {
    static const int v = env_knob("DPR_FIXED_POINT", 1, 0, 1);
    return v;
}

// ------------------------------------------------------------------ forward
// Two kernels.  k_co_splat covers every (chunk, pose) whose footprint fits the LDS tile in one
// pass -- nearly all of them on a sorted cloud -- with a short inner loop (at most 64 VGPRs for
// fp32: two workgroups per CU).  What does not fit (sparse tails of the cloud, incoherent input,
// grids much larger than the cloud's resolution) is only FLAGGED per block, and k_co_splat_wide,
// launched after it over the same grid, handles those poses of the flagged blocks in row bands.
template <typename T> struct COSplatOcc {
    static constexpr int value = sizeof(T) == 4 ? DPR_CO_SPLAT_OCC : 4;
};

template <typename T, int NI, bool HAS_PW>
__global__ __launch_bounds__(kCOThreads, COSplatOcc<T>::value) void k_co_splat(
    GridDesc<2> gd, int64_t P, int64_t B, int poses_per_slice, const T* __restrict__ points,
    const T* __restrict__ pw, const T* __restrict__ rot, const T* __restrict__ trans,
    const T* __restrict__ ow, T* __restrict__ out, uint32_t* __restrict__ wide_count,
    uint4* __restrict__ wide_items, int fixed) {
    __shared__ double acc[kCOCap];
    __shared__ T sbox[kCOWaves][6];
    __shared__ float smaxw[kCOWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int i = threadIdx.x; i < kCOCap; i += kCOThreads) acc[i] = 0.0;
    T pt[kCOPPT][NI], w[kCOPPT], c[NI], h[NI];
    bool live[kCOPPT];
    const bool any = co_load_chunk<T, NI>(points, HAS_PW ? pw : nullptr, P,
                                          (int64_t)blockIdx.x * kCOChunk, pt, w, live, c, h, sbox);
    const float maxpw = co_max_abs_weight<T, HAS_PW>(w, live, smaxw);
    const int64_t b_lo = (int64_t)blockIdx.y * poses_per_slice;
    const int64_t b_hi = (b_lo + poses_per_slice < B) ? b_lo + poses_per_slice : B;
    const int nbs = any ? (int)(b_hi - b_lo) : 0;  // no finite point in this chunk: nothing to do
    // Blocks start at different poses: chunks along one viewing ray project onto the same
    // pixels, and float atomics of many workgroups into the same rows at the same time run an
    // order of magnitude slower than spread ones.
    const int rot0 = nbs > 0 ? (int)(blockIdx.x % (unsigned)nbs) : 0;
#ifndef DPR_CO_NO_FOOT_TABLE
    __shared__ int foot[kCOMaxSlice][4];
    __shared__ int fexp[kCOMaxSlice];
    co_fill_footprints<T, NI>(foot, c, h, gd, rot, trans, b_lo, nbs, fexp, ow, maxpw, fixed);
#endif
    unsigned long long wide_mask = 0;  // poses of this slice whose footprint outgrows the tile
    static_assert(kCOMaxSlice <= 64, "one bit per pose of a slice");
    for (int jb = 0; jb < nbs; ++jb) {
        const int64_t b = b_lo + (jb + rot0) % nbs;
        const Pose<T, NI, 2> ps = load_pose<T, NI, 2>(rot, trans, ow, b);
        int lo[2], hi[2];
#ifndef DPR_CO_NO_FOOT_TABLE
        const int64_t cells = co_read_footprint(foot, (int)(b - b_lo), lo, hi);
#else
        const int64_t cells = co_footprint<T, NI>(c, h, ps, gd, lo, hi);
#endif
        if (cells == 0) continue;  // uniform
        if (cells > kCOCap) {  // uniform: left to k_co_splat_wide
            wide_mask |= 1ull << (unsigned)(b - b_lo);
            continue;
        }

        // the footprint is clipped to the grid, so a neighbour inside it is in the grid: two
        // unsigned compares per neighbour; whatever falls outside the bound takes the cold path
        const int W = hi[0] - lo[0] + 1, H = hi[1] - lo[1] + 1;
        T* o = out + b * gd.G;
        // fp32 data: exact 64-bit fixed-point sums in the LDS tile (FixScale, dpr_device.h): the
        // kernel was bound by ds_add_f64 (2.56 G atomics / 1544 G/s = 1.66 of its 2.11 ms at C4's
        // share); ds_add_u64 retires twice as fast
#ifndef DPR_CO_NO_FOOT_TABLE
        const int fe = fexp[(int)(b - b_lo)];
#else
        const int fe0 = fix_exponent(sizeof(T) == 4 ? fabsf((float)ps.ow) * maxpw : __builtin_inff(),
                                     (uint32_t)kCOChunk, fixed);
        const int fe = fe0 == kFixNone ? kFixOff : fe0;
#endif
        const FixScale fs = fix_scale_from_exponent(fe == kFixOff ? kFixNone : fe);
        auto pose_points = [&](auto fix_tag) {
            constexpr bool FIX = decltype(fix_tag)::value;
#pragma unroll
            for (int k = 0; k < kCOPPT; ++k) {
                int ref0[2];
                T dlo[2];
                const bool ok = ref_and_deltas<T, NI, 2>(pt[k], ps, gd, ref0, dlo);  // (NaN beyond the cloud)
                const T wk = HAS_PW ? ps.ow * w[k] : ps.ow * T(1);  // src/raster.jl:52
                const int lx0 = ref0[0] - lo[0], ly0 = ref0[1] - lo[1];
                // one range test for all four neighbours (the footprint is clipped to the grid; round 3:
                // C4 forward 4.04 -> 4.00 ms; the same change in the gather kernel: 4.69 -> 4.16 ms)
                if (ok && (unsigned)lx0 < (unsigned)(W - 1) && (unsigned)ly0 < (unsigned)(H - 1)) {
                    double* b0 = &acc[ly0 * W + lx0];
                    cell_add<FIX, T>(b0, voxel_weight<T, 2>(dlo, 0, wk), fs);
                    cell_add<FIX, T>(b0 + 1, voxel_weight<T, 2>(dlo, 1, wk), fs);
                    cell_add<FIX, T>(b0 + W, voxel_weight<T, 2>(dlo, 2, wk), fs);
                    cell_add<FIX, T>(b0 + W + 1, voxel_weight<T, 2>(dlo, 3, wk), fs);
                } else if (ok) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        int lx = lx0 + (s & 1), ly = ly0 + (s >> 1);
                        asm volatile("" : "+v"(lx), "+v"(ly));  // keep this path out of the hot one
                        const T v = voxel_weight<T, 2>(dlo, s, wk);
                        if ((unsigned)lx < (unsigned)W && (unsigned)ly < (unsigned)H) {
                            cell_add<FIX, T>(&acc[ly * W + lx], v, fs);
                        } else {
                            const int ix = lx + lo[0], iy = ly + lo[1];
                            if (ix >= 0 && ix < gd.n[0] && iy >= 0 && iy < gd.n[1])
                                atomic_add<T>(o + (size_t)iy * gd.n[0] + ix, v);
                        }
                    }
                }
            }
        };
        if (fs.mul != 0.0) pose_points(std::true_type{});  // (uniform)
        else pose_points(std::false_type{});
        lds_barrier();
        // flush + re-zero: one wave per image row segment, contiguous x across the lanes
        for (int r = wave; r < H; r += kCOWaves) {
            T* orow = o + (size_t)(lo[1] + r) * gd.n[0] + lo[0];
            for (int x = lane; x < W; x += kWave) {
                const double a = acc[r * W + x];
                if (__double_as_longlong(a) != 0) {
                    atomic_add<T>(orow + x, (T)fix_value(a, fs));
                    acc[r * W + x] = 0.0;
                }
            }
        }
        lds_barrier();  // LDS only: the flush's atomics stay in flight
    }
    // Work items of k_co_splat_wide: (chunk, first pose, mask of up to kCOWideGroup poses).  One
    // item per GROUP of poses rather than per pose: the wide kernel loads the chunk (and reduces
    // its bounding box) once per item, and a sparse chunk is wide for nearly every pose; groups
    // of 8 still spread the few sparse chunks over the whole chip (64 poses: 8 items per chunk).
    if (threadIdx.x == 0 && wide_mask) {
        for (int g0 = 0; g0 < nbs; g0 += kCOWideGroup) {
            const unsigned m = (unsigned)((wide_mask >> g0) & ((1ull << kCOWideGroup) - 1ull));
            if (m)
                wide_items[atomicAdd(wide_count, 1u)] =
                    make_uint4(blockIdx.x, (unsigned)(b_lo + g0), m, (unsigned)((b_lo + g0) >> 32));
        }
    }
}

template <typename T, int NI, bool HAS_PW>
__global__ __launch_bounds__(kCOThreads) void k_co_splat_wide(
    GridDesc<2> gd, int64_t P, const T* __restrict__ points, const T* __restrict__ pw,
    const T* __restrict__ rot, const T* __restrict__ trans, const T* __restrict__ ow,
    T* __restrict__ out, const uint32_t* __restrict__ wide_count,
    const uint4* __restrict__ wide_items, int fixed) {
    // one workgroup per CU here (the register budget of the general path): twice the tile
    __shared__ double acc[kCOWideCap];
    __shared__ T sbox[kCOWaves][6];
    __shared__ float smaxw[kCOWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t n_items = *wide_count;  // 0 in the usual case
    if (blockIdx.x >= n_items) return;
    for (int i = threadIdx.x; i < kCOWideCap; i += kCOThreads) acc[i] = 0.0;
    // one (chunk, group of poses) item at a time, so that a few sparse chunks with every pose wide
    // still spread over the whole chip
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint4 item = wide_items[it];
        __syncthreads();  // sbox of the previous item has been read
        T pt[kCOPPT][NI], w[kCOPPT], c[NI], h[NI];
        bool live[kCOPPT];
        co_load_chunk<T, NI>(points, HAS_PW ? pw : nullptr, P, (int64_t)item.x * kCOChunk, pt, w,
                             live, c, h, sbox);
        const float maxpw = co_max_abs_weight<T, HAS_PW>(w, live, smaxw);
        const int64_t b_first = (int64_t)item.y | ((int64_t)item.w << 32);
        for (unsigned pm = item.z; pm; pm &= pm - 1) {  // the wide poses of the group (uniform)
        const int64_t b = b_first + (__ffs((int)pm) - 1);
        const Pose<T, NI, 2> ps = load_pose<T, NI, 2>(rot, trans, ow, b);
        int lo[2], hi[2];
        const int64_t cells = co_footprint<T, NI>(c, h, ps, gd, lo, hi);
        if (cells <= kCOCap) continue;  // never: k_co_splat made the same decision
        // Passes over bands of rows (one pass when the footprint fits the larger tile); a footprint
        // WIDER than the whole tile goes to global memory directly.  The neighbours and their
        // weights are computed once and kept across the passes.
        const int W = hi[0] - lo[0] + 1, H = hi[1] - lo[1] + 1;
        const bool direct = W > kCOWideCap;
        const int band = direct ? H : kCOWideCap / W;
        T* o = out + b * gd.G;
        const FixScale fs = fix_scale(sizeof(T) == 4 ? fabsf((float)ps.ow) * maxpw : __builtin_inff(),
                                      (uint32_t)kCOChunk, fixed);
        int lx0[kCOPPT], ly0[kCOPPT];
        T v[kCOPPT][4];
#pragma unroll
        for (int k = 0; k < kCOPPT; ++k) {
            int ref0[2];
            T dlo[2];
            const bool ok = ref_and_deltas<T, NI, 2>(pt[k], ps, gd, ref0, dlo);  // (NaN beyond the cloud)
            const T wk = HAS_PW ? ps.ow * w[k] : ps.ow * T(1);  // src/raster.jl:52
            lx0[k] = ref0[0] - lo[0];
            ly0[k] = ok ? ref0[1] - lo[1] : (1 << 29);  // no point: outside every band, and the grid
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v[k][s] = voxel_weight<T, 2>(dlo, s, wk);
                // outside the bound (rounding), or no LDS tile at all: straight to the image
                const int lx = lx0[k] + (s & 1), ly = ly0[k] + (s >> 1);
                const bool in_foot = !direct && (unsigned)lx < (unsigned)W && (unsigned)ly < (unsigned)H;
                if (ok && !in_foot) {
                    int ix = lx + lo[0], iy = ly + lo[1];
                    asm volatile("" : "+v"(ix), "+v"(iy));  // keep the address math in here
                    if (ix >= 0 && ix < gd.n[0] && iy >= 0 && iy < gd.n[1])
                        atomic_add<T>(o + (size_t)iy * gd.n[0] + ix, v[k][s]);
                }
            }
        }
        if (direct) continue;  // uniform
        for (int y0 = 0; y0 < H; y0 += band) {
            const int rows = (H - y0 < band) ? H - y0 : band;
#pragma unroll
            for (int k = 0; k < kCOPPT; ++k) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int lx = lx0[k] + (s & 1), ly = ly0[k] + (s >> 1) - y0;
                    if ((unsigned)lx < (unsigned)W && (unsigned)ly < (unsigned)rows) {
                        if (fs.mul != 0.0) cell_add<true, T>(&acc[ly * W + lx], v[k][s], fs);  // (uniform)
                        else cell_add<false, T>(&acc[ly * W + lx], v[k][s], fs);
                    }
                }
            }
            lds_barrier();
            // flush + re-zero: one wave per image row segment, contiguous x across the lanes
            for (int r = wave; r < rows; r += kCOWaves) {
                T* orow = o + (size_t)(lo[1] + y0 + r) * gd.n[0] + lo[0];
                for (int x = lane; x < W; x += kWave) {
                    const double a = acc[r * W + x];
                    if (__double_as_longlong(a) != 0) {
                        atomic_add<T>(orow + x, (T)fix_value(a, fs));
                        acc[r * W + x] = 0.0;
                    }
                }
            }
            lds_barrier();  // LDS only: the flush's atomics stay in flight
        }
        }  // poses of the item
    }
}

// ------------------------------------------------------------------ pullback
// partials[(slice_pose) ...]: layout [NVAL][B][nblk] (f64), NVAL = 2 NI + 2 + 1
template <typename T, int NI, bool HAS_PW>
__global__ __launch_bounds__(kCOThreads) void k_co_gather(
    GridDesc<2> gd, int64_t P, int64_t B, int poses_per_slice, const T* __restrict__ g,
    const T* __restrict__ points, const T* __restrict__ pw, const T* __restrict__ rot,
    const T* __restrict__ trans, const T* __restrict__ ow, T* __restrict__ ds_dpoints,
    T* __restrict__ ds_dpw, double* __restrict__ partials, int accumulate_points, Residual<T> rs,
    SortHeader want, const SortHeader* hdr, const uint32_t* __restrict__ perm) {
    // perm != nullptr: the chunk is a run of the library's sorted copy and ds_dpoints / ds_dpw are
    // the CALLER's arrays -- the epilogue scatters the chunk's gradients through the permutation
    // itself (fire-and-forget stores spread over the whole kernel) instead of leaving them in a
    // sorted buffer for a separate un-sort pass (10 M points: 0.33 ms and 160 MB of workspace).
    constexpr int NVAL = 2 * NI + 2 + 1;  // dR | dt | d out_weight
    if (want.magic && !sort_header_ok(hdr, want)) {
        // REUSE_BINNING without the matching KEEP_BINNING forward: NaN partials and NaN point
        // gradients (k_co_unsort does the latter when it runs), nothing read through stale pointers
        const int64_t b_lo0 = (int64_t)blockIdx.y * poses_per_slice;
        const int64_t b_hi0 = (b_lo0 + poses_per_slice < B) ? b_lo0 + poses_per_slice : B;
        for (int i = threadIdx.x; i < (int)(b_hi0 - b_lo0) * NVAL; i += kCOThreads)
            partials[((size_t)(i % NVAL) * B + (b_lo0 + i / NVAL)) * gridDim.x + blockIdx.x] =
                __builtin_nan("");
        if (perm && blockIdx.y == 0) {  // any order will do: every entry becomes NaN
            const int64_t base0 = (int64_t)blockIdx.x * kCOChunk;
            const int n0 = (int)((P - base0 < kCOChunk) ? P - base0 : kCOChunk);
            const T nan = T(__builtin_nanf(""));
            for (int i = threadIdx.x; i < n0 * NI; i += kCOThreads) ds_dpoints[base0 * NI + i] = nan;
            if (ds_dpw)
                for (int i = threadIdx.x; i < n0; i += kCOThreads) ds_dpw[base0 + i] = nan;
        }
        return;
    }
    // tile: footprint of ds_dout | red: per-thread per-pose sums.  The epilogue reuses both as
    // one buffer to turn the chunk's point gradients into coalesced rows.
    // fp32 data (PIPE): two tiles and two `red` buffers -- the footprint of pose b + 1 is requested
    // into registers before pose b is gathered and written to the other tile after it, so its
    // latency hides behind the arithmetic and one barrier per pose is enough (the kernel runs one
    // workgroup per CU by its 1024 threads x ~100 VGPRs anyway, so the LDS is there).
#ifndef DPR_CO_NO_FOOT_TABLE
    constexpr bool PIPE = sizeof(T) == 4;
#else
    constexpr bool PIPE = false;
#endif
    constexpr int CAP = PIPE ? kCOGatherCap : kCOCap;
    constexpr int NBUF = PIPE ? 2 : 1;
    constexpr int kRed = NVAL * kCOThreads;
    __shared__ T smem[NBUF * (CAP + kRed)];
    T* const tile0 = smem;
    T* const red0 = smem + NBUF * CAP;
    __shared__ T sbox[kCOWaves][6];
    __shared__ double pacc[kCOMaxSlice][NVAL];
    static_assert(NVAL <= kCOWaves, "one wave per per-pose sum");
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int i = threadIdx.x; i < kCOMaxSlice * NVAL; i += kCOThreads) (&pacc[0][0])[i] = 0.0;
    T pt[kCOPPT][NI], w[kCOPPT], c[NI], h[NI];
    bool live[kCOPPT];
    const bool any = co_load_chunk<T, NI>(points, HAS_PW ? pw : nullptr, P,
                                          (int64_t)blockIdx.x * kCOChunk, pt, w, live, c, h, sbox);
    T dp[kCOPPT][NI], dpw[kCOPPT];
#pragma unroll
    for (int k = 0; k < kCOPPT; ++k) {
        dpw[k] = T(0);
#pragma unroll
        for (int j = 0; j < NI; ++j) dp[k][j] = T(0);
    }
    const int64_t b_lo = (int64_t)blockIdx.y * poses_per_slice;
    const int64_t b_hi = (b_lo + poses_per_slice < B) ? b_lo + poses_per_slice : B;
#ifndef DPR_CO_NO_FOOT_TABLE
    __shared__ int foot[kCOMaxSlice][4];
    co_fill_footprints<T, NI>(foot, c, h, gd, rot, trans, b_lo, any ? (int)(b_hi - b_lo) : 0);
#endif
    // stage the footprint of ds_dout (residual mode: scale * (out - target)); no barrier
    auto stage = [&](T* tile, const int (&lo)[2], int W, int H, int64_t b) {
        const T* gb = g + b * gd.G;
        const T* tb = rs.target ? rs.target + b * gd.G : nullptr;
        for (int r = wave; r < H; r += kCOWaves) {
            const size_t off = (size_t)(lo[1] + r) * gd.n[0] + lo[0];
            for (int x = lane; x < W; x += kWave) {
                T v = gb[off + x];
                if (tb) v = rs.scale * (v - tb[off + x]);
                tile[r * W + x] = v;
            }
        }
    };
    // one pose: the thread's kCOPPT points against the staged footprint (or global memory when it
    // does not fit); per-pose sums into vals, point gradients into dp / dpw
    auto gather_pose = [&](const Pose<T, NI, 2>& ps, const int (&lo)[2], const int (&hi)[2], bool fits,
                           const T* tile, int64_t b, T (&vals)[NVAL]) {
        const int W = hi[0] - lo[0] + 1, H = hi[1] - lo[1] + 1;
        const T* gb = g + b * gd.G;
        const T* tb = rs.target ? rs.target + b * gd.G : nullptr;
#pragma unroll
        for (int q = 0; q < NVAL; ++q) vals[q] = T(0);
#pragma unroll
        for (int k = 0; k < kCOPPT; ++k) {
            int ref0[2];
            T dlo[2];
            const bool ok = ref_and_deltas<T, NI, 2>(pt[k], ps, gd, ref0, dlo);  // (NaN beyond the cloud)
            if (!ok) continue;  // no in-range neighbour (or non-finite): empty gradient
            const T pwi = HAS_PW ? w[k] : T(1);
            T dcoord[2] = {T(0), T(0)}, dow_part = T(0), dpw_part = T(0);
            // all four neighbours inside the staged footprint (which is clipped to the grid): two
            // unsigned compares instead of eight signed range tests per neighbour
            T gq[4];
            {
                const int lx0 = ref0[0] - lo[0], ly0 = ref0[1] - lo[1];
                if (fits && (unsigned)lx0 < (unsigned)(W - 1) && (unsigned)ly0 < (unsigned)(H - 1)) {
                    const T* b0 = &tile[ly0 * W + lx0];
                    gq[0] = b0[0];
                    gq[1] = b0[1];
                    gq[2] = b0[W];
                    gq[3] = b0[W + 1];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        int ix = ref0[0] + (s & 1), iy = ref0[1] + (s >> 1);
                        asm volatile("" : "+v"(ix), "+v"(iy));  // keep this path out of the hot one
                        const bool in = ix >= 0 && ix < gd.n[0] && iy >= 0 && iy < gd.n[1];
                        const bool in_tile = fits && ix >= lo[0] && ix <= hi[0] && iy >= lo[1] && iy <= hi[1];
                        T gi = T(0);
                        if (in && in_tile) {
                            gi = tile[(iy - lo[1]) * W + (ix - lo[0])];
                        } else if (in) {
                            const size_t off = (size_t)iy * gd.n[0] + ix;
                            gi = gb[off];
                            if (tb) gi = rs.scale * (gi - tb[off]);
                        }
                        gq[s] = gi;
                    }
                }
            }
            // The bilinear form and its two partial derivatives in FACTORED form (src/raster_pullback.jl:
            // 51-65 sums g_s * sigma(s_n) * prod_{m != n} omega_m(s_m) over the four neighbours; with
            // omega(0) = 1 - d, omega(1) = d that is
            //   W     = g00 + dx (g10 - g00) + dy [(g01 - g00) + dx ((g11 - g10) - (g01 - g00))]
            //   dW/dx = (g10 - g00) + dy ((g11 - g01) - (g10 - g00))
            //   dW/dy = (g01 - g00) + dx ((g11 - g10) - (g01 - g00))
            // ): 14 instead of 36 vector instructions per (point, pose) of this VALU-bound loop, and
            // explicit FMAs (the library is built with -ffp-contract=off).  These sums run over
            // neighbours, points and poses in an order the reference does not share anyway; cell
            // choice and deltas above stay in the reference's operation order.  A dropped neighbour
            // arrives as g = 0, which is what dropping its term means.
            {
                const T a = gq[1] - gq[0], c = gq[2] - gq[0];
                const T b = gq[3] - gq[2], d = gq[3] - gq[1];
                const T dWdx = fma_t(dlo[1], b - a, a);
                const T dWdy = fma_t(dlo[0], d - c, c);
                const T W = fma_t(dlo[1], dWdy, fma_t(dlo[0], a, gq[0]));
                dow_part = W * pwi;
                dpw_part = W * ps.ow;
                const T owpw = ps.ow * pwi;
                dcoord[0] = owpw * dWdx;
                dcoord[1] = owpw * dWdy;
            }
            T scaled[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) scaled[n] = dcoord[n] * (T(gd.n[n]) / T(2));  // :67
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int j = 0; j < NI; ++j) vals[n + j * 2] = fma_t(scaled[n], pt[k][j], vals[n + j * 2]);  // :69
                vals[2 * NI + n] += scaled[n];                                                              // :68
            }
            vals[2 * NI + 2] += dow_part;
#pragma unroll
            for (int j = 0; j < NI; ++j)  // rotation' * scaled (:70)
                dp[k][j] = fma_t(ps.R[1 + j * 2], scaled[1], fma_t(ps.R[0 + j * 2], scaled[0], dp[k][j]));
            dpw[k] += dpw_part;
        }
    };
    // per-pose sums: T within the thread (kCOPPT points), f64 across the block.  Every thread
    // parks its NVAL sums in LDS; after a barrier wave q adds up value q.
    auto reduce_parked = [&](const T* red, int j) {
        if (wave < NVAL) {
#ifndef DPR_CO_RED_F64
            if constexpr (sizeof(T) == 4) {
                // fp32 data: the block's 4096 terms are summed in f32 as a tree (4 per thread, 16 per
                // reducer lane, 64 lanes: what the reference's pairwise `sum` does in the same
                // precision); across blocks in f64 (k_co_reduce).  The f64 version of this step --
                // 16 conversions and f64 adds per lane and six ds_bpermute round trips -- sat on the
                // critical path of every pose (nine of the sixteen waves reduce).
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < kCOThreads / kWave; ++i) sum += red[wave * kCOThreads + i * kWave + lane];
                sum = wave_sum_lane63(sum);
                if (lane == kWave - 1) pacc[j][wave] = (double)sum;
                return;
            }
#endif
            double sum = 0.0;
#pragma unroll
            for (int i = 0; i < kCOThreads / kWave; ++i) sum += (double)red[wave * kCOThreads + i * kWave + lane];
            sum = wave_sum<double>(sum);
            if (lane == 0) pacc[j][wave] = sum;
        }
    };
    if constexpr (PIPE) {
#ifndef DPR_CO_NO_FOOT_TABLE
        // Prefetch: thread t requests cells t, t + 1024, ... of the next footprint (row-major, width
        // Wn) -- all of a tile's loads are in flight at once, every lane takes part, the LDS writes are
        // linear.  (Measured against a wave-per-row mapping, which needs no division per cell: 3.09
        // against 3.41 ms for C4's pullback, and against the same mapping with LDS-DMA loads,
        // global_load_lds_dword: 3.33 ms -- experiments/r04_gather_prefetch_modes.patch.)
        constexpr int kSlots = (CAP + kCOThreads - 1) / kCOThreads;
        const int nbs = any ? (int)(b_hi - b_lo) : 0;
        int lo[2] = {0, 0}, hi[2] = {-1, -1};
        int64_t cells = 0;
        bool staged = false;  // tile[buf] holds the current pose's footprint
        if (nbs > 0) cells = co_read_footprint(foot, 0, lo, hi);
        __syncthreads();  // (pacc zeroed)
        const int n0 = gd.n[0];
        for (int j = 0; j < nbs; ++j) {
            const int64_t b = b_lo + j;
            const int buf = j & 1;
            const bool fits = cells > 0 && cells <= CAP;
            if (fits && !staged) {  // (uniform) the first pose of the block
                stage(tile0 + buf * CAP, lo, hi[0] - lo[0] + 1, hi[1] - lo[1] + 1, b);
                __syncthreads();
            }
            // the next pose's footprint: requested now, parked in the other tile after this pose
            int lo_n[2] = {0, 0}, hi_n[2] = {-1, -1};
            int64_t cells_n = 0;
            if (j + 1 < nbs) cells_n = co_read_footprint(foot, j + 1, lo_n, hi_n);
            const int Wn = hi_n[0] - lo_n[0] + 1;
            const bool pref_n = cells_n > 0 && cells_n <= CAP;
            T pf_g[kSlots], pf_t[kSlots];
            const T* gb_n = g + (b + 1) * gd.G;
            const T* tb_n = rs.target ? rs.target + (b + 1) * gd.G : nullptr;
            T* const tn = tile0 + (buf ^ 1) * CAP;
            if (pref_n) {
                const float inv_w = 1.0f / (float)Wn;
#pragma unroll
                for (int u = 0; u < kSlots; ++u) {
                    const int i = threadIdx.x + u * kCOThreads;
                    pf_g[u] = T(0);
                    pf_t[u] = T(0);
                    if (i < (int)cells_n) {
                        // row of cell i: i / Wn through the reciprocal (i <= 8192: exact after the fix-up)
                        int r = (int)(((float)i + 0.5f) * inv_w);
                        int x = i - r * Wn;
                        if (x < 0) { --r; x += Wn; }
                        if (x >= Wn) { ++r; x -= Wn; }
                        const size_t off = (size_t)(lo_n[1] + r) * n0 + lo_n[0] + x;
                        pf_g[u] = gb_n[off];
                        if (tb_n) pf_t[u] = tb_n[off];
                    }
                }
            }
            T* const red = red0 + buf * kRed;
            if (cells > 0) {  // (uniform; else: no neighbour of the chunk is in the grid, sums stay 0)
                const Pose<T, NI, 2> ps = load_pose<T, NI, 2>(rot, trans, ow, b);
                T vals[NVAL];
                gather_pose(ps, lo, hi, fits, tile0 + buf * CAP, b, vals);
#pragma unroll
                for (int q = 0; q < NVAL; ++q) red[q * kCOThreads + threadIdx.x] = vals[q];
            }
            if (pref_n) {
#pragma unroll
                for (int u = 0; u < kSlots; ++u) {
                    const int i = threadIdx.x + u * kCOThreads;
                    if (i < (int)cells_n) tn[i] = rs.target ? rs.scale * (pf_g[u] - pf_t[u]) : pf_g[u];
                }
            }
            // one barrier per pose: `red[buf]` is parked and the other tile is staged; both are next
            // written two poses on, behind the barrier of the pose in between
            lds_barrier();
            if (cells > 0) reduce_parked(red, j);
            lo[0] = lo_n[0]; lo[1] = lo_n[1]; hi[0] = hi_n[0]; hi[1] = hi_n[1];
            cells = cells_n;
            staged = pref_n;
        }
#endif
    } else {
        for (int64_t b = b_lo; any && b < b_hi; ++b) {
            const Pose<T, NI, 2> ps = load_pose<T, NI, 2>(rot, trans, ow, b);
            int lo[2], hi[2];
#ifndef DPR_CO_NO_FOOT_TABLE
            const int64_t cells = co_read_footprint(foot, (int)(b - b_lo), lo, hi);
#else
            const int64_t cells = co_footprint<T, NI>(c, h, ps, gd, lo, hi);
#endif
            if (cells == 0) continue;  // uniform: no neighbour of the chunk is in the grid (sums stay 0)
            // A footprint that does not fit the LDS tile (sparse tails of the cloud, incoherent
            // input) is gathered from global memory directly (L2-resident image; unlike the
            // forward's atomics these are plain loads).
            const bool fits = cells <= CAP;
            if (fits) {
                stage(tile0, lo, hi[0] - lo[0] + 1, hi[1] - lo[1] + 1, b);
                __syncthreads();
            } else {
                lds_barrier();  // `red` of the previous pose has been read
            }
            T vals[NVAL];
            gather_pose(ps, lo, hi, fits, tile0, b, vals);
            // after the barrier (which also releases the tile for the next pose) wave q adds up
            // value q -- it is done before it arrives at the next barrier, and `red` is next
            // written after that one
#pragma unroll
            for (int q = 0; q < NVAL; ++q) red0[q * kCOThreads + threadIdx.x] = vals[q];
            lds_barrier();
            reduce_parked(red0, (int)(b - b_lo));
        }
    }
    __syncthreads();
    const int nb = (int)(b_hi - b_lo);
    const int64_t nblk = gridDim.x;
    for (int i = threadIdx.x; i < nb * NVAL; i += kCOThreads) {
        const int j = i / NVAL, q = i % NVAL;
        partials[((size_t)q * B + (b_lo + j)) * nblk + blockIdx.x] = pacc[j][q];
    }
    // Point gradients: a thread's points are 64 apart from its neighbour lane's (co_point), so
    // writing them from the registers would touch 64 cache lines per instruction (and, with the
    // poses split over blockIdx.y, run at the one-lane-per-line atomic rate).  Through LDS
    // instead: element i of the chunk's [kCOChunk][NI] block sits at i + i / 64 (the padding
    // keeps the strided writes off one bank), the weights behind it.
    constexpr int kPwBase = kCOChunk * NI + kCOChunk * NI / 64;
    static_assert(kPwBase + kCOChunk + kCOChunk / 64 <= CAP + NVAL * kCOThreads, "epilogue fits");
    if (perm) {  // straight to the caller's order (host: only when the poses are not split over grid.y)
        const int64_t cbase = (int64_t)blockIdx.x * kCOChunk;
#pragma unroll
        for (int k = 0; k < kCOPPT; ++k) {
            const int64_t i = cbase + co_point(lane, wave, k);
            if (i >= P) continue;
            const size_t p = perm[i];
            if (p >= (size_t)P) continue;  // never for a permutation this library wrote
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[p * NI + j] = dp[k][j];
            if (ds_dpw) ds_dpw[p] = dpw[k];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kCOPPT; ++k) {
        const int q = co_point(lane, wave, k);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = q * NI + j;
            smem[i + (i >> 6)] = dp[k][j];
        }
        smem[kPwBase + q + (q >> 6)] = dpw[k];
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kCOChunk;
    const int n_here = (int)((P - base < kCOChunk) ? P - base : kCOChunk);
    for (int i = threadIdx.x; i < n_here * NI; i += kCOThreads) {
        const T v = smem[i + (i >> 6)];
        if (accumulate_points)
            atomic_add<T>(ds_dpoints + base * NI + i, v);
        else
            ds_dpoints[base * NI + i] = v;
    }
    if (!ds_dpw) return;  // (the caller declined ds_dpoint_weight)
    for (int q = threadIdx.x; q < n_here; q += kCOThreads) {
        const T v = smem[kPwBase + q + (q >> 6)];
        if (accumulate_points)
            atomic_add<T>(ds_dpw + base + q, v);
        else
            ds_dpw[base + q] = v;
    }
}

// partials[NVAL][B][nblk] -> ds_drotation | ds_dtranslation | ds_dout_weight.  Block per
// (scalar, pose).
template <typename T, int NI>
__global__ __launch_bounds__(256) void k_co_reduce(const double* __restrict__ partials, int64_t B,
                                                   int64_t b0, int64_t nblk, T* __restrict__ d_rot,
                                                   T* __restrict__ d_trans,
                                                   T* __restrict__ d_ow) {
    __shared__ double wsum[4];
    const int q = blockIdx.x;
    const int64_t b = b0 + blockIdx.y;  // grid.y holds at most 65535 poses per launch
    const double* src = partials + ((size_t)q * B + b) * nblk;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < nblk; i += 256) s += src[i];
    s = wave_sum<double>(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (q < 2 * NI)
            d_rot[b * (2 * NI) + q] = (T)tot;
        else if (q < 2 * NI + 2)
            d_trans[b * 2 + (q - 2 * NI)] = (T)tot;
        else
            d_ow[b] = (T)tot;
    }
}

// sorted gradients back to the caller's order: dst[perm[i]] = src[i]
template <typename T, int NI>
__global__ __launch_bounds__(256) void k_co_unsort(int64_t P, const uint32_t* __restrict__ perm,
                                                   const T* __restrict__ dp_sorted,
                                                   const T* __restrict__ dpw_sorted,
                                                   T* __restrict__ ds_dpoints,
                                                   T* __restrict__ ds_dpw, SortHeader want,
                                                   const SortHeader* hdr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (want.magic && !sort_header_ok(hdr, want)) {
        const T nan = T(__builtin_nanf(""));
#pragma unroll
        for (int j = 0; j < NI; ++j) ds_dpoints[i * NI + j] = nan;
        if (ds_dpw) ds_dpw[i] = nan;
        return;
    }
    const size_t p = perm[i];
    if (p >= (size_t)P) return;  // never for a permutation this library wrote
#pragma unroll
    for (int j = 0; j < NI; ++j) ds_dpoints[p * NI + j] = dp_sorted[i * NI + j];
    if (ds_dpw) ds_dpw[p] = dpw_sorted[i];
}

// ------------------------------------------------------------------ host side
static size_t co_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct COPlan {
    int64_t nblk;
    int slices, poses_per_slice;
    size_t off_hdr, off_pts, off_pw, off_perm, off_grad, off_gradw, off_part, off_sort, total;
};


// One layout for both operations (like DPR_ALGO_TILED): a workspace sized for `raster` also
// serves the pullback of the same problem.
static COPlan co_plan(size_t elem, int op, unsigned flags, int n_in, int64_t P, int64_t B) {
    op = DPR_OP_PULLBACK;
    COPlan pl;
    pl.nblk = (P + kCOChunk - 1) / kCOChunk;
    if (pl.nblk < 1) pl.nblk = 1;
    // poses per block: all of them when the chunks alone fill the chip, else split the poses
    // over grid.y (per-pose sums of at most kCOMaxSlice poses live in LDS)
    int64_t slices = 1;
    if (pl.nblk < 1024 && B > 1) slices = (1024 + pl.nblk - 1) / pl.nblk;
    if (slices > B) slices = B;
    if (slices < 1) slices = 1;  // B == 0 (accepted by check_common): no division by zero
    int64_t pps = (B + slices - 1) / slices;
    if (pps > kCOMaxSlice) pps = kCOMaxSlice;
    if (pps < 1) pps = 1;
    slices = (B + pps - 1) / pps;
    if (slices < 1) slices = 1;
    pl.slices = (int)slices;
    pl.poses_per_slice = (int)pps;
    const bool sort = !(flags & DPR_FLAG_COHERENT_POINTS);
    size_t o = 0;
    pl.off_hdr = o;
    o += co_align(sizeof(SortHeader));
    pl.off_pts = o;
    if (sort) o += co_align((size_t)P * n_in * elem);
    pl.off_pw = o;
    if (sort) o += co_align((size_t)P * elem);
    pl.off_perm = o;
    if (sort) o += co_align((size_t)P * 4);
    pl.off_grad = o;
    if (sort && op == DPR_OP_PULLBACK) o += co_align((size_t)P * n_in * elem);
    pl.off_gradw = o;
    if (sort && op == DPR_OP_PULLBACK) o += co_align((size_t)P * elem);
    pl.off_part = o;
    if (op == DPR_OP_PULLBACK) o += co_align((size_t)(2 * n_in + 3) * (size_t)(B > 0 ? B : 1) * pl.nblk * 8);
    pl.off_sort = o;
    if (sort) {
        const size_t a = sort_workspace_bytes(P), c = coarse_sort_scratch_bytes(elem, P);
        o += co_align(a > c ? a : c);
    }
    pl.total = o > 0 ? o : 256;
    return pl;
}

// The in-call sort of a cloud not known to be coherent.  What the kernels need is compact 4096-point chunks, not
// sorted neighbours (profiles/r05_experiments.md): a counting sort into 4096 Hilbert-numbered cells of the model
// frame (dpr_coarse.h: count, two small scans, one write-combining scatter) instead of keys + radix passes + a
// random gather.
// The cells are 1/16 of the frame wide, a chunk of the cell sort spans a whole cell: its footprints are larger
// than those of a Hilbert-sorted chunk, which costs the kernels ~2-3 us per pose at 10 M points (C4's share, 64 poses:
// k_co_gather 3.06 -> 3.15 ms, k_co_splat + wide 2.55 -> 2.67) against 0.21 ms saved once per call (0.43 -> 0.22).
// Steps of C4's share with the cell sort / the radix sort: 8 poses 1.34 / 1.58 ms, 32 poses 3.33 / 3.49, 64 poses
// 6.09 / 6.13; the 512-pose job 41.6 / 40.4.  The cell sort up to 96 poses, the radix sort (15 key bits, as compact
// as the full key) beyond.
template <typename T>
static int co_sort(hipStream_t st, int n_in, int64_t P, int64_t B, const T* points, const T* pw, T* spts, T* spw,
                   uint32_t* perm, char* scratch) {
    static const int mode = env_knob("DPR_CO_SORT", 0, 0, 2);  // 1: cells, 2: radix (A/B runs)
    const bool radix = mode == 2 || (mode == 0 && B > 96);
    if (!radix) return coarse_sort_with_perm<T>(st, n_in, P, points, pw, spts, spw, perm, scratch);
    return sort_points_impl<T>((void*)st, n_in, P, points, spts, perm, pw, spw, scratch, sort_workspace_bytes(P),
                               nullptr, false);
}

size_t chunkown_workspace_bytes(size_t elem, int op, unsigned flags, int n_in, int64_t P,
                                int64_t B) {
    if (P >= (int64_t)1 << 32) return (size_t)-1;
    return co_plan(elem, op, flags, n_in, P, B).total;
}

#define DPR_HIP(expr)                                                                \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            return fail(DPR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T, int NI>
int raster_chunkown(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                    int64_t B, T* out, const T* points, const T* rot, const T* trans, const T* bg,
                    const T* ow, const T* pw, void* ws_, size_t ws_bytes) {
    if (P >= (int64_t)1 << 32)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: P must be < 2^32");
    const COPlan pl = co_plan(sizeof(T), DPR_OP_RASTER, flags, NI, P, B);
    if (pl.slices > 65535)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: at most %d poses per call",
                    65535 * kCOMaxSlice);
    const bool sort = !(flags & DPR_FLAG_COHERENT_POINTS);
    if (!ws_ || ws_bytes < pl.total)
        return fail(DPR_ERR_WORKSPACE, "DPR_ALGO_CHUNKED raster needs %zu workspace bytes, got %zu",
                    pl.total, ws_ ? ws_bytes : (size_t)0);
    char* ws = (char*)ws_;
    GridDesc<2> gd;
    gd.n[0] = (int)grid[0];
    gd.n[1] = (int)grid[1];
    gd.G = G;
    const T* pts = points;
    const T* pws = pw;
    // list of the (chunk, group of <= 8 poses) items left to k_co_splat_wide: a counter + at most
    // nblk * B entries of 16 bytes, in the (forward-unused) region of the pullback's partial sums
    uint32_t* wide_count = (uint32_t*)(ws + pl.off_part);
    uint4* wide_items = (uint4*)(ws + pl.off_part + 16);
    if (sort && P > 0) {
        T* spts = (T*)(ws + pl.off_pts);
        T* spw = pw ? (T*)(ws + pl.off_pw) : (T*)nullptr;
        if (int rc = co_sort<T>(st, NI, P, B, points, pw, spts, spw, (uint32_t*)(ws + pl.off_perm), ws + pl.off_sort))
            return rc;
        pts = spts;
        pws = spw;
    }
    {
        SortHeader h{};
        h.magic = (sort && (flags & DPR_FLAG_KEEP_BINNING)) ? kSortMagic : 0u;  // else: nothing to reuse
        h.elem = (uint32_t)sizeof(T);
        h.n_in = NI;
        h.has_pw = pw ? 1u : 0u;
        h.P = P;
        h.points = (uint64_t)(uintptr_t)points;
        h.pw = (uint64_t)(uintptr_t)pw;
        hipLaunchKernelGGL(k_co_write_header, dim3(1), dim3(64), 0, st, h,
                           (SortHeader*)(ws + pl.off_hdr), wide_count);
    }
    stage_mark(st);
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {
        const int64_t nb = (B - b0 < 65535) ? B - b0 : 65535;
        const int64_t want = (G + kBlock - 1) / kBlock;
        dim3 gg((unsigned)(want < 4096 ? want : 4096), (unsigned)nb);
        hipLaunchKernelGGL(k_fill_background<T>, gg, dim3(kBlock), 0, st, out + b0 * G, G,
                           bg ? bg + b0 : nullptr);
    }
    stage_mark(st);
    if (P > 0) {
        dim3 gg((unsigned)pl.nblk, (unsigned)pl.slices);
        if (pws) {
            hipLaunchKernelGGL((k_co_splat<T, NI, true>), gg, dim3(kCOThreads), 0, st, gd, P, B,
                               pl.poses_per_slice, pts, pws, rot, trans, ow, out, wide_count,
                               wide_items, co_fixed_point());
            hipLaunchKernelGGL((k_co_splat_wide<T, NI, true>), dim3(kCOWideBlocks),
                               dim3(kCOThreads), 0, st, gd, P, pts, pws, rot, trans, ow, out,
                               wide_count, wide_items, co_fixed_point());
        } else {
            hipLaunchKernelGGL((k_co_splat<T, NI, false>), gg, dim3(kCOThreads), 0, st, gd, P, B,
                               pl.poses_per_slice, pts, pws, rot, trans, ow, out, wide_count,
                               wide_items, co_fixed_point());
            hipLaunchKernelGGL((k_co_splat_wide<T, NI, false>), dim3(kCOWideBlocks),
                               dim3(kCOThreads), 0, st, gd, P, pts, pws, rot, trans, ow, out,
                               wide_count, wide_items, co_fixed_point());
        }
    }
    stage_mark(st);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

template <typename T, int NI>
int pullback_chunkown(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                      int64_t B, const T* g, const T* points, const T* rot, const T* trans,
                      const T* ow, const T* pw, T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow,
                      T* d_pw, void* ws_, size_t ws_bytes, Residual<T> rs) {
    if (P >= (int64_t)1 << 32)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: P must be < 2^32");
    const COPlan pl = co_plan(sizeof(T), DPR_OP_PULLBACK, flags, NI, P, B);
    if (pl.slices > 65535)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: at most %d poses per call",
                    65535 * kCOMaxSlice);
    if (!ws_ || ws_bytes < pl.total)
        return fail(DPR_ERR_WORKSPACE,
                    "DPR_ALGO_CHUNKED pullback needs %zu workspace bytes, got %zu", pl.total,
                    ws_ ? ws_bytes : (size_t)0);
    const bool sort = !(flags & DPR_FLAG_COHERENT_POINTS);
    char* ws = (char*)ws_;
    GridDesc<2> gd;
    gd.n[0] = (int)grid[0];
    gd.n[1] = (int)grid[1];
    gd.G = G;
    const T* pts = points;
    const T* pws = pw;
    T* gp = d_pts;
    T* gw = d_pw;
    // DPR_FLAG_REUSE_BINNING: the sorted copy and the permutation of the preceding
    // KEEP_BINNING raster call are still in the workspace (validated on the device)
    const bool reuse = sort && (flags & DPR_FLAG_REUSE_BINNING);
    SortHeader want{};
    if (reuse) {
        want.magic = kSortMagic;
        want.elem = (uint32_t)sizeof(T);
        want.n_in = NI;
        want.has_pw = pw ? 1u : 0u;
        want.P = P;
        want.points = (uint64_t)(uintptr_t)points;
        want.pw = (uint64_t)(uintptr_t)pw;
    }
    const SortHeader* hdr = (const SortHeader*)(ws + pl.off_hdr);
    if (sort && P > 0) {
        T* spts = (T*)(ws + pl.off_pts);
        T* spw = pw ? (T*)(ws + pl.off_pw) : (T*)nullptr;
        if (!reuse)
            if (int rc = co_sort<T>(st, NI, P, B, points, pw, spts, spw, (uint32_t*)(ws + pl.off_perm),
                                    ws + pl.off_sort))
                return rc;
        pts = spts;
        pws = spw;
        gp = (T*)(ws + pl.off_grad);
        gw = d_pw ? (T*)(ws + pl.off_gradw) : (T*)nullptr;  // (NULL: the caller declined this gradient)
    }
    // one pose slice per chunk (the usual case from ~4 M points on): the gather kernel writes the
    // gradients through the permutation itself, no sorted gradient buffer and no un-sort pass
    const bool fused_unsort = sort && P > 0 && pl.slices == 1;
    const uint32_t* perm = fused_unsort ? (const uint32_t*)(ws + pl.off_perm) : (const uint32_t*)nullptr;
    if (fused_unsort) {
        gp = d_pts;
        gw = d_pw;
    }
    stage_mark(st);
    // ds_dbackground[b] = sum(ds_dout[.., b]) (and the loss of the residual form)
    if (rs.target && rs.loss) DPR_HIP(hipMemsetAsync(rs.loss, 0, sizeof(T) * (size_t)B, st));
    DPR_HIP(hipMemsetAsync(d_bg, 0, sizeof(T) * (size_t)B, st));
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {
        const int64_t nb = (B - b0 < 65535) ? B - b0 : 65535;
        const int64_t want = grid_sum_blocks(G, nb);
        Residual<T> rb = rs;
        if (rb.target) rb.target += b0 * G;
        if (rb.loss) rb.loss += b0;
        hipLaunchKernelGGL(k_grid_sum<T>, dim3((unsigned)want, (unsigned)nb), dim3(kBlock), 0, st,
                           g + b0 * G, G, d_bg + b0, rb);
    }
    stage_mark(st);
    double* partials = (double*)(ws + pl.off_part);
    const int accumulate = pl.slices > 1;
    if (P > 0) {
        if (accumulate) {
            DPR_HIP(hipMemsetAsync(gp, 0, sizeof(T) * (size_t)(P * NI), st));
            if (gw) DPR_HIP(hipMemsetAsync(gw, 0, sizeof(T) * (size_t)P, st));
        }
        dim3 gg((unsigned)pl.nblk, (unsigned)pl.slices);
        if (pws)
            hipLaunchKernelGGL((k_co_gather<T, NI, true>), gg, dim3(kCOThreads), 0, st, gd, P, B,
                               pl.poses_per_slice, g, pts, pws, rot, trans, ow, gp, gw, partials,
                               accumulate, rs, want, hdr, perm);
        else
            hipLaunchKernelGGL((k_co_gather<T, NI, false>), gg, dim3(kCOThreads), 0, st, gd, P, B,
                               pl.poses_per_slice, g, pts, pws, rot, trans, ow, gp, gw, partials,
                               accumulate, rs, want, hdr, perm);
    } else {
        DPR_HIP(hipMemsetAsync(partials, 0, (size_t)(2 * NI + 3) * (size_t)B * pl.nblk * 8, st));
    }
    stage_mark(st);
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {
        const int64_t nb = (B - b0 < 65535) ? B - b0 : 65535;
        hipLaunchKernelGGL((k_co_reduce<T, NI>), dim3(2 * NI + 3, (unsigned)nb), dim3(256), 0, st,
                           (const double*)partials, B, b0, pl.nblk, d_rot, d_trans, d_ow);
    }
    if (sort && P > 0 && !fused_unsort)
        hipLaunchKernelGGL((k_co_unsort<T, NI>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                           P, (const uint32_t*)(ws + pl.off_perm), (const T*)gp, (const T*)gw, d_pts,
                           d_pw, want, hdr);
    stage_mark(st);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

#define DPR_INST_CO(T, NI)                                                                        \
    template int raster_chunkown<T, NI>(hipStream_t, unsigned, const int64_t*, int64_t, int64_t,  \
                                        int64_t, T*, const T*, const T*, const T*, const T*,      \
                                        const T*, const T*, void*, size_t);                       \
    template int pullback_chunkown<T, NI>(hipStream_t, unsigned, const int64_t*, int64_t, int64_t, \
                                          int64_t, const T*, const T*, const T*, const T*,        \
                                          const T*, const T*, T*, T*, T*, T*, T*, T*, void*,      \
                                          size_t, Residual<T>);
DPR_INST_CO(float, 2)
DPR_INST_CO(float, 3)
DPR_INST_CO(double, 2)
DPR_INST_CO(double, 3)
}  // namespace dpr
