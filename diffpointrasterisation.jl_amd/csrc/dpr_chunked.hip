// DPR_ALGO_CHUNKED on 3-D grids, forward over SEVERAL POSES of a cloud that is SPARSE on the grid:
// small owner-computes voxel tiles (32 x 16 x 8 cells, 32 KB of LDS: four workgroups per CU) fed by
// CHUNK LISTS.  The other regimes of the 3-D algorithm are in dpr_owner.hip (large LDS tiles over a
// box hierarchy for clouds that fill the grid; the direct pullback); this kernel family stays for
// the regime where a tile holds a few hundred points and an item's fixed cost decides: 1e5 points x
// 64 poses -> 128^3 runs 0.45 ms here against 1.0 ms for the large tiles and for the tiled path
// (profiles/r03_auto_regret.txt coherent section, profiles/r05_experiments.md).
//
// For spatially coherent point order 64 consecutive points touch only a handful of tiles.  Only
// the chunk ids are binned:
//
//   sets         k_chunk_sets    per (chunk, pose): the exact set of tiles its points touch
//                                (wave-level union, chunk = one wave); per-tile counters
//   lists        k_list_scan     per pose: tile list offsets (+ heaviest-first tile order)
//                k_list_fill     chunk ids into the per-tile lists (<= 16 tiles per chunk)
//   forward      k_chunk_splat   block per (tile, pose): walks its chunk list, reads the
//                                points straight from the caller's array (coalesced), keeps
//                                ONLY the contributions that land in voxels the tile owns in
//                                an LDS tile of f64 accumulators; out = background + acc with
//                                plain stores.  No halo exchange, no global atomics.
//   divert       k_chunk_divert_fwd chunks touching more than 16 tiles (incoherent input) go
//                                through direct global atomics; always correct, slow only for
//                                input that should not use this algorithm.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>

#include "../../include/dpr.h"
#include "dpr_device.h"
#include "dpr_tiled.h"

namespace dpr {

constexpr int kChunk = 64;          // points per chunk = one wavefront
constexpr int kSetMax = 16;         // tiles a chunk may be listed in; more -> diverted
constexpr int kDiverted = 255;
constexpr int kCThreads = 512;      // tile kernels: eight chunks per iteration
constexpr int kCWaves = kCThreads / kWave;
constexpr int kCMaxTiles = 65536;

template <int NO> struct CTile;
template <> struct CTile<3> {
    static constexpr int T[3] = {32, 16, 8};
};
template <> struct CTile<2> {
    static constexpr int T[3] = {32, 32, 1};
};

template <int NO> struct CGeom {
    int nt[NO];
    int NT;
};

template <int NO> static bool make_cgeom(const int64_t* grid, CGeom<NO>* tg) {
    int64_t NT = 1;
    for (int d = 0; d < NO; ++d) {
        tg->nt[d] = (int)((grid[d] + CTile<NO>::T[d] - 1) / CTile<NO>::T[d]);
        NT *= tg->nt[d];
    }
    if (NT > kCMaxTiles) return false;
    tg->NT = (int)NT;
    return true;
}

template <int NO> __host__ __device__ constexpr int ctile_voxels() {
    int v = 1;
    for (int d = 0; d < NO; ++d) v *= CTile<NO>::T[d];
    return v;
}
template <int NO> __host__ __device__ constexpr int ctile_voxels_halo() {
    int v = 1;
    for (int d = 0; d < NO; ++d) v *= CTile<NO>::T[d] + 1;
    return v;
}

template <int NO>
__device__ __forceinline__ void ctile_origin(int tile, const CGeom<NO>& tg, int (&x0)[NO],
                                             int (&tc)[NO]) {
#pragma unroll
    for (int d = 0; d < NO; ++d) {
        tc[d] = tile % tg.nt[d];
        tile /= tg.nt[d];
        x0[d] = tc[d] * CTile<NO>::T[d];
    }
}

// ------------------------------------------------------------------ sets
// Exact set of tiles touched by the in-range neighbours of a chunk's 64 points, built by a
// wave-level union (no LDS, no block barrier).  Lane l keeps set element l.
template <typename T, int NI, int NO>
__global__ __launch_bounds__(256) void k_chunk_sets(GridDesc<NO> gd, CGeom<NO> tg, int64_t P,
                                                    int64_t n_chunks, const T* __restrict__ points,
                                                    const T* __restrict__ rot,
                                                    const T* __restrict__ trans, int64_t b0,
                                                    uint16_t* __restrict__ sets,
                                                    uint8_t* __restrict__ nset,
                                                    uint32_t* __restrict__ tile_count) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t c = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    const int64_t bl = blockIdx.y;
    if (c >= n_chunks) return;  // wave-uniform
    const int64_t p = c * kChunk + lane;
    bool valid = false;
    int base = 0, cross = 0;
    if (p < P) {
        const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, nullptr, b0 + bl);
        T pt[NI];
        load_point<T, NI>(points, p, pt);
        int ref0[NO];
        T dlo[NO];
        valid = ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo);
        if (valid) {
            int stride = 1;
#pragma unroll
            for (int d = 0; d < NO; ++d) {
                const int lo = ref0[d] < 0 ? 0 : ref0[d];  // lowest in-range neighbour
                const int hi = ref0[d] + 1;                // upper neighbour (may be out of range)
                base += (lo / CTile<NO>::T[d]) * stride;
                if (hi < gd.n[d] && hi / CTile<NO>::T[d] != lo / CTile<NO>::T[d]) cross |= 1 << d;
                stride *= tg.nt[d];
            }
        }
    }
    int myset = -1, n = 0;
    bool overflow = false;
#pragma unroll
    for (int m = 0; m < (1 << NO); ++m) {
        int cand = -1;
        if (valid && (m & ~cross) == 0) {
            cand = base;
            int stride = 1;
#pragma unroll
            for (int d = 0; d < NO; ++d) {
                if ((m >> d) & 1) cand += stride;
                stride *= tg.nt[d];
            }
        }
        unsigned long long mask = __ballot(cand >= 0);
        while (mask) {
            const int leader = __shfl(cand, __ffsll((long long)mask) - 1, kWave);
            const bool present = __ballot(lane < n && myset == leader) != 0ull;
            if (!present) {
                if (n < kSetMax) {
                    if (lane == n) myset = leader;
                    ++n;
                } else {
                    overflow = true;
                }
            }
            if (cand == leader) cand = -1;
            mask = __ballot(cand >= 0);
        }
    }
    if (!overflow && lane < n) {
        atomicAdd(&tile_count[bl * tg.NT + myset], 1u);
        sets[(bl * n_chunks + c) * kSetMax + lane] = (uint16_t)myset;
    }
    if (lane == 0) nset[bl * n_chunks + c] = overflow ? (uint8_t)kDiverted : (uint8_t)n;
}

// ------------------------------------------------------------------ lists
// per pose: exclusive scan of tile_count -> list_start[NT+1]; tile_order by decreasing count
__global__ __launch_bounds__(1024) void k_list_scan(const uint32_t* __restrict__ tile_count,
                                                    int NT, uint32_t* __restrict__ list_start,
                                                    uint32_t* __restrict__ tile_order) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t bcount[33], bstart[33];
    const uint32_t* totals = tile_count + (size_t)blockIdx.x * NT;
    uint32_t* start = list_start + (size_t)blockIdx.x * (NT + 1);
    uint32_t* order = tile_order + (size_t)blockIdx.x * NT;
    if (threadIdx.x < 33) bcount[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < NT; i += 1024) {
        const uint32_t c = totals[i];
        atomicAdd(&bcount[c ? 32 - __clz(c) : 0], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int k = 32; k >= 0; --k) {
            bstart[k] = s;
            s += bcount[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NT; i += 1024) {
        const uint32_t c = totals[i];
        order[atomicAdd(&bstart[c ? 32 - __clz(c) : 0], 1u)] = (uint32_t)i;
    }
    const int per = (NT + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    uint32_t s = 0;
    for (int i = i0; i < i0 + per && i < NT; ++i) s += totals[i];
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if ((threadIdx.x & 63) >= o) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wbase += wsum[w];
    uint32_t run = wbase + incl - s;
    for (int i = i0; i < i0 + per && i < NT; ++i) {
        start[i] = run;
        run += totals[i];
    }
    if (threadIdx.x == 1023) start[NT] = wbase + incl;
}

// chunk ids into the tile lists; tile_count is counted back down to zero
__global__ __launch_bounds__(256) void k_list_fill(int NT, int64_t n_chunks,
                                                   const uint16_t* __restrict__ sets,
                                                   const uint8_t* __restrict__ nset,
                                                   uint32_t* __restrict__ tile_count,
                                                   const uint32_t* __restrict__ list_start,
                                                   uint32_t* __restrict__ list, int64_t cap) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t bl = blockIdx.y;
    if (c >= n_chunks) return;
    const int n = nset[bl * n_chunks + c];
    if (n == 0 || n == kDiverted) return;
    uint32_t* tcnt = tile_count + bl * NT;
    const uint32_t* start = list_start + bl * (NT + 1);
    uint32_t* lst = list + bl * cap;
    for (int k = 0; k < n; ++k) {
        const int tile = sets[(bl * n_chunks + c) * kSetMax + k];
        const uint32_t slot = atomicSub(&tcnt[tile], 1u) - 1u;
        lst[start[tile] + slot] = (uint32_t)c;
    }
}

// ------------------------------------------------------------------ forward
template <typename T, int NI, int NO, bool HAS_PW>
__global__ __launch_bounds__(kCThreads) void k_chunk_splat(
    GridDesc<NO> gd, CGeom<NO> tg, int64_t P, const T* __restrict__ points,
    const T* __restrict__ pw, const T* __restrict__ rot, const T* __restrict__ trans,
    const T* __restrict__ ow, const T* __restrict__ bg, int64_t b0,
    const uint32_t* __restrict__ list_start, const uint32_t* __restrict__ tile_order,
    const uint32_t* __restrict__ list, int64_t cap, T* __restrict__ out) {
    constexpr int NV = ctile_voxels<NO>();
    __shared__ double acc[NV];
    // Staging of the 8 x 64 points of one iteration.  Consecutive sorted points hit the same
    // voxels, and same-address LDS atomics of one wave-instruction serialise; reading the
    // staged points back through an odd-stride permutation puts lane neighbours ~a chunk
    // apart.
    __shared__ T stage[NI + 1][kCThreads];
    for (int i = threadIdx.x; i < NV; i += kCThreads) acc[i] = 0.0;
    const int64_t bl = blockIdx.y, b = b0 + bl;
    const int tile = (int)tile_order[bl * tg.NT + blockIdx.x];
    int x0[NO], tc[NO];
    ctile_origin<NO>(tile, tg, x0, tc);
    const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b);
    const uint32_t* start = list_start + bl * (tg.NT + 1);
    const uint32_t l0 = start[tile], l1 = start[tile + 1];
    const uint32_t* lst = list + bl * cap;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int src = (threadIdx.x * 67) & (kCThreads - 1);  // 67 odd: a bijection on 512
    for (uint32_t i0 = l0; i0 < l1; i0 += kCWaves) {
        // stage: wave w loads chunk i0 + w (coalesced); absent points become NaN (rejected)
        {
            const uint32_t i = i0 + wave;
            int64_t p = -1;
            if (i < l1) p = (int64_t)lst[i] * kChunk + lane;
            const bool have = p >= 0 && p < P;
            const int64_t pl = have ? p : 0;
            T pt[NI];
            load_point<T, NI>(points, pl, pt);
            const T w = HAS_PW ? pw[pl] : T(1);
            const T nanv = T(__builtin_nanf(""));
#pragma unroll
            for (int j = 0; j < NI; ++j) stage[j][threadIdx.x] = have ? pt[j] : nanv;
            stage[NI][threadIdx.x] = w;
        }
        __syncthreads();
        {
            T pt[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) pt[j] = stage[j][src];
            const T w = ps.ow * stage[NI][src];  // src/raster.jl:52
            int ref0[NO];
            T dlo[NO];
            if (ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo)) {
                int lb[NO];
                bool touches = true;
#pragma unroll
                for (int d = 0; d < NO; ++d) {
                    lb[d] = ref0[d] - x0[d];
                    touches = touches && lb[d] >= -1 && lb[d] < CTile<NO>::T[d];
                }
                if (touches) {
#pragma unroll
                    for (int s = 0; s < (1 << NO); ++s) {
                        int idx = 0, stride = 1;
                        bool owned = true;
#pragma unroll
                        for (int d = 0; d < NO; ++d) {
                            const int l = lb[d] + ((s >> d) & 1);
                            owned = owned && l >= 0 && l < CTile<NO>::T[d];
                            idx += l * stride;
                            stride *= CTile<NO>::T[d];
                        }
                        // owned cells beyond the grid edge (partial tiles) are never flushed:
                        // the individual drop of src/raster.jl:62
                        if (owned) atomicAdd(&acc[idx], (double)voxel_weight<T, NO>(dlo, s, w));
                    }
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const double bgv = bg ? (double)bg[b] : 0.0;
    T* o = out + b * gd.G;
    for (int i = threadIdx.x; i < NV; i += kCThreads) {
        int rem = i, off = 0, stride = 1;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < NO; ++d) {
            const int l = rem % CTile<NO>::T[d];
            rem /= CTile<NO>::T[d];
            const int gcoord = x0[d] + l;
            ok = ok && gcoord < gd.n[d];
            off += gcoord * stride;
            stride *= gd.n[d];
        }
        if (ok) o[off] = (T)(bgv + acc[i]);
    }
}

// chunks that touch too many tiles: direct global atomics onto the finished grid
template <typename T, int NI, int NO>
__global__ __launch_bounds__(256) void k_chunk_divert_fwd(
    GridDesc<NO> gd, int64_t P, int64_t n_chunks, const T* __restrict__ points,
    const T* __restrict__ pw, const T* __restrict__ rot, const T* __restrict__ trans,
    const T* __restrict__ ow, int64_t b0, const uint8_t* __restrict__ nset,
    T* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    const int64_t bl = blockIdx.y, b = b0 + bl;
    if (c >= n_chunks || nset[bl * n_chunks + c] != kDiverted) return;  // wave-uniform
    const int64_t p = c * kChunk + (threadIdx.x & (kWave - 1));
    if (p >= P) return;
    const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b);
    T pt[NI];
    load_point<T, NI>(points, p, pt);
    int ref0[NO];
    T dlo[NO];
    if (!ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo)) return;
    const T w = ps.ow * (pw ? pw[p] : T(1));
    T* o = out + b * gd.G;
#pragma unroll
    for (int s = 0; s < (1 << NO); ++s) {
        const int off = nbr_offset<NO>(ref0, s, gd);
        if (off >= 0) atomic_add<T>(o + off, voxel_weight<T, NO>(dlo, s, w));
    }
}

// ------------------------------------------------------------------ host side
static size_t calign(size_t x) { return (x + 255) & ~(size_t)255; }

struct CPlan {
    int64_t n_chunks, cap, Bw;  // Bw = poses held in the workspace at once
    size_t off_sets, off_nset, off_count, off_start, off_order, off_list, total;
};

static CPlan make_cplan(int NT, int64_t P, int64_t B) {
    CPlan pl;
    pl.n_chunks = (P + kChunk - 1) / kChunk;
    if (pl.n_chunks < 1) pl.n_chunks = 1;
    pl.cap = pl.n_chunks * kSetMax;
    pl.Bw = B < 1 ? 1 : B;
    // bound the per-call workspace: at most 16 poses of sets/lists at a time
    if (pl.Bw > 16) pl.Bw = 16;
    size_t o = 0;
    pl.off_sets = o;
    o += calign((size_t)pl.Bw * pl.n_chunks * kSetMax * 2);
    pl.off_nset = o;
    o += calign((size_t)pl.Bw * pl.n_chunks);
    pl.off_count = o;
    o += calign((size_t)pl.Bw * NT * 4);
    pl.off_start = o;
    o += calign((size_t)pl.Bw * (NT + 1) * 4);
    pl.off_order = o;
    o += calign((size_t)pl.Bw * NT * 4);
    pl.off_list = o;
    o += calign((size_t)pl.Bw * pl.cap * 4);
    pl.total = o;
    return pl;
}

bool chunked_supported(int n_out, const int64_t* grid) {
    if (n_out == 3) {
        CGeom<3> tg;
        return make_cgeom<3>(grid, &tg);
    }
    CGeom<2> tg;
    return make_cgeom<2>(grid, &tg);
}

size_t chunked_workspace_bytes(int n_out, const int64_t* grid, int64_t P, int64_t B) {
    int NT;
    if (n_out == 3) {
        CGeom<3> tg;
        if (!make_cgeom<3>(grid, &tg)) return (size_t)-1;
        NT = tg.NT;
    } else {
        CGeom<2> tg;
        if (!make_cgeom<2>(grid, &tg)) return (size_t)-1;
        NT = tg.NT;
    }
    return make_cplan(NT, P, B).total;
}

#define DPR_HIP(expr)                                                                \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            return fail(DPR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <int NO> static GridDesc<NO> cgrid_desc(const int64_t* grid, int64_t G) {
    GridDesc<NO> gd;
    for (int d = 0; d < NO; ++d) gd.n[d] = (int)grid[d];
    gd.G = G;
    return gd;
}

// tile sets + lists for poses [b0, b0 + nb)
template <typename T, int NI, int NO>
static int build_lists(hipStream_t st, const GridDesc<NO>& gd, const CGeom<NO>& tg, const CPlan& pl,
                       char* ws, int64_t P, const T* points, const T* rot, const T* trans,
                       int64_t b0, int64_t nb) {
    uint16_t* sets = (uint16_t*)(ws + pl.off_sets);
    uint8_t* nset = (uint8_t*)(ws + pl.off_nset);
    uint32_t* count = (uint32_t*)(ws + pl.off_count);
    uint32_t* start = (uint32_t*)(ws + pl.off_start);
    uint32_t* order = (uint32_t*)(ws + pl.off_order);
    uint32_t* list = (uint32_t*)(ws + pl.off_list);
    DPR_HIP(hipMemsetAsync(count, 0, (size_t)nb * tg.NT * 4, st));
    const unsigned cblocks = (unsigned)((pl.n_chunks + 3) / 4);
    hipLaunchKernelGGL((k_chunk_sets<T, NI, NO>), dim3(cblocks, (unsigned)nb), dim3(256), 0, st, gd,
                       tg, P, pl.n_chunks, points, rot, trans, b0, sets, nset, count);
    stage_mark(st);
    hipLaunchKernelGGL(k_list_scan, dim3((unsigned)nb), dim3(1024), 0, st, count, tg.NT, start,
                       order);
    hipLaunchKernelGGL(k_list_fill, dim3((unsigned)((pl.n_chunks + 255) / 256), (unsigned)nb),
                       dim3(256), 0, st, tg.NT, pl.n_chunks, (const uint16_t*)sets,
                       (const uint8_t*)nset, count, start, list, pl.cap);
    stage_mark(st);
    return DPR_OK;
}

template <typename T, int NI, int NO>
int raster_chunked(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                   int64_t B, T* out, const T* points, const T* rot, const T* trans, const T* bg,
                   const T* ow, const T* pw, void* ws_, size_t ws_bytes) {
    CGeom<NO> tg;
    if (!make_cgeom<NO>(grid, &tg))
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: grid needs more than %d tiles",
                    kCMaxTiles);
    (void)flags;  // (DPR_FLAG_KEEP_BINNING: the 3-D pullback reads nothing a forward could leave)
    const CPlan pl = make_cplan(tg.NT, P, B);
    if (!ws_ || ws_bytes < pl.total)
        return fail(DPR_ERR_WORKSPACE, "DPR_ALGO_CHUNKED raster needs %zu workspace bytes, got %zu",
                    pl.total, ws_ ? ws_bytes : (size_t)0);
    char* ws = (char*)ws_;
    const GridDesc<NO> gd = cgrid_desc<NO>(grid, G);
    for (int64_t b0 = 0; b0 < B; b0 += pl.Bw) {
        const int64_t nb = (B - b0 < pl.Bw) ? B - b0 : pl.Bw;
        if (int rc = build_lists<T, NI, NO>(st, gd, tg, pl, ws, P, points, rot, trans, b0, nb))
            return rc;
        const dim3 tgrid((unsigned)tg.NT, (unsigned)nb);
#define DPR_SPLAT(HAS_PW)                                                                        \
    hipLaunchKernelGGL((k_chunk_splat<T, NI, NO, HAS_PW>), tgrid, dim3(kCThreads), 0, st, gd, tg, \
                       P, points, pw, rot, trans, ow, bg, b0,                                    \
                       (const uint32_t*)(ws + pl.off_start), (const uint32_t*)(ws + pl.off_order), \
                       (const uint32_t*)(ws + pl.off_list), pl.cap, out)
        if (pw) DPR_SPLAT(true);
        else DPR_SPLAT(false);
#undef DPR_SPLAT
        stage_mark(st);
        hipLaunchKernelGGL((k_chunk_divert_fwd<T, NI, NO>),
                           dim3((unsigned)((pl.n_chunks + 3) / 4), (unsigned)nb), dim3(256), 0, st,
                           gd, P, pl.n_chunks, points, pw, rot, trans, ow, b0,
                           (const uint8_t*)(ws + pl.off_nset), out);
        stage_mark(st);
    }
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

template int raster_chunked<float, 3, 3>(hipStream_t, unsigned, const int64_t*, int64_t, int64_t, int64_t, float*,
                                         const float*, const float*, const float*, const float*,
                                         const float*, const float*, void*, size_t);
template int raster_chunked<double, 3, 3>(hipStream_t, unsigned, const int64_t*, int64_t, int64_t, int64_t,
                                          double*, const double*, const double*, const double*,
                                          const double*, const double*, const double*, void*, size_t);
}  // namespace dpr
