// C-ABI entry points of libdpr (declared in include/dpr.h) and host-side dispatch.
// Host checks mirror the reference's @argcheck's (src/raster.jl:14-23,
// ext/DiffPointRasterisationCUDAExt.jl:246-262) but report through status codes.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <initializer_list>
#include <string>

#include "../../include/dpr.h"
#include "dpr_kernels_atomic.h"
#include "dpr_tiled.h"

namespace dpr {

static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define DPR_HIP(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(DPR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// AUTO and DPR_ALGO_CHUNKED on 2-D grids: a small cost model (ms on one MI355X) instead of fixed
// thresholds, because the crossover moves with three things at once -- the poses that share the
// Hilbert sort, the size of the cloud, and how far a 4096-point chunk spreads over the image
// (`spread`, in pixels: chunks whose footprint outgrows the LDS tile take the slower banded path,
// and beyond the double-size tile the direct one).  Round 3 refitted it to the regret table
// (tools/auto_regret.py, profiles/r03_auto_regret.txt: Gaussian, uniform and clustered clouds,
// 1e4-1e7 points, 1-64 poses, 128^2-1024^2): per-pose fixed costs of the chunk-owner kernels were
// 4x too high (1e4 points x 64 poses: 0.07 ms measured, 0.21 modelled), the tiled path's
// per-pose floor grows with the image (1024 tiles at 1024^2: +13 us per pose), and very sparse
// images (spread beyond ~2e4 pixels) cost the chunk-owner forward a second tier.
struct PairCost {
    double fwd, bwd;
};
static double chunk_spread(int n_in, int64_t G, int64_t P) {
    const double frac = P > 4096 ? 4096.0 / (double)P : 1.0;
    return (double)G * (n_in == 3 ? std::cbrt(frac * frac) : frac);
}
static double clamp01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
// The in-call sort of an unsorted cloud: since round 5 a counting sort into 4096 Hilbert-numbered cells
// (dpr_coarse.h: count, two small scans, one write-combining scatter; 1e7 points 0.2 ms, 1e6 points 0.05 ms --
// rounds 2-4: Hilbert keys + radix passes + a random gather, 0.62 / 0.17 ms)
// (coefficients: the measured sort plus what the coarser chunks cost the kernels, chosen against
// profiles/r05_auto_regret.txt with tools/regret_eval.py -- 0.03 + 0.017 pm, the sort alone, sends 1e5-1e6 points x
// 4-16 poses on 512^2-1024^2 to the chunk-owner forward, 1.4-1.8x behind the tiled one)
static double sort_cost(double pm) { return 0.05 + 0.025 * pm; }
static PairCost chunkown_cost(int n_in, int64_t G, int64_t P, int64_t B, bool coherent) {
    const double pm = (double)P * 1e-6, f = chunk_spread(n_in, G, P), gm = (double)G / 1048576.0;
    PairCost c;
    c.fwd = 0.03 + 0.004 * pm +
            (double)B * (0.0004 + 0.0035 * gm * gm +
                         pm * (0.0055 + 0.02 * clamp01((f - 2000) / 10000) +
                               0.06 * clamp01((f - 20000) / 100000)));
    c.bwd = 0.025 + 0.0075 * pm +
            (double)B * (0.0006 + 0.0022 * (pm < 1.0 ? pm : 1.0) + 0.0006 * gm +
                         pm * (0.0046 + 0.004 * clamp01((f - 2000) / 5000)));
    // (round 4: 0.007 / 0.006 until the gather kernel prefetched its footprints -- 1e7 points: 0.044 ms
    // per pose measured, 1e6 points: 0.007-0.009: one wave of blocks, ~2 us of latency per pose)
    if (!coherent) {
        c.fwd += sort_cost(pm);
        c.bwd += sort_cost(pm) + 0.01 + 0.015 * pm;  // + gradients back to the caller's order (large
                                                     // clouds: stored through the permutation by the gather kernel)
    }
    return c;
}
static PairCost other_cost(int n_out, const int64_t* grid, int64_t G, int64_t P, int64_t B) {
    const double pm = (double)P * 1e-6, gm = (double)G / 1048576.0;
    PairCost a, t;
    a.fwd = 0.01 + (double)B * (0.001 + 0.19 * pm);
    a.bwd = 0.05 + (double)B * (0.0003 + 0.05 * pm);
    if (!tiled_supported(n_out, grid) || P >= ((int64_t)1 << 32)) return a;
    // (refitted at the end of round 3: one slice of the cloud per CU and whole sub-chunks made the
    // tiled path 5-25 % faster below ~3e6 points, most of all with pose groups)
    const double sq = std::sqrt(pm);
    const double pf1 = 0.0063 + 0.0062 * sq, pf2 = 0.0025 + 0.0095 * pm;
    const double pb1 = 0.0076 + 0.0095 * std::pow(pm, 0.7), pb2 = 0.0184 * pm - 0.001;
    t.fwd = 0.05 + (double)B * ((pf1 > pf2 ? pf1 : pf2) + gm * (0.012 + 0.012 * sq));
    t.bwd = 0.04 + (double)B * ((pb1 > pb2 ? pb1 : pb2) + gm * (0.014 + 0.0076 * pm));
    PairCost c;
    c.fwd = a.fwd < t.fwd ? a.fwd : t.fwd;
    c.bwd = a.bwd < t.bwd ? a.bwd : t.bwd;
    return c;
}
// op < 0: the raster + pullback pair of a KEEP_BINNING / REUSE_BINNING call pair (one sort)
//
// The margin is on the chunk-owner side (1.2 from 32 poses on: taken even when modelled 20 % behind): the
// model is fitted to clouds that fill the image, where the alternatives are at their best; on a
// clustered cloud the chunk-owner footprints shrink and it wins by 2-3x (1e7 points x 64 poses on
// 1024^2: 3.5 vs 9.6 ms tiled), while nothing makes it lose by more than ~1.2x where the model
// calls a tie.  AUTO cannot see the cloud from the host, so it minimises the worst case.
static bool chunkown_preferred(int op, int n_in, int n_out, const int64_t* grid, int64_t G,
                               int64_t P, int64_t B, bool coherent) {
    if (n_out != 2 || P >= ((int64_t)1 << 32) || P < 1 || B < 1) return false;
    if (B > 65535 * 64) return false;  // the chunk-owner kernels' grid.y (pose slices of <= 64)
    const PairCost c = chunkown_cost(n_in, G, P, B, coherent), o = other_cost(n_out, grid, G, P, B);
    const double margin = B >= 32 ? 1.2 : (B >= 16 ? 1.0 : 0.9);
    if (op == DPR_OP_RASTER) return c.fwd < margin * o.fwd;
    if (op == DPR_OP_PULLBACK) return c.bwd < margin * o.bwd;
    const double pm = (double)P * 1e-6;
    return c.fwd + c.bwd - (coherent ? 0.0 : sort_cost(pm)) < margin * (o.fwd + o.bwd);
}

// DPR_ALGO_CHUNKED on 3-D grids (chunk lists per tile, points read in place, one launch for all
// poses): what AUTO picks for a FORWARD call over several poses of a cloud the caller vouches is
// coherent when the cloud is SPARSE on the grid (at most one point per ten voxels: 1e5 points
// into 128^3, 1e6 into 256^3, the reference README's 1e5 into 1024^3).  There a tile holds few
// points, the per-pose binning of the tiled path is all fixed cost, and the lists win by 1.5-3.5x
// on Gaussian, uniform and clustered clouds alike (profiles/r03_auto_regret.txt, coherent
// section: 1e5 points x 64 poses -> 128^3 0.44 vs 1.53 ms; 1e6 x 64 -> 256^3 2.8 vs 5.8 ms).
// Denser clouds are a tie on clouds that fill the grid and a 3-5x loss on clustered ones (heavy
// tiles are not split on this path), so they stay with the tiled path -- including the
// 50 M -> 512^3 fp64 share of config C5, where the lists measured 13.5 vs 16.7 ms on the Gaussian
// cloud.  Not for a KEEP/REUSE pair (the tiled pair with a shared binning wins the step) and not
// for the pullback (per-pose gather with read-modify-write of the point gradients).
// which forward the 3-D algorithm runs: chunk lists (small tiles) for a sparse cloud over several
// poses, owner-computes large tiles otherwise
static bool chunked3d_lists(const int64_t* grid, int64_t G, int64_t P, int64_t B) {
    static const bool off = env_knob("DPR_CHUNKED3D_NO_LISTS", 0, 0, 1) != 0;  // (experiments: owner tiles everywhere)
    return !off && B >= 4 && P * 10 <= G && chunked_supported(3, grid);
}
static bool chunked3d_preferred(int op, int n_out, const int64_t* grid, int64_t G, int64_t P,
                                int64_t B, unsigned flags) {
    if (op != DPR_OP_RASTER || n_out != 3 || (flags & 3u) || !(flags & DPR_FLAG_COHERENT_POINTS))
        return false;
    if (P >= ((int64_t)1 << 32) || !owner_supported(grid)) return false;
    // DENSE cloud, two poses or more, a grid of >= 1024 owner tiles (256^3: 1216): the owner-computes forward
    // (dpr_owner.hip) reads the points in place for every pose where the tiled path writes and re-reads a record
    // per (point, pose).  Measured on Hilbert-sorted clouds, 2-16 poses, fp32 and fp64
    // (profiles/r05_owner_batch_sweep*.txt): at 0.6-1.8 points per voxel 1.3-1.7x ahead on a Gaussian cloud,
    // 1.7-2.2x on a uniform one, 1.0-1.08x ... 0.73x on a clustered one (0.1 sigma; the host cannot tell) -- the
    // smaller worst case.  Below ~0.4 points per voxel (3e7 -> 512^3: 0.93-0.98 Gaussian, 0.6 clustered) and on
    // grids with fewer tiles than CUs (128^3: 160 tiles) the tiled path stays.
    if (B >= 2 && owner_tiles(grid) >= 1024 && P * 5 >= G * 2 && P <= 2 * G) return true;
    if (B < 4 || P < 30000) return false;
    if (!chunked_supported(n_out, grid)) return false;
    // (fewer than 16 poses: only the very sparse cloud -- on a clustered one the lists lose 2x at
    // one point per 17-21 voxels and 4 poses, where they win 1.3x on a Gaussian or uniform cloud)
    return P * (B >= 16 ? 10 : 25) <= G;
}

static bool pullback3d_sorts(unsigned flags, const int64_t* grid, int64_t P, int64_t B);  // (below)
static bool raster3d_sorts(unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B);
template <typename T>
static int raster_owner_sorted(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                               T* out, const T* points, const T* rot, const T* trans, const T* bg, const T* ow,
                               const T* pw, void* ws_, size_t ws_bytes);

// DPR_ALGO_CHUNKED pullback on 3-D grids: a thread per point in cloud order gathering straight from
// ds_dout (dpr_owner.hip).  On a cloud the caller vouches is coherent a wave's gathers share cache
// lines, nothing is binned, staged or un-permuted: 10 M points -> 256^3 0.136 ms (Gaussian; 0.11
// uniform and clustered clouds) against 0.24-0.28 ms for the tiled pipeline on the same clouds and
// 0.17 ms for its binning-reusing half of a KEEP / REUSE pair (profiles/r05_experiments.md) -- so
// such a pair has nothing to share either.  Batches run pose by pose (the point gradients accumulate
// through memory): ahead of the tiled path and of the direct kernel of DPR_ALGO_ATOMIC from 1e6
// points on up to ~32 poses (1e7 x 16 -> 256^3: 2.9 vs 4.4 / 8.2 ms; 1e6 x 16: 0.85 vs 1.11 / 1.12);
// from 32 poses on the ATOMIC kernel's registers across the poses win (1e6 x 64: 2.9 vs 3.4), and a
// small cloud pays per pose for the grid sum and two launches (3e5 x 16 -> 128^3: 0.45 vs 0.28 tiled).
static bool direct3d_preferred(int op, int n_out, const int64_t* grid, int64_t P, int64_t B,
                               unsigned flags) {
    if (op != DPR_OP_PULLBACK || n_out != 3 || !(flags & DPR_FLAG_COHERENT_POINTS) || grid[0] < 2 ||
        P >= ((int64_t)1 << 32) || !owner_supported(grid))
        return false;
    const int64_t G = grid[0] * grid[1] * grid[2];
    // (32 poses and more, measured with the in-kernel pose loop -- tools/batch_probe.py, Gaussian cloud, 128^3 and
    // 256^3: from 3e6 points on the direct kernel is 1.4-2.3x ahead of both alternatives in fp32 (1e7 x 32 -> 256^3:
    // 4.7 vs 8.8 tiled / 9.8 atomic) and level to 1.1x ahead in fp64; at 1e6 points the ATOMIC kernel keeps a
    // 1.05-1.3x lead on 256^3)
    // On small grids (<= 1024 tiles of the tiled path, e.g. 128^3) the alternative at 32+ poses is the tiled
    // pipeline, not the ATOMIC kernel: the direct kernel is 1.6x ahead of it in fp32 from 1e6 points on, 1.08-1.16x
    // behind the best in fp64.
    // Few poses: per pose the direct kernel costs ~13 ns per 1000 points + 2.1 ns per 1000 cells (its grid-sum
    // slices), the ATOMIC kernel 50 + 0.8 (profiles/r05_auto_regret.txt, coherent section: 1e5 x 16 -> 128^3
    // 0.12 vs 0.23 ms, 1e5 x 16 -> 256^3 0.58 vs 0.35): direct from P > G / 28 on.
    if (B == 1) return P >= 10000;
    if (B < 32) return P >= 10000 && P * 28 >= G;
    return P >= 3000000 || (P >= 1000000 && tiled_tiles(n_out, grid) <= 1024);
}

// (N_in, N_out) with 1 <= N_in, N_out <= 4, in any combination: the reference is generic in both
// (/root/reference/src/raster.jl:5-13, src/util.jl:26-27 -- 2^N_out neighbours, an N_out x N_in matrix per pose).
// The three shapes its tests use -- (2,2), (3,3), (3,2) -- have every algorithm; all the others (embeddings with
// N_out > N_in and 4-D points / grids included) run on the direct kernels (DPR_ALGO_ATOMIC), which are templates
// over both dimensions.
constexpr int kMaxDim = 4;
static bool dims_supported(int n_in, int n_out) {
    return n_out >= 1 && n_out <= kMaxDim && n_in >= 1 && n_in <= kMaxDim;
}
static bool dims_have_all_algos(int n_in, int n_out) {
    return (n_in == 2 && n_out == 2) || (n_in == 3 && n_out == 3) || (n_in == 3 && n_out == 2);
}

static int check_common(int n_in, int n_out, const int64_t* grid, int64_t P, int64_t B,
                        int64_t* G_out) {
    if (!dims_supported(n_in, n_out))
        return fail(DPR_ERR_UNSUPPORTED_DIMS,
                    "unsupported (n_in, n_out) = (%d, %d); supported: 1 <= n_in, n_out <= 4", n_in,
                    n_out);
    if (!grid) return fail(DPR_ERR_INVALID_ARG, "grid is NULL");
    if (P < 0 || B < 0) return fail(DPR_ERR_INVALID_ARG, "negative P (%lld) or B (%lld)",
                                    (long long)P, (long long)B);
    int64_t G = 1;
    for (int d = 0; d < n_out; ++d) {
        if (grid[d] < 1 || grid[d] > 32768)
            return fail(DPR_ERR_INVALID_ARG, "grid[%d] = %lld out of range [1, 32768]", d,
                        (long long)grid[d]);
        G *= grid[d];
    }
    if (G > (int64_t)0x7fffffff)
        return fail(DPR_ERR_INVALID_ARG, "voxels per pose (%lld) exceed 2^31-1", (long long)G);
    *G_out = G;
    return DPR_OK;
}

template <int NO> static GridDesc<NO> make_grid(const int64_t* grid, int64_t G) {
    GridDesc<NO> gd;
    for (int d = 0; d < NO; ++d) gd.n[d] = (int)grid[d];
    gd.G = G;
    return gd;
}

// DPR_ALGO_AUTO.  `flags` (in/out): a KEEP_BINNING / REUSE_BINNING call pair must run the SAME
// algorithm in both calls, so with either flag set the choice is made for the pair, from
// arguments both calls share -- and when the pair's algorithm cannot share (atomic; tiled with
// B > 1), AUTO drops the two flags instead of failing: each call then works on its own.
// An explicit algorithm keeps the strict behaviour (error).
//
// op == DPR_OP_RESIDUAL_PULLBACK: a pullback that forms its sensitivity from (out, target).  The 3-D
// DPR_ALGO_CHUNKED pullback (direct gathers) has no such variant, so AUTO never picks it for this op -- the
// rules that would are skipped and a KEEP / REUSE pair keeps its flags (the tiled pair shares as usual).
static int resolve_algo(int algo, int op, int n_in, int n_out, const int64_t* grid, int64_t P,
                        int64_t B, int64_t G, unsigned* flags) {
    const bool residual = op == DPR_OP_RESIDUAL_PULLBACK;
    if (residual) op = DPR_OP_PULLBACK;
    if (!dims_have_all_algos(n_in, n_out)) {
        // direct kernels only: AUTO drops the sharing flags (nothing to keep), an explicit other
        // algorithm is refused by the dispatch below
        if (algo == DPR_ALGO_AUTO) *flags &= ~3u;
        return algo == DPR_ALGO_AUTO ? DPR_ALGO_ATOMIC : algo;
    }
    if (algo != DPR_ALGO_AUTO) return algo;
    const bool coherent = (*flags & DPR_FLAG_COHERENT_POINTS) != 0;
    // a coherent cloud on a 3-D grid, one pose: the pullback gathers directly and reads nothing a
    // forward could keep -- the pair has nothing to share, each call picks its own best path
    // (a residual pullback cannot take that path: its pair stays a tiled pair)
    if ((*flags & 3u) && !residual && direct3d_preferred(DPR_OP_PULLBACK, n_out, grid, P, B, *flags)) *flags &= ~3u;
    if (*flags & 3u) {
        if (chunkown_preferred(-1, n_in, n_out, grid, G, P, B, coherent)) return DPR_ALGO_CHUNKED;
        // tiled: one pose, or a batch on a grid too large for pose groups (every pose keeps its
        // own binning)
        if ((B == 1 || tiled_batch_share_ok(n_out, grid, P, B)) &&
            tiled_preferred(DPR_OP_RASTER, n_out, grid, P, B, G) &&
            tiled_preferred(DPR_OP_PULLBACK, n_out, grid, P, B, G))
            return DPR_ALGO_TILED;
        *flags &= ~3u;
    }
    // many poses onto a 2-D grid: chunk-owned tiles with the pose loop inside
    if (chunkown_preferred(op, n_in, n_out, grid, G, P, B, coherent)) return DPR_ALGO_CHUNKED;
    // forward over several poses of a coherent cloud on a large 3-D grid: owner-computes tiles
    if (chunked3d_preferred(op, n_out, grid, G, P, B, *flags)) return DPR_ALGO_CHUNKED;
    // pullback of one pose of a coherent cloud on a 3-D grid: direct gathers in cloud order
    if (!residual && direct3d_preferred(op, n_out, grid, P, B, *flags)) return DPR_ALGO_CHUNKED;
    // pullback over many (>= 32) poses of a coherent cloud on a grid without pose groups: the direct
    // kernel (point in registers across the poses, cache-friendly gathers on sorted input) is
    // never more than ~6 % behind the tiled pipeline there and up to 1.9x ahead (clustered cloud,
    // 1e6 points x 64 poses -> 256^3: 2.9 vs 5.4 ms); from 8 poses on while the cloud is small
    // enough for the tiled path's per-pose fixed cost to show (1e6 x 16 -> 256^3: 1.1 vs 1.3 ms)
    if (op == DPR_OP_PULLBACK && coherent && n_out == 3 && tiled_tiles(n_out, grid) > 1024 &&
        (B >= 32 || (B >= 8 && P <= 1500000)))
        return DPR_ALGO_ATOMIC;
    // pullback over 16+ poses of a large 3-D cloud in ANY order: sort inside the call, direct kernels on the
    // sorted copy (pullback3d_sorts above): 1e7 x 16 -> 256^3 3.0 vs 4.4 ms tiled, x 64 10.3 vs 17.3; 3e6 x 16
    // 1.55 vs 1.77
    // (1e6 points: from 64 poses on -- 1.18 vs 1.60 ms on 128^3, 3.7 vs 4.6 on 256^3; fp64 on 128^3 1.08x behind)
    // (1e7 points: from 8 poses on -- 2.3 vs 2.7 ms on 256^3, level on 128^3; fp64 3.6 vs 4.5 / 3.1 vs 3.4)
    if (op == DPR_OP_PULLBACK && n_out == 3 && !coherent && !residual &&
        ((B >= 16 && P >= 3000000) || (B >= 64 && P >= 1000000) || (B >= 8 && P >= 10000000)) &&
        pullback3d_sorts(*flags, grid, P, B) && owner_supported(grid))
        return DPR_ALGO_CHUNKED;
    // forward over 16+ poses of a cloud in any order that is SPARSE on a 3-D grid: sort inside the call, chunk lists
    // on the sorted copy (profiles/r05_unsorted_batches.txt: 3e5-1e6 points -> 256^3 / 384^3, 16-64 poses: 1.1-2.5x
    // ahead of the tiled and ATOMIC kernels on clouds that fill the grid; the clustered cloud at 1e6 -> 256^3 0.82-0.88x)
    if (op == DPR_OP_RASTER && n_out == 3 && !coherent && !(*flags & 3u) && B >= 16 && P >= 200000 &&
        raster3d_sorts(*flags, grid, G, P, B) && chunked3d_lists(grid, G, P, B) && owner_supported(grid))
        return DPR_ALGO_CHUNKED;
    // forward over 32+ poses of a large DENSE 3-D cloud in any order: sort inside the call, owner tiles on the
    // sorted copy -- where the owner tiles win on coherent input (chunked3d_preferred's dense rule).  The tiled
    // path bins such batches in pose groups and is hard to beat in fp32: 1e7 points -> 256^3, as-generated
    // order, 16 / 32 / 64 poses: 3.05 / 5.40 / 10.2 ms against 2.97 / 5.93 / 11.8 (fp64: 4.1 / 7.6 / 14.5 against
    // 5.2 / 10.5 / 20.9) -- profiles/r05_unsorted_batches.txt
    if (op == DPR_OP_RASTER && n_out == 3 && !coherent && !(*flags & 3u) && B >= 32 && P >= 3000000 &&
        raster3d_sorts(*flags, grid, G, P, B) && owner_supported(grid) && owner_tiles(grid) >= 1024 &&
        P * 5 >= G * 2 && P <= 2 * G)
        return DPR_ALGO_CHUNKED;
    return tiled_preferred(op, n_out, grid, P, B, G) ? DPR_ALGO_TILED : DPR_ALGO_ATOMIC;
}

// Optional per-stage timing: the caller arms an array of hipEvent_t; every stage boundary
// records the next one on the launch stream (bench.py reads kernel durations this way).
static thread_local hipEvent_t* g_stage_events = nullptr;
static thread_local int g_stage_cap = 0, g_stage_n = 0;

void stage_mark(hipStream_t st) {
    if (g_stage_events && g_stage_n < g_stage_cap)
        (void)hipEventRecord(g_stage_events[g_stage_n++], st);
}

// ---------------------------------------------------------------- forward
template <typename T, int NI, int NO>
static int raster_atomic(hipStream_t st, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                         T* out, const T* points, const T* rot, const T* trans, const T* bg,
                         const T* ow, const T* pw) {
    const GridDesc<NO> gd = make_grid<NO>(grid, G);
    {
        const int64_t want = (G + kBlock - 1) / kBlock;
        dim3 g((unsigned)(want < 4096 ? want : 4096), (unsigned)(B < 65535 ? B : 65535));
        // B > 65535 poses: loop on the host in slabs of 65535
        for (int64_t b0 = 0; b0 < B; b0 += 65535) {
            const int64_t nb = (B - b0 < 65535) ? B - b0 : 65535;
            g.y = (unsigned)nb;
            hipLaunchKernelGGL(k_fill_background<T>, g, dim3(kBlock), 0, st, out + b0 * G, G,
                               bg ? bg + b0 : nullptr);
        }
    }
    stage_mark(st);
    if (P > 0) {
        dim3 g((unsigned)((P + kBlock - 1) / kBlock), (unsigned)(B < 65535 ? B : 65535));
        hipLaunchKernelGGL((k_fwd_atomic<T, NI, NO>), g, dim3(kBlock), 0, st, gd, P, B, out,
                           points, rot, trans, ow, pw);
    }
    stage_mark(st);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// The kernels move records as 16/32-byte vectors and scalars as T: a misaligned pointer
// would fault on the device, so it is refused here.
template <typename T>
static int check_alignment(const void* ws, std::initializer_list<const void*> data) {
    if (ws && ((uintptr_t)ws & 255))
        return fail(DPR_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    uintptr_t bits = 0;
    for (const void* p : data) bits |= (uintptr_t)p;  // NULL (optional argument) adds nothing
    if (bits & (sizeof(T) - 1))
        return fail(DPR_ERR_INVALID_ARG, "a data pointer is not aligned to its element type");
    return DPR_OK;
}

template <typename T>
static int raster_impl(void* stream, int algo, unsigned flags, int n_in, int n_out, const int64_t* grid, int64_t P,
                       int64_t B, T* out, const T* points, const T* rot, const T* trans,
                       const T* bg, const T* ow, const T* pw, void* ws, size_t ws_bytes) {
    int64_t G = 0;
    if (int rc = check_common(n_in, n_out, grid, P, B, &G)) return rc;
    if (B == 0) return DPR_OK;
    if (!out) return fail(DPR_ERR_INVALID_ARG, "out is NULL");
    if (!rot || !trans) return fail(DPR_ERR_INVALID_ARG, "rotation/translation is NULL");
    if (P > 0 && !points) return fail(DPR_ERR_INVALID_ARG, "points is NULL with P > 0");
    if ((P + kBlock - 1) / kBlock > 0x7fffffffLL)
        return fail(DPR_ERR_INVALID_ARG, "P too large");
    if (int rc = check_alignment<T>(ws, {out, points, rot, trans, bg, ow, pw})) return rc;
    hipStream_t st = (hipStream_t)stream;
    algo = resolve_algo(algo, DPR_OP_RASTER, n_in, n_out, grid, P, B, G, &flags);
    stage_mark(st);
#define DPR_CASE(NI, NO)                                                                       \
    if (n_in == NI && n_out == NO) {                                                           \
        if (algo == DPR_ALGO_ATOMIC && !(flags & 3u))                                           \
            return raster_atomic<T, NI, NO>(st, grid, G, P, B, out, points, rot, trans, bg, ow, \
                                            pw);                                               \
        if (algo == DPR_ALGO_TILED)                                                            \
            return raster_tiled<T, NI, NO>(st, flags, grid, G, P, B, out, points, rot, trans, bg, ow,  \
                                           pw, ws, ws_bytes);                                  \
        if (algo == DPR_ALGO_CHUNKED) {                                                        \
            if constexpr (NO == 2)                                                             \
                return raster_chunkown<T, NI>(st, flags, grid, G, P, B, out, points, rot, trans, \
                                              bg, ow, pw, ws, ws_bytes);                       \
            else if (P > 0 && raster3d_sorts(flags, grid, G, P, B))                           \
                return raster_owner_sorted<T>(st, flags, grid, G, P, B, out, points, rot,      \
                                              trans, bg, ow, pw, ws, ws_bytes);                \
            else if (chunked3d_lists(grid, G, P, B))                                           \
                return raster_chunked<T, NI, NO>(st, flags, grid, G, P, B, out, points, rot,   \
                                                 trans, bg, ow, pw, ws, ws_bytes);             \
            else                                                                               \
                return raster_owner<T>(st, flags, grid, G, P, B, out, points, rot, trans, bg,  \
                                       ow, pw, ws, ws_bytes);                                  \
        }                                                                                      \
    }
    DPR_CASE(2, 2)
    DPR_CASE(3, 3)
    DPR_CASE(3, 2)
#undef DPR_CASE
#define DPR_CASE_DIRECT(NI, NO)                                                                 \
    if (n_in == NI && n_out == NO && algo == DPR_ALGO_ATOMIC && !(flags & 3u))                  \
        return raster_atomic<T, NI, NO>(st, grid, G, P, B, out, points, rot, trans, bg, ow, pw);
    DPR_CASE_DIRECT(1, 1) DPR_CASE_DIRECT(2, 1) DPR_CASE_DIRECT(3, 1) DPR_CASE_DIRECT(4, 1)
    DPR_CASE_DIRECT(1, 2) DPR_CASE_DIRECT(4, 2)
    DPR_CASE_DIRECT(1, 3) DPR_CASE_DIRECT(2, 3) DPR_CASE_DIRECT(4, 3)
    DPR_CASE_DIRECT(1, 4) DPR_CASE_DIRECT(2, 4) DPR_CASE_DIRECT(3, 4) DPR_CASE_DIRECT(4, 4)
#undef DPR_CASE_DIRECT
    if (!dims_have_all_algos(n_in, n_out))
        return fail(DPR_ERR_UNSUPPORTED_ALGO,
                    "(n_in, n_out) = (%d, %d) runs on DPR_ALGO_ATOMIC only (no flags)", n_in, n_out);
    return fail(DPR_ERR_UNSUPPORTED_ALGO, "unknown algorithm %d (flags %u need DPR_ALGO_TILED)", algo,
                flags);
}

// ---------------------------------------------------------------- pullback
template <typename T, int NI, int NO>
static int pullback_atomic(hipStream_t st, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                           const T* g, const T* points, const T* rot, const T* trans, const T* ow,
                           const T* pw, T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow,
                           T* d_pw, Residual<T> rs) {
    const GridDesc<NO> gd = make_grid<NO>(grid, G);
    if (rs.target && rs.loss) DPR_HIP(hipMemsetAsync(rs.loss, 0, sizeof(T) * (size_t)B, st));
    DPR_HIP(hipMemsetAsync(d_rot, 0, sizeof(T) * (size_t)(B * NO * NI), st));
    DPR_HIP(hipMemsetAsync(d_trans, 0, sizeof(T) * (size_t)(B * NO), st));
    DPR_HIP(hipMemsetAsync(d_ow, 0, sizeof(T) * (size_t)B, st));
    DPR_HIP(hipMemsetAsync(d_bg, 0, sizeof(T) * (size_t)B, st));
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {
        const int64_t nb = (B - b0 < 65535) ? B - b0 : 65535;
        const int64_t want = grid_sum_blocks(G, nb);
        dim3 gg((unsigned)want, (unsigned)nb);
        Residual<T> rb = rs;
        if (rb.target) rb.target += b0 * G;
        if (rb.loss) rb.loss += b0;
        hipLaunchKernelGGL(k_grid_sum<T>, gg, dim3(kBlock), 0, st, g + b0 * G, G, d_bg + b0, rb);
    }
    stage_mark(st);
    if (P > 0) {
        const int64_t pblocks = (P + kBlock - 1) / kBlock;
        // enough blocks to fill 256 CUs x 8: split poses into slices when P is small
        int64_t slices = 1;
        if (pblocks < 2048 && B > 1) {
            slices = (2048 + pblocks - 1) / pblocks;
            if (slices > B) slices = B;
            if (slices > 65535) slices = 65535;
        }
        const int poses_per_slice = (int)((B + slices - 1) / slices);
        slices = (B + poses_per_slice - 1) / poses_per_slice;
        const int accumulate = slices > 1;
        if (accumulate) {
            DPR_HIP(hipMemsetAsync(d_pts, 0, sizeof(T) * (size_t)(P * NI), st));
            if (d_pw) DPR_HIP(hipMemsetAsync(d_pw, 0, sizeof(T) * (size_t)P, st));
        }
        dim3 gg((unsigned)pblocks, (unsigned)slices);
        hipLaunchKernelGGL((k_bwd_gather<T, NI, NO>), gg, dim3(kBlock), 0, st, gd, P, B, g, points,
                           rot, trans, ow, pw, d_pts, d_rot, d_trans, d_ow, d_pw, poses_per_slice,
                           accumulate, rs);
    }
    stage_mark(st);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// 3-D DPR_ALGO_CHUNKED pullback of a cloud that is NOT flagged coherent, over a batch large enough to pay for
// a sort: the cloud is Hilbert-sorted into the workspace (dpr_sort_points' kernels, 30-bit keys), the direct
// kernels run on the sorted copy (dpr_owner.hip: a wave's gathers share cache lines; fp32: pose loop inside)
// and the point gradients go back to the caller's order through the inverse permutation.  1e7 points x 16
// poses -> 256^3 in as-generated order: 3.0 ms against 4.4 for the tiled pipeline and 8.2 for the ATOMIC kernel.
static bool pullback3d_sorts(unsigned flags, const int64_t* grid, int64_t P, int64_t B) {
    return !(flags & DPR_FLAG_COHERENT_POINTS) && B >= 8 && P >= 200000 && P < ((int64_t)1 << 32) && grid[0] >= 2;
}
struct Sorted3dPlan {
    size_t off_pts, off_pw, off_inv, off_perm, off_g, off_gw, off_sort, off_own, own_bytes, total;
};
static size_t s3_align(size_t x) { return (x + 255) & ~(size_t)255; }
static Sorted3dPlan sorted3d_plan(size_t elem, const int64_t* grid, int64_t P, int64_t B) {
    Sorted3dPlan pl;
    size_t o = 0;
    pl.off_pts = o;  o += s3_align((size_t)P * 3 * elem);
    pl.off_pw = o;   o += s3_align((size_t)P * elem);
    pl.off_inv = o;  o += s3_align((size_t)P * 4);
    pl.off_perm = o; o += s3_align((size_t)P * 4);
    pl.off_g = o;    o += s3_align((size_t)P * 3 * elem);
    pl.off_gw = o;   o += s3_align((size_t)P * elem);
    pl.off_sort = o; o += s3_align(sort_workspace_bytes(P));
    pl.off_own = o;
    pl.own_bytes = owner_workspace_bytes(DPR_OP_PULLBACK, grid, P, B);
    o += s3_align(pl.own_bytes == (size_t)-1 ? 0 : pl.own_bytes);
    pl.total = o;
    return pl;
}
template <typename T>
__global__ __launch_bounds__(256) void k_unsort3(int64_t P, const uint32_t* __restrict__ inv,
                                                 const T* __restrict__ gs, const T* __restrict__ gws,
                                                 T* __restrict__ d_pts, T* __restrict__ d_pw) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const size_t q = inv[p];
    if (q >= (size_t)P) return;  // (never for a permutation this library wrote)
#pragma unroll
    for (int j = 0; j < 3; ++j) __builtin_nontemporal_store(gs[q * 3 + j], &d_pts[p * 3 + j]);
    if (d_pw) __builtin_nontemporal_store(gws[q], &d_pw[p]);
}
// The forward likewise (owner-computes tiles, or the chunk lists of a sparse cloud, on the sorted copy; nothing to
// bring back).
static bool raster3d_sorts(unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B) {
    (void)grid;
    (void)G;
    return !(flags & DPR_FLAG_COHERENT_POINTS) && B >= 8 && P >= 200000 && P < ((int64_t)1 << 32);
}
struct SortedFwdPlan {
    size_t off_pts, off_pw, off_perm, off_sort, off_own, own_bytes, total;
};
static SortedFwdPlan sorted_fwd_plan(size_t elem, const int64_t* grid, int64_t P, int64_t B) {
    SortedFwdPlan pl;
    size_t o = 0;
    pl.off_pts = o;  o += s3_align((size_t)P * 3 * elem);
    pl.off_pw = o;   o += s3_align((size_t)P * elem);
    pl.off_perm = o; o += s3_align((size_t)P * 4);
    pl.off_sort = o; o += s3_align(sort_workspace_bytes(P));
    pl.off_own = o;
    int64_t G = grid[0] * grid[1] * grid[2];
    pl.own_bytes = chunked3d_lists(grid, G, P, B) ? chunked_workspace_bytes(3, grid, P, B)
                                                  : owner_workspace_bytes(DPR_OP_RASTER, grid, P, B);
    o += s3_align(pl.own_bytes == (size_t)-1 ? 0 : pl.own_bytes);
    pl.total = o;
    return pl;
}
template <typename T>
static int raster_owner_sorted(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P, int64_t B,
                               T* out, const T* points, const T* rot, const T* trans, const T* bg, const T* ow,
                               const T* pw, void* ws_, size_t ws_bytes) {
    const SortedFwdPlan pl = sorted_fwd_plan(sizeof(T), grid, P, B);
    if (pl.own_bytes == (size_t)-1)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: grid needs too many tiles or P >= 2^32");
    if (!ws_ || ws_bytes < pl.total)
        return fail(DPR_ERR_WORKSPACE, "DPR_ALGO_CHUNKED forward (sorting inside the call) needs %zu workspace bytes, got %zu",
                    pl.total, ws_ ? ws_bytes : (size_t)0);
    char* ws = (char*)ws_;
    T* spts = (T*)(ws + pl.off_pts);
    T* spw = pw ? (T*)(ws + pl.off_pw) : (T*)nullptr;
    if (int rc = sort_points_impl<T>((void*)st, 3, P, points, spts, (uint32_t*)(ws + pl.off_perm), pw, spw,
                                     ws + pl.off_sort, sort_workspace_bytes(P), nullptr, true))
        return rc;
    stage_mark(st);
    const unsigned f = (flags | DPR_FLAG_COHERENT_POINTS) & ~3u;
    if (chunked3d_lists(grid, G, P, B))
        return raster_chunked<T, 3, 3>(st, f, grid, G, P, B, out, spts, rot, trans, bg, ow, spw, ws + pl.off_own,
                                       pl.own_bytes);
    return raster_owner<T>(st, f, grid, G, P, B, out, spts, rot, trans, bg, ow, spw, ws + pl.off_own, pl.own_bytes);
}

template <typename T>
static int pullback_owner_sorted(hipStream_t st, unsigned flags, const int64_t* grid, int64_t G, int64_t P,
                                 int64_t B, const T* g, const T* points, const T* rot, const T* trans,
                                 const T* ow, const T* pw, T* d_pts, T* d_rot, T* d_trans, T* d_bg, T* d_ow,
                                 T* d_pw, void* ws_, size_t ws_bytes) {
    const Sorted3dPlan pl = sorted3d_plan(sizeof(T), grid, P, B);
    if (pl.own_bytes == (size_t)-1)
        return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: grid needs too many tiles or P >= 2^32");
    if (!ws_ || ws_bytes < pl.total)
        return fail(DPR_ERR_WORKSPACE, "DPR_ALGO_CHUNKED pullback (sorting inside the call) needs %zu workspace bytes, got %zu",
                    pl.total, ws_ ? ws_bytes : (size_t)0);
    char* ws = (char*)ws_;
    T* spts = (T*)(ws + pl.off_pts);
    T* spw = pw ? (T*)(ws + pl.off_pw) : (T*)nullptr;
    if (int rc = sort_points_impl<T>((void*)st, 3, P, points, spts, (uint32_t*)(ws + pl.off_perm), pw, spw,
                                     ws + pl.off_sort, sort_workspace_bytes(P), (uint32_t*)(ws + pl.off_inv), true))
        return rc;
    stage_mark(st);
    T* gs = (T*)(ws + pl.off_g);
    T* gws = d_pw ? (T*)(ws + pl.off_gw) : (T*)nullptr;
    if (int rc = pullback_owner<T>(st, flags | DPR_FLAG_COHERENT_POINTS, grid, G, P, B, g, spts, rot, trans, ow, spw,
                                   gs, d_rot, d_trans, d_bg, d_ow, gws, ws + pl.off_own, pl.own_bytes))
        return rc;
    hipLaunchKernelGGL((k_unsort3<T>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, P,
                       (const uint32_t*)(ws + pl.off_inv), (const T*)gs, (const T*)gws, d_pts, d_pw);
    stage_mark(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DPR_ERR_HIP, "k_unsort3: %s", hipGetErrorString(e));
    return DPR_OK;
}

template <typename T>
static int pullback_impl(void* stream, int algo, unsigned flags, int n_in, int n_out, const int64_t* grid,
                         int64_t P, int64_t B, const T* g, const T* points, const T* rot,
                         const T* trans, const T* ow, const T* pw, T* d_pts, T* d_rot, T* d_trans,
                         T* d_bg, T* d_ow, T* d_pw, void* ws, size_t ws_bytes,
                         Residual<T> rs = Residual<T>{nullptr, T(0), nullptr}) {
    int64_t G = 0;
    if (int rc = check_common(n_in, n_out, grid, P, B, &G)) return rc;
    hipStream_t st = (hipStream_t)stream;
    // DPR_FLAG_NO_POINT_WEIGHT_GRAD: the caller does not want ds_dpoint_weight (the reference's
    // rrule drops that tangent whenever point_weight was defaulted,
    // ext/DiffPointRasterisationChainRulesCoreExt.jl:23,70): nothing is written through the pointer
    if (flags & DPR_FLAG_NO_POINT_WEIGHT_GRAD) d_pw = nullptr;
    if (P > 0 && (!d_pts || (!d_pw && !(flags & DPR_FLAG_NO_POINT_WEIGHT_GRAD))))
        return fail(DPR_ERR_INVALID_ARG, "ds_dpoints/ds_dpoint_weight is NULL with P > 0");
    if (P > 0 && !points) return fail(DPR_ERR_INVALID_ARG, "points is NULL with P > 0");
    if (B == 0) {
        // no poses: point gradients are empty sums
        if (P > 0) {
            DPR_HIP(hipMemsetAsync(d_pts, 0, sizeof(T) * (size_t)(P * n_in), st));
            if (d_pw) DPR_HIP(hipMemsetAsync(d_pw, 0, sizeof(T) * (size_t)P, st));
        }
        return DPR_OK;
    }
    if (!g) return fail(DPR_ERR_INVALID_ARG, rs.target ? "out is NULL" : "ds_dout is NULL");
    if (!rot || !trans) return fail(DPR_ERR_INVALID_ARG, "rotation/translation is NULL");
    if (!d_rot || !d_trans || !d_bg || !d_ow)
        return fail(DPR_ERR_INVALID_ARG, "a per-pose output pointer is NULL");
    if (int rc = check_alignment<T>(ws, {g, points, rot, trans, ow, pw, d_pts, d_rot, d_trans, d_bg,
                                         d_ow, d_pw, rs.target, rs.loss}))
        return rc;
    algo = resolve_algo(algo, rs.target ? DPR_OP_RESIDUAL_PULLBACK : DPR_OP_PULLBACK, n_in, n_out, grid, P, B, G,
                        &flags);
    stage_mark(st);
#define DPR_CASE(NI, NO)                                                                         \
    if (n_in == NI && n_out == NO) {                                                             \
        if (algo == DPR_ALGO_ATOMIC && !(flags & 3u))                                             \
            return pullback_atomic<T, NI, NO>(st, grid, G, P, B, g, points, rot, trans, ow, pw,   \
                                              d_pts, d_rot, d_trans, d_bg, d_ow, d_pw, rs);      \
        if (algo == DPR_ALGO_TILED)                                                              \
            return pullback_tiled<T, NI, NO>(st, flags, grid, G, P, B, g, points, rot, trans, ow, pw,    \
                                             d_pts, d_rot, d_trans, d_bg, d_ow, d_pw, ws,        \
                                             ws_bytes, rs);                                      \
        if (algo == DPR_ALGO_CHUNKED) {                                                          \
            if constexpr (NO == 2) {                                                             \
                return pullback_chunkown<T, NI>(st, flags, grid, G, P, B, g, points, rot, trans, \
                                                ow, pw, d_pts, d_rot, d_trans, d_bg, d_ow, d_pw, \
                                                ws, ws_bytes, rs);                               \
            } else {                                                                             \
                if (rs.target)                                                                   \
                    return fail(DPR_ERR_UNSUPPORTED_ALGO,                                        \
                                "the residual pullback has no 3-D DPR_ALGO_CHUNKED variant");    \
                if (grid[0] < 2) /* (its x-pair gathers need two cells per row) */               \
                    return pullback_atomic<T, NI, NO>(st, grid, G, P, B, g, points, rot, trans,  \
                                                      ow, pw, d_pts, d_rot, d_trans, d_bg, d_ow, \
                                                      d_pw, rs);                                 \
                if (P > 0 && pullback3d_sorts(flags, grid, P, B))                               \
                    return pullback_owner_sorted<T>(st, flags, grid, G, P, B, g, points, rot,    \
                                                    trans, ow, pw, d_pts, d_rot, d_trans, d_bg,  \
                                                    d_ow, d_pw, ws, ws_bytes);                   \
                return pullback_owner<T>(st, flags, grid, G, P, B, g, points, rot, trans, ow,    \
                                         pw, d_pts, d_rot, d_trans, d_bg, d_ow, d_pw, ws,        \
                                         ws_bytes);                                              \
            }                                                                                    \
        }                                                                                        \
    }
    DPR_CASE(2, 2)
    DPR_CASE(3, 3)
    DPR_CASE(3, 2)
#undef DPR_CASE
#define DPR_CASE_DIRECT(NI, NO)                                                                   \
    if (n_in == NI && n_out == NO && algo == DPR_ALGO_ATOMIC && !(flags & 3u))                    \
        return pullback_atomic<T, NI, NO>(st, grid, G, P, B, g, points, rot, trans, ow, pw, d_pts, \
                                          d_rot, d_trans, d_bg, d_ow, d_pw, rs);
    DPR_CASE_DIRECT(1, 1) DPR_CASE_DIRECT(2, 1) DPR_CASE_DIRECT(3, 1) DPR_CASE_DIRECT(4, 1)
    DPR_CASE_DIRECT(1, 2) DPR_CASE_DIRECT(4, 2)
    DPR_CASE_DIRECT(1, 3) DPR_CASE_DIRECT(2, 3) DPR_CASE_DIRECT(4, 3)
    DPR_CASE_DIRECT(1, 4) DPR_CASE_DIRECT(2, 4) DPR_CASE_DIRECT(3, 4) DPR_CASE_DIRECT(4, 4)
#undef DPR_CASE_DIRECT
    if (!dims_have_all_algos(n_in, n_out))
        return fail(DPR_ERR_UNSUPPORTED_ALGO,
                    "(n_in, n_out) = (%d, %d) runs on DPR_ALGO_ATOMIC only (no flags)", n_in, n_out);
    return fail(DPR_ERR_UNSUPPORTED_ALGO, "unknown algorithm %d", algo);
}

template <typename T>
static size_t workspace_impl(int op, int algo, unsigned flags, int n_in, int n_out,
                             const int64_t* grid, int64_t P, int64_t B) {
    int64_t G = 0;
    if (check_common(n_in, n_out, grid, P, B, &G)) return (size_t)-1;
    if (op != DPR_OP_RASTER && op != DPR_OP_PULLBACK && op != DPR_OP_RESIDUAL_PULLBACK) {
        fail(DPR_ERR_INVALID_ARG, "unknown op %d", op);
        return (size_t)-1;
    }
    algo = resolve_algo(algo, op, n_in, n_out, grid, P, B, G, &flags);
    if (op == DPR_OP_RESIDUAL_PULLBACK) {
        if (algo == DPR_ALGO_CHUNKED && n_out == 3) {
            fail(DPR_ERR_UNSUPPORTED_ALGO, "the residual pullback has no 3-D DPR_ALGO_CHUNKED variant");
            return (size_t)-1;
        }
        op = DPR_OP_PULLBACK;  // (same buffers as the plain pullback of the algorithm chosen for it)
    }
    if (algo == DPR_ALGO_ATOMIC) return 0;
    if (!dims_have_all_algos(n_in, n_out)) {
        fail(DPR_ERR_UNSUPPORTED_ALGO, "(n_in, n_out) = (%d, %d) runs on DPR_ALGO_ATOMIC only", n_in,
             n_out);
        return (size_t)-1;
    }
    if (algo == DPR_ALGO_TILED) {
        const size_t n = tiled_workspace_bytes(sizeof(T), op, flags, n_in, n_out, grid, P, B);
        if (n == (size_t)-1)
            fail(DPR_ERR_UNSUPPORTED_ALGO,
                 "DPR_ALGO_TILED: grid needs too many tiles or P >= 2^32 (the call would be refused)");
        return n;
    }
    if (algo == DPR_ALGO_CHUNKED) {
        if (n_out == 2) {
            const size_t n = chunkown_workspace_bytes(sizeof(T), op, flags, n_in, P, B);
            if (n == (size_t)-1) fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: P must be < 2^32");
            return n;
        }
        size_t n = owner_workspace_bytes(op, grid, P, B);
        if (op == DPR_OP_RASTER && chunked3d_lists(grid, G, P, B)) n = chunked_workspace_bytes(n_out, grid, P, B);
        if (op == DPR_OP_PULLBACK && n != (size_t)-1 && P > 0 && pullback3d_sorts(flags, grid, P, B))
            n = sorted3d_plan(sizeof(T), grid, P, B).total;
        if (op == DPR_OP_RASTER && n != (size_t)-1 && P > 0 && raster3d_sorts(flags, grid, G, P, B))
            n = sorted_fwd_plan(sizeof(T), grid, P, B).total;  // (owner tiles or chunk lists behind the sort)
        if (n == (size_t)-1)
            fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_CHUNKED: grid needs too many tiles or P >= 2^32");
        return n;
    }
    fail(DPR_ERR_UNSUPPORTED_ALGO, "unknown algorithm %d", algo);
    return (size_t)-1;
}

}  // namespace dpr

extern "C" {

int dpr_version(void) { return DPR_VERSION; }

const char* dpr_last_error(void) { return dpr::g_last_error.c_str(); }

int dpr_resolve_algo_ex(int op, unsigned flags, int n_in, int n_out, const int64_t* grid,
                        int64_t P, int64_t B) {
    int64_t G = 0;
    if (int rc = dpr::check_common(n_in, n_out, grid, P, B, &G)) return rc;
    if (op != DPR_OP_RASTER && op != DPR_OP_PULLBACK && op != DPR_OP_RESIDUAL_PULLBACK)
        return dpr::fail(DPR_ERR_INVALID_ARG, "unknown op %d", op);
    return dpr::resolve_algo(DPR_ALGO_AUTO, op, n_in, n_out, grid, P, B, G, &flags);
}

int dpr_resolve_flags_ex(int op, unsigned flags, int n_in, int n_out, const int64_t* grid,
                         int64_t P, int64_t B) {
    int64_t G = 0;
    if (int rc = dpr::check_common(n_in, n_out, grid, P, B, &G)) return rc;
    if (op != DPR_OP_RASTER && op != DPR_OP_PULLBACK && op != DPR_OP_RESIDUAL_PULLBACK)
        return dpr::fail(DPR_ERR_INVALID_ARG, "unknown op %d", op);
    (void)dpr::resolve_algo(DPR_ALGO_AUTO, op, n_in, n_out, grid, P, B, G, &flags);
    return (int)(flags & 0xffffu);
}

int dpr_resolve_algo(int op, int n_in, int n_out, const int64_t* grid, int64_t P, int64_t B) {
    return dpr_resolve_algo_ex(op, 0u, n_in, n_out, grid, P, B);
}

int dpr_stage_timing_begin(void** events, int capacity) {
    if (!events || capacity < 1) return dpr::fail(DPR_ERR_INVALID_ARG, "stage timing: no events");
    dpr::g_stage_events = (hipEvent_t*)events;
    dpr::g_stage_cap = capacity;
    dpr::g_stage_n = 0;
    return DPR_OK;
}

int dpr_stage_timing_end(void) {
    const int n = dpr::g_stage_n;
    dpr::g_stage_events = nullptr;
    dpr::g_stage_cap = dpr::g_stage_n = 0;
    return n;
}

size_t dpr_workspace_bytes_f32(int op, int algo, int n_in, int n_out, const int64_t* grid,
                               int64_t P, int64_t B) {
    return dpr::workspace_impl<float>(op, algo, 0u, n_in, n_out, grid, P, B);
}
size_t dpr_workspace_bytes_f64(int op, int algo, int n_in, int n_out, const int64_t* grid,
                               int64_t P, int64_t B) {
    return dpr::workspace_impl<double>(op, algo, 0u, n_in, n_out, grid, P, B);
}
size_t dpr_workspace_bytes_ex_f32(int op, int algo, unsigned flags, int n_in, int n_out,
                                  const int64_t* grid, int64_t P, int64_t B) {
    return dpr::workspace_impl<float>(op, algo, flags, n_in, n_out, grid, P, B);
}
size_t dpr_workspace_bytes_ex_f64(int op, int algo, unsigned flags, int n_in, int n_out,
                                  const int64_t* grid, int64_t P, int64_t B) {
    return dpr::workspace_impl<double>(op, algo, flags, n_in, n_out, grid, P, B);
}

#define DPR_DEFINE(SUF, T)                                                                        \
    int dpr_raster_ex_##SUF(void* stream, int algo, unsigned flags, int n_in, int n_out,         \
                            const int64_t* grid,                                                  \
                            int64_t P, int64_t B, T* out, const T* points, const T* rotation,     \
                            const T* translation, const T* background, const T* out_weight,       \
                            const T* point_weight, void* workspace, size_t workspace_bytes) {     \
        return dpr::raster_impl<T>(stream, algo, flags, n_in, n_out, grid, P, B, out, points,     \
                                   rotation,                                                      \
                                   translation, background, out_weight, point_weight, workspace,  \
                                   workspace_bytes);                                              \
    }                                                                                             \
    int dpr_raster_##SUF(void* stream, int n_in, int n_out, const int64_t* grid, int64_t P,       \
                         int64_t B, T* out, const T* points, const T* rotation,                   \
                         const T* translation, const T* background, const T* out_weight,          \
                         const T* point_weight, void* workspace, size_t workspace_bytes) {        \
        return dpr::raster_impl<T>(stream, DPR_ALGO_AUTO, 0u, n_in, n_out, grid, P, B, out,      \
                                   points,                                                        \
                                   rotation, translation, background, out_weight, point_weight,   \
                                   workspace, workspace_bytes);                                   \
    }                                                                                             \
    int dpr_raster_pullback_ex_##SUF(                                                             \
        void* stream, int algo, unsigned flags, int n_in, int n_out, const int64_t* grid,         \
        int64_t P, int64_t B,                                                                     \
        const T* ds_dout, const T* points, const T* rotation, const T* translation,               \
        const T* out_weight, const T* point_weight, T* ds_dpoints, T* ds_drotation,               \
        T* ds_dtranslation, T* ds_dbackground, T* ds_dout_weight, T* ds_dpoint_weight,            \
        void* workspace, size_t workspace_bytes) {                                                \
        return dpr::pullback_impl<T>(stream, algo, flags, n_in, n_out, grid, P, B, ds_dout,       \
                                     points,                                                      \
                                     rotation, translation, out_weight, point_weight, ds_dpoints, \
                                     ds_drotation, ds_dtranslation, ds_dbackground,               \
                                     ds_dout_weight, ds_dpoint_weight, workspace,                 \
                                     workspace_bytes);                                            \
    }                                                                                             \
    int dpr_raster_pullback_##SUF(                                                                \
        void* stream, int n_in, int n_out, const int64_t* grid, int64_t P, int64_t B,             \
        const T* ds_dout, const T* points, const T* rotation, const T* translation,               \
        const T* out_weight, const T* point_weight, T* ds_dpoints, T* ds_drotation,               \
        T* ds_dtranslation, T* ds_dbackground, T* ds_dout_weight, T* ds_dpoint_weight,            \
        void* workspace, size_t workspace_bytes) {                                                \
        return dpr::pullback_impl<T>(stream, DPR_ALGO_AUTO, 0u, n_in, n_out, grid, P, B,         \
                                     ds_dout,                                                     \
                                     points, rotation, translation, out_weight, point_weight,     \
                                     ds_dpoints, ds_drotation, ds_dtranslation, ds_dbackground,   \
                                     ds_dout_weight, ds_dpoint_weight, workspace,                 \
                                     workspace_bytes);                                            \
    }

DPR_DEFINE(f32, float)
DPR_DEFINE(f64, double)
#undef DPR_DEFINE

#define DPR_DEFINE_RESIDUAL(SUF, T)                                                               \
    int dpr_raster_residual_pullback_ex_##SUF(                                                    \
        void* stream, int algo, unsigned flags, int n_in, int n_out, const int64_t* grid,         \
        int64_t P, int64_t B, const T* out, const T* target, double residual_scale,               \
        const T* points, const T* rotation, const T* translation, const T* out_weight,            \
        const T* point_weight, T* loss, T* ds_dpoints, T* ds_drotation, T* ds_dtranslation,       \
        T* ds_dbackground, T* ds_dout_weight, T* ds_dpoint_weight, void* workspace,               \
        size_t workspace_bytes) {                                                                 \
        if (!target && B > 0)                                                                     \
            return dpr::fail(DPR_ERR_INVALID_ARG, "residual pullback: target is NULL");           \
        return dpr::pullback_impl<T>(stream, algo, flags, n_in, n_out, grid, P, B, out, points,   \
                                     rotation, translation, out_weight, point_weight, ds_dpoints, \
                                     ds_drotation, ds_dtranslation, ds_dbackground,               \
                                     ds_dout_weight, ds_dpoint_weight, workspace,                 \
                                     workspace_bytes,                                             \
                                     dpr::Residual<T>{target, (T)residual_scale, loss});          \
    }                                                                                             \
    int dpr_raster_residual_pullback_##SUF(                                                       \
        void* stream, int n_in, int n_out, const int64_t* grid, int64_t P, int64_t B,             \
        const T* out, const T* target, double residual_scale, const T* points,                    \
        const T* rotation, const T* translation, const T* out_weight, const T* point_weight,      \
        T* loss, T* ds_dpoints, T* ds_drotation, T* ds_dtranslation, T* ds_dbackground,           \
        T* ds_dout_weight, T* ds_dpoint_weight, void* workspace, size_t workspace_bytes) {        \
        return dpr_raster_residual_pullback_ex_##SUF(                                             \
            stream, DPR_ALGO_AUTO, 0u, n_in, n_out, grid, P, B, out, target, residual_scale,      \
            points, rotation, translation, out_weight, point_weight, loss, ds_dpoints,            \
            ds_drotation, ds_dtranslation, ds_dbackground, ds_dout_weight, ds_dpoint_weight,      \
            workspace, workspace_bytes);                                                          \
    }
DPR_DEFINE_RESIDUAL(f32, float)
DPR_DEFINE_RESIDUAL(f64, double)
#undef DPR_DEFINE_RESIDUAL

}  // extern "C"
