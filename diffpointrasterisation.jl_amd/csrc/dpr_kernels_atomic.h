// DPR_ALGO_ATOMIC: thread-per-point kernels that talk to the grid directly.
//   forward  : 2^N_out global float atomics per (point, pose)
//   backward : 2^N_out gathers per (point, pose); the pose loop runs inside the
//              thread so ds_dpoints / ds_dpoint_weight are accumulated in
//              registers and written once with plain stores; per-pose sums are
//              reduced wave -> block and leave the block as one atomic each.
// This is the general fallback and the right shape for small problems and for
// many poses over a small (cache-resident) grid.
#pragma once
#include "dpr_device.h"

namespace dpr {

constexpr int kBlock = 256;

// out[.., b] = background[b]   (src/raster.jl:27)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_fill_background(T* __restrict__ out, int64_t G,
                                                            const T* __restrict__ background) {
    const int64_t b = blockIdx.y;
    const T bg = background ? background[b] : T(0);
    T* o = out + b * G;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < G;
         i += (int64_t)gridDim.x * kBlock)
        o[i] = bg;
}

// src/raster.jl:36-66, one thread per (point, pose) handling all 2^N_out neighbours.
template <typename T, int NI, int NO>
__global__ __launch_bounds__(kBlock) void k_fwd_atomic(GridDesc<NO> gd, int64_t P, int64_t B,
                                                       T* __restrict__ out,
                                                       const T* __restrict__ points,
                                                       const T* __restrict__ rot,
                                                       const T* __restrict__ trans,
                                                       const T* __restrict__ ow,
                                                       const T* __restrict__ pw) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    T pt[NI];
    load_point<T, NI>(points, p, pt);
    const T pwi = pw ? pw[p] : T(1);
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b);
        int ref0[NO];
        T dlo[NO];
        if (!ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo)) continue;
        const T w = ps.ow * pwi;  // src/raster.jl:52
        T* o = out + b * gd.G;
#pragma unroll
        for (int s = 0; s < (1 << NO); ++s) {
            const int off = nbr_offset<NO>(ref0, s, gd);
            if (off >= 0) atomic_add<T>(o + off, voxel_weight<T, NO>(dlo, s, w));
        }
    }
}

// Blocks per pose of k_grid_sum: every block ends with ONE atomic on ds_dbackground[b], and
// thousands of atomics on one address serialise (8192 blocks: 116 us for a 67 MB grid, most of
// it the atomics) -- at most 1024 per pose.
static inline int64_t grid_sum_blocks(int64_t G, int64_t nb) {
    int64_t want = (G + (int64_t)kBlock * 16 - 1) / ((int64_t)kBlock * 16);
    if (want > 1024) want = 1024;
    if (want * nb > 8192) want = (8192 + nb - 1) / nb;
    return want < 1 ? 1 : want;
}

// ds_dbackground[b] = sum(ds_dout[.., b])  (src/raster_pullback.jl:78;
// ext/DiffPointRasterisationCUDAExt.jl:265-267).  ds_dbackground pre-zeroed.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_grid_sum(const T* __restrict__ g, int64_t G,
                                                     T* __restrict__ ds_dbackground,
                                                     Residual<T> rs) {
    __shared__ T part[2][kBlock / kWave];
    const int64_t b = blockIdx.y;
    const int64_t o = b * G;
    T acc = T(0), sq = T(0);
    const int64_t step = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (!rs.target) {  // four loads in flight per thread
        T a1 = T(0), a2 = T(0), a3 = T(0);
        for (; i + 3 * step < G; i += 4 * step) {
            acc += g[o + i];
            a1 += g[o + i + step];
            a2 += g[o + i + 2 * step];
            a3 += g[o + i + 3 * step];
        }
        acc += (a1 + a2) + a3;
    }
    for (; i < G; i += step) {
        const T x = g[o + i];
        if (rs.target) {
            const T d = x - rs.target[o + i];
            acc += rs.scale * d;
            sq += d * d;
        } else {
            acc += x;
        }
    }
    acc = wave_sum<T>(acc);
    sq = wave_sum<T>(sq);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        part[0][threadIdx.x / kWave] = acc;
        part[1][threadIdx.x / kWave] = sq;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        T s = part[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < kBlock / kWave; ++w) s += part[threadIdx.x][w];
        if (threadIdx.x == 0) atomic_add<T>(ds_dbackground + b, s);
        else if (rs.target && rs.loss) atomic_add<T>(rs.loss + b, s);
    }
}

// Pullback over the pose range [b_lo, b_hi).  Pre-zeroed: ds_drotation,
// ds_dtranslation, ds_dout_weight.  ds_dpoints / ds_dpoint_weight are written
// with plain stores when `accumulate_points` is 0 (single launch covering all
// poses) and with atomics onto pre-zeroed buffers otherwise.
template <typename T, int NI, int NO>
__global__ __launch_bounds__(kBlock) void k_bwd_gather(
    GridDesc<NO> gd, int64_t P, int64_t B, const T* __restrict__ g, const T* __restrict__ points,
    const T* __restrict__ rot, const T* __restrict__ trans, const T* __restrict__ ow,
    const T* __restrict__ pw, T* __restrict__ ds_dpoints, T* __restrict__ ds_drotation,
    T* __restrict__ ds_dtranslation, T* __restrict__ ds_dout_weight,
    T* __restrict__ ds_dpoint_weight, int poses_per_slice, int accumulate_points,
    Residual<T> rs) {
    constexpr int NV = NO * NI + NO + 1;  // dR | dt | d out_weight
    constexpr int NW = kBlock / kWave;
    __shared__ T red[NW][NV];

    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = p < P;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    T pt[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) pt[j] = T(0);
    if (live) load_point<T, NI>(points, p, pt);
    const T pwi = (live && pw) ? pw[p] : T(1);

    T acc_pt[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) acc_pt[j] = T(0);
    T acc_pw = T(0);

    const int64_t b_lo = (int64_t)blockIdx.y * poses_per_slice;
    const int64_t b_hi = (b_lo + poses_per_slice < B) ? b_lo + poses_per_slice : B;
    for (int64_t b = b_lo; b < b_hi; ++b) {
        const Pose<T, NI, NO> ps = load_pose<T, NI, NO>(rot, trans, ow, b);
        T vals[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) vals[k] = T(0);
        int ref0[NO];
        T dlo[NO];
        if (live && ref_and_deltas<T, NI, NO>(pt, ps, gd, ref0, dlo)) {
            const int64_t gb = b * gd.G;
            T scaled[NO], dow_part, dpw_part;
            if (rs.target)  // (uniform; decided once, not per gather)
                point_backward<T, NI, NO>(
                    ref0, dlo, gd, ps.ow, pwi,
                    [&](int off) { return rs.scale * (g[gb + off] - rs.target[gb + off]); }, scaled,
                    dow_part, dpw_part);
            else
                point_backward<T, NI, NO>(ref0, dlo, gd, ps.ow, pwi,
                                          [&](int off) { return g[gb + off]; }, scaled, dow_part,
                                          dpw_part);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
#pragma unroll
                for (int j = 0; j < NI; ++j) vals[n + j * NO] = scaled[n] * pt[j];  // :69
                vals[NO * NI + n] = scaled[n];                                      // :68
            }
            vals[NO * NI + NO] = dow_part;  // :57
#pragma unroll
            for (int j = 0; j < NI; ++j) {  // rotation' * scaled  (:70)
                T v = ps.R[0 + j * NO] * scaled[0];
#pragma unroll
                for (int n = 1; n < NO; ++n) v = v + ps.R[n + j * NO] * scaled[n];
                acc_pt[j] += v;
            }
            acc_pw += dpw_part;  // :58
        }
        // per-pose sums: wave -> block -> one atomic per scalar per block
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const T s = wave_sum<T>(vals[k]);
            if (lane == 0) red[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < NV) {
            T s = red[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][threadIdx.x];
            const int k = threadIdx.x;
            if (s != T(0)) {
                if (k < NO * NI)
                    atomic_add<T>(ds_drotation + b * (NO * NI) + k, s);
                else if (k < NO * NI + NO)
                    atomic_add<T>(ds_dtranslation + b * NO + (k - NO * NI), s);
                else
                    atomic_add<T>(ds_dout_weight + b, s);
            }
        }
        __syncthreads();
    }
    if (live) {
        if (accumulate_points) {
#pragma unroll
            for (int j = 0; j < NI; ++j) atomic_add<T>(ds_dpoints + p * NI + j, acc_pt[j]);
            if (ds_dpoint_weight) atomic_add<T>(ds_dpoint_weight + p, acc_pw);
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) ds_dpoints[p * NI + j] = acc_pt[j];
            if (ds_dpoint_weight) ds_dpoint_weight[p] = acc_pw;
        }
    }
}

}  // namespace dpr
