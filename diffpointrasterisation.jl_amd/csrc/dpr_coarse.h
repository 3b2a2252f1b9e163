// Coarse cell sort of the model-frame points INSIDE batched calls of the tiled path (included by
// dpr_tiled.hip after k_colscan and lds_barrier).
//
// What local binning (k_bin_local) needs from the order of the cloud is only that a sub-chunk of
// consecutive points covers a bounded region -- not a full spatial sort.  One counting-sort
// pass into 4096 cells of the model frame (16^3 in 3-D, 64^2 in 2-D, cells numbered along a
// Hilbert curve so that consecutive cells are neighbours) gives exactly that: a sub-chunk lies
// inside one cell (or two consecutive ones), i.e. inside a box of 1/16 of the frame, whatever the
// density of the cloud.  Cost: one counting pass + one write-combining scatter over the points
// (the machinery of the per-pose binning: per-slice LDS histograms, column prefix, exact
// atomic-free placement) instead of key generation + three radix passes + a random gather
// (50 M fp64 points: 3.8 ms).  Every point gets a slot: coordinates outside [-1, 1) and NaN are
// clamped into the border cells (a pose may still translate such a point into the grid).
//
//   k_cell_count    slice histograms by cell      -> counts[nblk][4096]
//   k_colscan       (dpr_tiled.hip) column prefix -> counts (in place), totals[4096]
//   k_cell_scan     exclusive scan of the totals  -> cell_start[4096]
//   k_cell_scatter  sub-chunks ordered by cell in LDS, written out as runs: points_sorted,
//                   point_weight_sorted, inv_perm[p] = position of point p in the sorted copy
#pragma once
#include "dpr_hilbert.h"

namespace dpr {

constexpr int kCells = 4096;
constexpr int kCellThreads = 1024;

template <typename T, int NI> __device__ __forceinline__ int cell_of(const T (&p)[NI]) {
    static_assert(NI == 2 || NI == 3, "cells exist for 2-D and 3-D clouds");
    constexpr int BITS = NI == 3 ? 4 : 6;
    uint32_t X[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const T x = (p[j] * T(0.5) + T(0.5)) * T(1u << BITS);
        X[j] = !(x > T(0)) ? 0u : (x >= T((1u << BITS) - 1) ? (1u << BITS) - 1 : (uint32_t)x);
    }
    hilbert_transpose<NI, BITS>(X);
    uint32_t key = 0;
#pragma unroll
    for (int j = 0; j < NI; ++j)  // X[0] carries the most significant bit of every level
        key |= (NI == 3 ? spread3(X[j]) : spread2(X[j])) << (NI - 1 - j);
    return (int)(key & (kCells - 1));
}

template <typename T, int NI>
__global__ __launch_bounds__(kCellThreads) void k_cell_count(int64_t P, int64_t chunk,
                                                             const T* __restrict__ points,
                                                             uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kCells];
    for (int i = threadIdx.x; i < kCells; i += kCellThreads) hist[i] = 0;
    __syncthreads();
    const unsigned slice = xcd_slice(blockIdx.x, gridDim.x);
    const int64_t lo = (int64_t)slice * chunk;
    const int64_t hi = (lo + chunk < P) ? lo + chunk : P;
    constexpr int kCU = 4;  // points in flight per thread
    for (int64_t base = lo + threadIdx.x; base < hi; base += (int64_t)kCU * kCellThreads) {
        T pt[kCU][NI];
        bool live[kCU];
#pragma unroll
        for (int u = 0; u < kCU; ++u) {
            const int64_t p = base + (int64_t)u * kCellThreads;
            live[u] = p < hi;
            load_point<T, NI>(points, live[u] ? p : hi - 1, pt[u]);
        }
#pragma unroll
        for (int u = 0; u < kCU; ++u)
            if (live[u]) atomicAdd(&hist[cell_of<T, NI>(pt[u])], 1u);
    }
    __syncthreads();
    uint32_t* row = counts + (size_t)slice * kCells;
    for (int i = threadIdx.x; i < kCells; i += kCellThreads) row[i] = hist[i];
}

// exclusive scan of totals[4096] -> cell_start[4096]; one block of 1024 threads
__global__ __launch_bounds__(1024) void k_cell_scan(const uint32_t* __restrict__ totals,
                                                    uint32_t* __restrict__ cell_start) {
    __shared__ uint32_t wsum[16];
    constexpr int PER = kCells / 1024;
    uint32_t c[PER], s = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        c[q] = totals[threadIdx.x * PER + q];
        s += c[q];
    }
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if ((threadIdx.x & 63) >= o) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - s;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        cell_start[threadIdx.x * PER + q] = run;
        run += c[q];
    }
}

// Write-combining scatter by cell (the structure of k_scatter_wc): S points per round.
//   LDS: cursor[4096] | lhist[4096] | spt[S * NI] | spw[S] | dest[S]
// PERM: also write perm[position in the sorted copy] = original index (what the chunk-owner kernels scatter
// their gradients through), coalesced with the points.
template <typename T, int NI, bool HAS_PW, int S, bool PERM = false>
__global__ __launch_bounds__(kCellThreads) void k_cell_scatter(
    int64_t P, int64_t chunk, const T* __restrict__ points, const T* __restrict__ pw,
    const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ cell_start,
    T* __restrict__ points_sorted, T* __restrict__ pw_sorted, uint32_t* __restrict__ inv_perm,
    uint32_t* __restrict__ perm = nullptr) {
    constexpr int PPT = S / kCellThreads;
    constexpr int BPT = kCells / kCellThreads;
    __shared__ uint32_t cursor[kCells], lhist[kCells];
    __shared__ T spt[S * NI];
    __shared__ T spw[HAS_PW ? S : 1];
    __shared__ uint32_t dest[S];
    __shared__ uint32_t sorig[PERM ? S : 1];
    uint32_t* const wsum = dest;  // per-wave sums of the scan: dead before phase c writes dest[]
    const unsigned slice = xcd_slice(blockIdx.x, gridDim.x);
    const uint32_t* row = prefix + (size_t)slice * kCells;
    for (int i = threadIdx.x; i < kCells; i += kCellThreads) {
        cursor[i] = cell_start[i] + row[i];
        lhist[i] = 0;
    }
    __syncthreads();
    const int64_t lo = (int64_t)slice * chunk;
    const int64_t hi = (lo + chunk < P) ? lo + chunk : P;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int bin0 = threadIdx.x * BPT;
    for (int64_t base = lo; base < hi; base += S) {
        T pt[PPT][NI], w[PPT];
        int cell[PPT];
        uint32_t lrank[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int64_t p = base + threadIdx.x + (int64_t)k * kCellThreads;
            const int64_t pl = p < hi ? p : hi - 1;
            load_point<T, NI>(points, pl, pt[k]);
            w[k] = HAS_PW ? pw[pl] : T(1);
        }
        // a. classify, rank inside (sub-chunk, cell)
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int64_t p = base + threadIdx.x + (int64_t)k * kCellThreads;
            cell[k] = p < hi ? cell_of<T, NI>(pt[k]) : -1;
            lrank[k] = 0;
            if (cell[k] >= 0) lrank[k] = atomicAdd(&lhist[cell[k]], 1u);
        }
        lds_barrier();
        // b. exclusive scan of lhist (in place)
        uint32_t cnt[BPT], cnt_sum = 0;
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            cnt[q] = lhist[bin0 + q];
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
        for (int wv = 0; wv < kCellThreads / kWave; ++wv) n_valid += wsum[wv];
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            lhist[bin0 + q] = run;
            run += cnt[q];
        }
        lds_barrier();
        // c. place into LDS in cell order; remember the global destination
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int64_t p = base + threadIdx.x + (int64_t)k * kCellThreads;
            if (cell[k] >= 0) {
                const uint32_t sidx = lhist[cell[k]] + lrank[k];
                const uint32_t d = cursor[cell[k]] + lrank[k];
#pragma unroll
                for (int j = 0; j < NI; ++j) spt[sidx * NI + j] = pt[k][j];
                if (HAS_PW) spw[sidx] = w[k];
                dest[sidx] = d;
                if (PERM) sorig[sidx] = (uint32_t)p;
                if (!PERM) __builtin_nontemporal_store(d, &inv_perm[p]);  // (the PERM caller has no use for it)
            }
        }
        lds_barrier();
        // d. write-out, scalar by scalar: a cell's run is contiguous in LDS and in memory
        for (uint32_t e = threadIdx.x; e < n_valid * NI; e += kCellThreads) {
            const uint32_t i = e / NI, j = e - i * NI;
            points_sorted[(size_t)dest[i] * NI + j] = spt[e];
        }
        if (HAS_PW)
            for (uint32_t i = threadIdx.x; i < n_valid; i += kCellThreads) pw_sorted[dest[i]] = spw[i];
        if (PERM)
            for (uint32_t i = threadIdx.x; i < n_valid; i += kCellThreads) perm[dest[i]] = sorig[i];
        // e. advance the cursors, clear the histogram
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            cursor[bin0 + q] += cnt[q];
            lhist[bin0 + q] = 0;
        }
        lds_barrier();
    }
}

// slices of the cloud for the coarse sort: one block per CU at most, whole sub-chunks
static void coarse_slices(size_t elem, int64_t P, int* nblk, int64_t* chunk) {
    const int64_t sub = elem == 4 ? 4096 : 2048;
    int64_t c = ((P + 255) / 256 + sub - 1) / sub * sub;
    if (c < sub) c = sub;
    *chunk = c;
    *nblk = (int)((P + c - 1) / c);
    if (*nblk < 1) *nblk = 1;
}
// workspace of the coarse sort: counts[nblk][4096] | totals[4096] | cell_start[4096]
static size_t coarse_workspace_bytes(size_t elem, int64_t P) {
    int nblk;
    int64_t chunk;
    coarse_slices(elem, P < 1 ? 1 : P, &nblk, &chunk);
    return ((size_t)nblk * kCells * 4 + 255) / 256 * 256 + 2 * (size_t)kCells * 4;
}

template <typename T, int NI>
static int coarse_sort_points(hipStream_t st, int64_t P, const T* points, const T* pw,
                              T* points_sorted, T* pw_sorted, uint32_t* inv_perm, char* ws,
                              uint32_t* perm = nullptr) {
    if (P <= 0) return DPR_OK;
    int nblk;
    int64_t chunk;
    coarse_slices(sizeof(T), P, &nblk, &chunk);
    uint32_t* counts = (uint32_t*)ws;
    uint32_t* totals = (uint32_t*)(ws + ((size_t)nblk * kCells * 4 + 255) / 256 * 256);
    uint32_t* cell_start = totals + kCells;
    hipLaunchKernelGGL((k_cell_count<T, NI>), dim3(nblk), dim3(kCellThreads), 0, st, P, chunk, points,
                       counts);
    hipLaunchKernelGGL(k_colscan, dim3(kCells / kScanTiles), dim3(1024), 0, st, counts, nblk, kCells,
                       totals);
    hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(1024), 0, st, (const uint32_t*)totals, cell_start);
    constexpr int S = sizeof(T) == 4 ? 4096 : 2048;
    if (perm) {
        if (pw)
            hipLaunchKernelGGL((k_cell_scatter<T, NI, true, S, true>), dim3(nblk), dim3(kCellThreads), 0, st, P,
                               chunk, points, pw, (const uint32_t*)counts, (const uint32_t*)cell_start,
                               points_sorted, pw_sorted, inv_perm, perm);
        else
            hipLaunchKernelGGL((k_cell_scatter<T, NI, false, S, true>), dim3(nblk), dim3(kCellThreads), 0, st, P,
                               chunk, points, pw, (const uint32_t*)counts, (const uint32_t*)cell_start,
                               points_sorted, pw_sorted, inv_perm, perm);
        return DPR_OK;
    }
    if (pw)
        hipLaunchKernelGGL((k_cell_scatter<T, NI, true, S>), dim3(nblk), dim3(kCellThreads), 0, st, P,
                           chunk, points, pw, (const uint32_t*)counts, (const uint32_t*)cell_start,
                           points_sorted, pw_sorted, inv_perm);
    else
        hipLaunchKernelGGL((k_cell_scatter<T, NI, false, S>), dim3(nblk), dim3(kCellThreads), 0, st, P,
                           chunk, points, pw, (const uint32_t*)counts, (const uint32_t*)cell_start,
                           points_sorted, pw_sorted, inv_perm);
    return DPR_OK;
}

}  // namespace dpr
