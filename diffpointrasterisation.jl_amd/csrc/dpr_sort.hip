// dpr_sort_points_*: pose-independent spatial pre-sort of the model-frame points along a
// HILBERT curve.  Not part of the reference; SURVEY.md 8(f) rank 3.  Every algorithm of this
// library is faster on spatially coherent input, and the sort depends on the points only, so a
// caller amortises it over poses and iterations.  Hilbert rather than Morton (Z-) order: the
// Hilbert curve has no jumps, so ANY run of consecutive points is a compact blob (a Z-order run
// that straddles a high octree boundary is two blobs far apart: 12 % of the 4096-point chunks
// of DPR_ALGO_CHUNKED then had footprints too large for LDS).
//
//   keys   30-bit (3-D) / 32-bit (2-D) Hilbert index of the point quantised on [-1, 1)^n
//          (Skilling's transpose algorithm, "Programming the Hilbert curve", AIP Conf. Proc.
//          707, 2004: axes -> transposed index by bit manipulation)
//   sort   rocPRIM radix sort of (key, index) pairs
//   gather points_sorted[i] = points[perm[i]]   (+ point weights)
//
// Gradients computed on the sorted cloud go back with ds_dpoints[perm[i]] = sorted_grad[i].
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/dpr.h"
#include "dpr_hilbert.h"
#include "dpr_tiled.h"

namespace dpr {

// FINE: 30-bit keys in 3-D (1024^3 cells, four radix passes) -- what the public dpr_sort_points_*
// uses: the box hierarchy of the owner-computes 3-D path (dpr_owner.hip) culls by groups of 16
// consecutive points, and with 256^3 cells the points inside a cell (4.7 of them at the centre of
// the 10 M-point / 256^3 headline cloud) come in arbitrary order, a group of 16 is a line of 3-4
// cells rather than a 1.5-voxel blob, and a tile looks at 2.0 points per point it needs instead of
// 1.6.  Coarse 24-bit keys (three passes) stay for the sorts INSIDE calls, which are timed: chunks
// of thousands of points only need cells of a voxel or so.
template <typename T, int NI, bool FINE>
__global__ __launch_bounds__(256) void k_hilbert_keys(int64_t P, const T* __restrict__ points,
                                                      uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ idx) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    constexpr int BITS = NI == 3 ? (FINE ? 10 : 8) : 12;
    uint32_t X[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        T x = points[p * NI + j];
        // [-1, 1) -> [0, 2^BITS); NaN and out-of-range points go to the ends
        x = (x * T(0.5) + T(0.5)) * T(1u << BITS);
        X[j] = !(x > T(0)) ? 0u : (x >= T((1u << BITS) - 1) ? (1u << BITS) - 1 : (uint32_t)x);
    }
    hilbert_transpose<NI, BITS>(X);
    uint32_t key = 0;
#pragma unroll
    for (int j = 0; j < NI; ++j)  // X[0] carries the most significant bit of every level
        key |= (NI == 3 ? spread3(X[j]) : spread2(X[j])) << (NI - 1 - j);
    keys[p] = key;
    idx[p] = (uint32_t)p;
}

// kGatherPer elements per thread, all index loads and then all random point loads issued before
// the first store: one dependent load chain per thread left the random reads at a third of the
// rate the un-permute of the tiled path reaches with the same access pattern.
constexpr int kGatherPer = 4;
template <typename T, int NI>
__global__ __launch_bounds__(256) void k_gather_points(int64_t P, const uint32_t* __restrict__ perm,
                                                       const T* __restrict__ points,
                                                       const T* __restrict__ pw,
                                                       T* __restrict__ points_sorted,
                                                       T* __restrict__ pw_sorted,
                                                       uint32_t* __restrict__ inv_perm) {
    // inv_perm (optional): inv_perm[perm[i]] = i, for consumers that bring results back to the
    // caller's order by GATHERING (random reads run ~1.7x faster than scattered stores here)
    const int64_t i0 = (int64_t)blockIdx.x * (256 * kGatherPer) + threadIdx.x;
    uint32_t p[kGatherPer];
#pragma unroll
    for (int k = 0; k < kGatherPer; ++k) {
        const int64_t i = i0 + k * 256;
        p[k] = perm[i < P ? i : P - 1];
    }
    T v[kGatherPer][NI], w[kGatherPer];
#pragma unroll
    for (int k = 0; k < kGatherPer; ++k) {
#pragma unroll
        for (int j = 0; j < NI; ++j) v[k][j] = points[(size_t)p[k] * NI + j];
        w[k] = pw_sorted ? pw[p[k]] : T(0);
    }
#pragma unroll
    for (int k = 0; k < kGatherPer; ++k) {
        const int64_t i = i0 + k * 256;
        if (i >= P) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) points_sorted[i * NI + j] = v[k][j];
        if (pw_sorted) pw_sorted[i] = w[k];
        if (inv_perm) inv_perm[p[k]] = (uint32_t)i;
    }
}

static size_t salign(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t radix_temp_bytes(int64_t P) {
    size_t temp = 0;
    uint32_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, temp, nul, nul, nul, nul, (size_t)P, 0, 32,
                                    (hipStream_t)0);
    return temp;
}

size_t sort_workspace_bytes(int64_t P) {
    if (P < 1) P = 1;
    return salign((size_t)P * 4) * 3 + salign(radix_temp_bytes(P));  // keys in/out, idx in, temp
}

template <typename T>
int sort_points_impl(void* stream, int n_in, int64_t P, const T* points, T* points_sorted,
                     uint32_t* perm, const T* pw, T* pw_sorted, void* ws_, size_t ws_bytes,
                     uint32_t* inv_perm, bool fine) {
    if (n_in != 2 && n_in != 3)
        return fail(DPR_ERR_UNSUPPORTED_DIMS, "dpr_sort_points: n_in must be 2 or 3 (got %d)", n_in);
    if (P < 0 || P >= ((int64_t)1 << 32))
        return fail(DPR_ERR_INVALID_ARG, "dpr_sort_points: P out of range");
    if (P == 0) return DPR_OK;
    if (!points || !points_sorted || !perm)
        return fail(DPR_ERR_INVALID_ARG, "dpr_sort_points: NULL points / points_sorted / perm");
    if ((pw == nullptr) != (pw_sorted == nullptr))
        return fail(DPR_ERR_INVALID_ARG,
                    "dpr_sort_points: point_weight and point_weight_sorted go together");
    if (points == points_sorted)
        return fail(DPR_ERR_INVALID_ARG, "dpr_sort_points: in-place sorting is not supported");
    const size_t need = sort_workspace_bytes(P);
    if (!ws_ || ws_bytes < need)
        return fail(DPR_ERR_WORKSPACE, "dpr_sort_points needs %zu workspace bytes, got %zu", need,
                    ws_ ? ws_bytes : (size_t)0);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)ws_;
    // workspace: keys_in | keys_out | idx_in | rocPRIM temporary storage
    uint32_t* keys_in = (uint32_t*)ws;
    uint32_t* keys_out = (uint32_t*)(ws + salign((size_t)P * 4));
    uint32_t* idx_in = (uint32_t*)(ws + 2 * salign((size_t)P * 4));
    void* temp = ws + 3 * salign((size_t)P * 4);
    size_t temp_bytes = radix_temp_bytes(P);
    const dim3 grid((unsigned)((P + 255) / 256));
    if (n_in == 3 && fine)
        hipLaunchKernelGGL((k_hilbert_keys<T, 3, true>), grid, dim3(256), 0, st, P, points, keys_in, idx_in);
    else if (n_in == 3)
        hipLaunchKernelGGL((k_hilbert_keys<T, 3, false>), grid, dim3(256), 0, st, P, points, keys_in, idx_in);
    else if (fine)
        hipLaunchKernelGGL((k_hilbert_keys<T, 2, true>), grid, dim3(256), 0, st, P, points, keys_in, idx_in);
    else
        hipLaunchKernelGGL((k_hilbert_keys<T, 2, false>), grid, dim3(256), 0, st, P, points, keys_in, idx_in);
    // In-call sorts of 3-D clouds order by the top 5 bits per axis of the 24-bit key only (32^3 cells; a stable
    // sort: points of one cell keep their order): the chunk-owner kernels need compact 4096-point chunks, not
    // sorted neighbours -- C4's share measured 6.20 / 6.22 / 6.12 / 6.31 / 10.4 ms per step at 8 / 6 / 5 / 4 / 3 bits
    // -- and 15 key bits are two radix passes instead of three (DPR_SORT_BITS: experiments).
    static const int cbits = env_knob("DPR_SORT_BITS", 5, 2, 8);
    const unsigned begin_bit = (!fine && n_in == 3) ? (unsigned)(3 * (8 - cbits)) : 0u;
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, idx_in, perm,
                                             (size_t)P, begin_bit, (fine && n_in == 3) ? 30 : 24, st);
    if (e != hipSuccess)
        return fail(DPR_ERR_HIP, "rocprim::radix_sort_pairs failed: %s", hipGetErrorString(e));
    const dim3 ggrid((unsigned)((P + 256 * kGatherPer - 1) / (256 * kGatherPer)));
    if (n_in == 3)
        hipLaunchKernelGGL((k_gather_points<T, 3>), ggrid, dim3(256), 0, st, P, perm, points, pw,
                           points_sorted, pw_sorted, inv_perm);
    else
        hipLaunchKernelGGL((k_gather_points<T, 2>), ggrid, dim3(256), 0, st, P, perm, points, pw,
                           points_sorted, pw_sorted, inv_perm);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(DPR_ERR_HIP, "dpr_sort_points: %s", hipGetErrorString(e));
    return DPR_OK;
}

template int sort_points_impl<float>(void*, int, int64_t, const float*, float*, uint32_t*,
                                     const float*, float*, void*, size_t, uint32_t*, bool);
template int sort_points_impl<double>(void*, int, int64_t, const double*, double*, uint32_t*,
                                      const double*, double*, void*, size_t, uint32_t*, bool);

}  // namespace dpr

extern "C" {

size_t dpr_sort_points_workspace_bytes(int64_t P) { return dpr::sort_workspace_bytes(P); }

int dpr_sort_points_f32(void* stream, int n_in, int64_t P, const float* points,
                        float* points_sorted, uint32_t* perm, const float* point_weight,
                        float* point_weight_sorted, void* workspace, size_t workspace_bytes) {
    return dpr::sort_points_impl<float>(stream, n_in, P, points, points_sorted, perm, point_weight,
                                        point_weight_sorted, workspace, workspace_bytes, nullptr, true);
}

int dpr_sort_points_f64(void* stream, int n_in, int64_t P, const double* points,
                        double* points_sorted, uint32_t* perm, const double* point_weight,
                        double* point_weight_sorted, void* workspace, size_t workspace_bytes) {
    return dpr::sort_points_impl<double>(stream, n_in, P, points, points_sorted, perm,
                                         point_weight, point_weight_sorted, workspace,
                                         workspace_bytes, nullptr, true);
}
}
