// Hilbert-curve index helpers shared by dpr_sort.hip (the full sort behind dpr_sort_points_*)
// and dpr_coarse.h (the coarse cell sort inside batched calls of the tiled path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpr {

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t spread2(uint32_t v) {  // 16 bits -> every second bit
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// Skilling's AxesToTranspose: X[0..N) with BITS bits each -> the Hilbert index in "transposed"
// form (its bits interleaved over X[0], X[1], .., most significant first)
template <int N, int BITS> __device__ __forceinline__ void hilbert_transpose(uint32_t (&X)[N]) {
    constexpr uint32_t M = 1u << (BITS - 1);
#pragma unroll
    for (uint32_t Q = M; Q > 1; Q >>= 1) {  // inverse undo
        const uint32_t Pm = Q - 1;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (X[i] & Q) {
                X[0] ^= Pm;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & Pm;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
#pragma unroll
    for (int i = 1; i < N; ++i) X[i] ^= X[i - 1];  // Gray encode
    uint32_t t = 0;
#pragma unroll
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[N - 1] & Q) t ^= Q - 1;
#pragma unroll
    for (int i = 0; i < N; ++i) X[i] ^= t;
}

}  // namespace dpr
