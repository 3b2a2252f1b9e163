// Design micro-benchmarks for the splat pipeline on MI355X (not part of the product).
// Measures the primitive rates that decide between direct atomics, binning and LDS tiles.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void k_copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}

__global__ void k_read_aos3(const float* __restrict__ a, float* __restrict__ sink, size_t n) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += a[3 * i] + a[3 * i + 1] + a[3 * i + 2];
    if (acc == 1.2345f) sink[0] = acc;
}

// 8 atomics per "point": 4 pairs of x-adjacent voxels around a random ref (like the 3D splat)
__global__ void k_atomic_scatter(float* grid, int n, size_t P, int coherent) {
    size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= P) return;
    uint32_t h = coherent ? (uint32_t)(p / 4) * 2654435761u : hash32((uint32_t)p);
    int x, y, z;
    if (coherent) {  // consecutive points walk along x
        size_t q = p / 2;
        x = q % (n - 1); y = (q / (n - 1)) % (n - 1); z = (q / ((size_t)(n - 1) * (n - 1))) % (n - 1);
    } else {
        x = h % (n - 1); y = (h >> 8) % (n - 1); z = hash32(h) % (n - 1);
    }
    for (int s = 0; s < 8; ++s) {
        size_t off = (size_t)(x + (s & 1)) + (size_t)n * ((y + ((s >> 1) & 1)) + (size_t)n * (z + (s >> 2)));
        unsafeAtomicAdd(grid + off, 0.125f);
    }
}

__global__ void k_gather8(const float* __restrict__ grid, float* __restrict__ outp, int n, size_t P) {
    size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= P) return;
    uint32_t h = hash32((uint32_t)p);
    int x = h % (n - 1), y = (h >> 8) % (n - 1), z = hash32(h) % (n - 1);
    float acc = 0;
    for (int s = 0; s < 8; ++s) {
        size_t off = (size_t)(x + (s & 1)) + (size_t)n * ((y + ((s >> 1) & 1)) + (size_t)n * (z + (s >> 2)));
        acc += grid[off];
    }
    outp[p] = acc;
}

// LDS atomics: each block owns a TILE-float tile, every thread does `iters` x 8 ds_add_f32
template <int TILE>
__global__ void k_lds_atomic(float* outp, int iters) {
    __shared__ float tile[TILE];
    for (int i = threadIdx.x; i < TILE; i += blockDim.x) tile[i] = 0;
    __syncthreads();
    uint32_t h = hash32(blockIdx.x * 1024 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        h = hash32(h);
        int base = h % (TILE - 320);
        for (int s = 0; s < 8; ++s) {
            int off = base + (s & 1) + 17 * ((s >> 1) & 1) + 289 * (s >> 2);
            atomicAdd(&tile[off], 0.5f);
        }
    }
    __syncthreads();
    float acc = 0;
    for (int i = threadIdx.x; i < TILE; i += blockDim.x) acc += tile[i];
    if (acc == 1.2345f) outp[0] = acc;
}

// scattered record writes: record i (16 B) goes to slot perm(i)
__global__ void k_scatter16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, uint32_t mul) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t j = ((uint64_t)i * mul) % n;  // mul coprime with n -> permutation
    dst[j] = src[i];
}
__global__ void k_gather16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, uint32_t mul) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t j = ((uint64_t)i * mul) % n;
    dst[i] = src[j];
}
__global__ void k_scatter12_4(const float4* __restrict__ src, float* __restrict__ d3, float* __restrict__ d1, size_t n, uint32_t mul) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t j = ((uint64_t)i * mul) % n;
    float4 v = src[i];
    d3[3 * j] = v.x; d3[3 * j + 1] = v.y; d3[3 * j + 2] = v.z;
    d1[j] = v.w;
}
// binned scatter: many bins, write frontier per bin (simulates counting-sort scatter):
// record i goes to bin (hash(i) % nbins) at position cursor -- we emulate with precomputed
// slot = bin * cap + (i / nbins) so writes to a bin's frontier advance together.
__global__ void k_scatter_bins(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int nbins) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t bin = hash32((uint32_t)i) % nbins;
    size_t cap = n / nbins + 1;
    size_t j = (size_t)bin * cap + (i / nbins) % cap;
    dst[j] = src[i];
}
// returning int atomics on `nbins` hot counters
__global__ void k_cursor_atomics(uint32_t* cursors, uint32_t* outp, size_t n, int nbins) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t bin = hash32((uint32_t)i) % nbins;
    uint32_t r = atomicAdd(&cursors[bin], 1u);
    if (r == 0xffffffffu) outp[0] = r;
}

template <typename F> float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main() {
    const size_t P = 10000000;
    const int n = 256;
    const size_t G = (size_t)n * n * n;
    float *grid, *pts, *sink, *o1;
    float4 *recA, *recB;
    uint32_t* cursors;
    CK(hipMalloc(&grid, G * 4)); CK(hipMalloc(&pts, P * 16)); CK(hipMalloc(&sink, 1024));
    CK(hipMalloc(&o1, P * 16)); CK(hipMalloc(&recA, P * 16)); CK(hipMalloc(&recB, P * 16 + (1 << 20)));
    CK(hipMalloc(&cursors, 1 << 20));
    CK(hipMemset(grid, 0, G * 4)); CK(hipMemset(pts, 0, P * 16)); CK(hipMemset(recA, 0, P * 16));
    CK(hipMemset(cursors, 0, 1 << 20));
    const int TB = 256;
    const unsigned nb = (unsigned)((P + TB - 1) / TB);
    float ms;

    ms = time_ms([&] { k_copy4<<<4096, TB>>>((const float4*)recA, recB, P); });
    printf("copy float4 160MB+160MB      : %8.3f ms  %7.1f GB/s\n", ms, 2 * P * 16 / ms * 1e-6);
    ms = time_ms([&] { k_read_aos3<<<4096, TB>>>(pts, sink, P); });
    printf("read AoS3 120MB (3 dword/lane): %8.3f ms  %7.1f GB/s\n", ms, P * 12 / ms * 1e-6);
    ms = time_ms([&] { k_atomic_scatter<<<nb, TB>>>(grid, n, P, 0); });
    printf("global atomics random 8/pt   : %8.3f ms  %7.2f G atom/s\n", ms, 8.0 * P / ms * 1e-6);
    ms = time_ms([&] { k_atomic_scatter<<<nb, TB>>>(grid, n, P, 1); });
    printf("global atomics coherent 8/pt : %8.3f ms  %7.2f G atom/s\n", ms, 8.0 * P / ms * 1e-6);
    ms = time_ms([&] { k_gather8<<<nb, TB>>>(grid, o1, n, P); });
    printf("global gathers random 8/pt   : %8.3f ms  %7.2f G gather/s\n", ms, 8.0 * P / ms * 1e-6);
    {
        const int iters = 64, blocks = 256 * 8;
        ms = time_ms([&] { k_lds_atomic<4096 + 512><<<blocks, 256>>>(sink, iters); });
        double n_at = (double)blocks * 256 * iters * 8;
        printf("LDS atomics (18KB tile, 8 blk/CU): %8.3f ms  %7.2f G atom/s\n", ms, n_at / ms * 1e-6);
        ms = time_ms([&] { k_lds_atomic<32768><<<256 * 4, 1024>>>(sink, iters); });
        n_at = (double)256 * 4 * 1024 * iters * 8;
        printf("LDS atomics (128KB tile, 1 blk/CU x1024): %8.3f ms  %7.2f G atom/s\n", ms, n_at / ms * 1e-6);
    }
    ms = time_ms([&] { k_scatter16<<<nb, TB>>>(recA, recB, P, 2654435761u); });
    printf("scatter 16B records (perm)   : %8.3f ms  %7.1f GB/s useful\n", ms, P * 32 / ms * 1e-6);
    ms = time_ms([&] { k_gather16<<<nb, TB>>>(recA, recB, P, 2654435761u); });
    printf("gather 16B records (perm)    : %8.3f ms  %7.1f GB/s useful\n", ms, P * 32 / ms * 1e-6);
    ms = time_ms([&] { k_scatter12_4<<<nb, TB>>>(recA, (float*)recB, o1, P, 2654435761u); });
    printf("scatter 12B+4B (perm)        : %8.3f ms  %7.1f GB/s useful\n", ms, P * 32 / ms * 1e-6);
    for (int nbins : {512, 4096, 32768}) {
        ms = time_ms([&] { k_scatter_bins<<<nb, TB>>>(recA, recB, P, nbins); });
        printf("scatter 16B into %5d bins   : %8.3f ms  %7.1f GB/s useful\n", nbins, ms, P * 32 / ms * 1e-6);
    }
    for (int nbins : {512, 4096, 32768}) {
        ms = time_ms([&] { k_cursor_atomics<<<nb, TB>>>(cursors, (uint32_t*)sink, P, nbins); });
        printf("returning int atomics %5d ctr: %8.3f ms  %7.2f G atom/s\n", nbins, ms, P / ms * 1e-6);
    }
    return 0;
}
