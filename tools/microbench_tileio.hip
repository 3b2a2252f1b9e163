// Fixed per-tile cost of the tile kernels: blocks of 1024 / 256 threads with an 80 / 40 KB LDS
// tile that (A) do nothing, (B) zero the tile, (C) also store 32 KB of rows, (D) also read them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void k_tile(float* __restrict__ out, const float* __restrict__ in, int nx, int ny, int nz) {
    __shared__ double acc[9945];
    if (MODE >= 1) {
        for (int i = threadIdx.x; i < 9945; i += THREADS) acc[i] = 0.0;
        __syncthreads();
    }
    // tile 64 x 16 x 8 of a 256^3 grid
    const int t = blockIdx.x % 2048;
    const int tx = t % 4, ty = (t / 4) % 16, tz = t / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE == 3) {
        for (int row = wave; row < 128; row += THREADS / 64) {
            const int y = ty * 16 + (row & 15), z = tz * 8 + (row >> 4);
            acc[row * 65 + lane] = in[((size_t)z * ny + y) * nx + tx * 64 + lane];
        }
        __syncthreads();
    }
    if (MODE >= 2) {
        for (int row = wave; row < 128; row += THREADS / 64) {
            const int y = ty * 16 + (row & 15), z = tz * 8 + (row >> 4);
            out[((size_t)z * ny + y) * nx + tx * 64 + lane] = (float)acc[row * 65 + lane];
        }
    }
    if (MODE == 0 && out == nullptr) acc[threadIdx.x] = 1.0;
}

template <int THREADS, int MODE> float run(float* out, const float* in, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tile<THREADS, MODE>), dim3(blocks), dim3(THREADS), 0, 0, out, in, 256, 256, 256);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_tile<THREADS, MODE>), dim3(blocks), dim3(THREADS), 0, 0, out, in, 256, 256, 256);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}

int main() {
    float *out, *in;
    CK(hipMalloc(&out, 256ull * 256 * 256 * 4));
    CK(hipMalloc(&in, 256ull * 256 * 256 * 4));
    CK(hipMemset(in, 0, 256ull * 256 * 256 * 4));
    for (int blocks : {2048, 2563}) {
        printf("blocks %d, 1024 threads: empty %.1f us | zero %.1f | zero+store %.1f | load+store %.1f\n", blocks,
               run<1024, 0>(out, in, blocks), run<1024, 1>(out, in, blocks), run<1024, 2>(out, in, blocks), run<1024, 3>(out, in, blocks));
        printf("blocks %d,  512 threads: empty %.1f us | zero %.1f | zero+store %.1f | load+store %.1f\n", blocks,
               run<512, 0>(out, in, blocks), run<512, 1>(out, in, blocks), run<512, 2>(out, in, blocks), run<512, 3>(out, in, blocks));
        printf("blocks %d,  256 threads: empty %.1f us | zero %.1f | zero+store %.1f | load+store %.1f\n", blocks,
               run<256, 0>(out, in, blocks), run<256, 1>(out, in, blocks), run<256, 2>(out, in, blocks), run<256, 3>(out, in, blocks));
    }
    return 0;
}
