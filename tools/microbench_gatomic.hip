// Global float atomic add shapes: how row-segment atomics must look to run at the fast rate.
// Each wave adds `LANES` contiguous floats starting at a row-dependent offset (ALIGN: multiple of
// 64 floats / arbitrary), rows picked pseudo-randomly from an image of `rows` x 512 floats.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
template <int LANES, int ALIGNED, int SKIPZERO>
__global__ __launch_bounds__(1024) void k(float* img, int rows, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 1024 + threadIdx.x) >> 6;
    uint32_t h = hash32(wave + 1);
    for (int it = 0; it < iters; ++it) {
        h = hash32(h);
        const int row = h % rows;
        int x0 = (h >> 16) % (512 - 128);
        if (ALIGNED) x0 &= ~63;
        else x0 |= 1;
        if (lane < LANES) {
            float v = (SKIPZERO && (lane & 3) == 0) ? 0.0f : 0.5f;
            if (!SKIPZERO || v != 0.0f) unsafeAtomicAdd(&img[(size_t)row * 512 + x0 + lane], v);
        }
    }
}
template <int LANES, int ALIGNED, int SKIPZERO> void run(const char* name, float* img, int rows) {
    const int iters = 64, blocks = 512;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<LANES, ALIGNED, SKIPZERO><<<blocks, 1024>>>(img, rows, iters); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); k<LANES, ALIGNED, SKIPZERO><<<blocks, 1024>>>(img, rows, iters); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    double instr = (double)blocks * 16 * iters;
    printf("%-44s rows=%6d : %8.3f ms  %7.2f ns/wave-instr chip-wide  %8.1f G lane-adds/s\n", name, rows, best,
           best * 1e6 / instr, instr * LANES / best * 1e-6);
}
int main() {
    float* img; CK(hipMalloc(&img, (size_t)131072 * 512 * 4)); CK(hipMemset(img, 0, (size_t)131072 * 512 * 4));
    for (int rows : {512, 32768, 131072}) {
        run<64, 1, 0>("64 lanes, 256B-aligned", img, rows);
        run<64, 0, 0>("64 lanes, misaligned (+4 B)", img, rows);
        run<32, 1, 0>("32 lanes, 128B-aligned", img, rows);
        run<33, 0, 0>("33 lanes, misaligned", img, rows);
        run<16, 1, 0>("16 lanes, 64B-aligned", img, rows);
        run<64, 1, 1>("64 lanes aligned, every 4th lane skipped", img, rows);
        run<1, 0, 0>("1 lane", img, rows);
    }
    return 0;
}
