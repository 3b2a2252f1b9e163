// Pure-write and mixed read/write bandwidth, round 6: is the forward tile kernel's "skeleton" (38 of 71 us at
// C3 for 94 MB of stores) bound by WRITE bandwidth?  Fill kernels of 67 / 160 / 640 MB with 4-, 16-byte
// plain and non-temporal stores, and a copy for reference.
// Build: hipcc -O3 --offload-arch=gfx950 microbench_write.hip -o microbench_write
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void k_fill(float* __restrict__ p, size_t n4, float v) {
    // grid-stride over float4 elements
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f4 x = {v, v + 1, v + 2, v + 3};
        if (MODE == 0) ((f4*)p)[i] = x;
        if (MODE == 1) __builtin_nontemporal_store(x, (f4*)p + i);
        if (MODE == 2) { p[4 * i] = v; p[4 * i + 1] = v; p[4 * i + 2] = v; p[4 * i + 3] = v; }  // lane-strided dwords (bad)
    }
}
// dword per lane, 256-byte rows per wave (what the tile flush does), nontemporal or not
template <int NT> __global__ __launch_bounds__(256) void k_fill_rows(float* __restrict__ p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, p + i);
        else p[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ a, float* out, size_t n4) {
    f4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += a[i];
    if (s.x + s.y + s.z + s.w == 1.2345f) out[0] = 1;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    const size_t maxb = 640ull << 20;
    float *a, *b; CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
    CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
    for (size_t mb : {67, 160, 640}) {
        const size_t bytes = mb << 20, n4 = bytes / 16, n = bytes / 4;
        for (int blocks : {2048, 8192}) {
            float t;
            t = timeit([&] { k_fill<0><<<blocks, 256>>>(a, n4, 1.f); });
            printf("fill %4zu MB float4 plain        blocks %5d: %7.1f us  %7.1f GB/s\n", mb, blocks, t * 1e3, bytes / t * 1e-6);
            t = timeit([&] { k_fill<1><<<blocks, 256>>>(a, n4, 1.f); });
            printf("fill %4zu MB float4 nontemporal  blocks %5d: %7.1f us  %7.1f GB/s\n", mb, blocks, t * 1e3, bytes / t * 1e-6);
            t = timeit([&] { k_fill_rows<0><<<blocks, 256>>>(a, n, 1.f); });
            printf("fill %4zu MB dword  plain        blocks %5d: %7.1f us  %7.1f GB/s\n", mb, blocks, t * 1e3, bytes / t * 1e-6);
            t = timeit([&] { k_fill_rows<1><<<blocks, 256>>>(a, n, 1.f); });
            printf("fill %4zu MB dword  nontemporal  blocks %5d: %7.1f us  %7.1f GB/s\n", mb, blocks, t * 1e3, bytes / t * 1e-6);
            t = timeit([&] { k_copy<<<blocks, 256>>>((const f4*)a, (f4*)b, n4); });
            printf("copy %4zu MB + %4zu MB             blocks %5d: %7.1f us  %7.1f GB/s (read + write)\n", mb, mb, blocks, t * 1e3, 2 * bytes / t * 1e-6);
            t = timeit([&] { k_read<<<blocks, 256>>>((const f4*)a, b, n4); });
            printf("read %4zu MB float4              blocks %5d: %7.1f us  %7.1f GB/s\n", mb, blocks, t * 1e3, bytes / t * 1e-6);
        }
    }
    return 0;
}
