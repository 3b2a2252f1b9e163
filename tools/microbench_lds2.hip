// LDS f64 atomic probes, round 2: how ds_add_f64 behaves when lanes of a wave share addresses
// (spatially sorted input), and what wave-level alternatives cost.  Design aid, not product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
constexpr int TILE = 9945;  // 65*17*9 f64 = 80 KB like k_tile_splat
// MODE: number of distinct addresses per wave-instruction = 64 / DUP (lanes l and l^1.. share)
// KIND 0: ds_add_f64  1: ds_add_u64  2: ds_add_f32 3: ds_add_u32
template <int DUP, int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* outp, int iters) {
    __shared__ double tile[TILE];
    for (int i = threadIdx.x; i < TILE; i += THREADS) tile[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t h = hash32(blockIdx.x * 4096 + (threadIdx.x / 64) * 64 + lane / DUP + 1);
    int base = h % (TILE - 1200);
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        base = (base + 977) % (TILE - 1200);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int idx = base + (s & 1) + 65 * ((s >> 1) & 1) + 1105 * (s >> 2);
            if (KIND == 0) atomicAdd(&tile[idx], 0.5);
            if (KIND == 1) atomicAdd(&((unsigned long long*)tile)[idx], 12345ull);
            if (KIND == 2) atomicAdd(&((float*)tile)[idx], 0.5f);
            if (KIND == 3) atomicAdd(&((uint32_t*)tile)[idx], 3u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TILE; i += THREADS) acc += tile[i];
    if (acc == 1.2345) outp[0] = (float)acc;
}
template <int DUP, int KIND, int THREADS> void run(const char* name, float* sink) {
    const int iters = 128, blocks = 512;  // 2 blocks per CU like k_tile_splat
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<DUP, KIND, THREADS><<<blocks, THREADS>>>(sink, iters); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); k<DUP, KIND, THREADS><<<blocks, THREADS>>>(sink, iters); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    double ops = (double)blocks * THREADS * iters * 8;
    double wave_instr_per_cu = ops / 64 / 256;
    printf("%-12s thr=%4d distinct/wave=%2d : %8.3f ms  %8.1f G op/s  ~%6.1f cyc/wave-instr/CU\n", name, THREADS, 64 / DUP, best,
           ops / best * 1e-6, best * 1e-3 * 2.4e9 / wave_instr_per_cu);
}
template <int KIND> void sweep(const char* name, float* sink) {
    run<1, KIND, 1024>(name, sink); run<2, KIND, 1024>(name, sink); run<4, KIND, 1024>(name, sink);
    run<8, KIND, 1024>(name, sink); run<16, KIND, 1024>(name, sink); run<64, KIND, 1024>(name, sink);
}
int main() {
    float* sink; CK(hipMalloc(&sink, 1024));
    sweep<0>("ds_add_f64", sink);
    sweep<1>("ds_add_u64", sink);
    sweep<2>("ds_add_f32", sink);
    sweep<3>("ds_add_u32", sink);
    return 0;
}
