// LDS ds_add_u64 probes, round 5 (owner-computes splat): cost of a wave instruction as a function of
// (a) how many lanes are active, (b) how local the 64 addresses are (all lanes inside a cube of R^3
// cells of a 34 x 34 x 16 tile, as the sub-chunks of one 1024-point chunk are).  Design aid.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
constexpr int PX = 34, PY = 34, PZ = 16, TILE = PX * PY * PZ;
// ACTIVE lanes of 64 take part; every lane's base cell lies in a cube of R^3 cells (R = 0: anywhere)
template <int ACTIVE, int R, int KIND>
__global__ __launch_bounds__(1024) void k(float* outp, int iters) {
    extern __shared__ double tile[];
    for (int i = threadIdx.x; i < TILE; i += 1024) tile[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = ((lane * 37) & 63) < ACTIVE;
    uint32_t h = hash32(blockIdx.x * 4096 + threadIdx.x + 1);
    uint32_t hw = hash32(blockIdx.x * 64 + wave + 77);
    for (int it = 0; it < iters; ++it) {
        h = hash32(h + it);
        hw = hash32(hw + it);
        int x, y, z;
        if (R == 0) {
            x = h % (PX - 1); y = (h >> 8) % (PY - 1); z = (h >> 16) % (PZ - 1);
        } else {
            const int cx = hw % (PX - R), cy = (hw >> 8) % (PY - R), cz = (hw >> 16) % (PZ - (R < PZ - 1 ? R : PZ - 2));
            x = cx + h % R; y = cy + (h >> 8) % R; z = cz + (h >> 16) % (R < PZ - 1 ? R : PZ - 2);
        }
        const int base = x + PX * y + PX * PY * z;
        if (on) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int idx = base + (s & 1) + PX * ((s >> 1) & 1) + PX * PY * (s >> 2);
                if (KIND == 1) atomicAdd(&((unsigned long long*)tile)[idx], 12345ull);
                if (KIND == 0) atomicAdd(&tile[idx], 0.5);
                if (KIND == 3) atomicAdd(&((uint32_t*)tile)[idx * 2], 3u);
            }
        }
    }
    __syncthreads();
    double acc = 0;
    for (int i = threadIdx.x; i < TILE; i += 1024) acc += tile[i];
    if (acc == 1.2345) outp[0] = (float)acc;
}
template <int ACTIVE, int R, int KIND> void run(const char* name, float* sink) {
    const int iters = 256, blocks = 512;
    CK(hipFuncSetAttribute((const void*)k<ACTIVE, R, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<ACTIVE, R, KIND><<<blocks, 1024, TILE * 8>>>(sink, iters); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); k<ACTIVE, R, KIND><<<blocks, 1024, TILE * 8>>>(sink, iters); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double winstr_per_cu = (double)blocks * 16 * iters * 8 / 256;
    printf("%-10s active=%2d cube=%2d : %7.3f ms  %6.1f cyc/wave-instr/CU  %7.1f G lane-adds/s\n", name, ACTIVE, R, best,
           best * 1e-3 * 2.4e9 / winstr_per_cu, (double)blocks * 1024 * iters * 8 * ACTIVE / 64 / best * 1e-6);
}
template <int KIND> void sweep(const char* name, float* sink) {
    run<64, 0, KIND>(name, sink); run<48, 0, KIND>(name, sink); run<36, 0, KIND>(name, sink); run<32, 0, KIND>(name, sink);
    run<16, 0, KIND>(name, sink); run<8, 0, KIND>(name, sink);
    run<64, 12, KIND>(name, sink); run<64, 8, KIND>(name, sink); run<64, 6, KIND>(name, sink); run<64, 4, KIND>(name, sink);
    run<36, 8, KIND>(name, sink); run<36, 6, KIND>(name, sink); run<36, 4, KIND>(name, sink);
}
int main() {
    float* sink; CK(hipMalloc(&sink, 1024));
    sweep<1>("ds_add_u64", sink);
    sweep<3>("ds_add_u32", sink);
    sweep<0>("ds_add_f64", sink);
    return 0;
}
