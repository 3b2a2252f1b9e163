// Is ds_add_f32's rate on gfx950 (193 cycles per wave-instruction in r01_microbench_lds_atomics.txt, against 10 for
// ds_add_u32 and 26 for ds_add_f64) a property of the fp32 DENORMAL mode of the wave?  Same probe as microbench_lds.hip
// (256-thread blocks, 8 LDS atomics per thread and iteration on random cells of a 16 KB tile) with the MODE register's
// FP_DENORM field for single precision set to "flush" (0) or "preserve" (3) by s_setreg at kernel entry.
// Design aid, not product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
constexpr int TILE = 4096;

// OP 0: ds_add_f32   1: ds_add_u32   2: ds_add_rtn_f32   3: ds_add_f64   4: ds_add_u64   5: ds_pk_add_f16 (as 2 x half)
// DEN: -1 leave the mode alone, 0 flush fp32 denormals, 3 preserve them
template <int OP, int DEN>
__global__ __launch_bounds__(256) void k(float* outp, int iters) {
    __shared__ double tile_d[TILE / 2];
    float* tile = (float*)tile_d;
    uint32_t* tile_u = (uint32_t*)tile_d;
    // hwreg(HW_REG_MODE = 1, offset 4, size 2): id | offset << 6 | (size - 1) << 11
    if (DEN == 0) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);
    if (DEN == 3) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 3);
    for (int i = threadIdx.x; i < TILE; i += 256) tile[i] = 0;
    __syncthreads();
    uint32_t h = hash32(blockIdx.x * 256 + threadIdx.x + 1);
    int a[8];
    for (int s = 0; s < 8; ++s) { h = hash32(h); a[s] = h & (TILE - 1); }
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int idx = (a[s] + it * 67) & (TILE - 1);
            if (OP == 0) asm volatile("ds_add_f32 %0, %1" :: "v"(idx * 4), "v"(0.5f + (float)s) : "memory");
            if (OP == 1) atomicAdd(&tile_u[idx], 3u);
            if (OP == 2) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(idx * 4), "v"(0.5f) : "memory"); acc += r; }
            if (OP == 3) atomicAdd(&tile_d[idx >> 1], 0.5);
            if (OP == 4) atomicAdd(&((unsigned long long*)tile_d)[idx >> 1], 12345ull);
            if (OP == 5) asm volatile("ds_pk_add_f16 %0, %1" :: "v"(idx * 4), "v"(0x38003800u) : "memory");
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TILE; i += 256) acc += tile[i];
    if (acc == 1.2345f) outp[0] = acc;
}

template <int OP, int DEN> void run(const char* name, float* sink, int blocks_per_cu) {
    const int iters = 256, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<OP, DEN><<<blocks, 256>>>(sink, iters); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); k<OP, DEN><<<blocks, 256>>>(sink, iters); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double ops = (double)blocks * 256 * iters * 8;
    const double wave_instr_per_cu = ops / 64 / 256;
    printf("%-44s blk/CU=%d : %8.3f ms  %8.1f G op/s  ~%6.1f cyc/wave-instr/CU\n", name, blocks_per_cu, best,
           ops / best * 1e-6, best * 1e-3 * 2.4e9 / wave_instr_per_cu);
}

int main() {
    float* sink; CK(hipMalloc(&sink, 1024));
    for (int bpc : {1, 4}) {
        run<0, -1>("ds_add_f32, mode as launched", sink, bpc);
        run<0, 0>("ds_add_f32, fp32 denormals FLUSHED", sink, bpc);
        run<0, 3>("ds_add_f32, fp32 denormals preserved", sink, bpc);
        run<2, 0>("ds_add_rtn_f32, flushed", sink, bpc);
        run<1, -1>("ds_add_u32", sink, bpc);
        run<3, -1>("ds_add_f64", sink, bpc);
        run<4, -1>("ds_add_u64", sink, bpc);
        run<5, -1>("ds_pk_add_f16", sink, bpc);
    }
    return 0;
}
