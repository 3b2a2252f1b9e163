// Do global float atomics run faster when every address is only ever touched from ONE XCD?
// (design aid, not product)  Row-shaped atomics (64 lanes x 4 B contiguous) into `nimg` images of
// 512 x 512 floats:
//   shared       every block hits all images
//   xcd-owned    a block on XCD x (HW_REG_XCC_ID) hits images x, x+8, ... only
//   blk%8-owned  the same with blockIdx.x % 8 in place of the hardware id (is the mapping round-robin?)
// Also prints the XCC_ID histogram per blockIdx % 8.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf;
}
// mode 0 shared, 1 xcd-owned, 2 blk%8-owned; lanes: active lanes per instruction
template <int MODE>
__global__ __launch_bounds__(256) void k(float* img, int nimg, int iters, int lanes, int scatter) {
    const unsigned x = MODE == 1 ? xcc_id() : (blockIdx.x & 7);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t h = hash32(blockIdx.x * 4 + wave + 1);
    const int per = nimg / 8;
    for (int it = 0; it < iters; ++it) {
        h = hash32(h + it);
        int im = MODE == 0 ? (h % nimg) : (x + 8 * ((h >> 3) % per));
        const uint32_t row = (h >> 8) % 512, col = ((h >> 20) % 7) * 64;
        size_t off = (size_t)im * 262144 + row * 512 + col + lane;
        if (scatter) off = (size_t)im * 262144 + (hash32(h ^ (lane * 0x9e3779b9u)) % 262144);
        if (lane < lanes) unsafeAtomicAdd(img + off, 1.0f);
    }
}
__global__ void k_map(unsigned* hist) { if (threadIdx.x == 0) atomicAdd(&hist[(blockIdx.x & 7) * 16 + xcc_id()], 1u); }

// element type of the atomic (row shape, shared images): 0 f32, 1 u32, 2 u64, 3 f64
template <int TY>
__global__ __launch_bounds__(256) void k_ty(float* img, int nimg, int iters, int lanes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t h = hash32(blockIdx.x * 4 + wave + 1);
    for (int it = 0; it < iters; ++it) {
        h = hash32(h + it);
        const int im = h % nimg;
        const uint32_t row = (h >> 8) % 512, col = ((h >> 20) % 3) * 64;
        const size_t off = (size_t)im * 262144 + row * 512 + col * (TY >= 2 ? 2 : 1);
        if (lane < lanes) {
            if (TY == 0) unsafeAtomicAdd(img + off + lane, 1.0f);
            if (TY == 1) atomicAdd((unsigned*)img + off + lane, 3u);
            if (TY == 2) atomicAdd((unsigned long long*)(img + off) + lane, 3ull);
            if (TY == 3) unsafeAtomicAdd((double*)(img + off) + lane, 1.0);
        }
    }
}
template <int TY> void run_ty(const char* name, float* img, int nimg, int lanes) {
    const int iters = 2000, blocks = 2048;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_ty<TY><<<blocks, 256>>>(img, nimg, 10, lanes); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); k_ty<TY><<<blocks, 256>>>(img, nimg, iters, lanes); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double winstr = (double)blocks * 4 * iters;
    printf("%-8s row nimg=%3d lanes=%2d : %8.3f ms  %7.2f G wave-instr/s  %8.1f G lane-adds/s\n", name, nimg, lanes,
           best, winstr / best * 1e-6, winstr * lanes / best * 1e-6);
}
template <int MODE> void run(const char* name, float* img, int nimg, int lanes, int scatter) {
    const int iters = 2000, blocks = 2048;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<blocks, 256>>>(img, nimg, 10, lanes, scatter); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); k<MODE><<<blocks, 256>>>(img, nimg, iters, lanes, scatter); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double winstr = (double)blocks * 4 * iters;
    printf("%-14s nimg=%3d lanes=%2d %s : %8.3f ms  %7.2f G wave-instr/s  %8.1f G lane-adds/s\n", name, nimg, lanes,
           scatter ? "scattered" : "row      ", best, winstr / best * 1e-6, winstr * lanes / best * 1e-6);
}
int main() {
    const int maximg = 512;
    float* img; CK(hipMalloc(&img, (size_t)maximg * 262144 * 4)); CK(hipMemset(img, 0, (size_t)maximg * 262144 * 4));
    unsigned* hist; CK(hipMalloc(&hist, 128 * 4)); CK(hipMemset(hist, 0, 128 * 4));
    k_map<<<4096, 64>>>(hist); CK(hipDeviceSynchronize());
    unsigned hh[128]; CK(hipMemcpy(hh, hist, sizeof(hh), hipMemcpyDeviceToHost));
    for (int b = 0; b < 8; ++b) { printf("blockIdx%%8=%d -> xcc:", b); for (int x = 0; x < 16; ++x) if (hh[b * 16 + x]) printf(" %d:%u", x, hh[b * 16 + x]); printf("\n"); }
    for (int lanes : {64, 32, 8}) {
        run_ty<0>("f32", img, 64, lanes);
        run_ty<1>("u32", img, 64, lanes);
        run_ty<2>("u64", img, 64, lanes);
        run_ty<3>("f64", img, 64, lanes);
    }
    for (int nimg : {64})
        for (int lanes : {64, 16, 1})
            for (int scatter : {0, 1}) {
                if (scatter && lanes != 64) continue;
                run<0>("shared", img, nimg, lanes, scatter);
                run<1>("xcd-owned", img, nimg, lanes, scatter);
                run<2>("blk%8-owned", img, nimg, lanes, scatter);
            }
    return 0;
}
