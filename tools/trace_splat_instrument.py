"""Instruments k_tile_splat with per-block wall_clock64() stamps (diagnostic build only):
  cp csrc/dpr_tiled.hip /tmp/keep.hip; python tools/trace_splat_instrument.py; build a variant
  library from the patched file (see profiles/r01_experiments.md), restore the source, then run
  DPR_LIB_OVERRIDE=<variant.so> python tools/trace_splat.py [P]."""
p="diffpointrasterisation.jl_amd/csrc/dpr_tiled.hip"; s=open(p).read()
def rep(a,b):
    global s
    assert a in s, a[:70]
    s=s.replace(a,b,1)
rep("namespace dpr {\n","namespace dpr {\n__device__ unsigned long long g_trace[8192 * 8];\n#define TR(k) if (threadIdx.x == 0 && blockIdx.x < 8192) g_trace[blockIdx.x * 8 + (k)] = wall_clock64();\n")
rep('''    const WorkItem item = items[blockIdx.x];
    if (blockIdx.x >= *n_items) return;  // the grid is sized for the worst case
    for (int i = threadIdx.x; i < NVH; i += kSplatThreads) acc[i] = 0.0;''','''    TR(0)
    const WorkItem item = items[blockIdx.x];
    if (blockIdx.x >= *n_items) return;  // the grid is sized for the worst case
    TR(1)
    for (int i = threadIdx.x; i < NVH; i += kSplatThreads) acc[i] = 0.0;''')
rep('''    lds_barrier();  // LDS phases only: prefetched records stay in flight
    while (r < r1) {
        Rec4<T> cur[kPF];''','''    lds_barrier();  // LDS phases only: prefetched records stay in flight
    TR(2)
    while (r < r1) {
        Rec4<T> cur[kPF];''')
rep('''    lds_barrier();  // LDS phases only: prefetched records stay in flight
    if ((item.part_nparts >> 16) > 1) {''','''    TR(3)
    lds_barrier();  // LDS phases only: prefetched records stay in flight
    TR(4)
    if ((item.part_nparts >> 16) > 1) {''')
rep('''    for (int row = threadIdx.x; row < ROWS; row += kSplatThreads)
        hb[row] = (T)acc[row * (TX + 1) + TX];
}''','''    for (int row = threadIdx.x; row < ROWS; row += kSplatThreads)
        hb[row] = (T)acc[row * (TX + 1) + TX];
    TR(5)
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_trace[blockIdx.x * 8 + 6] = ((unsigned long long)xcc << 32) | hwid;
        g_trace[blockIdx.x * 8 + 7] = item.end - item.begin;
    }
}''')
s=s.replace("}  // namespace dpr","}  // namespace dpr\nextern \"C\" int dpr_debug_trace(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dpr::g_trace), sizeof(unsigned long long) * 8192 * 8); }\n",1)
open(p,"w").write(s)
