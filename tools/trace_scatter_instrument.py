"""Instruments k_scatter_wc with per-block, per-round wall_clock64() stamps (diagnostic build only):
  cp <csrc>/dpr_tiled.hip /tmp/keep.hip; python tools/trace_scatter_instrument.py;
  bash tools/build_variant.sh trace "" dpr_tiled.hip; cp /tmp/keep.hip <csrc>/dpr_tiled.hip
then on the GPU box: DPR_LIB_OVERRIDE=<variant.so> python tools/trace_scatter.py."""
p = "diffpointrasterisation.jl_amd/csrc/dpr_tiled.hip"; s = open(p).read()
def rep(a, b):
    global s
    assert s.count(a) >= 1, a[:70]
    s = s.replace(a, b, 1)
rep("namespace dpr {\n", "namespace dpr {\n__device__ unsigned long long g_trace[512 * 8 * 8];\n__device__ int g_round_dummy;\n#define TR(k) if (threadIdx.x == 0 && blockIdx.x < 512 && tr_round < 8) g_trace[(blockIdx.x * 8 + tr_round) * 8 + (k)] = wall_clock64();\n")
# inside k_scatter_wc only: the round loop
a = s.index("void k_scatter_wc(")
b = s.index("// ------------------------------------------------------------------ local binning: K1")
body = s[a:b]
def brep(x, y):
    global body
    assert body.count(x) == 1, x[:70]
    body = body.replace(x, y)
brep("    for (int64_t base = lo; base < hi; base += S) {\n", "    int tr_round = -1;\n    for (int64_t base = lo; base < hi; base += S) {\n        ++tr_round;\n        TR(0)\n")
brep("            lds_barrier();\n            // b. exclusive scan of lhist (in place)\n", "            TR(1)\n            lds_barrier();\n            TR(2)\n            // b. exclusive scan of lhist (in place)\n")
brep("            lds_barrier();\n            // c. place into LDS in tile order; remember the global destination\n", "            lds_barrier();\n            TR(3)\n            // c. place into LDS in tile order; remember the global destination\n")
brep("            lds_barrier();\n            // d. write-out in LDS (= tile) order; e. advance cursors, clear the histogram\n", "            TR(4)\n            lds_barrier();\n            TR(5)\n            // d. write-out in LDS (= tile) order; e. advance cursors, clear the histogram\n")
brep("                    lhist[i] = 0;\n                }\n            }\n            lds_barrier();\n", "                    lhist[i] = 0;\n                }\n            }\n            TR(6)\n            lds_barrier();\n            TR(7)\n")
s = s[:a] + body + s[b:]
s = s.replace("}  // namespace dpr", "}  // namespace dpr\nextern \"C\" int dpr_debug_trace(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dpr::g_trace), sizeof(unsigned long long) * 512 * 8 * 8); }\n", 1)
open(p, "w").write(s)
