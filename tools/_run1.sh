set -e
mkdir -p gpurun_out
python -m pytest tests/test_parity_gpu.py -x -q -k "chunked" > gpurun_out/t2.log 2>&1 || (tail -60 gpurun_out/t2.log; exit 1)
tail -3 gpurun_out/t2.log
python tools/own_probe.py --bwd > gpurun_out/p2.log 2>&1 || (tail -30 gpurun_out/p2.log; exit 1)
cat gpurun_out/p2.log
DPR_LIB_OVERRIDE=$PWD/diffpointrasterisation.jl_amd/libdpr_stats.so python tools/own_probe.py > gpurun_out/p3.log 2>&1 || (tail -30 gpurun_out/p3.log; exit 1)
head -40 gpurun_out/p3.log
