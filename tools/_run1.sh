set -e
python -m pytest tests/test_owner_gpu.py -x -q > gpurun_out/t3.log 2>&1 || (tail -60 gpurun_out/t3.log; exit 1)
tail -3 gpurun_out/t3.log
