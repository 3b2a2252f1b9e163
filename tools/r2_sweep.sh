for c in 4096 8192 16384 32768 65536; do
  echo "== cap_min $c"; DPR_CAP_MIN=$c python tools/stage_probe.py --P 1000000 --grid 128 128 128 2>&1 | grep -v amdgpu | tail -2
done
for c in 4096 16384 32768; do
  echo "== cap_min $c (300k -> 96^3)"; DPR_CAP_MIN=$c python tools/stage_probe.py --P 300000 --grid 96 96 96 2>&1 | grep -v amdgpu | tail -2
done
