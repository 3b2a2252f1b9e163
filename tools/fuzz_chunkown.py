"""Seeded fuzz of DPR_ALGO_CHUNKED on 2-D grids against the oracle, many seeds: the per-seed case is
tests/test_configs_gpu.py::chunk_owner_fuzz_case (16 of them run in the test suite)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle
from tests.test_configs_gpu import chunk_owner_fuzz_case
oracle.build()
dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fails = 0
t0 = time.time()
for seed in range(n_seeds):
    try:
        chunk_owner_fuzz_case(oracle, dev, seed)
    except AssertionError as e:
        fails += 1
        print("FAIL seed", seed, str(e)[:300], flush=True)
    if seed % 20 == 19:
        print(f"seed {seed} done, {time.time() - t0:.0f} s, fails {fails}", flush=True)
print("done, fails =", fails)
sys.exit(1 if fails else 0)
