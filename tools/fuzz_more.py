import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pytest
import tests.test_parity_gpu as tp
from oracle import oracle
oracle.build()
dev = torch.device("cuda:0")
fails = 0
for seed in range(24, 424):
    try:
        tp.test_random_configurations_against_oracle(oracle, dev, seed)
    except AssertionError as e:
        fails += 1
        print("FAIL seed", seed, str(e)[:200], flush=True)
print("done, fails =", fails)
