"""How many pairs of tests/test_gpu_parity.py::test_class_bet_random_configurations' batches take the re-run in the row-tiled class (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H
import tests.test_gpu_parity as T
orig = H.BatchAligner.run
def run(self, *a, **k):
    ms = orig(self, *a, **k)
    print("   retried", self.retried(), "of", self.n, "lds/wave", self.info()["lds_bytes_per_wave"], flush=True)
    return ms
H.BatchAligner.run = run
from oracle.oracle_py import Oracle
o = Oracle("avx2")
for seed in range(10):
    print("seed", seed, flush=True)
    T.test_class_bet_random_configurations(H, o, seed)
