#!/usr/bin/env python3
"""One-off stress run (GPU box): tests/test_gpu_parity.py::test_random_configurations over many more seeds.
usage: python3 tools/stress_random.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import tests.test_gpu_parity as T   # noqa: E402
from block_aligner_amd import hip as H   # noqa: E402
from oracle.oracle_py import Oracle   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 24
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
o = Oracle("avx2")
bad = 0
for seed in range(first, first + count):
    try:
        T.test_random_configurations(H, o, seed)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL seed", seed, repr(e)[:300], flush=True)
print("done,", count, "configurations, failures:", bad)
