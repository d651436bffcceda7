import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from block_aligner_amd import hip, scores as S, synth
import test_gpu_parity as T
size = (32, 128); mode = ("trace",)
rng = np.random.default_rng(41 + size[1] + len(mode))
cases = [T._pssm_case(rng, int(rng.integers(1, 500)), size[1]) for _ in range(120)]
cases.append((b"", cases[0][1]))
pool = np.frombuffer(b"".join(q for q, _ in cases) + b"\0" * 8, np.uint8)
q_len = np.array([len(q) for q, _ in cases], np.uint32)
q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.uint64)
out = {}
for env in ("", "1"):
    if env: os.environ["BA_FORCE_QUAD"] = "1"
    b = hip.ProfileBatchAligner([p for _, p in cases], size, 30, hip.TRACE, pool, q_off, q_len)
    for rnd in range(2):
        b.run(); res = b.results()
        out[(env, rnd)] = res
        print(env, rnd, "status", res["status"].sum(), "empty cigars", [(k, int(q_len[k]), cases[k][1].str_len) for k in range(len(cases)) if res["cigar_len"][k] == 0][:20])
    b.close()
for k in ("score", "cells", "cigar_len"):
    print(k, np.array_equal(out[("", 0)][k], out[("1", 0)][k]), np.nonzero(out[("", 0)][k] != out[("1", 0)][k])[0][:10])
