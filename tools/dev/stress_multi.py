"""k_multi at production batch sizes over random parameters, every pair against the oracle (GPU box): python tools/dev/stress_multi.py [seed0] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
from oracle.oracle_py import Oracle
from tests.test_gpu_pipelines import run_and_compare
o = Oracle("avx2")
if os.environ.get("STRESS_GEOM"):   # the development library reads BA_MQ_GEOM / BA_FORCE_MULTI
    H.use_library(H.DEV_LIB_PATH); os.environ["BA_FORCE_MULTI"] = "1"
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for seed in range(seed0, seed0 + count):
    rng = np.random.default_rng(seed)
    hi = 128 << int(rng.integers(0, 5))
    lo_len, hi_len = int(rng.integers(50, 400)), int(rng.integers(600, 2500))
    edits = (int(rng.integers(0, 40)), int(rng.integers(60, 300)))
    tails = int(rng.integers(0, 300))
    x_drop = int(rng.integers(20, 200))
    mode = [(), ("x_drop",), ("trace",), ("trace", "x_drop")][int(rng.integers(0, 4))]
    if os.environ.get("STRESS_SPECIAL"):   # round 5: the special modes the multi-pair kernels take, on pairs with unrelated heads
        mode = mode + (("local_start",) if rng.random() < 0.6 else ("free_query_start_gaps",))
    kind = int(rng.integers(0, 3))
    ext = -int(rng.integers(1, 4)); opn = ext - int(rng.integers(2, 12))
    if kind == 0:
        alpha, matrix = synth.DNA, S.NucMatrix.new_simple(int(rng.integers(1, 4)), -int(rng.integers(1, 5)))
    elif kind == 1:
        alpha, matrix = synth.AMINO, S.BLOSUM62
    else:
        alpha, matrix = np.frombuffer(b"abcdefgh", np.uint8), S.BYTES1
        mode = tuple(m for m in mode if m != "x_drop"); opn, ext = -2, -1
    pairs = synth.make_pairs(17000, (lo_len, hi_len), edits, tails, alpha, seed=seed, indels=int(rng.integers(0, 3)), indel_len=(5, 150))
    if os.environ.get("STRESS_SPECIAL"):
        lists = []
        for p in range(len(pairs)):
            q, r = np.frombuffer(pairs.query(p), np.uint8), np.frombuffer(pairs.reference(p), np.uint8)
            if p % 3 == 0:
                q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 200)), alpha), q]); r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 200)), alpha), r])
            elif p % 3 == 1:
                r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 400)), alpha), r])
            lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
        pairs = synth.PairSet.from_lists(lists)
    lo = 128
    if os.environ.get("STRESS_WIDE"):   # round 6: k_multi's 256-cell slots (DNA, the plain modes, 256..512 / 1024 / 2048), longer pairs
        lo, hi = 256, 512 << int(rng.integers(0, 3))
        alpha, matrix, kind = synth.DNA, S.NucMatrix.new_simple(int(rng.integers(1, 4)), -int(rng.integers(1, 5))), 0
        mode = tuple(m for m in mode if m in ("trace", "x_drop"))
        pairs = synth.make_pairs(5000, (2 * lo_len + 3000, 2 * hi_len + 4000), (2 * edits[0], 3 * edits[1]), tails, alpha, seed=seed, indels=int(rng.integers(0, 3)), indel_len=(5, 300), workers=8)
    if os.environ.get("STRESS_M512"):   # round 6: k_multi's 512-cell slot (DNA, the plain modes, 512..1024 / 2048 / 4096 -- the last one through the class bet), long pairs
        lo, hi = 512, 1024 << int(rng.integers(0, 3))
        alpha, matrix, kind = synth.DNA, S.NucMatrix.new_simple(int(rng.integers(1, 4)), -int(rng.integers(1, 5))), 0
        mode = tuple(m for m in mode if m in ("trace", "x_drop"))
        pairs = synth.make_pairs(int(rng.integers(300, 1500)), (4 * lo_len + 5000, 4 * hi_len + 8000), (4 * edits[0], 6 * edits[1]), tails, alpha, seed=seed, indels=int(rng.integers(0, 3)), indel_len=(20, 1500), workers=8)
    if os.environ.get("STRESS_GEOM"):   # round 6: k_multi in four-wave workgroups at two / three waves per SIMD (DNA, the plain modes, 128..512 / 1024), ~8 k pairs
        hi = 512 << int(rng.integers(0, 2))
        alpha, matrix, kind = synth.DNA, S.NucMatrix.new_simple(int(rng.integers(1, 4)), -int(rng.integers(1, 5))), 0
        mode = tuple(m for m in mode if m in ("trace", "x_drop"))
        os.environ["BA_MQ_GEOM"] = str(2 + seed % 2)
        pairs = synth.make_pairs(int(rng.integers(3000, 9000)), (lo_len + 1000, hi_len + 2000), (edits[0], 2 * edits[1]), tails, alpha, seed=seed, indels=int(rng.integers(0, 3)), indel_len=(5, 200), workers=8)
    what = (seed, kind, (lo, hi), (opn, ext), x_drop, mode) + ((("geom", os.environ["BA_MQ_GEOM"]),) if os.environ.get("STRESS_GEOM") else ())
    try:
        run_and_compare(H, o, pairs, matrix, (opn, ext), (lo, hi), x_drop if "x_drop" in mode else 0, mode, kind == 0, what)
        print("ok", what, flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL", what, repr(e)[:300], flush=True)
print("done, failures:", bad)
