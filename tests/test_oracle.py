"""CPU tests of the oracle itself: it must reproduce every known answer the reference's own tests hold
(tests/golden/reference_kats.json, transcribed from scan_block.rs:1908-2230, avx2.rs:469-489, lib.rs:8-35),
and its scalar lane model (the kernel's specification) must equal the intrinsic version."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from tests.common import check_expect, kat_matrix, kat_profile


@pytest.mark.parametrize("backend", ["avx2", "scalar"])
def test_reference_kats(backend, oracle, oracle_scalar, kats):
    o = oracle if backend == "avx2" else oracle_scalar
    for k in kats["lane"]:
        out = o.lane_op(0, k["input"], [k["gap"]] * 16)
        assert list(out) == k["expect"], k["name"]
    for k in kats["align"] + kats["inferred"]:
        e = k["expect"]
        if k["kind"] == "profile":
            res = o.align_profile(k["q"].encode(), kat_profile(k), k["size"], k["x_drop"], k["mode"])
            check_expect(k["name"], e, res, cigar=res["cigar"])
        else:
            m = kat_matrix(k)
            res = o.align(m, k["q"].encode(), k["r"].encode(), k["gaps"], k["size"], k["x_drop"], k["mode"])
            ceq = None
            if "cigar_eq" in e:
                ceq = o.align(m, k["q"].encode(), k["r"].encode(), k["gaps"], k["size"], k["x_drop"], k["mode"], cigar_eq=True)["cigar"]
            check_expect(k["name"], e, res, cigar=res["cigar"], cigar_eq=ceq)


def test_percent_len(oracle):
    # lib.rs:109-111 and README.md:96-101: 1 % / 10 % of 10 kbp
    assert oracle.percent_len(10000, 0.01) == 128
    assert oracle.percent_len(10000, 0.1) == 1024
    assert oracle.percent_len(100, 0.01) == 32
    assert oracle.percent_len(10 ** 7, 0.1) == 16384


def test_lane_model_matches_intrinsics(oracle, oracle_scalar):
    rng = np.random.default_rng(7)
    for it in range(3000):
        kind = it % 4
        x = [rng.integers(-32768, 32768, 16), rng.integers(-600, 100, 16), rng.integers(-32768, -32000, 16),
             rng.integers(16000, 32768, 16)][kind]
        y = rng.integers(-32768, 32768, 16)
        g = int(rng.integers(-128, 0))
        for op, b in ((0, [g] * 16), (1, y), (2, y), (3, y), (4, y), (5, y), (6, y), (7, [int(x.max())] * 16), (8, y), (9, y),
                      (10, [g] * 16), (11, y)):
            assert np.array_equal(oracle.lane_op(op, x, b), oracle_scalar.lane_op(op, x, b)), (op, x, b)


MODES = [(), ("x_drop",), ("trace",), ("trace", "x_drop"), ("trace", "local_start"), ("trace", "local_start", "x_drop"),
         ("trace", "free_query_start_gaps"), ("x_drop", "local_start")]


def test_backends_agree_on_random_pairs(oracle, oracle_scalar):
    rng = np.random.default_rng(11)
    for it in range(400):
        dna = it % 2 == 0
        alpha = synth.DNA if dna else synth.AMINO
        L = int(rng.integers(0, 600))
        r = synth.rand_str(rng, L, alpha)
        q = np.concatenate([synth.mutate(rng, r, int(rng.integers(0, max(1, L // 5))), alpha), synth.rand_str(rng, int(rng.integers(0, 40)), alpha)])
        r = np.concatenate([r, synth.rand_str(rng, int(rng.integers(0, 40)), alpha)])
        m = S.NucMatrix.new_simple(int(rng.integers(1, 4)), int(rng.integers(-4, 0))) if dna else S.BLOSUM62
        ge = int(rng.integers(-3, 0)); go = ge - int(rng.integers(1, 12))
        mn = 16 << int(rng.integers(0, 3)); mx = mn << int(rng.integers(0, 4))
        mode = MODES[int(rng.integers(len(MODES)))]
        xd = int(rng.integers(0, 120))
        a = oracle.align(m, q.tobytes(), r.tobytes(), (go, ge), (mn, mx), xd, mode, cigar_eq=True)
        b = oracle_scalar.align(m, q.tobytes(), r.tobytes(), (go, ge), (mn, mx), xd, mode, cigar_eq=True)
        assert a == b, (mode, mn, mx)


def test_cigar_is_consistent_with_score(oracle):
    """A global alignment's CIGAR must consume both sequences and re-score to the reported score."""
    from tests.common import rescore
    rng = np.random.default_rng(5)
    m = S.NucMatrix.new_simple(2, -3)
    for _ in range(50):
        L = int(rng.integers(1, 400))
        r = synth.rand_str(rng, L, synth.DNA)
        q = synth.mutate(rng, r, L // 10, synth.DNA)
        res = oracle.align(m, q.tobytes(), r.tobytes(), (-5, -1), (32, 128), 0, ("trace",))
        runs = []
        num = ""
        for ch in res["cigar"]:
            if ch.isdigit():
                num += ch
            else:
                runs.append((int(num) << 4) | " M=XID".index(ch)); num = ""
        assert rescore(runs, q.tobytes(), r.tobytes(), m, (-5, -1)) == res["score"]
