"""GPU parity proper: the HIP path vs the CPU oracle on the same seeded inputs, bit-exact (all-integer work):
score, end indices, CIGAR runs and the computed-cell count, through the batch entry points of the C ABI."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from oracle.oracle_py import cigar_runs_to_string

pytestmark = pytest.mark.gpu


def compare(H, oracle, pairs, matrix, gaps, size, x_drop, mode_names, cigar_eq=True, threads=8):
    mode = 0
    for m in mode_names:
        mode |= {"trace": H.TRACE, "x_drop": H.X_DROP}[m]
    if cigar_eq and "trace" in mode_names:
        mode |= H.CIGAR_EQ
    b = H.BatchAligner(matrix, gaps, size, x_drop, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    b.run()
    res = b.results()
    assert not res["status"].any(), np.nonzero(res["status"])[0][:10]
    ref = oracle.batch_align(matrix, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, gaps, size, x_drop,
                             mode_names, cigar_eq=cigar_eq, threads=threads)
    bad = np.nonzero((res["score"] != ref["scores"]) | (res["query_idx"] != ref["query_idx"]) | (res["reference_idx"] != ref["reference_idx"]))[0]
    assert bad.size == 0, (bad[:10], res["score"][bad[:5]], ref["scores"][bad[:5]], res["query_idx"][bad[:5]], ref["query_idx"][bad[:5]],
                           res["reference_idx"][bad[:5]], ref["reference_idx"][bad[:5]])
    assert int(res["cells"].sum()) == ref["cells"]
    if "trace" in mode_names:
        assert np.array_equal(res["cigar_len"], ref["cig_len"])
        runs, off = b.cigars(res["cigar_len"])
        for p in range(len(pairs)):
            want = ref["cig_ops"][int(ref["cig_off"][p]): int(ref["cig_off"][p]) + int(ref["cig_len"][p])]
            got = runs[int(off[p]): int(off[p + 1])]
            assert np.array_equal(got, want), (p, cigar_runs_to_string(got)[:200], cigar_runs_to_string(want)[:200])
    b.close()
    return res


NUC = S.NucMatrix.new_simple(2, -3)


@pytest.mark.parametrize("mode", [(), ("x_drop",), ("trace",), ("trace", "x_drop")])
@pytest.mark.parametrize("size", [(16, 16), (32, 32), (32, 128), (128, 128)])
def test_dna_small_blocks(hip, oracle, mode, size):
    pairs = synth.make_pairs(300, (0, 700), (0, 80), 25, synth.DNA, seed=100 + size[0] + size[1])
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 50, mode)


@pytest.mark.parametrize("mode", [("x_drop",), ("trace", "x_drop"), ("trace",)])
@pytest.mark.parametrize("size", [(32, 256), (64, 512), (128, 1024), (32, 2048)])
def test_dna_growing_blocks(hip, oracle, mode, size):
    """Long insertions/deletions force grow + checkpoint restore + shrink (C3b of SURVEY 8d)."""
    pairs = synth.make_pairs(120, (1500, 3000), (100, 300), 100, synth.DNA, seed=7 + size[1], indels=3, indel_len=(20, 200))
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, mode)
    assert res["cells"].max() > 0


@pytest.mark.parametrize("mode", [(), ("x_drop",), ("trace",), ("trace", "x_drop")])
def test_protein_blosum62(hip, oracle, mode):
    """uc_bench-shaped: BLOSUM62, gaps (-11,-1), block 32..256 (examples/uc_bench.rs:85-100)."""
    pairs = synth.make_pairs(300, (22, 900), (0, 200), 0, synth.AMINO, seed=31)
    compare(hip, oracle, pairs, S.BLOSUM62, (-11, -1), (32, 256), 50, mode)


def test_bytes_matrix(hip, oracle):
    pairs = synth.make_pairs(100, (0, 300), (0, 40), 5, np.frombuffer(b"abcdefghij\x01\xff", np.uint8), seed=5)
    compare(hip, oracle, pairs, S.BYTES1, (-2, -1), (16, 64), 0, ())
    compare(hip, oracle, pairs, S.BYTES1, (-2, -1), (16, 64), 0, ("trace",))


def test_edge_cases(hip, oracle):
    """Empty and ragged inputs, N bases, one-sided emptiness, identical and unrelated pairs."""
    rng = np.random.default_rng(3)
    lists = [(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A", b"A"), (b"ACGTNNNNACGT", b"ACGTACGT"),
             (b"A" * 500, b"A" * 500), (b"A" * 300, b"T" * 300), (b"ACGT" * 100, b"ACGT" * 100 + b"TTTT" * 30),
             (synth.rand_str(rng, 1000, synth.DNA).tobytes(), synth.rand_str(rng, 17, synth.DNA).tobytes()),
             (synth.rand_str(rng, 15, synth.DNA).tobytes(), synth.rand_str(rng, 900, synth.DNA).tobytes())]
    pairs = synth.PairSet.from_lists(lists)
    for mode in [(), ("x_drop",), ("trace",), ("trace", "x_drop")]:
        for size in [(16, 16), (32, 256)]:
            compare(hip, oracle, pairs, S.NW1, (-2, -1), size, 20, mode)


def test_config2_shape(hip, oracle):
    """BASELINE config 2 shape, reduced count: 1 kbp DNA, ~90 % identity, X-drop, block 32..256."""
    pairs = synth.make_pairs(1000, 1000, 100, 50, synth.DNA, seed=1234)
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 100, ("x_drop",))


def test_config3_shape(hip, oracle):
    """BASELINE config 3 shape, reduced count: 10 kbp DNA, k = 1000, +500 tails, X-drop 100, block 128..1024, traceback."""
    pairs = synth.make_pairs(96, 10000, 1000, 500, synth.DNA, seed=1234)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), (128, 1024), 100, ("trace", "x_drop"))
    assert (res["query_idx"] > 9000).all()


def test_traceback_consumer_path(hip, oracle, monkeypatch):
    """Large TRACE batches hand finished trace stacks to dedicated traceback workgroups inside the launch
    (agent-scope release/acquire queue, slot reuse). Force that path on a batch small enough for the oracle and big
    enough that every fill wave recycles its trace slots several times."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_SLOTS_PER_WAVE", "2")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    pairs = synth.make_pairs(6000, (200, 1500), (10, 150), 40, synth.DNA, seed=77, indels=1, indel_len=(10, 80))
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 60, ("trace", "x_drop"))
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 128), 0, ("trace",))
