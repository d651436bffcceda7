"""GPU parity of the small-block kernel (k_small, ba_small.hpp): batches that start at 32 cells run sixteen pairs per wave -- slots of
four lanes x eight cells that take their pairs from the batch themselves, run the first block, plain shift steps and the last step of
a global alignment -- and hand a pair to the same wave's solo mode (the per-pair driver) for grows, larger blocks, shrinks and X-drop
termination. Every pair is compared with the oracle: score, end positions, computed cells, CIGAR runs (and a spread of the CIGARs is
re-walked and re-scored without the oracle: tests/test_gpu_parity.py compare)."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from block_aligner_amd import workloads as W
from tests.test_gpu_parity import NUC, compare
from tests.test_gpu_pipelines import mode_bits, run_and_compare

pytestmark = pytest.mark.gpu

MODES = [(), ("x_drop",), ("trace",), ("trace", "x_drop")]


@pytest.fixture
def force_small(devlib, monkeypatch):
    monkeypatch.setenv("BA_FORCE_SMALL", "1")


def kernel_of(hip, matrix, gaps, size, x_drop, mode_bits, pairs):
    b = hip.BatchAligner(matrix, gaps, size, x_drop, mode_bits, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    k = b.info()["kernel"]
    b.close()
    return k


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("size", [(32, 32), (32, 64), (32, 256), (32, 1024)])
def test_small_dna(hip, oracle, force_small, mode, size):
    """Indels of 5 .. 120 bases force grows, checkpoint restores and shrinks: pairs move between slot and solo mode many times."""
    pairs = synth.make_pairs(700, (0, 2500), (0, 250), 60, synth.DNA, seed=300 + size[1], indels=2, indel_len=(5, 120))
    assert kernel_of(hip, NUC, (-5, -1), size, 80, 0, pairs) == "k_small"
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 80, mode)
    assert res["cells"].max() > 0


@pytest.mark.parametrize("mode", MODES)
def test_small_ragged_and_short(hip, oracle, force_small, mode):
    """Pairs shorter than a block, empty sequences and one-sided pairs share waves with ordinary ones."""
    rng = np.random.default_rng(12)
    lists = [(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A", b"A"), (b"A" * 31, b"A" * 33), (b"A" * 32, b"A" * 32), (b"ACGT" * 10, b"ACGT" * 300),
             (b"ACGT" * 300, b"ACGT" * 7)]
    for _ in range(300):
        n = int(rng.integers(0, 400))
        a = synth.rand_str(rng, n, synth.DNA)
        b = synth.mutate(rng, a, int(rng.integers(0, 1 + n // 8)), synth.DNA) if n else a
        lists.append((a.tobytes(), b.tobytes()))
    pairs = synth.PairSet.from_lists(lists)
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 128), 60, mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("size", [(32, 32), (32, 256)])
def test_small_protein(hip, oracle, force_small, mode, size):
    """The uc_bench shape (examples/uc_bench.rs:85-100): global alignments end inside their slot, the pairs that grow go solo."""
    w = W.config4(2500, seed=5, trace="trace" in mode)
    compare(hip, oracle, w.pairs, w.matrix, w.gaps, size, 60, mode)


def test_small_bytes(hip, oracle, force_small):
    pairs = synth.make_pairs(400, (0, 900), (0, 60), 5, np.frombuffer(b"abcdefghij\x01\xff", np.uint8), seed=6)
    compare(hip, oracle, pairs, S.BYTES1, (-2, -1), (32, 128), 0, ())
    compare(hip, oracle, pairs, S.BYTES1, (-2, -1), (32, 128), 0, ("trace",))


@pytest.mark.parametrize("mode", [("trace",), ("trace", "x_drop")])
def test_small_trace_regions_overflow_and_rerun(hip, oracle, force_small, monkeypatch, mode):
    """Pair-slot trace regions cut far too small: slots and the solo driver report the overflow, the batch re-runs those pairs."""
    monkeypatch.setenv("BA_TRACE_MARGIN_PCT", "30")
    pairs = synth.make_pairs(600, (200, 1500), (10, 150), 40, synth.DNA, seed=41, indels=2, indel_len=(10, 100))
    b = hip.BatchAligner(NUC, (-5, -1), (32, 256), 80, hip.TRACE | hip.CIGAR_EQ | (hip.X_DROP if "x_drop" in mode else 0), pairs.pool, pairs.q_off, pairs.q_len,
                         pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_small"
    b.run()
    assert b.retried() > 0
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 80, mode)


def test_small_trace_overflow_at_the_minimum_size(hip, oracle, force_small, monkeypatch):
    """Global alignment at (32, 32): a slot keeps no checkpoint (its buffer-0 scalars are only ever the zeros written when it takes its pair), and a
    slot that leaves because its trace region is exhausted hands exactly that to the solo driver (round-4 advisor finding)."""
    monkeypatch.setenv("BA_TRACE_MARGIN_PCT", "30")
    pairs = synth.make_pairs(600, (200, 1500), (10, 150), 40, synth.DNA, seed=43)
    b = hip.BatchAligner(NUC, (-5, -1), (32, 32), 0, hip.TRACE | hip.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_small"
    b.run()
    assert b.retried() > 0
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 32), 0, ("trace",))


SPECIAL_MODES = [("trace", "local_start"), ("trace", "x_drop", "local_start"), ("local_start",), ("x_drop", "local_start"), ("trace", "free_query_start_gaps"),
                 ("trace", "x_drop", "free_query_start_gaps"), ("free_query_start_gaps",)]


def _flanked_pairs(n, seed):
    """Related cores behind unrelated flanks of different lengths (what LOCAL_START skips), plus queries cut out of a longer reference (what
    FREE_QUERY_START_GAPS skips), plus ordinary related pairs with indels."""
    rng = np.random.default_rng(seed)
    lists = []
    for k in range(n):
        core = synth.rand_str(rng, int(rng.integers(40, 900)), synth.DNA)
        other = synth.mutate(rng, core, int(rng.integers(0, 1 + len(core) // 10)), synth.DNA)
        if k % 3 == 0:
            q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 120)), synth.DNA), core])
            r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 120)), synth.DNA), other])
        elif k % 3 == 1:
            q = core
            r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 300)), synth.DNA), other, synth.rand_str(rng, int(rng.integers(0, 100)), synth.DNA)])
        else:
            q, r = core, other
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    return synth.PairSet.from_lists(lists)


@pytest.mark.parametrize("mode", SPECIAL_MODES)
@pytest.mark.parametrize("size", [(32, 32), (32, 256)])
def test_small_local_and_free_start(hip, oracle, force_small, mode, size):
    """Round 5: the slots take LOCAL_START / FREE_QUERY_START_GAPS steps too (k_small's special instantiations): every cell at least the
    relative zero, a zero mask of one bit per cell behind a slot rectangle's trace words (scan_block.rs:1130-1136, 1184-1187, 1597-1611)."""
    m = 0
    for k in mode:
        m |= {"trace": hip.TRACE, "x_drop": hip.X_DROP, "local_start": hip.LOCAL_START, "free_query_start_gaps": hip.FREE_QUERY_START_GAPS}[k]
    pairs = _flanked_pairs(600, 70 + size[1])
    assert kernel_of(hip, NUC, (-5, -1), size, 60, m, pairs) == "k_small"
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 60, mode)
    pairs = synth.make_pairs(400, (0, 1500), (0, 150), 40, synth.DNA, seed=71 + size[1], indels=2, indel_len=(5, 100))
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 60, mode)
    w = W.config4(800, seed=9, trace="trace" in mode)
    compare(hip, oracle, w.pairs, w.matrix, w.gaps, size, 60, mode)


def test_small_free_query_end_gaps_stays_per_pair(hip, force_small):
    """FREE_QUERY_END_GAPS needs the mode's running column maxima (scan_block.rs:1189-1201): not a slot's."""
    pairs = synth.make_pairs(100, (5, 25), (0, 3), 0, synth.DNA, seed=3)
    assert kernel_of(hip, NUC, (-5, -1), (32, 128), 0, hip.FREE_QUERY_END_GAPS, pairs) != "k_small"
    assert kernel_of(hip, NUC, (-5, -1), (32, 128), 0, hip.FREE_QUERY_END_GAPS | hip.LOCAL_START, pairs) != "k_small"


@pytest.mark.parametrize("mode", MODES)
def test_small_dna_at_production_threshold(hip, oracle, mode):
    """No forcing, the release library: 60 k DNA pairs at 32..256 (the library takes k_small from 49152 pairs, 57344 with traceback)."""
    assert hip.lib().ba_dev_build() == 0
    pairs = synth.make_pairs(60000, (0, 1500), (0, 150), 40, synth.DNA, seed=812, indels=1, indel_len=(5, 60))
    assert kernel_of(hip, NUC, (-5, -1), (32, 256), 100, mode_bits(hip, mode, True), pairs) == "k_small"
    run_and_compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 100, mode, True, ("dna 32..256 k_small", mode))


@pytest.mark.parametrize("mode", [("trace", "x_drop", "local_start"), ("x_drop", "local_start")])
def test_small_local_start_at_production_threshold(hip, oracle, mode):
    """No forcing, the release library: LOCAL_START batches take k_small from 196608 pairs with traceback, 262144 without; FREE_QUERY_START_GAPS batches
    stay with the per-pair kernel (ba_host.cpp: measured)."""
    assert hip.lib().ba_dev_build() == 0
    n = 200000 if "trace" in mode else 265000
    pairs = _flanked_pairs(n, 5)
    assert kernel_of(hip, NUC, (-5, -1), (32, 256), 50, mode_bits(hip, mode, True), pairs) == "k_small"
    assert kernel_of(hip, NUC, (-5, -1), (32, 256), 50, mode_bits(hip, ("trace", "x_drop", "free_query_start_gaps"), True), pairs) != "k_small"
    run_and_compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 50, mode, True, ("dna local 32..256 k_small", mode))


@pytest.mark.parametrize("mode", MODES)
def test_small_protein_at_production_threshold(hip, oracle, mode):
    """No forcing: 100 k protein pairs (the library takes k_small from 32768 pairs, 98304 with traceback), BLOSUM62, block 32..256."""
    w = W.config4(100000, seed=92, trace="trace" in mode)
    assert kernel_of(hip, w.matrix, w.gaps, w.size, 60 if "x_drop" in mode else 0, mode_bits(hip, mode, False), w.pairs) == "k_small"
    run_and_compare(hip, oracle, w.pairs, w.matrix, w.gaps, w.size, 60 if "x_drop" in mode else 0, mode, False, ("protein 32..256 k_small", mode))


def _pssm_case(rng, length, block_max, indel=False):
    """examples/pssm_bench.rs:43-98 shaped: PSSM rows = BLOSUM62 rows of a random consensus, position-specific gap costs (scan_block.rs:658-676)."""
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    cons = aa[rng.integers(0, 20, length)]
    p = S.AAProfile(length, block_max, -1)
    for i, c in enumerate(cons):
        for b in aa:
            p.set(i + 1, int(b), S.BLOSUM62.get(int(c), int(b)))
    for i in range(length + 1):
        p.set_gap_open_C(i, int(rng.integers(-14, -7)))
        p.set_gap_open_R(i, int(rng.integers(-14, -7)))
        if i >= 1:
            p.set_gap_close_C(i, int(rng.integers(-3, 1)))
    q = synth.mutate(rng, cons, int(0.3 * length), aa)
    if indel and len(q) > 80:   # a long insertion or deletion: the pair grows in solo mode and comes back to its slot
        at = int(rng.integers(20, len(q) - 20)); ln = int(rng.integers(10, 90))
        q = np.concatenate([q[:at], synth.rand_str(rng, ln, aa), q[at:]]) if rng.random() < 0.5 else np.concatenate([q[:at], q[at + ln:]])
    return q.astype(np.uint8).tobytes(), p


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("size", [(32, 32), (32, 256)])
def test_small_profile(hip, oracle, force_small, mode, size):
    """Sequence-to-profile slots (round 5): right steps read the profile's rows of the lane's residues and per-column gap costs, down steps the
    rows of the column residues and per-cell gap costs with the roles of C and R exchanged (scan_block.rs:658-682); every pair its own PSSM and
    position-specific costs; empty and shorter-than-a-block pairs share the waves. Compared with the oracle pair by pair."""
    rng = np.random.default_rng(77 + size[1] + len(mode))
    cases = [_pssm_case(rng, int(rng.integers(1, 600)), size[1], indel=(k % 3 == 0)) for k in range(500)]
    cases.append((b"", cases[0][1]))
    cases.append((cases[1][0], S.AAProfile(0, size[1], -1)))
    pool = np.frombuffer(b"".join(q for q, _ in cases) + b"\0" * 8, np.uint8)
    q_len = np.array([len(q) for q, _ in cases], np.uint32)
    q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.uint64)
    m = 0
    for name in mode:
        m |= {"trace": hip.TRACE, "x_drop": hip.X_DROP}[name]
    b = hip.ProfileBatchAligner([p for _, p in cases], size, 30, m, pool, q_off, q_len)
    assert b.info()["kernel"] == "k_small"
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"]) if "trace" in mode else (None, None)
    for k, (q, p) in enumerate(cases):
        ref = oracle.align_profile(q, p, size, 30, mode)
        got = (int(res["score"][k]), int(res["query_idx"][k]), int(res["reference_idx"][k]), int(res["cells"][k]))
        assert got == (ref["score"], ref["query_idx"], ref["reference_idx"], ref["cells"]), (k, len(q), p.str_len, got, ref)
        if "trace" in mode:
            assert hip.runs_to_string(runs[int(off[k]): int(off[k + 1])]) == ref["cigar"], k
    b.close()


def test_small_profile_at_production_threshold(hip, oracle):
    """No forcing, the release library: 20 k PSSM alignments with traceback (the library takes k_small from 10000 profile pairs), every pair compared."""
    assert hip.lib().ba_dev_build() == 0
    w = W.config5(20000, seed=9)
    b = W.make_batch(hip, w)
    assert b.info()["kernel"] == "k_small"
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    ref = oracle.batch_align_profile(w.pairs.pool, w.pairs.q_off, w.pairs.q_len, w.profiles, w.size, w.x_drop, w.mode, threads=8)
    assert np.array_equal(ref["scores"], res["score"]) and np.array_equal(ref["cells"], res["cells"].astype(np.uint64))
    assert np.array_equal(ref["cig_len"], res["cigar_len"])
    ln = ref["cig_len"].astype(np.int64)
    start = np.repeat(ref["cig_off"].astype(np.int64), ln)
    within = np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
    assert np.array_equal(ref["cig_ops"][start + within], runs[: int(off[len(w.profiles)])])
    b.close()
