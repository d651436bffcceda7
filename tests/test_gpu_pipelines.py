"""The multi-kernel pipelines at the batch sizes from which the library itself picks them -- no development switch, the release
library: k_quad -> queue -> per-pair kernel -> k_walk for batches that start at 32 cells (from 8192 DNA pairs or PSSMs, 65536
protein pairs: ba_host.cpp batch_build), k_multi for batches that start at 128 cells (from 16384 pairs). Every pair of every
batch is compared with the oracle: score, end positions, computed cells, every CIGAR run."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from block_aligner_amd import workloads as W
from tests.gotoh import check_cigar
from tests.test_gpu_parity import NUC

pytestmark = pytest.mark.gpu

MODES = [(), ("x_drop",), ("trace",), ("trace", "x_drop")]


def mode_bits(H, mode, cigar_eq):
    m = 0
    for k in mode:
        m |= {"trace": H.TRACE, "x_drop": H.X_DROP, "local_start": H.LOCAL_START, "free_query_start_gaps": H.FREE_QUERY_START_GAPS}[k]
    if cigar_eq and "trace" in mode:
        m |= H.CIGAR_EQ
    return m


def flat_oracle_runs(ref, n):
    ln = ref["cig_len"].astype(np.int64)
    start = np.repeat(ref["cig_off"].astype(np.int64), ln)
    within = np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
    return ref["cig_ops"][start + within]


def assert_batch_equals(H, b, res, ref, pairs, matrix, gaps, mode, what, sample=97):
    assert not res["status"].any(), (what, np.nonzero(res["status"])[0][:10])
    bad = np.nonzero((res["score"] != ref["scores"]) | (res["query_idx"] != ref["query_idx"]) | (res["reference_idx"] != ref["reference_idx"]))[0]
    assert bad.size == 0, (what, bad[:10], res["score"][bad[:5]], ref["scores"][bad[:5]])
    assert int(res["cells"].sum()) == int(np.asarray(ref["cells"]).sum()), what
    if "trace" in mode:
        assert np.array_equal(res["cigar_len"], ref["cig_len"]), what
        runs, off = b.cigars(res["cigar_len"])
        assert np.array_equal(runs[: int(off[len(pairs)])], flat_oracle_runs(ref, len(pairs))), what
        if matrix is not None:   # oracle-independent: a spread of the HIP CIGARs re-walked and re-scored
            for p in range(0, len(pairs), sample):
                check_cigar(runs[int(off[p]): int(off[p + 1])], pairs.query(p), pairs.reference(p), matrix, gaps, int(res["score"][p]),
                            int(res["query_idx"][p]), int(res["reference_idx"][p]), mode, what=(what, p))


def run_and_compare(H, oracle, pairs, matrix, gaps, size, x_drop, mode, cigar_eq, what):
    b = H.BatchAligner(matrix, gaps, size, x_drop, mode_bits(H, mode, cigar_eq), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    b.run()
    res = b.results()
    ref = oracle.batch_align(matrix, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, gaps, size, x_drop, mode,
                             cigar_eq=cigar_eq and "trace" in mode, threads=16)
    assert_batch_equals(H, b, res, ref, pairs, matrix, gaps, mode, what)
    b.close()


@pytest.mark.parametrize("mode", MODES)
def test_dna_small_blocks_at_threshold(hip, oracle, mode):
    """9 k DNA pairs at 32..256 (threshold 8192): k_quad, the queue, the per-pair kernel and (with traceback) k_walk."""
    assert hip.lib().ba_dev_build() == 0
    pairs = synth.make_pairs(9000, (0, 1500), (0, 150), 40, synth.DNA, seed=811, indels=1, indel_len=(5, 60))
    run_and_compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 100, mode, True, ("dna 32..256", mode))


@pytest.mark.parametrize("mode", [(), ("trace",), ("x_drop",), ("trace", "x_drop")])
def test_protein_small_blocks_at_threshold(hip, oracle, mode):
    """70 k protein pairs (threshold 65536), BLOSUM62, block 32..256: the uc_bench shape (examples/uc_bench.rs:85-100)."""
    w = W.config4(70000, seed=91, trace="trace" in mode)
    run_and_compare(hip, oracle, w.pairs, w.matrix, w.gaps, w.size, 60 if "x_drop" in mode else 0, mode, False, ("protein 32..256", mode))


@pytest.mark.parametrize("trace", [False, True])
def test_pssm_small_blocks_at_threshold(hip, oracle, trace):
    """10 k sequence-to-PSSM alignments (threshold 8192), block 32..256 (examples/pssm_bench.rs:94-100)."""
    w = W.config5(10000, seed=17)
    mode = ("trace",) if trace else ()
    b = hip.ProfileBatchAligner(w.profiles, w.size, w.x_drop, hip.TRACE if trace else 0, w.pairs.pool, w.pairs.q_off, w.pairs.q_len)
    b.run()
    res = b.results()
    ref = oracle.batch_align_profile(w.pairs.pool, w.pairs.q_off, w.pairs.q_len, w.profiles, w.size, w.x_drop, mode, threads=16)
    assert np.array_equal(ref["cells"], res["cells"].astype(np.uint64))
    assert_batch_equals(hip, b, res, ref, w.pairs, None, None, mode, ("pssm 32..256", mode))
    b.close()


@pytest.mark.parametrize("mode", MODES)
def test_dna_128_cell_start_at_threshold(hip, oracle, mode):
    """13 k pairs of 1500..3000 bases that start at 128 cells: four pairs per wave with in-kernel solo mode (k_multi: from 12288 pairs of 3000
    residues and more per pair); 20 k shorter ones (300..1500): the per-pair kernel (round 5: k_multi's traced solo driver loses there)."""
    pairs = synth.make_pairs(13000, (1500, 3000), (60, 300), 80, synth.DNA, seed=2025, indels=1, indel_len=(10, 120))
    b = hip.BatchAligner(NUC, (-5, -1), (128, 512), 80, mode_bits(hip, mode, True), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi"
    b.close()
    run_and_compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 80, mode, True, ("dna 128..512", mode))
    pairs = synth.make_pairs(20000, (300, 1500), (20, 150), 80, synth.DNA, seed=2025, indels=1, indel_len=(10, 120))
    b = hip.BatchAligner(NUC, (-5, -1), (128, 512), 80, mode_bits(hip, mode, True), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_align"
    b.close()
    run_and_compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 80, mode, True, ("dna 128..512 short", mode))


@pytest.mark.parametrize("mode", [("trace", "x_drop", "local_start"), ("local_start",), ("trace", "free_query_start_gaps"), ("x_drop", "free_query_start_gaps")])
def test_special_modes_128_cell_start_at_threshold(hip, oracle, mode):
    """Round 5, no forcing, the release library: LOCAL_START / FREE_QUERY_START_GAPS batches that start at 128 cells take k_multi from 12288 long
    pairs like the plain modes (its special instantiations, the traceback waves' records with the zero-mask bits)."""
    assert hip.lib().ba_dev_build() == 0
    rng = np.random.default_rng(77)
    base = synth.make_pairs(14000, (1500, 3000), (60, 300), 80, synth.DNA, seed=2026, indels=1, indel_len=(10, 120))   # (k_multi: from 12288 pairs of 3000 residues and more per pair)
    lists = []
    for p in range(len(base)):   # every third pair behind unrelated heads, every third a query inside a longer reference
        q, r = np.frombuffer(base.query(p), np.uint8), np.frombuffer(base.reference(p), np.uint8)
        if p % 3 == 0:
            q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 300)), synth.DNA), q]); r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 300)), synth.DNA), r])
        elif p % 3 == 1:
            r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 500)), synth.DNA), r])
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    pairs = synth.PairSet.from_lists(lists)
    b = hip.BatchAligner(NUC, (-5, -1), (128, 512), 80, mode_bits(hip, mode, True), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi"
    b.close()
    run_and_compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 80, mode, True, ("dna special 128..512 k_multi", mode))


def test_four_pipeline_batches_in_flight_round_after_round(hip, oracle):
    """Four small-block batches launched together, 20 rounds: queue hand-offs, the launch beside k_quad and k_walk of different
    batches overlap on the device; every round of every batch must equal the oracle."""
    cases = []
    for k, (alpha, matrix, gaps, mode, xd, n) in enumerate([(synth.AMINO, S.BLOSUM62, (-11, -1), ("trace",), 0, 66000), (synth.AMINO, S.BLOSUM62, (-11, -1), (), 0, 66000),
                                                            (synth.DNA, NUC, (-5, -1), ("trace", "x_drop"), 60, 6000), (synth.DNA, NUC, (-5, -1), ("x_drop",), 60, 6000)]):
        ps = synth.make_pairs(n, (0, 600 if alpha is synth.AMINO else 1200), (0, 100), 20, alpha, seed=4000 + k, indels=1, indel_len=(5, 80))
        eq = alpha is synth.DNA
        b = hip.BatchAligner(matrix, gaps, (32, 256), xd, mode_bits(hip, mode, eq), ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len)
        ref = oracle.batch_align(matrix, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len, gaps, (32, 256), xd, mode, cigar_eq=eq and "trace" in mode, threads=16)
        cases.append((b, ps, ref, mode, matrix, gaps))
    for rnd in range(20):
        for c in cases:
            c[0].launch()
        for b, ps, ref, mode, matrix, gaps in cases:
            b.wait()
            assert_batch_equals(hip, b, b.results(), ref, ps, matrix if rnd == 0 else None, gaps, mode, ("round", rnd, mode), sample=997)
    for c in cases:
        c[0].close()


def test_release_library_ignores_the_development_switches(hip, oracle, monkeypatch):
    """The shipped library reads no environment variables: with every development switch set -- some of which skip work or change
    what is computed in the development build -- its output is bit-identical."""
    assert hip.lib().ba_dev_build() == 0
    pairs = synth.make_pairs(3000, (200, 900), (10, 90), 40, synth.DNA, seed=5150, indels=1, indel_len=(5, 60))
    mode = ("trace", "x_drop")

    def run():
        b = hip.BatchAligner(NUC, (-5, -1), (32, 256), 80, mode_bits(hip, mode, True), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        b.run()
        res = b.results()
        runs, off = b.cigars(res["cigar_len"])
        b.close()
        return res, runs

    base, base_runs = run()
    for name in ("BA_SKIP_WALK", "BA_NO_TRACEBACK", "BA_NO_FAST", "BA_NO_SPEC", "BA_NO_QUAD", "BA_FORCE_PIPE", "BA_NO_MULTI", "BA_CALLER_ORDER", "BA_HOST_PACK",
                 "BA_INLINE_TRACEBACK", "BA_FULL_TRACE_SLOTS"):
        monkeypatch.setenv(name, "1")
    monkeypatch.setenv("BA_TRACE_MARGIN_PCT", "5")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    again, again_runs = run()
    for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"):
        assert np.array_equal(base[k], again[k]), k
    assert np.array_equal(base_runs, again_runs)
    ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (32, 256), 80, mode, cigar_eq=True, threads=8)
    assert np.array_equal(ref["scores"], base["score"]) and np.array_equal(ref["cig_len"], base["cigar_len"])


def test_cigars_gathered_behind_the_launch(hip, oracle):
    """ba_batch_compact_cigars between launch and wait: ba_batch_cigars is then a plain copy of the same runs, in the caller's order
    (the device works on the pairs longest first)."""
    pairs = synth.make_pairs(3000, (100, 1200), (10, 120), 40, synth.DNA, seed=606)
    mode = ("trace", "x_drop")
    b = hip.BatchAligner(NUC, (-5, -1), (128, 512), 80, mode_bits(hip, mode, True), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    b.run()
    res0 = b.results()
    runs0, off0 = b.cigars(res0["cigar_len"])          # gathered on demand
    pinned = hip.pinned_array(runs0.size + 1000)
    for out in (None, None, pinned, pinned):
        b.launch(); b.compact_cigars(out); b.wait()
        res = b.results()
        runs, off = b.cigars(res["cigar_len"], out=out)   # gathered behind the launch: one copy / already in host memory
        assert np.array_equal(off, off0) and np.array_equal(runs, runs0)
        if out is not None:
            out[:] = 0
    ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (128, 512), 80, mode, cigar_eq=True, threads=8)
    assert np.array_equal(runs0, flat_oracle_runs(ref, len(pairs)))
    b.close()


@pytest.mark.parametrize("kind", ["dna-xdrop", "protein"])
def test_whole_wave_walks_for_every_path_of_a_pipeline_batch(hip, oracle, devlib, monkeypatch, kind):
    """k_walk gives a whole wave only to the batch's longest paths (ba_host.cpp plan_walks); here the development switches hand it every
    pair of at least 512 residues, so thousands of random paths -- with =/X runs (DNA) and without (protein), finished by k_quad and by
    the per-pair kernel -- go through walk_wave and are compared run by run."""
    monkeypatch.setenv("BA_FORCE_QUAD", "1")
    monkeypatch.setenv("BA_FORCE_PIPE", "1")
    monkeypatch.setenv("BA_WALK_WAVE_FRAC", "1000")
    monkeypatch.setenv("BA_WALK_WAVE_MAX", "1000000")
    if kind == "dna-xdrop":
        pairs = synth.make_pairs(3000, (200, 1500), (10, 150), 40, synth.DNA, seed=31, indels=1, indel_len=(5, 60))
        run_and_compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 80, ("trace", "x_drop"), True, ("wave walks", kind))
    else:
        pairs = synth.make_pairs(3000, (100, 1200), (0, 250), 0, synth.AMINO, seed=32, indels=1, indel_len=(3, 40))
        run_and_compare(hip, oracle, pairs, S.BLOSUM62, (-11, -1), (32, 256), 0, ("trace",), False, ("wave walks", kind))


def test_wait_is_bounded_on_the_host(hip, oracle):
    """The kernels of a launch wait for each other without a give-up (round 4), so the host bounds ba_batch_wait (round-4 advisor finding):
    with a limit shorter than the launch the call fails with a message instead of blocking; the launch itself is not disturbed -- waiting
    again with the default limit returns its results."""
    pairs = synth.make_pairs(20000, 3000, 300, 100, synth.DNA, seed=99, workers=1)
    b = hip.BatchAligner(NUC, (-5, -1), (128, 512), 100, mode_bits(hip, ("trace", "x_drop"), True), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    try:
        hip.lib().ba_set_wait_limit_ms(1)
        b.launch()
        with pytest.raises(RuntimeError, match="did not finish within 1 ms"):
            b.wait()
    finally:
        hip.lib().ba_set_wait_limit_ms(600000)
    ms = b.wait()
    assert ms > 1.0
    res = b.results()
    assert not res["status"].any()
    sub = pairs.subset(np.arange(0, 20000, 200))
    ref = oracle.batch_align(NUC, sub.pool, sub.q_off, sub.q_len, sub.r_off, sub.r_len, (-5, -1), (128, 512), 100, ("trace", "x_drop"), cigar_eq=True, threads=8)
    assert np.array_equal(res["score"][::200], ref["scores"])
    b.close()
