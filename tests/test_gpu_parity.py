"""GPU parity proper: the HIP path vs the CPU oracle on the same seeded inputs, bit-exact (all-integer work):
score, end indices, CIGAR runs and the computed-cell count, through the batch entry points of the C ABI."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from oracle.oracle_py import cigar_runs_to_string
from tests.gotoh import check_cigar

pytestmark = pytest.mark.gpu


def compare(H, oracle, pairs, matrix, gaps, size, x_drop, mode_names, cigar_eq=True, threads=8):
    mode = 0
    for m in mode_names:
        mode |= {"trace": H.TRACE, "x_drop": H.X_DROP, "local_start": H.LOCAL_START, "free_query_start_gaps": H.FREE_QUERY_START_GAPS,
                 "free_query_end_gaps": H.FREE_QUERY_END_GAPS}[m]
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
            # oracle-independent: the HIP CIGAR is a path that ends at the HIP end position, re-scores to the HIP score by the
            # affine-gap definition and whose =/X runs agree with the bytes (examples/verify_trace.rs:8-31)
            check_cigar(got, pairs.query(p), pairs.reference(p), matrix, gaps, int(res["score"][p]), int(res["query_idx"][p]),
                        int(res["reference_idx"][p]), mode_names, what=("pair", p, mode_names, size))
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


def test_traceback_consumer_path(hip, oracle, devlib, monkeypatch):
    """Large TRACE batches hand finished trace stacks to dedicated traceback workgroups inside the launch
    (agent-scope release/acquire queue, slot reuse). Force that path on a batch small enough for the oracle and big
    enough that every fill wave recycles its trace slots several times."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_SLOTS_PER_WAVE", "2")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    pairs = synth.make_pairs(6000, (200, 1500), (10, 150), 40, synth.DNA, seed=77, indels=1, indel_len=(10, 80))
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 256), 60, ("trace", "x_drop"))
    compare(hip, oracle, pairs, NUC, (-5, -1), (32, 128), 0, ("trace",))
    # the traceback lanes also implement the early stops of LOCAL_START / FREE_QUERY_START_GAPS
    sub = pairs.subset(np.arange(1500))
    compare(hip, oracle, sub, NUC, (-5, -1), (32, 128), 60, ("trace", "local_start", "x_drop"))
    compare(hip, oracle, sub, NUC, (-5, -1), (32, 128), 0, ("trace", "free_query_start_gaps"))


def _substring_pairs(n, seed, qlen=(40, 120), rlen=(150, 700), edits=(0, 12)):
    """Queries that are mutated slices of a longer reference (+ unrelated flanks on the query for the local tests)."""
    rng = np.random.default_rng(seed)
    lists = []
    for _ in range(n):
        r = synth.rand_str(rng, int(rng.integers(rlen[0], rlen[1] + 1)), synth.DNA)
        ql = int(rng.integers(qlen[0], qlen[1] + 1))
        st = int(rng.integers(0, max(1, len(r) - ql)))
        q = synth.mutate(rng, r[st: st + ql], int(rng.integers(edits[0], edits[1] + 1)), synth.DNA)
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    return synth.PairSet.from_lists(lists)


@pytest.mark.parametrize("mode", [("trace", "local_start"), ("trace", "local_start", "x_drop"), ("x_drop", "local_start"),
                                  ("local_start",), ("trace", "free_query_start_gaps"), ("free_query_start_gaps", "x_drop")])
@pytest.mark.parametrize("size", [(32, 32), (32, 256), (128, 512)])
def test_local_and_free_start_modes(hip, oracle, mode, size):
    """LOCAL_START / FREE_QUERY_START_GAPS (scan_block.rs:1130-1136, 1597-1611) on random related pairs."""
    pairs = synth.make_pairs(150, (50, 900), (0, 90), 30, synth.DNA, seed=900 + size[1], indels=1, indel_len=(10, 60))
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 40, mode)
    compare(hip, oracle, _substring_pairs(100, 17 + size[0]), S.NW1, (-2, -1), size, 30, mode)


@pytest.mark.parametrize("mode", [("free_query_end_gaps",), ("trace", "free_query_end_gaps"),
                                  ("trace", "free_query_end_gaps", "free_query_start_gaps"), ("free_query_end_gaps", "local_start")])
@pytest.mark.parametrize("size", [(128, 128), (128, 512), (256, 1024)])
def test_free_query_end_gaps(hip, oracle, mode, size):
    """FREE_QUERY_END_GAPS needs min block size > query length (scan_block.rs:860-862); the result is the best cell of
    the last query row, found through the reference's per-lane bookkeeping (scan_block.rs:333-368, 1189-1201)."""
    pairs = _substring_pairs(200, 5 + size[1], qlen=(1, 110), rlen=(100, 900), edits=(0, 15))
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 0, mode)
    compare(hip, oracle, pairs, S.NW1, (-2, -1), size, 0, mode)


AA20 = b"ACDEFGHIKLMNPQRSTVWY"


def _pssm_case(rng, length, block_max):
    """examples/pssm_bench.rs:43-98 shaped: PSSM rows = BLOSUM62 rows of a random consensus, per-position gap costs."""
    cons = bytes(AA20[i] for i in rng.integers(0, 20, length))
    p = S.AAProfile(length, block_max, -1)
    for i, c in enumerate(cons):
        for b in AA20:
            p.set(i + 1, b, S.BLOSUM62.get(c, b))
    for i in range(length + 1):
        p.set_gap_open_C(i, int(rng.integers(-14, -7)))
        p.set_gap_open_R(i, int(rng.integers(-14, -7)))
        if i >= 1:
            p.set_gap_close_C(i, int(rng.integers(-3, 1)))
    q = synth.mutate(rng, np.frombuffer(cons, np.uint8), int(0.3 * length), np.frombuffer(AA20, np.uint8)).astype(np.uint8).tobytes()
    return q, p


@pytest.mark.parametrize("pipe", ["", "1", "quad"])
@pytest.mark.parametrize("mode", [(), ("x_drop",), ("trace",), ("trace", "x_drop")])
@pytest.mark.parametrize("size", [(16, 16), (32, 128), (32, 256), (128, 1024)])
def test_profile_batch(hip, oracle, devlib, monkeypatch, mode, size, pipe):
    """Sequence-to-profile alignment (place_block_profile_*, scan_block.rs:612-783) as a batch: every pair has its own
    PSSM and per-position gap open / close costs; compared with the oracle pair by pair. pipe: the pair-slot form large TRACE
    batches take (a trace region per pair, all paths walked by k_walk after the fill), forced on this small batch."""
    if pipe == "quad":   # the small-block pipeline (k_quad: four pairs per wave at 32 cells), forced on this small batch
        if size[0] != 32:
            pytest.skip("the small-block pipeline starts at 32 cells")
        monkeypatch.setenv("BA_FORCE_QUAD", "1")
    elif pipe:
        if "trace" not in mode:
            pytest.skip("pair-slot batches are TRACE batches")
        monkeypatch.setenv("BA_FORCE_PIPE", "1")
    rng = np.random.default_rng(41 + size[1] + len(mode))
    cases = [_pssm_case(rng, int(rng.integers(1, 500)), size[1]) for _ in range(120)]
    cases.append((b"", cases[0][1]))
    pool = np.frombuffer(b"".join(q for q, _ in cases) + b"\0" * 8, np.uint8)
    q_len = np.array([len(q) for q, _ in cases], np.uint32)
    q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.uint64)
    m = 0
    for name in mode:
        m |= {"trace": hip.TRACE, "x_drop": hip.X_DROP}[name]
    b = hip.ProfileBatchAligner([p for _, p in cases], size, 30, m, pool, q_off, q_len)
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


def test_batch_align_exp(hip, oracle):
    """Block::align_exp as kernel passes over the shrinking subset of pairs below the target (scan_block.rs:884-902):
    long indels make the small block sizes miss the optimum, so several passes run."""
    pairs = synth.make_pairs(200, (600, 1500), (20, 120), 30, synth.DNA, seed=321, indels=2, indel_len=(30, 150))
    # targets between "always reached" and "never reached"
    base = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (512, 512), 0, ())
    for frac in (0.5, 0.98, 1.5):
        target = int(np.median(base["scores"]) * frac)
        sc, qi, ri, reached = hip.batch_align_exp(NUC, (-5, -1), (32, 512), 0, target, 0, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        for p in range(len(pairs)):
            ref = oracle.align_exp(NUC, pairs.query(p), pairs.reference(p), (-5, -1), (32, 512), 0, target)
            assert (int(sc[p]), int(qi[p]), int(ri[p]), int(reached[p]) or None) == (ref["score"], ref["query_idx"], ref["reference_idx"], ref["reached"]), (p, target)


def test_trace_blocks(hip, oracle):
    """Trace::blocks() (scan_block.rs:1676-1691): the rectangle list of the HIP trace stack equals the oracle's, rectangle by
    rectangle in fill order -- on plain shift steps, with grow / checkpoint-restore / shrink (long indels), and in X-drop mode."""
    rng = np.random.default_rng(9)
    cases = []
    r = synth.rand_str(rng, 900, synth.DNA)
    cases.append((synth.mutate(rng, r, 90, synth.DNA), r, (32, 256), 0, False))
    ps = synth.make_pairs(6, (1500, 3000), (100, 300), 100, synth.DNA, seed=77, indels=3, indel_len=(20, 200))
    for p in range(len(ps)):
        cases.append((np.frombuffer(ps.query(p), np.uint8), np.frombuffer(ps.reference(p), np.uint8), [(32, 256), (64, 512), (128, 1024)][p % 3], 100, p % 2 == 0))
    grew = 0
    for q, r, size, x_drop, xd in cases:
        qb, rb = q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()
        a = hip.Block(len(qb), len(rb), size[1], trace=True, x_drop=xd)
        a.align(hip.PaddedBytes.from_bytes(qb, size[1], S.NucMatrix), hip.PaddedBytes.from_bytes(rb, size[1], S.NucMatrix), NUC, S.Gaps(-5, -1), size, x_drop)
        blocks = a.trace().blocks()
        mode = ("trace", "x_drop") if xd else ("trace",)
        ref = oracle.align(NUC, qb, rb, (-5, -1), size, x_drop, mode)
        want = oracle.align_blocks(NUC, qb, rb, (-5, -1), size, x_drop, mode)
        assert a.res().score == ref["score"]
        assert blocks == want, (size, len(blocks), len(want), [(k, x, y) for k, (x, y) in enumerate(zip(blocks, want)) if x != y][:3])
        assert sum(w * h for _, _, w, h in blocks) == ref["surviving_cells"]
        assert blocks[0][:2] == (0, 0)
        grew += any(h > size[0] or w > size[0] for _, _, w, h in blocks)
    assert grew >= 3          # the indel cases really exercise grown blocks


def test_long_pairs(hip, oracle):
    """150 kbp pairs (tens of thousands of driver steps each, trace stacks of tens of MB): 32-bit trace offsets, the
    rectangle list and the CIGAR capacity at scale; a few pairs so both the inline and the in-launch traceback run."""
    pairs = synth.make_pairs(3, 150000, 12000, 2000, synth.DNA, seed=4242, indels=4, indel_len=(50, 400))
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 1024), 100, ("trace", "x_drop"))      # (X-drop may stop at a long indel)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), (32, 2048), 0, ("trace",))             # global: always to the end
    assert (res["query_idx"] == pairs.q_len).all() and (res["cigar_len"] > 10000).all()


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations(hip, oracle, seed):
    """Seeded sweep over the whole parameter space of the path at once: matrix kind, gap costs, block range, X-drop
    threshold, mode combination and sequence shapes are all drawn at random; every draw must match the oracle exactly."""
    rng = np.random.default_rng(1000 + seed)
    kind = ["nuc", "aa", "bytes"][int(rng.integers(0, 3))]
    ext = -int(rng.integers(1, 6)); opn = ext - int(rng.integers(1, 16))
    lo = 16 << int(rng.integers(0, 4)); hi = lo << int(rng.integers(0, 4))            # 16..128 up to x8
    x_drop = int(rng.integers(0, 120))
    modes = [(), ("x_drop",), ("trace",), ("trace", "x_drop"), ("trace", "local_start"), ("local_start", "x_drop"),
             ("trace", "free_query_start_gaps"), ("trace", "free_query_end_gaps"), ("free_query_end_gaps", "local_start")]
    mode = modes[int(rng.integers(0, len(modes)))]
    if kind == "bytes":   # the byte matrix does not penalise the padding (scores.rs:235-239): X-drop / free-end maxima can
        mode = tuple(m for m in mode if m not in ("x_drop", "free_query_end_gaps"))   # land past the sequence end, where the reference's cigar() panics
    if kind == "nuc":
        matrix, alpha = S.NucMatrix.new_simple(int(rng.integers(1, 6)), -int(rng.integers(1, 8))), synth.DNA
    elif kind == "aa":
        matrix, alpha = [S.BLOSUM62, S.static_matrix("PAM120"), S.static_matrix("BLOSUM90")][int(rng.integers(0, 3))], synth.AMINO
    else:
        matrix, alpha = S.ByteMatrix.new_simple(int(rng.integers(1, 5)), -int(rng.integers(1, 5))), np.frombuffer(b"abcdxyz\x01\xfe", np.uint8)   # (not byte 0: it is the pad byte and would match the padding)
    fqe = "free_query_end_gaps" in mode
    lists = []
    for _ in range(40):
        rl = int(rng.integers(0, 700))
        r = synth.rand_str(rng, rl, alpha)
        if fqe:   # min block size must exceed the query length (scan_block.rs:860-862)
            ql = int(rng.integers(0, lo))
            st = int(rng.integers(0, max(1, rl - ql + 1)))
            q = synth.mutate(rng, r[st: st + ql], int(rng.integers(0, 4)), alpha)[: lo - 1]
        else:
            q = synth.mutate(rng, r, int(rng.integers(0, max(1, rl // 6 + 1))), alpha)
            if rng.random() < 0.3:
                q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 40)), alpha), q])
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    pairs = synth.PairSet.from_lists(lists)
    compare(hip, oracle, pairs, matrix, (opn, ext), (lo, hi), x_drop, mode, cigar_eq=bool(rng.integers(0, 2)))


def test_batch_reload(hip, oracle):
    """ba_batch_reload: new pairs through an existing batch's device buffers (the trace arena is allocated once)."""
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    first = synth.make_pairs(400, (300, 1200), (5, 100), 30, synth.DNA, seed=50)
    b = hip.BatchAligner(NUC, (-5, -1), (32, 256), 60, mode, first.pool, first.q_off, first.q_len, first.r_off, first.r_len)
    for seed, count in ((51, 400), (52, 123), (53, 1)):
        nxt = synth.make_pairs(count, (0, 1000), (0, 80), 20, synth.DNA, seed=seed)
        b.reload(nxt.pool, nxt.q_off, nxt.q_len, nxt.r_off, nxt.r_len)
        b.run()
        res = b.results()
        ref = oracle.batch_align(NUC, nxt.pool, nxt.q_off, nxt.q_len, nxt.r_off, nxt.r_len, (-5, -1), (32, 256), 60, ("trace", "x_drop"), cigar_eq=True, threads=8)
        assert not res["status"].any()
        assert np.array_equal(res["score"], ref["scores"]) and np.array_equal(res["query_idx"], ref["query_idx"]) and np.array_equal(res["cigar_len"], ref["cig_len"])
        runs, off = b.cigars(res["cigar_len"])
        for p in range(count):
            want = ref["cig_ops"][int(ref["cig_off"][p]): int(ref["cig_off"][p]) + int(ref["cig_len"][p])]
            assert np.array_equal(runs[int(off[p]): int(off[p + 1])], want), p
    too_many = synth.make_pairs(401, 100, 5, 0, synth.DNA, seed=54)
    with pytest.raises(RuntimeError):
        b.reload(too_many.pool, too_many.q_off, too_many.q_len, too_many.r_off, too_many.r_len)
    too_long = synth.make_pairs(2, 5000, 5, 0, synth.DNA, seed=55)
    with pytest.raises(RuntimeError):
        b.reload(too_long.pool, too_long.q_off, too_long.q_len, too_long.r_off, too_long.r_len)
    b.close()


def test_trace_slots_larger_than_memory_share(hip, oracle, devlib, monkeypatch):
    """330 pairs of 2 x 400 kbp at max block 2048: one full-size trace slot is 0.8 GB, so the full launch (4096 waves) would need
    3 TB; the launch shrinks to the waves whose slots fit in device memory instead of failing to allocate. (Full-size slots by the development
    switch: since round 6 batches of long pairs get slots sized by the expected stack from 256 pairs on -- the next test.)"""
    monkeypatch.setenv("BA_FULL_TRACE_SLOTS", "1")
    pairs = synth.make_pairs(330, 400000, 12000, 100, synth.DNA, seed=777, workers=8)
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    b = hip.BatchAligner(NUC, (-5, -1), (128, 2048), 200, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    info = b.info()
    assert info["grid"] < 330 and info["trace_arena_bytes"] > 100e9        # fewer waves than pairs, arena near the memory size
    b.run()
    res = b.results()
    assert not res["status"].any()
    ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (128, 2048), 200, ("trace", "x_drop"), cigar_eq=True, threads=8)
    assert np.array_equal(res["score"], ref["scores"]) and np.array_equal(res["query_idx"], ref["query_idx"]) and np.array_equal(res["cigar_len"], ref["cig_len"])
    assert int(res["cells"].sum()) == ref["cells"]
    b.close()


def test_long_pairs_get_trace_slots_by_the_expected_stack(hip, oracle):
    """Round 6: 300 pairs of 2 x 60 kbp at 128..2048 -- full-size slots would be 37 GB, slots sized by the expected stack (with the re-run of pairs that
    outgrow theirs) a fraction of it; results as the oracle's."""
    pairs = synth.make_pairs(300, 60000, 3000, 100, synth.DNA, seed=778, indels=2, indel_len=(100, 2500), workers=8)
    b = hip.BatchAligner(NUC, (-5, -1), (128, 2048), 200, hip.TRACE | hip.X_DROP | hip.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["trace_arena_bytes"] < 12e9, b.info()
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 2048), 200, ("trace", "x_drop"), threads=16)


@pytest.mark.parametrize("mode", [("trace", "local_start"), ("local_start", "x_drop"), ("trace", "free_query_start_gaps"),
                                  ("trace", "free_query_end_gaps"), ("free_query_end_gaps",)])
def test_profile_special_modes(hip, oracle, mode):
    """The special modes on the sequence-to-profile path (the const generics of Block apply to align_profile too)."""
    rng = np.random.default_rng(91 + len(mode[0]) + len(mode))
    fqe = "free_query_end_gaps" in mode
    size = (128, 256) if fqe else (32, 128)
    cases = []
    for _ in range(60):
        q, p = _pssm_case(rng, int(rng.integers(1, 300)), size[1])
        if fqe:
            q = q[: int(rng.integers(0, 100))]          # min block size must exceed the query length
        elif rng.random() < 0.5:
            q = bytes(AA20[i] for i in rng.integers(0, 20, int(rng.integers(0, 30)))) + q   # unrelated prefix
        cases.append((q, p))
    pool = np.frombuffer(b"".join(q for q, _ in cases) + b"\0" * 8, np.uint8)
    q_len = np.array([len(q) for q, _ in cases], np.uint32)
    q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.uint64)
    m = 0
    for name in mode:
        m |= {"trace": hip.TRACE, "x_drop": hip.X_DROP, "local_start": hip.LOCAL_START,
              "free_query_start_gaps": hip.FREE_QUERY_START_GAPS, "free_query_end_gaps": hip.FREE_QUERY_END_GAPS}[name]
    b = hip.ProfileBatchAligner([p for _, p in cases], size, 25, m, pool, q_off, q_len)
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"]) if "trace" in mode else (None, None)
    for k, (q, p) in enumerate(cases):
        ref = oracle.align_profile(q, p, size, 25, mode)
        got = (int(res["score"][k]), int(res["query_idx"][k]), int(res["reference_idx"][k]), int(res["cells"][k]))
        assert got == (ref["score"], ref["query_idx"], ref["reference_idx"], ref["cells"]), (k, len(q), p.str_len, got, ref)
        if "trace" in mode:
            assert hip.runs_to_string(runs[int(off[k]): int(off[k + 1])]) == ref["cigar"], k
    b.close()


@pytest.mark.parametrize("mode,size", [(("trace", "local_start", "x_drop"), (32, 2048)), (("trace", "free_query_start_gaps"), (64, 1024)),
                                       (("trace", "free_query_end_gaps"), (2048, 2048)), (("free_query_end_gaps", "local_start"), (1024, 2048))])
def test_special_modes_large_blocks(hip, oracle, mode, size):
    """The special-mode kernels keep whole columns in registers up to 2048 cells (8 and 16 chunks): long indels make the
    block grow all the way."""
    if "free_query_end_gaps" in mode:
        pairs = _substring_pairs(60, 3 + size[0], qlen=(200, size[0] - 40), rlen=(1500, 4000), edits=(0, 30))
    else:
        pairs = synth.make_pairs(60, (1500, 4000), (50, 300), 60, synth.DNA, seed=size[1], indels=3, indel_len=(50, 500))
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 80, mode)
    prot = synth.make_pairs(40, (300, 1500), (20, 200), 0, synth.AMINO, seed=5 + size[1], indels=2, indel_len=(30, 200))
    if "free_query_end_gaps" not in mode:
        compare(hip, oracle, prot, S.BLOSUM62, (-11, -1), size, 60, mode, cigar_eq=False)


def test_profile_traceback_lanes_and_exp(hip, oracle, devlib, monkeypatch):
    """Profile batches through the in-launch traceback hand-off (forced on a small batch), and align_profile_exp as a batch."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    rng = np.random.default_rng(2024)
    cases = [_pssm_case(rng, int(rng.integers(20, 400)), 256) for _ in range(3000)]
    pool = np.frombuffer(b"".join(q for q, _ in cases) + b"\0" * 8, np.uint8)
    q_len = np.array([len(q) for q, _ in cases], np.uint32)
    q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.uint64)
    profiles = [p for _, p in cases]
    b = hip.ProfileBatchAligner(profiles, (32, 256), 0, hip.TRACE, pool, q_off, q_len)
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for k in range(0, len(cases), 7):
        ref = oracle.align_profile(cases[k][0], cases[k][1], (32, 256), 0, ("trace",))
        assert (int(res["score"][k]), int(res["query_idx"][k]), int(res["reference_idx"][k])) == (ref["score"], ref["query_idx"], ref["reference_idx"]), k
        assert hip.runs_to_string(runs[int(off[k]): int(off[k + 1])]) == ref["cigar"], k
    b.close()
    monkeypatch.delenv("BA_FORCE_TB"); monkeypatch.delenv("BA_WGS_PER_CU")
    # align_profile_exp over the first 300 pairs: the median score of the fixed-size run as target
    sub = slice(0, 300)
    target = int(np.median(res["score"][sub]))
    sc, qi, ri, reached = hip.batch_align_profile_exp(profiles[sub], (32, 256), 0, target, 0, pool, q_off[sub], q_len[sub])
    lib = oracle.lib
    import ctypes as C
    for k in range(300):
        # the oracle's align_profile_exp (scan_block.rs:974-992), written out with align_profile
        mn, got = 32, None
        while mn <= 256:
            r = oracle.align_profile(cases[k][0], cases[k][1], (mn, 256), 0, ())
            if r["score"] >= target:
                got = mn
                break
            mn *= 2
        assert (int(sc[k]), int(qi[k]), int(ri[k]), int(reached[k]) or None) == (r["score"], r["query_idx"], r["reference_idx"], got), k


def test_longest_first_order_is_invisible(hip, oracle, devlib, monkeypatch):
    """The library hands the pairs to the waves longest first (ba_host.cpp Packed::order); scores, end positions, cells,
    CIGAR lengths and the CIGAR runs come back in the caller's order, identical to a batch kept in the caller's order."""
    pairs = synth.make_pairs(500, (0, 3000), (0, 150), 40, synth.DNA, seed=123)
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    out = []
    for keep in (False, True):
        if keep:
            monkeypatch.setenv("BA_CALLER_ORDER", "1")
        else:
            monkeypatch.delenv("BA_CALLER_ORDER", raising=False)
        b = hip.BatchAligner(NUC, (-5, -1), (32, 256), 80, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        b.run()
        res = b.results()
        out.append((res,) + tuple(b.cigars(res["cigar_len"])))
        b.close()
    monkeypatch.delenv("BA_CALLER_ORDER", raising=False)
    for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"):
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (32, 256), 80, ("trace", "x_drop"), cigar_eq=True, threads=8)
    assert np.array_equal(out[0][0]["score"], ref["scores"]) and np.array_equal(out[0][0]["cigar_len"], ref["cig_len"])


def test_device_side_packing(hip, oracle, devlib, monkeypatch):
    """Pooled batches of 256+ pairs are padded and converted on the device (k_pack_sequences); the images must equal the
    host packer's (BA_HOST_PACK=1): same results on mixed-case input, same error for a byte outside the alphabet."""
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    for matrix, alphabet, gaps in ((NUC, synth.DNA, (-5, -1)), (S.BLOSUM62, synth.AMINO, (-11, -1))):
        pairs = synth.make_pairs(700, (0, 900), (0, 60), 25, alphabet, seed=91)
        pool = pairs.pool.copy()
        lower = np.random.default_rng(5).random(pool.size) < 0.3
        pool[lower] = pool[lower] | 0x20          # lower case: convert_char upper-cases (scores.rs:130-134, 212-216)
        out = []
        for host in ("0", "1"):
            if host == "1":
                monkeypatch.setenv("BA_HOST_PACK", "1")
            else:
                monkeypatch.delenv("BA_HOST_PACK", raising=False)
            b = hip.BatchAligner(matrix, gaps, (32, 128), 50, mode, pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
            b.run()
            res = b.results()
            runs, off = b.cigars(res["cigar_len"])
            out.append((res, runs, off))
            if host == "0":   # a reload goes the same way
                nxt = synth.make_pairs(300, (10, 500), (0, 30), 10, alphabet, seed=92)
                b.reload(nxt.pool, nxt.q_off, nxt.q_len, nxt.r_off, nxt.r_len)
                b.run()
                r2 = b.results()
                ref2 = oracle.batch_align(matrix, nxt.pool, nxt.q_off, nxt.q_len, nxt.r_off, nxt.r_len, gaps, (32, 128), 50, ("trace", "x_drop"), cigar_eq=True, threads=8)
                assert np.array_equal(r2["score"], ref2["scores"]) and np.array_equal(r2["cigar_len"], ref2["cig_len"])
            b.close()
        monkeypatch.delenv("BA_HOST_PACK", raising=False)
        (ra, runs_a, off_a), (rb, runs_b, off_b) = out
        for k in ("score", "query_idx", "reference_idx", "cigar_len", "status"):
            assert np.array_equal(ra[k], rb[k]), k
        assert np.array_equal(runs_a, runs_b) and np.array_equal(off_a, off_b)
        ref = oracle.batch_align(matrix, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, gaps, (32, 128), 50, ("trace", "x_drop"), cigar_eq=True, threads=8)
        assert np.array_equal(ra["score"], ref["scores"]) and np.array_equal(ra["cigar_len"], ref["cig_len"])
        bad = pairs.pool.copy()
        bad[int(pairs.r_off[400]) + 2] = ord("!")
        assert int(pairs.r_len[400]) > 2
        with pytest.raises(RuntimeError, match="pair 400: byte 0x21 is outside the matrix alphabet"):
            hip.BatchAligner(matrix, gaps, (32, 128), 50, mode, bad, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        # a reload that fails half way leaves the batch empty, not half updated
        b = hip.BatchAligner(matrix, gaps, (32, 128), 50, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        with pytest.raises(RuntimeError, match="pair 400"):
            b.reload(bad, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        with pytest.raises(RuntimeError, match="holds no pairs"):
            b.run()
        b.reload(pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        b.run()
        assert np.array_equal(b.results()["score"], ref["scores"])
        b.close()


def test_batch_api_errors_and_coexisting_batches(hip, oracle):
    """Argument errors come back as errors (never aborts, never a silent fallback); two TRACE batches alive at once share
    the device memory that is left."""
    good = synth.make_pairs(64, (100, 600), (0, 40), 10, synth.DNA, seed=8)
    args = (good.pool, good.q_off, good.q_len, good.r_off, good.r_len)
    with pytest.raises(RuntimeError, match="pair 3: byte 0x21 is outside the matrix alphabet"):   # (the caller's pair index)
        bad = good.pool.copy(); bad[int(good.q_off[3]) + 1] = ord("!")
        hip.BatchAligner(NUC, (-5, -1), (32, 64), 0, 0, bad, *args[1:])
    with pytest.raises(RuntimeError, match="negative"):
        hip.BatchAligner(NUC, (5, -1), (32, 64), 0, 0, *args)
    with pytest.raises(RuntimeError, match="powers of two"):
        hip.BatchAligner(NUC, (-5, -1), (48, 64), 0, 0, *args)
    hip.BatchAligner(NUC, (-5, -1), (32, 4096), 0, hip.LOCAL_START, *args).close()   # (accepted since round 3: test_special_modes_in_the_tiled_block_class)
    with pytest.raises(RuntimeError, match="smaller than 2\\^16"):
        hip.BatchAligner(NUC, (-5, -1), (32, 65536), 0, 0, *args)
    with pytest.raises(RuntimeError, match="LOCAL_START"):
        hip.BatchAligner(NUC, (-5, -1), (32, 64), 0, hip.LOCAL_START | hip.FREE_QUERY_START_GAPS, *args)
    with pytest.raises(RuntimeError):
        hip.BatchAligner(NUC, (-5, -1), (32, 64), 0, 0, good.pool, good.q_off[:0], good.q_len[:0], good.r_off[:0], good.r_len[:0])
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    a = hip.BatchAligner(NUC, (-5, -1), (32, 256), 50, mode, *args)
    other = synth.make_pairs(80, (100, 900), (0, 60), 10, synth.DNA, seed=9)
    b = hip.BatchAligner(NUC, (-5, -1), (32, 256), 50, mode, other.pool, other.q_off, other.q_len, other.r_off, other.r_len)
    b.launch(); a.launch()          # both in flight on their own streams
    with pytest.raises(RuntimeError, match="in flight"):
        a.reload(*args)
    a.wait(); b.wait()
    for batch, ps in ((a, good), (b, other)):
        res = batch.results()
        ref = oracle.batch_align(NUC, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len, (-5, -1), (32, 256), 50, ("trace", "x_drop"), cigar_eq=True, threads=4)
        assert np.array_equal(res["score"], ref["scores"]) and np.array_equal(res["cigar_len"], ref["cig_len"])
    a.close(); b.close()


@pytest.mark.parametrize("margin", ["3", "40", "175"])
def test_adaptive_trace_slots_rerun_overflows(hip, oracle, devlib, monkeypatch, margin):
    """Large TRACE batches size their trace slots for the expected stack (block at its minimum size, one grow sequence);
    pairs that outgrow a slot come back with BA_ST_TRACE_OVERFLOW and are re-run with the reference's full bound inside
    the same ba_batch_run. Forced here on a small batch with long indels and artificially small margins: whatever share
    of the pairs overflows, every result is the oracle's."""
    monkeypatch.setenv("BA_ADAPTIVE_TRACE", "1")
    monkeypatch.setenv("BA_TRACE_MARGIN_PCT", margin)
    monkeypatch.setenv("BA_FORCE_TB", "1")
    pairs = synth.make_pairs(400, (800, 2500), (50, 250), 60, synth.DNA, seed=31 + int(margin), indels=3, indel_len=(30, 300))
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    b = hip.BatchAligner(NUC, (-5, -1), (32, 512), 80, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    for rnd in range(2):      # the second run re-uses the slots and re-runs the same pairs again
        b.run()
        res = b.results()
        assert not res["status"].any()
        retried = b.retried()
        assert retried > 0 if margin == "3" else retried >= 0
        ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (32, 512), 80, ("trace", "x_drop"), cigar_eq=True, threads=8)
        assert np.array_equal(res["score"], ref["scores"]) and np.array_equal(res["query_idx"], ref["query_idx"]) and np.array_equal(res["reference_idx"], ref["reference_idx"])
        assert np.array_equal(res["cigar_len"], ref["cig_len"]) and int(res["cells"].sum()) == ref["cells"]
        runs, off = b.cigars(res["cigar_len"])
        for p in range(len(pairs)):
            want = ref["cig_ops"][int(ref["cig_off"][p]): int(ref["cig_off"][p]) + int(ref["cig_len"][p])]
            assert np.array_equal(runs[int(off[p]): int(off[p + 1])], want), (margin, rnd, p)
        assert int(b.surviving_cells().sum()) > 0
    print(f"\nmargin {margin} %: {retried} of {len(pairs)} pairs re-run")
    b.close()


def test_multibatch_slices_merge_in_caller_order(hip, oracle):
    """ba_multibatch_*: the library cuts the pair list into cost-balanced contiguous slices, one batch per entry of
    `devices`, built by concurrent host threads and launched on their own streams; results and CIGAR runs come back in the
    caller's order, identical to a single batch. (A 1-GPU box runs the slices on the same device.)"""
    pairs = synth.make_pairs(900, (0, 2500), (0, 200), 30, synth.DNA, seed=808)
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    one = hip.BatchAligner(NUC, (-5, -1), (32, 256), 70, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    one.run()
    want = one.results()
    want_runs, want_off = one.cigars(want["cigar_len"])
    one.close()
    ndev = hip.device_count()
    for devices in ([0, 0], [0, 0, 0, 0, 0], list(range(ndev)) * 2):
        m = hip.MultiBatchAligner(NUC, (-5, -1), (32, 256), 70, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, devices)
        bounds = m.parts()
        assert len(bounds) == len(devices) + 1 and bounds[0] == 0 and bounds[-1] == len(pairs)
        assert [int(x) for x in bounds] == [int(x) for x in hip.shard_slices(pairs.q_len, pairs.r_len, len(devices))]
        ms = m.run()
        assert ms > 0
        got = m.results()
        for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"):
            assert np.array_equal(got[k], want[k]), (devices, k)
        runs, off = m.cigars(got["cigar_len"])
        assert np.array_equal(runs, want_runs) and np.array_equal(off, want_off)
        m.close()
    ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (32, 256), 70, ("trace", "x_drop"), cigar_eq=True, threads=8)
    assert np.array_equal(want["score"], ref["scores"]) and np.array_equal(want["cigar_len"], ref["cig_len"])
    with pytest.raises(RuntimeError, match="out of range"):
        hip.MultiBatchAligner(NUC, (-5, -1), (32, 256), 70, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, [0, 99])


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ("trace",), ("x_drop",), ()])
@pytest.mark.parametrize("size", [(128, 4096), (1024, 8192), (2048, 16384)])
def test_big_blocks(hip, oracle, mode, size):
    """Block sizes above 2048 (the reference takes any power of two below 2^16 - 1, scan_block.rs:855; percent_len returns up to
    16384, lib.rs:109-111; nanopore_bench_global runs 1 % - 10 % of < 50 kbp reads, examples/nanopore_bench_global.rs:144-183):
    borders in the L2-resident arena, rectangles filled in row tiles of 2048 cells. 60 kbp pairs with kb-scale insertions and
    deletions make the block grow through every size up to the maximum."""
    pairs = synth.make_pairs(5, (40000, 60000), (2000, 5000), 300, synth.DNA, seed=size[1] + len(mode), indels=4, indel_len=(800, 6000), workers=4)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 400, mode)
    assert res["cells"].max() > 60000 * size[0]


def test_block_ranges_into_the_tiled_class_start_in_the_2048_cell_class(hip, oracle):
    """Round 6 (ba_host.cpp batch_build, opt_class): a block range that starts at 1024 cells or below and ends above 2048 is launched in the 2048-cell
    class; pairs whose block never passes 2048 cells end there, a pair that wants to grow further is run again in the row-tiled class by
    ba_batch_wait -- the caller sees one run, and every pair equals the oracle's run with the full range."""
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    quiet = synth.make_pairs(24, (20000, 30000), (2000, 3000), 300, synth.DNA, seed=31, workers=4)
    b = hip.BatchAligner(NUC, (-5, -1), (256, 4096), 100, mode, quiet.pool, quiet.q_off, quiet.q_len, quiet.r_off, quiet.r_len)
    b.run()
    assert b.retried() == 0
    b.close()
    compare(hip, oracle, quiet, NUC, (-5, -1), (256, 4096), 100, ("trace", "x_drop"))
    growers = synth.make_pairs(6, (40000, 60000), (2000, 5000), 300, synth.DNA, seed=32, indels=4, indel_len=(2500, 6000), workers=4)
    both = synth.PairSet.from_lists([(quiet.query(p), quiet.reference(p)) for p in range(6)] + [(growers.query(p), growers.reference(p)) for p in range(len(growers))])
    b = hip.BatchAligner(NUC, (-5, -1), (128, 4096), 400, mode, both.pool, both.q_off, both.q_len, both.r_off, both.r_len)
    b.run()
    assert 0 < b.retried() <= len(growers), b.retried()
    b.close()
    for m in (("trace", "x_drop"), ("x_drop",), ("trace",)):
        compare(hip, oracle, both, NUC, (-5, -1), (128, 4096), 400, m)


def test_a_batch_that_loses_the_class_bet_runs_in_the_tiled_class_from_then_on(hip, oracle):
    """More than an eighth of the pairs grew past 2048 cells: the batch's next run is the row-tiled class's own (no first pass, nothing re-run),
    with the same results."""
    pairs = synth.make_pairs(6, (40000, 60000), (2000, 5000), 300, synth.DNA, seed=33, indels=4, indel_len=(2500, 6000), workers=4)
    b = hip.BatchAligner(NUC, (-5, -1), (128, 4096), 400, hip.TRACE | hip.X_DROP | hip.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    lds0 = b.info()["lds_bytes_per_wave"]
    b.run()
    first, n_again = b.results(), b.retried()
    runs1, off1 = b.cigars(first["cigar_len"])
    assert n_again >= 1
    b.run()
    second = b.results()
    runs2, off2 = b.cigars(second["cigar_len"])
    assert b.retried() == 0 and b.info()["lds_bytes_per_wave"] < lds0, (b.retried(), b.info(), lds0)
    for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"):
        assert np.array_equal(first[k], second[k]), k
    assert np.array_equal(runs1, runs2) and np.array_equal(off1, off2)
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 4096), 400, ("trace", "x_drop"))


@pytest.mark.parametrize("seed", range(10))
def test_class_bet_random_configurations(hip, oracle, seed):
    """Random block ranges that start at 128 .. 1024 cells and end at 4096 / 8192, random modes (the special ones too), DNA and protein, pairs with and
    without indels long enough to push the block past 2048 cells: the first pass in the 2048-cell class + the re-run of the pairs that grow equal
    the oracle's run with the full range, pair by pair."""
    rng = np.random.default_rng(9000 + seed)
    lo = 128 << int(rng.integers(0, 4)); hi = 4096 << int(rng.integers(0, 2))
    protein = seed % 3 == 2
    alpha, matrix, gaps = (synth.AMINO, S.BLOSUM62, (-11, -1)) if protein else (synth.DNA, NUC, (-int(rng.integers(4, 9)), -1))
    mode = [("trace", "x_drop"), ("trace",), ("x_drop",), (), ("trace", "local_start"), ("trace", "x_drop", "free_query_start_gaps")][int(rng.integers(0, 4 if protein else 6))]
    lists = []
    for k in range(10):
        n = int(rng.integers(6000, 20000))
        pr = synth.make_pairs(1, n, (n // 20, n // 8), int(rng.integers(0, 600)), alpha, seed=int(rng.integers(1 << 30)),
                              indels=int(rng.integers(0, 3)) if k % 2 else 0, indel_len=(1500, 5000))
        lists.append((pr.query(0), pr.reference(0)))
    pairs = synth.PairSet.from_lists(lists)
    compare(hip, oracle, pairs, matrix, gaps, (lo, hi), int(rng.integers(50, 400)), mode, cigar_eq=not protein)


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ()])
def test_big_blocks_without_the_2048_cell_first_pass(hip, oracle, devlib, monkeypatch, mode):
    """The row-tiled class on a range it no longer gets first (development switch BA_NO_OPT_CLASS): the same pairs as test_big_blocks."""
    monkeypatch.setenv("BA_NO_OPT_CLASS", "1")
    pairs = synth.make_pairs(5, (40000, 60000), (2000, 5000), 300, synth.DNA, seed=4096 + len(mode), indels=4, indel_len=(800, 6000), workers=4)
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 4096), 400, mode)


def test_percent_len_sizes_are_accepted(hip, oracle):
    """Every size block_percent_len can return is a size the batch constructor takes (lib.rs:109-111: up to 16384)."""
    assert hip.percent_len(10 ** 7, 0.1) == 16384 and hip.percent_len(60000, 0.1) == 8192
    pairs = synth.make_pairs(3, 3000, 200, 50, synth.DNA, seed=4)
    for mx in (4096, 8192, 16384, 32768):
        compare(hip, oracle, pairs, NUC, (-5, -1), (32, mx), 60, ("trace", "x_drop"))
    prot = synth.make_pairs(3, (2000, 5000), (200, 900), 0, synth.AMINO, seed=6, indels=2, indel_len=(300, 1500))
    compare(hip, oracle, prot, S.BLOSUM62, (-11, -1), (32, 4096), 0, ("trace",), cigar_eq=False)


@pytest.mark.parametrize("mode", [("x_drop",), (), ("trace", "x_drop"), ("trace",)])
def test_small_block_pipeline(hip, oracle, devlib, monkeypatch, mode):
    """Batches that start at 32 cells run in two fill passes: k_quad (four pairs per wave, one per 16-lane DPP row) starts every
    pair -- its first block as four 8-column sub-steps -- and runs its plain shift steps; the per-pair kernel then does whatever
    else a pair needs (grow, termination, matrix edge; pairs shorter than a block from scratch); pairs travel between the passes
    as PairCont records. With TRACE every pair stacks its trace words and rectangle records in its own region of the arenas
    across the passes and a last kernel (k_walk) walks all paths, one pair per lane. Forced on small batches here. DNA, protein
    and byte pairs, with growth (indels), tiny and empty sequences."""
    monkeypatch.setenv("BA_FORCE_QUAD", "1")
    dna = synth.make_pairs(700, (0, 1500), (0, 150), 40, synth.DNA, seed=61, indels=1, indel_len=(10, 120))
    for size in [(32, 32), (32, 64), (32, 256), (32, 2048)]:
        compare(hip, oracle, dna, NUC, (-5, -1), size, 60, mode)
    prot = synth.make_pairs(500, (22, 900), (0, 250), 0, synth.AMINO, seed=62)
    compare(hip, oracle, prot, S.BLOSUM62, (-11, -1), (32, 256), 40, mode)
    if "x_drop" not in mode:   # (the byte matrix is documented as inaccurate with X-drop, scores.rs:235-239)
        byt = synth.make_pairs(200, (0, 400), (0, 50), 5, np.frombuffer(b"abcdefghij\x01\xff", np.uint8), seed=63)
        compare(hip, oracle, byt, S.BYTES1, (-2, -1), (32, 128), 0, mode, cigar_eq=False)
    edge = synth.PairSet.from_lists([(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A" * 40, b"A" * 40), (b"ACGT" * 30, b"ACGT" * 30 + b"TTTT" * 20),
                                     (b"A" * 33, b"T" * 300), (b"ACGTNNNNACGT" * 5, b"ACGTACGT" * 6)] * 8)
    compare(hip, oracle, edge, S.NW1, (-2, -1), (32, 128), 20, mode)


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ("trace",)])
def test_pair_slot_batches_without_small_blocks(hip, oracle, devlib, monkeypatch, mode):
    """TRACE batches of many short pairs keep every pair's trace stack in its own region and walk all paths after the fill
    (k_walk, one pair per lane) whatever their block range; forced on small batches here."""
    monkeypatch.setenv("BA_FORCE_PIPE", "1")
    dna = synth.make_pairs(400, (0, 1200), (0, 120), 40, synth.DNA, seed=71, indels=1, indel_len=(10, 120))
    for size in [(16, 16), (64, 512), (128, 128), (128, 2048)]:
        compare(hip, oracle, dna, NUC, (-5, -1), size, 60, mode)
    prot = synth.make_pairs(400, (22, 900), (0, 250), 0, synth.AMINO, seed=72)
    compare(hip, oracle, prot, S.BLOSUM62, (-11, -1), (32, 256), 40, mode, cigar_eq=False)


@pytest.mark.parametrize("mode", [(), ("trace",)])
def test_small_block_batches_in_flight_together(hip, oracle, devlib, monkeypatch, mode):
    """Three small-block batches of global alignments launched back to back, each with its k_quad, the per-pair kernel beside it
    (which waits on k_quad's queue) and the one after it, on their own streams; then collected. Whatever order the device runs
    the kernels in, every batch's results are the oracle's -- a side launch that cannot get its k_quad going gives way."""
    monkeypatch.setenv("BA_FORCE_QUAD", "1")
    m = hip.TRACE if mode else 0
    sets = [synth.make_pairs(3000, (22, 700), (0, 200), 0, synth.AMINO, seed=900 + k) for k in range(3)]
    batches = [hip.BatchAligner(S.BLOSUM62, (-11, -1), (32, 256), 0, m, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len) for ps in sets]
    for rnd in range(2):
        for b in batches:
            b.launch()
        for b in batches:
            b.wait()
        for b, ps in zip(batches, sets):
            res = b.results()
            assert not res["status"].any()
            ref = oracle.batch_align(S.BLOSUM62, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len, (-11, -1), (32, 256), 0, mode, cigar_eq=False, threads=8)
            assert np.array_equal(res["score"], ref["scores"]) and int(res["cells"].sum()) == ref["cells"]
            if mode:
                assert np.array_equal(res["cigar_len"], ref["cig_len"])
                runs, off = b.cigars(res["cigar_len"])
                for p in range(0, len(ps), 7):
                    want = ref["cig_ops"][int(ref["cig_off"][p]): int(ref["cig_off"][p]) + int(ref["cig_len"][p])]
                    assert np.array_equal(runs[int(off[p]): int(off[p + 1])], want), (rnd, p)
    for b in batches:
        b.close()


@pytest.mark.parametrize("margin", ["3", "60"])
def test_pair_slot_regions_rerun_overflows_and_reload(hip, oracle, devlib, monkeypatch, margin):
    """The TRACE form of the small-block pipeline cuts the trace arena into one region per pair, sized for the pair's expected
    stack; a pair that outgrows its region leaves k_quad, reports BA_ST_TRACE_OVERFLOW from the per-pair kernel and is re-run
    with the reference's full bound inside the same ba_batch_run. Artificially small margins here; then the same batch object
    is reloaded with other pairs (the regions are cut again inside the existing arenas)."""
    monkeypatch.setenv("BA_FORCE_QUAD", "1")
    monkeypatch.setenv("BA_TRACE_MARGIN_PCT", margin)
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    sets = [synth.make_pairs(500 - 100 * k, (300, 2500 - 500 * k), (30, 250), 60, synth.DNA, seed=131 + k, indels=3, indel_len=(30, 300)) for k in range(2)]
    b = hip.BatchAligner(NUC, (-5, -1), (32, 512), 80, mode, sets[0].pool, sets[0].q_off, sets[0].q_len, sets[0].r_off, sets[0].r_len)
    for k, pairs in enumerate(sets):
        if k:
            b.reload(pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        b.run()
        res = b.results()
        assert not res["status"].any()
        assert b.retried() > 0 if margin == "3" else b.retried() >= 0
        ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (32, 512), 80, ("trace", "x_drop"), cigar_eq=True, threads=8)
        assert np.array_equal(res["score"], ref["scores"]) and np.array_equal(res["query_idx"], ref["query_idx"]) and np.array_equal(res["reference_idx"], ref["reference_idx"])
        assert np.array_equal(res["cigar_len"], ref["cig_len"]) and int(res["cells"].sum()) == ref["cells"]
        runs, off = b.cigars(res["cigar_len"])
        for p in range(len(pairs)):
            want = ref["cig_ops"][int(ref["cig_off"][p]): int(ref["cig_off"][p]) + int(ref["cig_len"][p])]
            assert np.array_equal(runs[int(off[p]): int(off[p + 1])], want), (margin, k, p)
        assert int(b.surviving_cells().sum()) > 0
    b.close()


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ("trace",)])
def test_speculative_grows_are_invisible(hip, oracle, devlib, monkeypatch, mode):
    """TRACE batches run grow steps without trace flags while nothing says their rectangles can be on a path, and go back to the
    chain's checkpoint -- re-running the steps since, traced -- when a later step raises the best score or a global alignment
    ends on top of them (ba_driver.hpp, "speculative grows"). Pairs with long indels grow in mid-alignment, improve afterwards
    and so take the roll-back; results, cell counts and CIGARs equal the oracle's and those of a run with BA_NO_SPEC=1."""
    pairs = synth.make_pairs(300, (1500, 4000), (100, 400), 150, synth.DNA, seed=17, indels=4, indel_len=(30, 400))
    size = (32, 1024)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 120, mode)
    monkeypatch.setenv("BA_NO_SPEC", "1")
    res2 = compare(hip, oracle, pairs, NUC, (-5, -1), size, 120, mode)
    for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len"):
        assert np.array_equal(res[k], res2[k]), k


@pytest.mark.parametrize("size", [(32, 128), (128, 512)])
def test_fill_waves_walk_when_the_traceback_waves_are_missing(hip, oracle, devlib, monkeypatch, size):
    """A launch whose traceback waves are not on the device (here: a development switch makes them leave at once) must not fail pairs
    with BA_ST_SLOT_TIMEOUT: a fill wave that finds nobody taking tracebacks walks pending ones itself (traceback_help_one), and the
    emptied fill waves walk the rest. (128, 512) takes the multi-pair kernel, (32, 128) the per-pair one."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_NO_TB_WAVES", "1")
    monkeypatch.setenv("BA_FORCE_MULTI", "1")
    monkeypatch.setenv("BA_NO_QUAD", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    monkeypatch.setenv("BA_SLOTS_PER_WAVE", "1" if size[0] == 32 else "6")
    pairs = synth.make_pairs(4000, (400, 1500), (30, 150), 60, synth.DNA, seed=77 + size[0])
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, ("trace", "x_drop"))


@pytest.mark.parametrize("mode", [("trace", "local_start"), ("trace", "free_query_start_gaps"), ("trace", "free_query_end_gaps"), ("free_query_end_gaps",),
                                  ("trace", "x_drop", "local_start"), ("trace", "free_query_end_gaps", "free_query_start_gaps")])
@pytest.mark.parametrize("size", [(128, 4096), (2048, 16384)])
def test_special_modes_in_the_tiled_block_class(hip, oracle, mode, size):
    """LOCAL_START / FREE_QUERY_START_GAPS / FREE_QUERY_END_GAPS with blocks above 2048 cells (the reference takes any power of two
    below 2^16 - 1 for every mode, scan_block.rs:855): rectangles filled in row tiles, the zero masks in the whole rectangle's layout,
    the per-lane FREE_QUERY_END_GAPS bookkeeping put together from per-column arrays after the last tile."""
    if "free_query_end_gaps" in mode:
        pairs = _substring_pairs(8, 3 + size[0], qlen=(40, size[0] - 20), rlen=(15000, 30000), edits=(0, 30))
    else:
        # (LOCAL_START keeps a zero-mask word per trace word: the reference's trace bound for a pair -- x 2 -- has to stay below the
        # library's 2^30 words per pair; these lengths also fit the earlier layout of 4 mask words per trace word)
        lens = (25000, 40000) if size[1] == 4096 else (11000, 15000)
        pairs = synth.make_pairs(5, lens, (1500, 3000), 300, synth.DNA, seed=size[1] + len(mode), indels=4, indel_len=(800, 5000), workers=4)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 400, mode)
    assert res["cells"].max() > 0
