"""Oracle-independent pin of the HIP path (no oracle/ involved): with min = max block size > max(|q|, |r|) the block covers
the whole DP matrix, so the HIP *global* score must equal the exact affine-gap optimum of a from-scratch full-matrix DP
(tests/gotoh.py; the idea of /root/reference/examples/x_drop_accuracy.rs:108-160), and the HIP CIGAR must be a path that
re-scores to that optimum (examples/verify_trace.rs:8-31). DNA, BLOSUM62 and PSSMs, block sizes 16 .. 2048."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from tests.gotoh import check_cigar, free_query_end_score, global_score, global_score_profile, last_row_scores

pytestmark = pytest.mark.gpu
AA20 = b"ACDEFGHIKLMNPQRSTVWY"
BLOCKS = [16, 32, 64, 128, 256, 512, 1024, 2048]


def _pairs_for_block(rng, B, n, alpha):
    lists = []
    for k in range(n):
        L = int(rng.integers(max(0, B // 2 - 8), B))          # lengths just below the block size: the block is needed in full
        r = synth.rand_str(rng, L, alpha)
        q = synth.mutate(rng, r, int(rng.integers(0, L // 5 + 1)), alpha)[: B - 1]
        if k % 4 == 0:
            q = synth.rand_str(rng, int(rng.integers(0, B)), alpha)       # unrelated: the optimum is mostly gaps / mismatches
        if k % 7 == 0:
            q, r = r, q
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    return synth.PairSet.from_lists(lists)


@pytest.mark.parametrize("B", BLOCKS)
@pytest.mark.parametrize("kind", ["dna", "protein"])
def test_full_block_global_score_is_the_exact_optimum(hip, kind, B):
    rng = np.random.default_rng(100 + B)
    alpha = synth.DNA if kind == "dna" else synth.AMINO
    n = 48 if B <= 256 else (16 if B <= 1024 else 6)
    pairs = _pairs_for_block(rng, B, n, alpha)
    matrix = S.NucMatrix.new_simple(2, -3) if kind == "dna" else S.BLOSUM62
    gaps = (-5, -1) if kind == "dna" else (-11, -1)
    mode = hip.TRACE | hip.CIGAR_EQ
    b = hip.BatchAligner(matrix, gaps, (B, B), 0, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for p in range(len(pairs)):
        q, r = pairs.query(p), pairs.reference(p)
        want = global_score(q, r, matrix, gaps)
        assert int(res["score"][p]) == want, (kind, B, p, len(q), len(r), int(res["score"][p]), want)
        assert (int(res["query_idx"][p]), int(res["reference_idx"][p])) == (len(q), len(r))
        check_cigar(runs[int(off[p]): int(off[p + 1])], q, r, matrix, gaps, want, len(q), len(r), what=(kind, B, p))
    b.close()


@pytest.mark.parametrize("B", [16, 64, 256, 1024])
def test_full_block_profile_score_is_the_exact_optimum(hip, B):
    """PSSM with uniform gap costs (examples/pssm_accuracy.rs:48-67's set-up): equals the plain affine optimum over
    position-specific scores."""
    rng = np.random.default_rng(7 + B)
    cases = []
    for k in range(24 if B <= 256 else 8):
        L = int(rng.integers(1, B - 1))
        cons = bytes(AA20[i] for i in rng.integers(0, 20, L))
        p = S.AAProfile(L, B, -1)
        for i, c in enumerate(cons):
            for a in AA20:
                p.set(i + 1, a, S.BLOSUM62.get(c, a))
        go = -int(rng.integers(5, 14))
        for i in range(L + 1):
            p.set_gap_open_C(i, go); p.set_gap_close_C(i, 0); p.set_gap_open_R(i, go)
        q = synth.mutate(rng, np.frombuffer(cons, np.uint8), L // 3, np.frombuffer(AA20, np.uint8)).astype(np.uint8).tobytes()[: B - 1]
        cases.append((q, p, go))
    pool = np.frombuffer(b"".join(q for q, _, _ in cases) + b"\0" * 8, np.uint8)
    q_len = np.array([len(q) for q, _, _ in cases], np.uint32)
    q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.uint64)
    b = hip.ProfileBatchAligner([p for _, p, _ in cases], (B, B), 0, hip.TRACE, pool, q_off, q_len)
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for k, (q, p, go) in enumerate(cases):
        assert int(res["score"][k]) == global_score_profile(q, p, go), (B, k, len(q), p.str_len)
        cq, cr = check_cigar(runs[int(off[k]): int(off[k + 1])], q, b"\0" * p.str_len, None, None, int(res["score"][k]), len(q), p.str_len)
        assert (cq, cr) == (len(q), p.str_len)
    b.close()


@pytest.mark.parametrize("B", [16, 32, 64, 256, 1024])
def test_full_block_profile_with_position_specific_gaps(hip, B):
    """Non-uniform PSSMs -- every position its own gap_open_C / gap_close_C / gap_open_R (scan_block.rs:658-706) -- at block sizes that cover
    the matrix: the HIP global score equals the from-scratch DP (tests/gotoh.py global_score_profile_pos) and the HIP CIGAR is a path from
    the origin to (|q|, |p|) that re-scores to it by the same definition. No oracle involved."""
    from tests.gotoh import global_score_profile_pos, rescore_profile_cigar
    from tests.test_gotoh import pos_profile_case
    rng = np.random.default_rng(70 + B)
    ge = -1 - (B // 16) % 2   # (one gap_extend per profile batch)
    cases = [pos_profile_case(rng, B, ge) for _ in range(32 if B <= 256 else 10)]
    pool = np.frombuffer(b"".join(q for q, _ in cases) + b"\0" * 8, np.uint8)
    q_len = np.array([len(q) for q, _ in cases], np.uint32)
    q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.uint64)
    b = hip.ProfileBatchAligner([p for _, p in cases], (B, B), 0, hip.TRACE, pool, q_off, q_len)
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for k, (q, p) in enumerate(cases):
        want = global_score_profile_pos(q, p)
        assert int(res["score"][k]) == want, (B, k, len(q), p.str_len, int(res["score"][k]), want)
        assert rescore_profile_cigar(runs[int(off[k]): int(off[k + 1])], q, p) == (want, len(q), p.str_len), (B, k)
    b.close()


def test_config3_cigars_rescore(hip):
    """Config-3 shaped pairs (10 kbp, X-drop, block 128..1024 incl. the closing grow sequence): every HIP CIGAR is a valid
    path from the origin to the HIP end position that re-scores to the HIP score."""
    pairs = synth.make_pairs(64, 10000, 1000, 500, synth.DNA, seed=99)
    m = S.NucMatrix.new_simple(2, -3)
    b = hip.BatchAligner(m, (-5, -1), (128, 1024), 100, hip.TRACE | hip.X_DROP | hip.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for p in range(len(pairs)):
        check_cigar(runs[int(off[p]): int(off[p + 1])], pairs.query(p), pairs.reference(p), m, (-5, -1), int(res["score"][p]),
                    int(res["query_idx"][p]), int(res["reference_idx"][p]), ("x_drop",), what=p)
    b.close()


@pytest.mark.parametrize("B", [32, 64, 256])
@pytest.mark.parametrize("mode", [("local_start",), ("free_query_start_gaps",), ("free_query_end_gaps",)])
def test_full_block_start_and_end_modes_against_the_full_dp(hip, mode, B):
    """Round 5 (the round-4 review: FREE_QUERY_END_GAPS was left with the oracle alone): with one block covering the matrix, the HIP score
    in each start / end mode equals a from-scratch full-matrix DP of that mode -- LOCAL_START: every cell floored at 0, global end;
    FREE_QUERY_START_GAPS: row 0 free; FREE_QUERY_END_GAPS: as the reference computes it, lane |q| % 16 of a maximum over all vectors
    (tests/gotoh.py free_query_end_score, read off scan_block.rs:332-337, 1189) -- and the CIGAR of the start modes re-scores to it."""
    rng = np.random.default_rng(300 + B + len(mode[0]))
    lists = []
    for k in range(60):
        L = int(rng.integers(1, B))
        r = synth.rand_str(rng, L, synth.DNA)
        if k % 3 == 0:
            a = int(rng.integers(0, L)); b = int(rng.integers(a, L)) + 1
            q = synth.mutate(rng, r[a:b], int(rng.integers(0, (b - a) // 5 + 1)), synth.DNA)
        elif k % 3 == 1:
            q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, B // 4)), synth.DNA), synth.mutate(rng, r, int(rng.integers(0, L // 6 + 1)), synth.DNA)])
            r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, B // 4)), synth.DNA), r])
        else:
            q = synth.mutate(rng, r, int(rng.integers(0, L // 4 + 1)), synth.DNA)
        q, r = q[: B - 1], r[: B - 1]
        if len(q) == 0:
            q = synth.rand_str(rng, 1, synth.DNA)
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    pairs = synth.PairSet.from_lists(lists)
    m = S.NucMatrix.new_simple(2, -3)
    gaps = (-5, -1)
    bits = hip.TRACE | hip.CIGAR_EQ | {"local_start": hip.LOCAL_START, "free_query_start_gaps": hip.FREE_QUERY_START_GAPS, "free_query_end_gaps": hip.FREE_QUERY_END_GAPS}[mode[0]]
    b = hip.BatchAligner(m, gaps, (B, B), 0, bits, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for p in range(len(pairs)):
        q, r = pairs.query(p), pairs.reference(p)
        got = (int(res["score"][p]), int(res["query_idx"][p]), int(res["reference_idx"][p]))
        if mode[0] == "free_query_end_gaps":
            assert got[0] == free_query_end_score(q, r, m, gaps, B, m.NULL), (mode, B, p, len(q), len(r), got)
        else:
            row = last_row_scores(q, r, m, gaps, local_start=mode[0] == "local_start", free_reference_start=mode[0] == "free_query_start_gaps")
            assert got == (int(row[len(r)]), len(q), len(r)), (mode, B, p, len(q), len(r), got, int(row[len(r)]))
            check_cigar(runs[int(off[p]): int(off[p + 1])], q, r, m, gaps, got[0], got[1], got[2], mode, what=(mode, B, p))
    b.close()
