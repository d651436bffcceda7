"""CPU tests of the oracle-independent checker (tests/gotoh.py): the full-matrix DP must reproduce the reference-held
global scores (so the checker itself is pinned by the reference's own known answers), agree with the oracle whenever
the block covers the whole matrix, and the CIGAR walker must accept every oracle CIGAR in every mode."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from tests.common import kat_matrix, kat_profile, parse_cigar
from tests.gotoh import check_cigar, free_query_end_score, global_score, global_score_profile, last_row_scores

AA20 = b"ACDEFGHIKLMNPQRSTVWY"


def test_full_dp_reproduces_reference_global_kats(kats):
    """Every global known answer whose block covers the whole DP matrix (all of scan_block.rs:1909-1992, the trace KATs
    and the doc-test) is a full-matrix optimum: the independent DP must give the reference's number."""
    n = 0
    for k in kats["align"] + kats["inferred"]:
        if k["mode"] and set(k["mode"]) - {"trace"}:
            continue
        rlen = len(k["profile"][0]) if k["kind"] == "profile" else len(k["r"])
        if max(len(k["q"]), rlen) >= k["size"][0]:      # the first (min-size) block must cover the whole matrix
            continue
        if k["kind"] == "profile":
            b, ma, mi, goc, gcc, gor, ge = k["profile"]
            if gcc != 0 or goc != gor or k["set_gap_close_C"]:
                continue
            got = global_score_profile(k["q"].encode(), kat_profile(k), goc)
        else:
            got = global_score(k["q"].encode(), k["r"].encode(), kat_matrix(k), k["gaps"])
        assert got == k["expect"]["score"], (k["name"], got, k["expect"])
        n += 1
    assert n >= 20, n


def _pow2_above(n):
    b = 16
    while b <= n:
        b *= 2
    return b


@pytest.mark.parametrize("kind", ["dna", "protein"])
def test_oracle_full_block_equals_full_dp(oracle, kind):
    rng = np.random.default_rng(12)
    for it in range(60):
        alpha = synth.DNA if kind == "dna" else synth.AMINO
        L = int(rng.integers(0, [14, 30, 60, 120, 250, 500, 1000][it % 7]))
        r = synth.rand_str(rng, L, alpha)
        q = synth.mutate(rng, r, int(rng.integers(0, L // 4 + 1)), alpha)
        if it % 5 == 0:
            q = synth.rand_str(rng, int(rng.integers(0, L + 1)), alpha)
        m = S.NucMatrix.new_simple(int(rng.integers(1, 4)), -int(rng.integers(1, 5))) if kind == "dna" else S.BLOSUM62
        ge = -int(rng.integers(1, 4)); go = ge - int(rng.integers(1, 12))
        B = _pow2_above(max(len(q), len(r)))
        res = oracle.align(m, q.tobytes(), r.tobytes(), (go, ge), (B, B), 0, ("trace",), cigar_eq=True)
        assert res["score"] == global_score(q.tobytes(), r.tobytes(), m, (go, ge)), (it, len(q), len(r), B)
        check_cigar(parse_cigar(res["cigar"]), q.tobytes(), r.tobytes(), m, (go, ge), res["score"], res["query_idx"], res["reference_idx"])


SPECIAL = [("local_start",), ("free_query_start_gaps",), ("free_query_end_gaps",)]


def special_optimum(q: bytes, r: bytes, m, gaps, mode, block=None):
    """(score, the end positions at which it may be reported -- None: not modelled) of the start / end modes, from the full-matrix DP
    (tests/gotoh.py)."""
    if "free_query_end_gaps" in mode:
        # the query consumed, the rest of the reference free -- as the reference computes it: see free_query_end_score. For queries shorter
        # than 16 (one vector, what the reference's own tests use) that is the best cell of the last query row, or 0 at (0, 0).
        row = last_row_scores(q, r, m, gaps)
        if len(q) < 16 and block is None:
            best = int(row.max())
            ends = {(len(q), int(j)) for j in np.nonzero(row == best)[0]} if best >= 0 else set()
            if best <= 0:
                ends.add((0, 0))
            return max(best, 0), ends
        return free_query_end_score(q, r, m, gaps, block, m.NULL), None
    row = last_row_scores(q, r, m, gaps, local_start="local_start" in mode, free_reference_start="free_query_start_gaps" in mode)
    return int(row[len(r)]), {(len(q), len(r))}


def test_start_end_mode_dp_reproduces_reference_kats(kats):
    """The reference's own known answers for LOCAL_START / FREE_QUERY_START_GAPS / FREE_QUERY_END_GAPS (scan_block.rs:2181-2229) pin the
    independent DP's reading of those modes."""
    n = 0
    for k in kats["align"]:
        mode = tuple(x for x in k["mode"] if x != "trace")
        if mode not in SPECIAL:
            continue
        score, ends = special_optimum(k["q"].encode(), k["r"].encode(), kat_matrix(k), tuple(k["gaps"]), mode)
        assert score == k["expect"]["score"] and (k["expect"]["query_idx"], k["expect"]["reference_idx"]) in ends, (k["name"], score, ends)
        n += 1
    assert n == 5


@pytest.mark.parametrize("mode", SPECIAL)
def test_oracle_full_block_special_modes_equal_full_dp(oracle, mode):
    """Round 5 (the round-4 review: FREE_QUERY_END_GAPS was left with the oracle alone): with the block covering the whole matrix the oracle's
    score in each start / end mode is the optimum of the full-matrix DP for that mode, and its end position attains it."""
    rng = np.random.default_rng(21 + len(mode[0]))
    for it in range(60):
        L = int(rng.integers(1, [14, 30, 60, 120, 250, 500][it % 6]))
        r = synth.rand_str(rng, L, synth.DNA)
        if it % 3 == 0:     # a query inside the reference (what the free-gap modes are for)
            a = int(rng.integers(0, L)); b = int(rng.integers(a, L)) + 1
            q = synth.mutate(rng, r[a:b], int(rng.integers(0, (b - a) // 5 + 1)), synth.DNA)
        elif it % 3 == 1:   # unrelated heads on both (what LOCAL_START skips)
            q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 20)), synth.DNA), synth.mutate(rng, r, int(rng.integers(0, L // 6 + 1)), synth.DNA)])
            r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 20)), synth.DNA), r])
        else:
            q = synth.mutate(rng, r, int(rng.integers(0, L // 4 + 1)), synth.DNA)
        if len(q) == 0:
            continue
        m = S.NucMatrix.new_simple(int(rng.integers(1, 4)), -int(rng.integers(1, 5)))
        ge = -int(rng.integers(1, 4)); go = ge - int(rng.integers(1, 12))
        B = _pow2_above(max(len(q), len(r)))
        res = oracle.align(m, q.tobytes(), r.tobytes(), (go, ge), (B, B), 0, mode + ("trace",), cigar_eq=True)
        score, ends = special_optimum(q.tobytes(), r.tobytes(), m, (go, ge), mode, block=B)
        assert res["score"] == score, (mode, it, len(q), len(r), B, res["score"], score)
        assert ends is None or (res["query_idx"], res["reference_idx"]) in ends, (mode, it, res["query_idx"], res["reference_idx"], sorted(ends)[:5])


def test_oracle_profile_full_block_equals_full_dp(oracle):
    rng = np.random.default_rng(3)
    for it in range(30):
        L = int(rng.integers(1, [14, 60, 250, 500][it % 4]))
        cons = bytes(AA20[i] for i in rng.integers(0, 20, L))
        B = _pow2_above(L + 40)
        p = S.AAProfile(L, B, -1)
        for i, c in enumerate(cons):
            for b in AA20:
                p.set(i + 1, b, S.BLOSUM62.get(c, b))
        go = -int(rng.integers(5, 14))
        for i in range(L + 1):
            p.set_gap_open_C(i, go); p.set_gap_close_C(i, 0); p.set_gap_open_R(i, go)
        q = synth.mutate(rng, np.frombuffer(cons, np.uint8), L // 3, np.frombuffer(AA20, np.uint8)).astype(np.uint8).tobytes()[: L + 30]
        res = oracle.align_profile(q, p, (B, B), 0, ())
        assert res["score"] == global_score_profile(q, p, go), (it, L, len(q))


def pos_profile_case(rng, B, ge=None):
    """A PSSM with position-specific gap_open_C / gap_close_C / gap_open_R (gap_extend -1 or -2 unless given) that fits one block."""
    L = int(rng.integers(1, B - 1))
    aa = np.frombuffer(AA20, np.uint8)
    cons = aa[rng.integers(0, 20, L)]
    p = S.AAProfile(L, B, int(rng.integers(-2, 0)) if ge is None else ge)
    for i, c in enumerate(cons):
        for a in aa:
            p.set(i + 1, int(a), S.BLOSUM62.get(int(c), int(a)))
    for i in range(L + 1):
        p.set_gap_open_C(i, int(rng.integers(-14, -3))); p.set_gap_open_R(i, int(rng.integers(-14, -3))); p.set_gap_close_C(i, int(rng.integers(-4, 1)))
    q = synth.mutate(rng, cons, L // 3, aa).astype(np.uint8).tobytes()[: B - 1]
    return q, p


def test_full_block_profile_with_position_specific_gaps(oracle):
    """Non-uniform PSSM gap costs (scan_block.rs:658-706): with the block covering the matrix the oracle's global score equals the
    from-scratch DP of tests/gotoh.py global_score_profile_pos, and its CIGAR re-scores to it by the same definition."""
    from tests.gotoh import global_score_profile_pos, rescore_profile_cigar
    rng = np.random.default_rng(31)
    for B in (16, 32, 64, 256, 1024):
        for it in range(30 if B <= 256 else 8):
            q, p = pos_profile_case(rng, B)
            res = oracle.align_profile(q, p, (B, B), 0, ("trace",))
            want = global_score_profile_pos(q, p)
            assert res["score"] == want, (B, it, len(q), p.str_len, res["score"], want)
            assert rescore_profile_cigar(parse_cigar(res["cigar"]), q, p) == (want, len(q), p.str_len), (B, it)


MODES = [("trace",), ("trace", "x_drop"), ("trace", "local_start"), ("trace", "local_start", "x_drop"), ("trace", "free_query_start_gaps")]


def test_cigar_walker_accepts_oracle_cigars(oracle):
    rng = np.random.default_rng(8)
    for it in range(200):
        dna = it % 2 == 0
        alpha = synth.DNA if dna else synth.AMINO
        L = int(rng.integers(0, 700))
        r = synth.rand_str(rng, L, alpha)
        q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 30)) * (it % 3 == 0), alpha), synth.mutate(rng, r, int(rng.integers(0, L // 6 + 1)), alpha)])
        m = S.NucMatrix.new_simple(2, -3) if dna else S.BLOSUM62
        gaps = (-5, -1) if dna else (-11, -1)
        mode = MODES[it % len(MODES)]
        size = [(16, 16), (32, 128), (32, 512)][it % 3]
        eq = bool(it % 4)
        res = oracle.align(m, q.tobytes(), r.tobytes(), gaps, size, 60, mode, cigar_eq=eq)
        check_cigar(parse_cigar(res["cigar"]), q.tobytes(), r.tobytes(), m, gaps, res["score"], res["query_idx"], res["reference_idx"],
                    mode, what=(it, mode, size))


def test_cigar_walker_rejects_wrong_paths():
    m = S.NucMatrix.new_simple(2, -3)
    q, r = b"ACGTACGT", b"ACGTTCGT"
    good = parse_cigar("4=1X3=")
    check_cigar(good, q, r, m, (-5, -1), 2 * 7 - 3, 8, 8)
    with pytest.raises(AssertionError):
        check_cigar(parse_cigar("8="), q, r, m, (-5, -1), 16, 8, 8)          # an X called =
    with pytest.raises(AssertionError):
        check_cigar(good, q, r, m, (-5, -1), 12, 8, 8)                         # wrong score
    with pytest.raises(AssertionError):
        check_cigar(parse_cigar("4=1X2="), q, r, m, (-5, -1), 9, 8, 8)         # does not reach the origin
    with pytest.raises(AssertionError):
        check_cigar(parse_cigar("4=1I1X3="), q, r, m, (-5, -1), 6, 8, 8)       # consumes more than the end position
