"""Self-checks on alignment results that need nothing but the sequences (the role of /root/reference/examples/verify_trace.rs:8-31
for this backend): a CIGAR is an explicit path, so re-walking it over the two sequences must consume exactly what the reported
end position says, re-score to the reported score by the affine-gap definition (a gap of length n costs open + (n - 1) extend,
scores.rs:329-338), and its = / X runs must agree with the bytes. Used by the tests and by bench.py on the HIP output."""
from __future__ import annotations

import numpy as np


def _upper(b: np.ndarray) -> np.ndarray:
    b = b.astype(np.int64)
    return np.where((b >= 97) & (b <= 122), b - 32, b)


def score_table(matrix) -> np.ndarray:
    """256 x 256 table of matrix.get(a, b) over the bytes the matrix defines (others: -128, the reference's fill value)."""
    t = np.full((256, 256), -128, np.int64)
    kind = getattr(matrix, "KIND", None)
    if kind == 2:                                  # ByteMatrix: equality on raw bytes
        t[:, :] = matrix.mismatch_score
        t[np.arange(256), np.arange(256)] = matrix.match_score
        return t
    hi = 91 if kind == 0 else 90                   # 'A'..'[' (AA) / 'A'..'Z' (nucleotides)
    for a in range(65, hi + 1):
        for b in range(65, hi + 1):
            v = matrix.get(a, b)
            for aa in (a, a + 32) if a <= 90 else (a,):
                for bb in (b, b + 32) if b <= 90 else (b,):
                    t[aa, bb] = v
    return t


def check_cigar(runs, q: bytes, r: bytes, matrix, gaps, score: int, query_idx: int, reference_idx: int, mode=(), what=""):
    """Assert that packed runs (len << 4 | op; op 1 M, 2 =, 3 X, 4 I, 5 D) spell out a path that ends at
    (query_idx, reference_idx), re-scores to `score`, and whose = / X runs agree with the bytes.
    Global and X-drop alignments start at (0, 0); LOCAL_START may start anywhere; FREE_QUERY_START_GAPS starts in row 0."""
    runs = np.asarray(runs, dtype=np.int64)
    ops, lens = runs & 15, runs >> 4
    assert ((ops >= 1) & (ops <= 5)).all() and (lens > 0).all(), (what, "malformed run")
    assert (ops[1:] != ops[:-1]).all(), (what, "adjacent runs with the same op")
    is_m = ops <= 3
    cq = int(lens[is_m | (ops == 4)].sum())
    cr = int(lens[is_m | (ops == 5)].sum())
    i0, j0 = query_idx - cq, reference_idx - cr
    assert i0 >= 0 and j0 >= 0, (what, "CIGAR consumes more than the end position", cq, cr, query_idx, reference_idx)
    if "local_start" in mode:
        pass
    elif "free_query_start_gaps" in mode:
        assert i0 == 0, (what, "free-query-start path must start in row 0", i0, j0)
    else:
        assert (i0, j0) == (0, 0), (what, "path does not start at the origin", i0, j0)
    assert query_idx <= len(q) and reference_idx <= len(r), (what, "end position out of bounds")
    # per-cell expansion of the match-type runs
    rep_op = np.repeat(ops, lens)
    di = (rep_op != 5).astype(np.int64)
    dj = (rep_op != 4).astype(np.int64)
    ipos = i0 + np.cumsum(di) - di                 # position consumed by each step
    jpos = j0 + np.cumsum(dj) - dj
    mm = rep_op <= 3
    qa = np.frombuffer(q, np.uint8)[ipos[mm]].astype(np.int64)
    ra = np.frombuffer(r, np.uint8)[jpos[mm]].astype(np.int64)
    total = 0
    if matrix is not None:
        tab = score_table(matrix)
        total += int(tab[qa, ra].sum())
        gl = lens[~is_m]
        total += int((gaps[0] + gaps[1] * (gl - 1)).sum())
        # (FREE_QUERY_END_GAPS: the reference takes the score from vector lane |q| % 16 of D_max, which merges every 16th row
        # of a block taller than 16 cells -- scan_block.rs:333-339 -- so the reported score need not belong to the reported
        # end cell; reproduced bit for bit, hence not re-scored here)
        if "free_query_end_gaps" not in mode:
            assert total == score, (what, "re-scored CIGAR differs from the reported score", total, score)
    mop = rep_op[mm]
    if (mop != 1).any():
        same = (qa == ra) if getattr(matrix, "KIND", 1) == 2 else (_upper(qa) == _upper(ra))
        assert (same[mop == 2]).all() and (~same[mop == 3]).all(), (what, "=/X runs disagree with the sequences")
    return cq, cr
