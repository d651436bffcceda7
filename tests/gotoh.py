"""Oracle-independent checks for alignment results (test infrastructure; own code, no oracle/ and no reference code).

Two things are checked WITHOUT the restatement in oracle/:

* `global_score` / `global_score_profile`: textbook full-matrix affine-gap (Gotoh) optimum, row by row in numpy. When the
  block covers the whole DP matrix (min = max block size > max(|q|, |r|)) the block aligner computes every cell, so its
  *global* score must equal this optimum exactly (the idea of /root/reference/examples/x_drop_accuracy.rs:108-160 and
  examples/accuracy.rs, which compare against a scalar DP; here it is an assertion, not a statistic).
* `check_cigar`: a CIGAR is an explicit path; re-walking it over the two sequences must consume exactly what the reported
  end position says, re-score to the reported score by the affine-gap definition (gap of length n = open + (n - 1) extend),
  and its =/X runs must agree with the bytes (the idea of /root/reference/examples/verify_trace.rs:8-31).
"""
from __future__ import annotations

import numpy as np

from block_aligner_amd.verify import _upper, check_cigar, score_table  # noqa: F401  (re-exported for the tests)

NEG = -(1 << 40)


def _gotoh_rows(sub, nq: int, nr: int, gap_open: int, gap_extend: int) -> int:
    """sub(i) -> int64[nr]: substitution scores of query position i (0-based) against reference positions 0..nr-1."""
    o, e = int(gap_open), int(gap_extend)
    j = np.arange(nr + 1, dtype=np.int64)
    H = np.empty(nr + 1, np.int64)
    H[0] = 0
    if nr:
        H[1:] = o + (j[1:] - 1) * e
    F = np.full(nr + 1, NEG, np.int64)             # best score ending in a vertical gap (consumes query only)
    for i in range(1, nq + 1):
        F = np.maximum(F + e, H + o)
        T = np.empty(nr + 1, np.int64)
        T[0] = o + (i - 1) * e
        if nr:
            T[1:] = np.maximum(H[:-1] + sub(i - 1), F[1:])
        # horizontal gaps: E[j] = max_{k<j} T[k] + o + (j-k-1) e  (opening from an E-derived cell never wins since o < e)
        pm = np.maximum.accumulate(T - j * e)
        Hn = T.copy()
        if nr:
            Hn[1:] = np.maximum(T[1:], pm[:-1] + o + (j[1:] - 1) * e)
        H = Hn
        F[0] = NEG
    return int(H[nr])


def global_score(q: bytes, r: bytes, matrix, gaps) -> int:
    """Optimal global affine-gap score of q vs r (gaps = (open, extend), open includes the first extend)."""
    tab = score_table(matrix)
    qa = np.frombuffer(q, np.uint8).astype(np.int64)
    ra = np.frombuffer(r, np.uint8).astype(np.int64)
    return _gotoh_rows(lambda i: tab[qa[i], ra], len(qa), len(ra), gaps[0], gaps[1])


def global_score_profile(q: bytes, profile, gap_open: int) -> int:
    """Optimal global score of q against a PSSM whose gap costs are uniform: every gap_open_C = gap_open_R = gap_open and
    every gap_close_C = 0 (how /root/reference/examples/pssm_accuracy.rs:48-67 sets a profile up to compare with a
    plain PSSM aligner). A gap of length n then costs gap_open + n * gap_extend."""
    qa = _upper(np.frombuffer(q, np.uint8)) - 65
    rows = profile.pos_aa[1: profile.str_len + 1].astype(np.int64)      # [position][residue]
    e = int(profile.gap_extend)
    return _gotoh_rows(lambda i: rows[:, qa[i]], len(qa), profile.str_len, int(gap_open) + e, e)
