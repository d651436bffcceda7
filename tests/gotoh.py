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


def last_row_scores(q: bytes, r: bytes, matrix, gaps, local_start: bool = False, free_reference_start: bool = False) -> np.ndarray:
    """H[|q|][0 .. |r|] of the full-matrix affine-gap DP, for the reference's start modes (scan_block.rs:1130-1136): local_start -- an
    alignment may start at any cell (every H floored at 0) --, free_reference_start (FREE_QUERY_START_GAPS) -- row 0 is 0 in every column:
    reference bases before the alignment are free. FREE_QUERY_END_GAPS reads its answer off this row: max over j (the query consumed, the
    rest of the reference free). Own code, textbook recurrences, no oracle/."""
    tab = score_table(matrix)
    qa = np.frombuffer(q, np.uint8).astype(np.int64)
    ra = np.frombuffer(r, np.uint8).astype(np.int64)
    nq, nr = len(qa), len(ra)
    o, e = int(gaps[0]), int(gaps[1])
    j = np.arange(nr + 1, dtype=np.int64)
    H = np.empty(nr + 1, np.int64)
    H[0] = 0
    if nr:
        H[1:] = 0 if (local_start or free_reference_start) else o + (j[1:] - 1) * e
    F = np.full(nr + 1, NEG, np.int64)
    for i in range(1, nq + 1):
        F = np.maximum(F + e, H + o)
        T = np.empty(nr + 1, np.int64)
        T[0] = 0 if local_start else o + (i - 1) * e
        if nr:
            T[1:] = np.maximum(H[:-1] + tab[qa[i - 1], ra], F[1:])
        if local_start:
            T = np.maximum(T, 0)
        pm = np.maximum.accumulate(T - j * e)
        Hn = T.copy()
        if nr:
            Hn[1:] = np.maximum(T[1:], pm[:-1] + o + (j[1:] - 1) * e)
        H = Hn
        F[0] = NEG
    return H


def free_query_end_score(q: bytes, r: bytes, matrix, gaps, block: int, pad: int) -> int:
    """What Block::<.., FREE_QUERY_END_GAPS>::align returns as its score when ONE block of `block` cells covers the whole matrix -- read off
    the reference, not off oracle/: the best score is taken from lane |q| % 16 of a 16-lane running maximum over EVERY vector of every column
    of the block (scan_block.rs:332-337 `simd_slow_extract_i16(D_max, query.len() % L)`, D_max = max over all D11 of place_block,
    scan_block.rs:1189), i.e. over all rows i = |q| mod 16 of the block -- the last query row, but also the rows 16, 32, ... above and below
    it, the padded ones included -- and it starts at 0 (cell (0, 0)). Only its POSITION bookkeeping is restricted to the last vectors
    (scan_block.rs:1191-1199). Both sequences are padded to the block with the matrix's NULL byte as PaddedBytes does."""
    tab = score_table(matrix)
    qa = np.concatenate([np.frombuffer(q, np.uint8), np.full(block - 1 - len(q), pad, np.uint8)]).astype(np.int64)
    ra = np.concatenate([np.frombuffer(r, np.uint8), np.full(block - 1 - len(r), pad, np.uint8)]).astype(np.int64)
    nq, nr = len(qa), len(ra)
    o, e = int(gaps[0]), int(gaps[1])
    j = np.arange(nr + 1, dtype=np.int64)
    H = np.empty(nr + 1, np.int64)
    H[0] = 0
    H[1:] = o + (j[1:] - 1) * e
    F = np.full(nr + 1, NEG, np.int64)
    k = len(q) % 16
    best = int(H.max()) if k == 0 else NEG
    for i in range(1, nq + 1):
        F = np.maximum(F + e, H + o)
        T = np.empty(nr + 1, np.int64)
        T[0] = o + (i - 1) * e
        T[1:] = np.maximum(H[:-1] + tab[qa[i - 1], ra], F[1:])
        pm = np.maximum.accumulate(T - j * e)
        Hn = T.copy()
        Hn[1:] = np.maximum(T[1:], pm[:-1] + o + (j[1:] - 1) * e)
        H = Hn
        F[0] = NEG
        if i % 16 == k:
            best = max(best, int(H.max()))
    return max(best, 0)


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


def global_score_profile_pos(q: bytes, profile) -> int:
    """Optimal global score of q against a PSSM with POSITION-SPECIFIC gap costs, by the definition the sequence-to-profile recurrence of
    /root/reference/src/scan_block.rs:658-706 states for its `right` orientation (vectors along the query, one profile position per column)
    -- the orientation that fills the whole matrix when one block covers it (the first step is a grow from size 0: an empty down
    rectangle, then one right rectangle, scan_block.rs:247-305):
      a gap that consumes profile positions j .. j + n - 1 (CIGAR D) costs gap_open_C[j] + n * gap_extend + gap_close_C[j + n - 1];
      a gap that consumes n query residues between profile positions j and j + 1 (CIGAR I) costs gap_open_R[j] + n * gap_extend.
    Written from that definition as a full-matrix DP, one profile column at a time; no oracle/, no reference code."""
    qa = _upper(np.frombuffer(q, np.uint8)) - 65
    nq, nr = len(qa), profile.str_len
    sc = profile.pos_aa.astype(np.int64)                                  # [position][residue]
    goC, clC, goR = (np.asarray(x, np.int64) for x in (profile.pos_gap_open_C, profile.pos_gap_close_C, profile.pos_gap_open_R))
    e = int(profile.gap_extend)
    i = np.arange(nq + 1, dtype=np.int64)
    D = np.empty(nq + 1, np.int64)
    D[0] = 0
    D[1:] = goR[0] + i[1:] * e                                            # column 0: one gap over the query's first i residues
    C = np.full(nq + 1, NEG, np.int64)                                    # best score ending in a gap over profile positions, before its close cost
    for j in range(1, nr + 1):
        C = np.maximum(C + e, D + goC[j] + e)
        T = C + clC[j]
        if nq:
            T[1:] = np.maximum(T[1:], D[:-1] + sc[j, qa])
        pm = np.maximum.accumulate(T - i * e)                              # gaps over query residues inside column j
        Dn = T.copy()
        if nq:
            Dn[1:] = np.maximum(T[1:], pm[:-1] + goR[j] + i[1:] * e)
        D = Dn
    return int(D[nq])


def rescore_profile_cigar(runs, q: bytes, profile) -> tuple[int, int, int]:
    """(score, query residues, profile positions) of the path a CIGAR spells out from the origin, by the same definition."""
    qa = _upper(np.frombuffer(q, np.uint8)) - 65
    e = int(profile.gap_extend)
    i = j = 0
    total = 0
    for x in np.asarray(runs, np.int64):
        op, n = int(x) & 15, int(x) >> 4
        if op in (1, 2, 3):
            for _ in range(n):
                total += int(profile.pos_aa[j + 1, qa[i]]); i += 1; j += 1
        elif op == 4:      # I: query residues against no profile position (between positions j and j + 1)
            total += int(profile.pos_gap_open_R[j]) + n * e; i += n
        else:              # D: profile positions j + 1 .. j + n against no residue
            total += int(profile.pos_gap_open_C[j + 1]) + n * e + int(profile.pos_gap_close_C[j + n]); j += n
    return total, i, j
