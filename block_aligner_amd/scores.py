"""Scoring matrices, gap costs and the amino-acid profile (PSSM): host-side mirror of the reference's
`scores` module (src/scores.rs), same names and argument meaning.

These objects only hold the *tables* in the reference's layouts (AAMatrix 27x32 i8, scores.rs:40-61;
NucMatrix 8x16 i8, scores.rs:142-165; ByteMatrix {match, mismatch}, scores.rs:220-232; AAProfile
scores.rs:452-505). All arithmetic on them happens in the HIP library.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "aa_matrices.json")


@dataclass(frozen=True)
class Gaps:
    """Open and extend gap costs; `open` includes the first extend (scores.rs:329-338)."""
    open: int
    extend: int


def _upper(c: int) -> int:
    return c - 32 if 97 <= c <= 122 else c


class AAMatrix:
    """Amino acid scoring matrix over 'A'..'Z' (+ the pad byte '['); scores.rs:40-135."""
    KIND = 0
    NULL = ord("A") + 26

    def __init__(self, scores: np.ndarray | None = None):
        self.scores = np.full(27 * 32, -128, dtype=np.int8) if scores is None else np.ascontiguousarray(scores, dtype=np.int8)
        assert self.scores.size == 27 * 32

    @classmethod
    def new_simple(cls, match_score: int, mismatch_score: int) -> "AAMatrix":
        m = cls()
        t = m.scores.reshape(27, 32)
        t[:26, :26] = mismatch_score
        t[np.arange(26), np.arange(26)] = match_score
        return m

    def set(self, a, b, score: int) -> None:
        a, b = _upper(_byte(a)), _upper(_byte(b))
        assert 65 <= a <= 91 and 65 <= b <= 91
        t = self.scores.reshape(27, 32)
        t[a - 65, b - 65] = score
        t[b - 65, a - 65] = score

    def get(self, a, b) -> int:
        a, b = _upper(_byte(a)), _upper(_byte(b))
        assert 65 <= a <= 91 and 65 <= b <= 91
        return int(self.scores.reshape(27, 32)[a - 65, b - 65])

    @staticmethod
    def convert_char(c: int) -> int:
        c = _upper(c)
        assert 65 <= c <= AAMatrix.NULL, "AAMatrix bytes must be in 'A'..='['"
        return c - 65

    def raw(self) -> np.ndarray:
        return self.scores


class NucMatrix:
    """Nucleotide scoring matrix (A, C, G, N, T); row (c & 7), column (c & 15); scores.rs:142-217."""
    KIND = 1
    NULL = ord("Z")

    def __init__(self, scores: np.ndarray | None = None):
        self.scores = np.full(8 * 16, -128, dtype=np.int8) if scores is None else np.ascontiguousarray(scores, dtype=np.int8)
        assert self.scores.size == 8 * 16

    @classmethod
    def new_simple(cls, match_score: int, mismatch_score: int) -> "NucMatrix":
        m = cls()
        for a in b"ATCGN":
            for b in b"ATCGN":
                m.scores[(a & 7) * 16 + (b & 15)] = match_score if a == b else mismatch_score
        return m

    def set(self, a, b, score: int) -> None:
        a, b = _upper(_byte(a)), _upper(_byte(b))
        assert 65 <= a <= 90 and 65 <= b <= 90
        self.scores[(a & 7) * 16 + (b & 15)] = score
        self.scores[(b & 7) * 16 + (a & 15)] = score

    def get(self, a, b) -> int:
        a, b = _upper(_byte(a)), _upper(_byte(b))
        return int(self.scores[(a & 7) * 16 + (b & 15)])

    @staticmethod
    def convert_char(c: int) -> int:
        c = _upper(c)
        assert 65 <= c <= NucMatrix.NULL, "NucMatrix bytes must be in 'A'..='Z'"
        return c

    def raw(self) -> np.ndarray:
        return self.scores


class ByteMatrix:
    """Arbitrary-bytes match/mismatch matrix; scores.rs:220-273."""
    KIND = 2
    NULL = 0

    def __init__(self, match_score: int = -128, mismatch_score: int = -128):
        self.match_score, self.mismatch_score = match_score, mismatch_score

    @classmethod
    def new_simple(cls, match_score: int, mismatch_score: int) -> "ByteMatrix":
        return cls(match_score, mismatch_score)

    def get(self, a, b) -> int:
        return self.match_score if _byte(a) == _byte(b) else self.mismatch_score

    @staticmethod
    def convert_char(c: int) -> int:
        return c

    def raw(self) -> np.ndarray:
        return np.array([self.match_score, self.mismatch_score], dtype=np.int8)


def _byte(x) -> int:
    if isinstance(x, (bytes, bytearray, str)):
        assert len(x) == 1
        return x[0] if not isinstance(x, str) else ord(x)
    return int(x)


def _load_statics():
    with open(_DATA) as f:
        doc = json.load(f)
    letters = [ord(c) - 65 for c in doc["letters"]]
    out = {}
    for name, tri in doc["matrices"].items():
        m = AAMatrix()
        t = m.scores.reshape(27, 32)
        for k, a in enumerate(letters):
            for l, b in enumerate(letters[: k + 1]):
                t[a, b] = tri[k][l]
                t[b, a] = tri[k][l]
        m.scores.setflags(write=False)
        out[name] = m
    return out


_STATICS = _load_statics()
BLOSUM45, BLOSUM50, BLOSUM62, BLOSUM80, BLOSUM90 = (_STATICS[n] for n in ("BLOSUM45", "BLOSUM50", "BLOSUM62", "BLOSUM80", "BLOSUM90"))
PAM100, PAM120, PAM160, PAM200, PAM250 = (_STATICS[n] for n in ("PAM100", "PAM120", "PAM160", "PAM200", "PAM250"))
NW1 = NucMatrix.new_simple(1, -1)        # scores.rs:275-277
BYTES1 = ByteMatrix.new_simple(1, -1)    # scores.rs:309-311


def static_matrix(name: str):
    if name == "NW1":
        return NW1
    if name == "BYTES1":
        return BYTES1
    return _STATICS[name]


class AAProfile:
    """Amino-acid position specific scoring matrix with per-position gap open/close costs
    (scores.rs:452-715). Position 0 is the padding column of the DP matrix; set scores from i = 1."""
    NULL = ord("A") + 26

    def __init__(self, str_len: int, block_size: int, gap_extend: int):
        self.max_len = str_len + block_size + 1
        self.curr_len = self.max_len
        self.str_len = str_len
        self.block_size = block_size
        self.gap_extend = gap_extend
        self.pos_aa = np.full((self.max_len, 32), -128, dtype=np.int8)
        self.pos_gap_open_C = np.full(self.max_len, -128, dtype=np.int8)
        self.pos_gap_close_C = np.full(self.max_len, -128, dtype=np.int8)
        self.pos_gap_open_R = np.full(self.max_len, -128, dtype=np.int8)

    new = classmethod(lambda cls, str_len, block_size, gap_extend: cls(str_len, block_size, gap_extend))

    @classmethod
    def from_bytes(cls, b: bytes, block_size: int, match_score: int, mismatch_score: int, gap_open_C: int,
                   gap_close_C: int, gap_open_R: int, gap_extend: int) -> "AAProfile":
        p = cls(len(b), block_size, gap_extend)
        for i, ch in enumerate(b):
            for c in range(65, 91):
                p.set(i + 1, c, match_score if c == ch else mismatch_score)
        for i in range(len(b) + 1):
            p.set_gap_open_C(i, gap_open_C)
            p.set_gap_close_C(i, gap_close_C)
            p.set_gap_open_R(i, gap_open_R)
        return p

    def len(self) -> int:
        return self.str_len

    __len__ = len

    def clear(self, str_len: int, block_size: int) -> None:
        curr_len = str_len + block_size + 1
        assert curr_len <= self.max_len
        self.pos_aa[:curr_len] = -128
        self.pos_gap_open_C[:curr_len] = -128
        self.pos_gap_close_C[:curr_len] = -128
        self.pos_gap_open_R[:curr_len] = -128
        self.str_len, self.curr_len, self.block_size = str_len, curr_len, block_size

    def set(self, i: int, b, score: int) -> None:
        b = _upper(_byte(b))
        assert 65 <= b <= 91
        self.pos_aa[i, b - 65] = score

    def set_all(self, order: bytes, scores, left_shift: int = 0, right_shift: int = 0) -> None:
        self._set_all(order, scores, left_shift, right_shift, rev=False)

    def set_all_rev(self, order: bytes, scores, left_shift: int = 0, right_shift: int = 0) -> None:
        self._set_all(order, scores, left_shift, right_shift, rev=True)

    def _set_all(self, order, scores, left_shift, right_shift, rev):
        assert 0 < len(order) <= 32
        o = [_upper(x) - 65 for x in bytes(order)]
        assert all(0 <= x <= 26 for x in o)
        sc = np.asarray(scores, dtype=np.int8).reshape(-1)
        assert sc.size // len(order) == self.str_len
        # i8 wrapping shl then arithmetic shr, as scores.rs:698
        v = ((sc.astype(np.int16) << left_shift).astype(np.int8) >> right_shift).astype(np.int8)
        v = v[: self.str_len * len(order)].reshape(self.str_len, len(order))
        rows = np.arange(self.str_len, 0, -1) if rev else np.arange(1, self.str_len + 1)
        for j, b in enumerate(o):
            self.pos_aa[rows, b] = v[:, j]

    def set_gap_open_C(self, i: int, gap: int) -> None:
        assert gap < 0, "Gap open cost must be negative!"
        self.pos_gap_open_C[i] = gap

    def set_gap_close_C(self, i: int, gap: int) -> None:
        self.pos_gap_close_C[i] = gap

    def set_gap_open_R(self, i: int, gap: int) -> None:
        assert gap < 0, "Gap open cost must be negative!"
        self.pos_gap_open_R[i] = gap

    def set_all_gap_open_C(self, gap: int) -> None:
        assert gap < 0, "Gap open cost must be negative!"
        self.pos_gap_open_C[: self.str_len + 1] = gap

    def set_all_gap_close_C(self, gap: int) -> None:
        self.pos_gap_close_C[: self.str_len + 1] = gap

    def set_all_gap_open_R(self, gap: int) -> None:
        assert gap < 0, "Gap open cost must be negative!"
        self.pos_gap_open_R[: self.str_len + 1] = gap

    def get(self, i: int, b) -> int:
        b = _upper(_byte(b))
        return int(self.pos_aa[i, b - 65])

    def get_gap_extend(self) -> int:
        return self.gap_extend

    @staticmethod
    def convert_char(c: int) -> int:
        return AAMatrix.convert_char(c)
