"""Helpers shared by the CPU and GPU tests."""
from __future__ import annotations

import numpy as np

from block_aligner_amd import scores as S

MODE_BITS = {"trace": 1, "x_drop": 2, "local_start": 4, "free_query_start_gaps": 8, "free_query_end_gaps": 16}


def kat_matrix(k):
    if isinstance(k["matrix"], str):
        return S.static_matrix(k["matrix"])
    cls = {"nuc": S.NucMatrix, "aa": S.AAMatrix, "bytes": S.ByteMatrix}[k["kind"]]
    return cls.new_simple(*k["matrix"][1:])


def kat_profile(k):
    b, ma, mi, goc, gcc, gor, ge = k["profile"]
    p = S.AAProfile.from_bytes(b.encode(), k["size"][1], ma, mi, goc, gcc, gor, ge)
    for i, g in k["set_gap_close_C"]:
        p.set_gap_close_C(i, g)
    return p


def mode_flags(mode) -> int:
    f = 0
    for m in mode:
        f |= MODE_BITS[m]
    return f


def check_expect(name, expect, res, cigar=None, cigar_eq=None):
    assert res["score"] == expect["score"], (name, res, expect)
    for f in ("query_idx", "reference_idx"):
        if f in expect:
            assert res[f] == expect[f], (name, f, res, expect)
    if "cigar" in expect:
        assert cigar == expect["cigar"], (name, cigar, expect)
    if "cigar_eq" in expect:
        assert cigar_eq == expect["cigar_eq"], (name, cigar_eq, expect)


def cigar_consumes(runs):
    """(query bases, reference bases) consumed by packed runs."""
    q = r = 0
    for x in runs:
        op, n = int(x) & 15, int(x) >> 4
        if op in (1, 2, 3):
            q += n; r += n
        elif op == 4:
            q += n
        elif op == 5:
            r += n
    return q, r


def rescore(runs, q: bytes, r: bytes, matrix, gaps) -> int:
    """Score of the alignment a CIGAR spells out, by the affine-gap definition (open includes the first extend)."""
    i = j = 0
    total = 0
    for x in runs:
        op, n = int(x) & 15, int(x) >> 4
        if op in (1, 2, 3):
            for _ in range(n):
                total += matrix.get(q[i], r[j]); i += 1; j += 1
        elif op == 4:
            total += gaps[0] + gaps[1] * (n - 1); i += n
        elif op == 5:
            total += gaps[0] + gaps[1] * (n - 1); j += n
    return total


def parse_cigar(text: str):
    """CIGAR text -> packed (len << 4 | op) runs."""
    import numpy as np
    runs, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            runs.append((int(num) << 4) | " M=XID".index(ch)); num = ""
    return np.array(runs, dtype=np.int64)
