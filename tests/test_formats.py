"""Input formats of the reference's batch callers (block_aligner_amd/formats.py): host-only parsing, plus (GPU) the
parsed batches going through the launchers with the reference's README / test vectors as content."""
import numpy as np
import pytest

from block_aligner_amd import formats as F
from block_aligner_amd import scores as S


def test_two_line_text(tmp_path):
    p = tmp_path / "pairs.txt"
    p.write_bytes(b"ttttacgt\nacgtacgt\nAAAA\nAAnA\n")     # reference line first (nanopore_bench_global.rs:30-31)
    ps = F.pairs_from_two_line_text(p)
    assert len(ps) == 2
    assert ps.query(0) == b"ACGTACGT" and ps.reference(0) == b"TTTTACGT"
    assert ps.query(1) == b"AANA" and ps.reference(1) == b"AAAA"
    (tmp_path / "odd.txt").write_bytes(b"A\nC\nG\n")
    with pytest.raises(ValueError):
        F.pairs_from_two_line_text(tmp_path / "odd.txt")


def test_m8(tmp_path):
    p = tmp_path / "x.m8"
    p.write_bytes(b"q1 r1 0.95 10 mkvla MKVLA\n\nq2\tr2\t0.5\tarnd\tARND\n")
    ps = F.pairs_from_m8([p])
    assert len(ps) == 2
    assert ps.query(0) == b"MKVLA" and ps.reference(0) == b"MKVLA"
    assert ps.query(1) == b"ARND" and ps.reference(1) == b"ARND"


def _write_pssm(path, cases):
    with open(path, "wb") as f:
        for seq, cns, rows in cases:
            f.write(b">" + seq + b"\n>" + cns + b"\n")
            f.write(b"pos cns " + b" ".join(bytes([c]) for c in F.PSSM_ORDER) + b"\n")
            for i, r in enumerate(rows):
                f.write(f"{i + 1} {chr(cns[i])} ".encode() + " ".join(str(int(v)) for v in r).encode() + b"\n")


def test_pssm(tmp_path):
    rng = np.random.default_rng(1)
    cases = []
    for n in (5, 12):
        cns = bytes(F.PSSM_ORDER[i] for i in rng.integers(0, 20, n))
        cases.append((cns[::-1], cns, rng.integers(-8, 9, (n, 20))))
    _write_pssm(tmp_path / "pairs.pssm", cases)
    profiles, queries, consensus = F.profiles_from_pssm(tmp_path / "pairs.pssm", 64, -10, -1)
    assert [len(p) for p in profiles] == [5, 12]
    assert queries[0] == cases[0][0] and consensus[1] == cases[1][1]
    p = profiles[1]
    assert p.get(3, b"D") == int(cases[1][2][2][2]) and p.get(12, b"Y") == int(cases[1][2][11][19])
    assert int(p.pos_gap_open_C[0]) == -128 and int(p.pos_gap_open_C[1]) == -10 and int(p.pos_gap_close_C[5]) == 0
    assert p.get(0, b"A") == -128 and p.get(3, b"B") == -128          # position 0 and letters outside the order stay at MIN
    pool, offs, lens = F.pool_from_sequences(queries)
    assert bytes(pool[int(offs[1]): int(offs[1]) + int(lens[1])]) == queries[1]


@pytest.mark.gpu
def test_parsed_batches_align(tmp_path, hip, oracle):
    # README.md:44-53 pair through the two-line reader (reference line first)
    p = tmp_path / "readme.txt"
    p.write_bytes(b"TTAAAAAAATTTTTTTTTTTT".lower() + b"\n" + b"TTTTTTTTAAAAAAATTTTTTTTT".lower() + b"\n")
    ps = F.pairs_from_two_line_text(p)
    b = hip.BatchAligner(S.NW1, (-2, -1), (32, 256), 0, hip.TRACE | hip.CIGAR_EQ, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len)
    b.run()
    res = b.results()
    runs, off = b.cigars(res["cigar_len"])
    assert (int(res["score"][0]), int(res["query_idx"][0]), int(res["reference_idx"][0])) == (7, 24, 21)
    assert hip.runs_to_string(runs) == "2=6I16=3D"
    # a PSSM file through the profile reader and the profile batch launcher
    rng = np.random.default_rng(3)
    cases = []
    for n in (40, 77, 130):
        cns = bytes(F.PSSM_ORDER[i] for i in rng.integers(0, 20, n))
        rows = np.array([[S.BLOSUM62.get(c, a) for a in F.PSSM_ORDER] for c in cns])
        seq = bytes(cns[k] if rng.random() > 0.2 else F.PSSM_ORDER[int(rng.integers(0, 20))] for k in range(n))
        cases.append((seq, cns, rows))
    _write_pssm(tmp_path / "pairs.pssm", cases)
    profiles, queries, _ = F.profiles_from_pssm(tmp_path / "pairs.pssm", 128, -10, -1)
    pool, offs, lens = F.pool_from_sequences(queries)
    pb = hip.ProfileBatchAligner(profiles, (32, 128), 0, hip.TRACE, pool, offs, lens)
    pb.run()
    r = pb.results()
    runs, off = pb.cigars(r["cigar_len"])
    for k in range(3):
        ref = oracle.align_profile(queries[k], profiles[k], (32, 128), 0, ("trace",))
        assert int(r["score"][k]) == ref["score"]
        assert hip.runs_to_string(runs[int(off[k]): int(off[k + 1])]) == ref["cigar"]
