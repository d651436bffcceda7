"""Writes tests/golden/crate_golden_input.tsv: the cases rust/examples/dump_golden.rs runs through the REAL block-aligner crate (its CPU
backend) to produce tests/golden/crate_golden.tsv -- golden vectors from the reference itself, for whoever has a Rust toolchain (this
image has none). tests/test_crate_golden.py compares the oracle and the HIP path with that file when it exists.

Seeded; the cases aim at what the reference's own 48 known answers do not reach: grows, checkpoint restores, shrinks and X-drop
termination after growth at (16, 64), (32, 256), (128, 1024), all mode bits; and -- round 5 -- sequence-to-profile cases (Block::align_profile over
PSSMs whose rows are BLOSUM62 rows of a consensus, examples/pssm_bench.rs:43-98, with position-specific gap_open_C / gap_close_C / gap_open_R).
Format: rust/examples/dump_golden.rs.

    python tests/golden/make_crate_golden_input.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from block_aligner_amd import synth   # noqa: E402
from block_aligner_amd import scores as S   # noqa: E402
from oracle.oracle_py import Oracle   # noqa: E402  (test infrastructure: a case on which the restated crate raises -- a traceback from an end
#                                       position in the padding, which Trace::cigar asserts on -- would abort dump_golden.rs: it is left out)

MODES = ["-", "x_drop", "trace", "trace+x_drop", "trace+local_start", "trace+x_drop+local_start", "trace+free_query_start_gaps",
         "trace+x_drop+free_query_start_gaps", "trace+free_query_end_gaps", "trace+local_start+free_query_end_gaps",
         "trace+free_query_start_gaps+free_query_end_gaps", "local_start", "free_query_end_gaps"]
SIZES = [(16, 16), (16, 64), (32, 32), (32, 256), (128, 1024), (64, 2048)]
BYTE_ALPHA = np.frombuffer(b"abcdefghij", np.uint8)


def main():
    rng = np.random.default_rng(20261003)
    oracle = Oracle("avx2")
    dropped = 0
    lines = ["# id\tkind\tmatrix\tgap_open\tgap_extend\tmin_size\tmax_size\tx_drop\tmode\tquery\treference"]
    cid = 0
    for kind, alpha, matrices, gaps in (("nuc", synth.DNA, ["NW1", "simple:2:-3"], [(-5, -1), (-2, -1)]),
                                        ("aa", synth.AMINO, ["BLOSUM62", "PAM120", "simple:3:-2"], [(-11, -1), (-6, -2)]),
                                        ("bytes", BYTE_ALPHA, ["BYTES1", "simple:2:-1"], [(-2, -1)])):
        for size in SIZES:
            for mode in MODES:
                if kind == "bytes" and "x_drop" in mode:
                    continue   # (documented as inaccurate by the reference, scores.rs:235-239)
                for rep in range(2):
                    n = int(rng.integers(0, 40)) if rep == 0 and size[1] <= 64 else int(rng.integers(60, 2200 if size[1] >= 1024 else 900))
                    r = synth.rand_str(rng, n, alpha)
                    q = synth.mutate(rng, r, int(rng.integers(0, 1 + n // 6)), alpha) if n else r
                    if n > 120 and rng.random() < 0.7:   # one long insertion or deletion: grow, checkpoint restore, shrink
                        at = int(rng.integers(20, len(q) - 20)); ln = int(rng.integers(10, min(400, 4 * size[1])))
                        q = np.concatenate([q[:at], synth.rand_str(rng, ln, alpha), q[at:]]) if rng.random() < 0.5 else np.concatenate([q[:at], q[at + ln:]])
                    if rng.random() < 0.5:   # unrelated tails: X-drop termination after growth
                        q = np.concatenate([q, synth.rand_str(rng, int(rng.integers(0, 300)), alpha)])
                        r = np.concatenate([r, synth.rand_str(rng, int(rng.integers(0, 300)), alpha)])
                    if "free_query_end_gaps" in mode:   # (the reference's precondition in this mode: min block size > query length, scan_block.rs:864-866)
                        k = int(rng.integers(0, size[0]))
                        at = int(rng.integers(0, max(1, len(q) - k)))
                        q = q[at: at + k]
                    m = matrices[int(rng.integers(0, len(matrices)))]
                    go, ge = gaps[int(rng.integers(0, len(gaps)))]
                    xd = int(rng.integers(10, 120)) if "x_drop" in mode else 0
                    qs = q.astype(np.uint8).tobytes().decode("ascii") or "-"
                    rs = r.astype(np.uint8).tobytes().decode("ascii") or "-"
                    mobj = S.static_matrix(m) if ":" not in m else {"aa": S.AAMatrix, "nuc": S.NucMatrix, "bytes": S.ByteMatrix}[kind].new_simple(*[int(v) for v in m.split(":")[1:]])
                    try:
                        oracle.align(mobj, q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes(), (go, ge), size, xd, tuple(mode.split("+")) if mode != "-" else ())
                    except RuntimeError:
                        dropped += 1
                        continue
                    lines.append("\t".join(str(x) for x in (cid, kind, m, go, ge, size[0], size[1], xd, mode, qs, rs)))
                    cid += 1
    # ---- sequence-to-profile (kind "profile"): matrix = "pssm" (row i = BLOSUM62 row of consensus[i] over the 20 standard residues), the gap_open field
    # holds one "gap_open_C/gap_close_C/gap_open_R" triple per position 0 .. len (comma-separated), gap_extend the profile's, reference = the consensus
    aa20 = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    prng = np.random.default_rng(20261004)
    n_prof = 0
    for size in [(16, 16), (16, 64), (32, 32), (32, 256), (32, 128), (128, 1024)]:
        for mode in MODES:
            n = int(prng.integers(0, 40)) if size[1] <= 16 and prng.random() < 0.3 else int(prng.integers(40, 900 if size[1] >= 1024 else 500))
            cons = aa20[prng.integers(0, 20, n)]
            ge = int(prng.integers(-2, 0))
            prof = S.AAProfile(n, size[1], ge)
            for i, c in enumerate(cons):
                for b in aa20:
                    prof.set(i + 1, int(b), S.BLOSUM62.get(int(c), int(b)))
            triples = []
            for i in range(n + 1):
                t = (int(prng.integers(-14, -3)), int(prng.integers(-4, 1)), int(prng.integers(-14, -3)))
                prof.set_gap_open_C(i, t[0]); prof.set_gap_close_C(i, t[1]); prof.set_gap_open_R(i, t[2])
                triples.append("%d/%d/%d" % t)
            q = synth.mutate(prng, cons, int(prng.uniform(0, 0.3) * n), aa20) if n else cons
            if len(q) > 80 and prng.random() < 0.8:   # a long insertion or deletion: grow, checkpoint restore, shrink
                at = int(prng.integers(20, len(q) - 20)); ln = int(prng.integers(max(8, size[0] // 2), min(250, 3 * size[1])))
                q = np.concatenate([q[:at], synth.rand_str(prng, ln, aa20), q[at:]]) if prng.random() < 0.5 else np.concatenate([q[:at], q[at + ln:]])
            if prng.random() < 0.4:
                q = np.concatenate([q, synth.rand_str(prng, int(prng.integers(0, 150)), aa20)])
            if "free_query_end_gaps" in mode:
                k = int(prng.integers(0, size[0]))
                at = int(prng.integers(0, max(1, len(q) - k)))
                q = q[at: at + k]
            xd = int(prng.integers(10, 120)) if "x_drop" in mode else 0
            qb = q.astype(np.uint8).tobytes()
            try:
                oracle.align_profile(qb, prof, size, xd, tuple(mode.split("+")) if mode != "-" else ())
            except RuntimeError:
                dropped += 1
                continue
            lines.append("\t".join(str(x) for x in (cid, "profile", "pssm", ",".join(triples), ge, size[0], size[1], xd, mode, qb.decode("ascii") or "-",
                                                     cons.astype(np.uint8).tobytes().decode("ascii") or "-")))
            cid += 1; n_prof += 1
    print("profile cases:", n_prof)
    path = os.path.join(ROOT, "tests", "golden", "crate_golden_input.tsv")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(path, cid, "cases", dropped, "dropped", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
