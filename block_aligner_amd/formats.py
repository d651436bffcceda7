"""Input formats of the reference's batch callers, packed straight into the layouts the batch launchers take
(a byte pool + per-pair offsets / lengths, or a list of profiles), so real datasets drop in where bench.py uses
synthetic ones.

* two lines per pair, reference first (examples/nanopore_bench_global.rs:23-38, also nanopore_bench.rs / nanopore_accuracy.rs)
* `.m8`-style rows whose last two whitespace-separated columns are query and reference (examples/uc_bench.rs:43-60)
* `pairs.pssm`: per pair a `>sequence` line, a `>consensus` line and len + 1 PSSM rows of which the first is a header
  and the others hold two label columns + 20 scores in `ACDEFGHIKLMNPQRSTVWY` order (examples/pssm_bench.rs:43-91)

Bytes are upper-cased as the reference's readers do; alphabet checks happen in the library when the batch is built.
"""
from __future__ import annotations

import numpy as np

from . import scores as S
from .synth import PairSet

PSSM_ORDER = b"ACDEFGHIKLMNPQRSTVWY"   # examples/pssm_bench.rs:43


def pairs_from_two_line_text(path) -> PairSet:
    """examples/nanopore_bench_global.rs:23-38: line 2k = reference, line 2k+1 = query."""
    with open(path, "rb") as f:
        lines = f.read().splitlines()
    if len(lines) % 2:
        raise ValueError(f"{path}: odd number of lines ({len(lines)}); the format is two lines per pair")
    return PairSet.from_lists((lines[k + 1].upper(), lines[k].upper()) for k in range(0, len(lines), 2))


def pairs_from_m8(paths) -> PairSet:
    """examples/uc_bench.rs:43-60: the last column is the reference, the one before it the query."""
    if isinstance(paths, (str, bytes)) or hasattr(paths, "__fspath__"):
        paths = [paths]
    pairs = []
    for path in paths:
        with open(path, "rb") as f:
            for n, line in enumerate(f):
                cols = line.split()
                if not cols:
                    continue
                if len(cols) < 2:
                    raise ValueError(f"{path}:{n + 1}: fewer than two columns")
                pairs.append((cols[-2].upper(), cols[-1].upper()))
    return PairSet.from_lists(pairs)


def profiles_from_pssm(path, padding: int, gap_open: int, gap_extend: int):
    """examples/pssm_bench.rs:45-91 -> (profiles, queries, consensus sequences).

    profiles[k] is the AAProfile of pair k (scores from row i >= 1, gap open C/R = gap_open and gap close C = 0 at
    positions >= 1, position 0 left at its defaults exactly as the reference leaves it), queries[k] the sequence
    aligned to it and consensus[k] the profile's consensus string (used by the reference for the sequence-sequence
    comparison runs)."""
    profiles, queries, consensus = [], [], []
    with open(path, "rb") as f:
        while True:
            seq = f.readline()
            if not seq:
                break
            seq = seq.rstrip()
            cns = f.readline().rstrip()
            length = len(cns) - 1
            p = S.AAProfile(length, padding, gap_extend)
            for i in range(length + 1):
                row = f.readline().rstrip()
                if i == 0:
                    continue
                vals = row.split()[2:]
                if len(vals) < len(PSSM_ORDER):
                    raise ValueError(f"{path}: PSSM row {i} of pair {len(profiles)} has {len(vals)} scores, expected 20")
                for j, c in enumerate(PSSM_ORDER):
                    p.set(i, c, int(vals[j]))
                p.set_gap_open_C(i, gap_open)
                p.set_gap_close_C(i, 0)
                p.set_gap_open_R(i, gap_open)
            profiles.append(p)
            queries.append(seq[1:].upper())
            consensus.append(cns[1:].upper())
    return profiles, queries, consensus


def pool_from_sequences(seqs):
    """Concatenate byte strings into (pool, offsets, lengths) for the batch launchers."""
    lens = np.array([len(s) for s in seqs], np.uint32)
    offs = np.zeros(len(seqs), np.uint64)
    if len(seqs) > 1:
        np.cumsum(lens[:-1], out=offs[1:])
    pool = np.frombuffer(b"".join(bytes(s) for s in seqs) + b"\0" * 8, np.uint8)
    return pool, offs, lens
