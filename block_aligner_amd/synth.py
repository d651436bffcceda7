"""Seeded synthetic sequence pairs for the bench and the parity tests.

The reference's harnesses draw inputs from the un-vendored `simulate-seqs` crate (`rand_str` + `rand_mutate` with
`StdRng::seed_from_u64(1234)`, examples/nanopore_bench.rs:20-50), which cannot be reproduced here; this module
defines our own generator with the same *shape*: a uniform random reference, a query derived from it by `k`
edits (each uniformly a substitution, insertion or deletion at a uniform position), and an unrelated random
tail appended to both (nanopore_bench.rs:43-48). numpy PCG64 streams, seed = base_seed + chunk index.
"""
from __future__ import annotations

import numpy as np

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
AMINO = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def rand_str(rng: np.random.Generator, n: int, alphabet: np.ndarray) -> np.ndarray:
    return alphabet[rng.integers(0, len(alphabet), n)]


def mutate(rng: np.random.Generator, ref: np.ndarray, k: int, alphabet: np.ndarray) -> np.ndarray:
    """k edits at positions drawn on the original string; edits that hit the same position collapse (last wins)."""
    n = len(ref)
    if n == 0 or k == 0:
        return ref.copy()
    pos = rng.integers(0, n, k)
    kind = rng.integers(0, 3, k)           # 0 substitute, 1 insert after, 2 delete
    base = alphabet[rng.integers(0, len(alphabet), k)]
    out = ref.copy()
    keep = np.ones(n, dtype=np.int64)      # copies of position p emitted (0 = deleted, 2 = base + inserted base)
    ins = np.zeros(n, dtype=np.uint8)
    sub = kind == 0
    out[pos[sub]] = base[sub]
    dele = kind == 2
    keep[pos[dele]] = 0
    insm = kind == 1
    keep[pos[insm]] = 2
    ins[pos[insm]] = base[insm]
    idx = np.repeat(np.arange(n), keep)
    res = out[idx]
    # second copy of an index = the inserted base
    second = np.zeros(len(idx), dtype=bool)
    second[1:] = idx[1:] == idx[:-1]
    res[second] = ins[idx[second]]
    return res


class PairSet:
    """Packed pool of raw sequence bytes plus per-pair offsets/lengths (the batch API's input layout)."""

    def __init__(self, pool, q_off, q_len, r_off, r_len):
        self.pool, self.q_off, self.q_len, self.r_off, self.r_len = pool, q_off, q_len, r_off, r_len

    def __len__(self):
        return len(self.q_len)

    def query(self, p: int) -> bytes:
        return self.pool[int(self.q_off[p]): int(self.q_off[p]) + int(self.q_len[p])].tobytes()

    def reference(self, p: int) -> bytes:
        return self.pool[int(self.r_off[p]): int(self.r_off[p]) + int(self.r_len[p])].tobytes()

    def subset(self, idx) -> "PairSet":
        idx = np.asarray(idx)
        return PairSet(self.pool, self.q_off[idx], self.q_len[idx], self.r_off[idx], self.r_len[idx])

    @classmethod
    def from_lists(cls, pairs) -> "PairSet":
        """pairs: iterable of (query bytes, reference bytes)."""
        chunks, qo, ql, ro, rl, off = [], [], [], [], [], 0
        for q, r in pairs:
            qo.append(off); ql.append(len(q)); chunks.append(np.frombuffer(bytes(q), np.uint8)); off += len(q)
            ro.append(off); rl.append(len(r)); chunks.append(np.frombuffer(bytes(r), np.uint8)); off += len(r)
        pool = np.concatenate(chunks) if off else np.zeros(0, np.uint8)
        pool = np.concatenate([pool, np.zeros(8, np.uint8)])   # keeps pointers valid for empty tails
        return cls(pool, np.array(qo, np.uint64), np.array(ql, np.uint32), np.array(ro, np.uint64), np.array(rl, np.uint32))


CHUNK = 1024   # pairs per independently seeded chunk (seed = base seed + chunk index), so results do not depend on workers


def _make_chunk(args):
    n, length, k_edits, tail, alphabet, seed, indels, indel_len = args
    rng = np.random.default_rng(seed)
    seqs, lens = [], np.zeros((n, 2), np.uint32)
    for p in range(n):
        L = int(rng.integers(length[0], length[1] + 1)) if isinstance(length, tuple) else int(length)
        k = int(rng.integers(k_edits[0], k_edits[1] + 1)) if isinstance(k_edits, tuple) else int(k_edits)
        ref = rand_str(rng, L, alphabet)
        qry = mutate(rng, ref, k, alphabet)
        for _e in range(indels):
            if len(qry) < 2:
                break
            ln = int(rng.integers(indel_len[0], indel_len[1] + 1))
            pos = int(rng.integers(0, len(qry)))
            if rng.integers(0, 2):
                qry = np.concatenate([qry[:pos], rand_str(rng, ln, alphabet), qry[pos:]])
            else:
                qry = np.concatenate([qry[:pos], qry[pos + ln:]])
        ref = np.concatenate([ref, rand_str(rng, tail, alphabet)])
        qry = np.concatenate([qry, rand_str(rng, tail, alphabet)])
        lens[p] = (len(qry), len(ref))
        seqs.append(qry); seqs.append(ref)
    return (np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)), lens


def make_pairs(n: int, length, k_edits, tail: int, alphabet: np.ndarray = DNA, seed: int = 1234,
               indels: int = 0, indel_len=(20, 100), workers: int = 1) -> PairSet:
    """n pairs: reference = random(length), query = mutate(reference, k_edits) [+ `indels` long insert/delete
    events of length U[indel_len] to exercise block growth], both + random(tail).
    `length` and `k_edits` may be ints or (lo, hi) ranges drawn uniformly per pair.
    workers > 1 generates chunks in forked processes (call before any GPU initialisation)."""
    jobs = [(min(CHUNK, n - s), length, k_edits, tail, alphabet, seed + c, indels, indel_len)
            for c, s in enumerate(range(0, n, CHUNK))]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = pool.map(_make_chunk, jobs)
    else:
        parts = [_make_chunk(j) for j in jobs]
    lens = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, 2), np.uint32)
    pool_arr = np.concatenate([p[0] for p in parts] + [np.zeros(8, np.uint8)])
    flat = lens.astype(np.uint64).reshape(-1)
    offs = np.zeros(flat.size, np.uint64)
    np.cumsum(flat[:-1], out=offs[1:])
    return PairSet(pool_arr, offs[0::2].copy(), lens[:, 0].copy(), offs[1::2].copy(), lens[:, 1].copy())
