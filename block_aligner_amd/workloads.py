"""The BASELINE.json configurations as seeded synthetic workloads: inputs, scoring and block parameters in
one place, shared by bench.py, the tools and the tests. The reference's harnesses read downloaded datasets
(/root/reference/data/README.md) or draw from an un-vendored crate, so every workload here is generated.

  C2  1 kbp DNA, ~90 % identity, X-drop 100, block 32..256                   (examples/nanopore_bench.rs:73-79 parameters)
  C3  10 kbp DNA, 1000 edits, +500 bp random tails, X-drop 100, block 1 %..10 % of the length, traceback
                                                                              (examples/nanopore_bench.rs:16-18,43-49)
  C4  protein pairs, BLOSUM62 (-11,-1), global, block 32..256                 (examples/uc_bench.rs:85-100); lengths lognormal,
      clipped to the Uniclust30 set's 22..8881 with a mean near its 300-330 (vis/block_aligner_accuracy_vis.ipynb:1005,1044)
  C5  sequence-to-PSSM, gap open -10 / extend -1, block 32..256, traceback    (examples/pssm_bench.rs:43-100)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import scores as S
from . import synth

AA20 = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
# algorithmic int16 operations per DP cell: 11 global, +3 X-drop arg-max bookkeeping, +6 trace flags
OPS_PER_CELL = {(): 11, ("x_drop",): 14, ("trace",): 17, ("trace", "x_drop"): 20}


@dataclass
class Workload:
    name: str
    pairs: synth.PairSet | None
    matrix: object
    gaps: tuple
    size: tuple
    x_drop: int
    mode: tuple                      # subset of ("trace", "x_drop"), sorted
    cigar_eq: bool = True
    profiles: list = field(default_factory=list)     # C5: one AAProfile per pair (pairs then holds only the queries)

    @property
    def ops_per_cell(self) -> int:
        base = tuple(m for m in self.mode if m in ("trace", "x_drop"))
        return OPS_PER_CELL[base] + (1 if "local_start" in self.mode else 0)   # (+ the max with the relative zero, scan_block.rs:1134-1136)

    def full_matrix_cells(self) -> int:
        ql = self.pairs.q_len.astype(np.int64)
        rl = np.array([p.str_len for p in self.profiles], np.int64) if self.profiles else self.pairs.r_len.astype(np.int64)
        return int(((ql + 1) * (rl + 1)).sum())


def config2(n: int = 10000, seed: int = 1234, trace: bool = False, workers: int = 1) -> Workload:
    pairs = synth.make_pairs(n, 1000, 100, 50, synth.DNA, seed=seed, workers=workers)
    return Workload("C2: %d x 1 kbp DNA, 100 edits, +50 bp tails, X-drop 100, block 32..256%s" % (n, ", traceback" if trace else ""),
                    pairs, S.NucMatrix.new_simple(2, -3), (-5, -1), (32, 256), 100, ("trace", "x_drop") if trace else ("x_drop",))


def config3(n: int = 100000, length: int = 10000, edits: int = 1000, tail: int = 500, seed: int = 1234, trace: bool = True,
            workers: int = 1, size=None) -> Workload:
    pairs = synth.make_pairs(n, length, edits, tail, synth.DNA, seed=seed, workers=workers)
    return Workload("C3: %d x %d bp DNA, %d edits, +%d bp tails, X-drop 100%s" % (n, length, edits, tail, ", traceback" if trace else ""),
                    pairs, S.NucMatrix.new_simple(2, -3), (-5, -1), size, 100, ("trace", "x_drop") if trace else ("x_drop",))


def protein_lengths(rng: np.random.Generator, n: int) -> np.ndarray:
    """Lognormal, clipped to 22..8881; median ~250, mean ~310 (the Uniclust30 pair sets' statistics)."""
    return np.clip(np.exp(rng.normal(np.log(250.0), 0.65, n)), 22, 8881).astype(np.int64)


def config4(n: int = 50000, seed: int = 77, trace: bool = False) -> Workload:
    rng = np.random.default_rng(seed)
    lens = protein_lengths(rng, n)
    ident = rng.uniform(0.3, 1.0, n)
    lists = []
    for L, idn in zip(lens, ident):
        r = synth.rand_str(rng, int(L), synth.AMINO)
        q = synth.mutate(rng, r, int((1.0 - idn) * L), synth.AMINO)
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    return Workload("C4: %d protein pairs (lognormal lengths 22..8881), BLOSUM62 (-11,-1), global, block 32..256%s" % (n, ", traceback" if trace else ""),
                    synth.PairSet.from_lists(lists), S.BLOSUM62, (-11, -1), (32, 256), 0, ("trace",) if trace else (), cigar_eq=False)


def config5(n: int = 5000, seed: int = 5) -> Workload:
    rng = np.random.default_rng(seed)
    rows = {int(c): np.array([S.BLOSUM62.get(int(c), int(a)) for a in AA20], np.int8) for c in AA20}
    profiles, qs = [], []
    for _ in range(n):
        length = int(rng.integers(50, 501))
        cons = AA20[rng.integers(0, 20, length)]
        pr = S.AAProfile(length, 256, -1)
        pr.pos_aa[1: length + 1, AA20 - 65] = np.stack([rows[int(c)] for c in cons])
        pr.pos_gap_open_C[: length + 1] = -10
        pr.pos_gap_open_R[: length + 1] = -10
        pr.pos_gap_close_C[1: length + 1] = 0
        profiles.append(pr)
        qs.append(synth.mutate(rng, cons, int(0.3 * length), AA20).astype(np.uint8).tobytes())
    pool = np.frombuffer(b"".join(qs) + b"\0" * 8, np.uint8)
    ql = np.array([len(q) for q in qs], np.uint32)
    qo = np.concatenate([[0], np.cumsum(ql[:-1])]).astype(np.uint64)
    pairs = synth.PairSet(pool, qo, ql, np.zeros(n, np.uint64), np.zeros(n, np.uint32))
    return Workload("C5: %d sequence-to-PSSM pairs (50..500 positions), gap open -10 / extend -1, block 32..256, traceback" % n,
                    pairs, None, (0, -1), (32, 256), 0, ("trace",), cigar_eq=False, profiles=profiles)


def config_local(n: int = 50000, seed: int = 31) -> Workload:
    """Seed extension with LOCAL_START (Block::<true, true, true>, scan_block.rs:825-846): 1 kbp DNA pairs whose first 100..300 bases are
    unrelated, X-drop 50, block 32..256, traceback -- the alignment starts wherever the path's score first leaves zero."""
    rng = np.random.default_rng(seed)
    base = synth.make_pairs(n, 1000, 100, 50, synth.DNA, seed=seed)
    lists = []
    for p in range(n):
        hq = synth.rand_str(rng, int(rng.integers(100, 300)), synth.DNA).astype(np.uint8).tobytes()
        hr = synth.rand_str(rng, int(rng.integers(100, 300)), synth.DNA).astype(np.uint8).tobytes()
        lists.append((hq + base.query(p), hr + base.reference(p)))
    return Workload("LOCAL_START: %d x ~1.2 kbp DNA with unrelated heads, X-drop 50, block 32..256, traceback" % n, synth.PairSet.from_lists(lists),
                    S.NucMatrix.new_simple(2, -3), (-5, -1), (32, 256), 50, ("local_start", "trace", "x_drop"))


def config_free_end(n: int = 50000, seed: int = 32) -> Workload:
    """FREE_QUERY_END_GAPS (scan_block.rs:825-846; precondition: min block size > query length): 40..120-base queries -- the start of
    a 300..1500-base reference, mutated -- global in the reference's start, free gaps after the query's end; block 128..512, traceback."""
    rng = np.random.default_rng(seed)
    lists = []
    for _ in range(n):
        r = synth.rand_str(rng, int(rng.integers(300, 1500)), synth.DNA)
        k = int(rng.integers(40, 121))
        q = synth.mutate(rng, r[:k], int(rng.integers(0, 1 + k // 8)), synth.DNA)[:127]
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    return Workload("FREE_QUERY_END_GAPS: %d x 40..120-base queries on 300..1500-base references, global start, block 128..512, traceback" % n,
                    synth.PairSet.from_lists(lists), S.NucMatrix.new_simple(2, -3), (-5, -1), (128, 512), 0, ("free_query_end_gaps", "trace"))


def mixed_reads(n: int = 20000, lo: int = 1000, hi: int = 40000, seed: int = 7) -> synth.PairSet:
    """A mixed-length read set (lengths log-uniform in lo .. hi, 10 % edits, tails of 0..500 bases): the input of
    examples/nanopore_bench_global.rs:144-171, where every pair gets its own block range percent_len(len, 1 %) ..= percent_len(len, 10 %)."""
    rng = np.random.default_rng(seed)
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(int)
    lists = []
    for L in lens:
        r = synth.rand_str(rng, int(L), synth.DNA)
        q = synth.mutate(rng, r, int(L) // 10, synth.DNA)
        t = int(rng.integers(0, 500))
        lists.append((np.concatenate([q, synth.rand_str(rng, t, synth.DNA)]).astype(np.uint8).tobytes(),
                      np.concatenate([r, synth.rand_str(rng, t, synth.DNA)]).astype(np.uint8).tobytes()))
    return synth.PairSet.from_lists(lists)


def make_batch(H, w: Workload):
    """The HIP batch object for a workload (block_aligner_amd.hip.BatchAligner / ProfileBatchAligner)."""
    mode = (H.TRACE if "trace" in w.mode else 0) | (H.X_DROP if "x_drop" in w.mode else 0)
    mode |= (H.LOCAL_START if "local_start" in w.mode else 0) | (H.FREE_QUERY_START_GAPS if "free_query_start_gaps" in w.mode else 0)
    mode |= H.FREE_QUERY_END_GAPS if "free_query_end_gaps" in w.mode else 0
    if w.profiles:
        return H.ProfileBatchAligner(w.profiles, w.size, w.x_drop, mode, w.pairs.pool, w.pairs.q_off, w.pairs.q_len)
    if "trace" in w.mode and w.cigar_eq:
        mode |= H.CIGAR_EQ
    p = w.pairs
    return H.BatchAligner(w.matrix, w.gaps, w.size, w.x_drop, mode, p.pool, p.q_off, p.q_len, p.r_off, p.r_len)
