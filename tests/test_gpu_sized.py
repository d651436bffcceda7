"""Every pair with its own block range (ba_sized_batch_*): a mixed-length read set aligned with percent_len(max(|q|, |r|), 1 %) ..=
percent_len(max(|q|, |r|), 10 %) per pair -- /root/reference/examples/nanopore_bench_global.rs:144-171 -- equals the oracle run pair by pair with
each pair's own range; results and CIGAR runs come back in the caller's order."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from oracle.oracle_py import cigar_runs_to_string

pytestmark = pytest.mark.gpu
NUC = S.NucMatrix.new_simple(2, -3)


def mixed_pairs(rng, n, lo, hi):
    lists = []
    for k in range(n):
        L = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        r = synth.rand_str(rng, L, synth.DNA)
        q = synth.mutate(rng, r, L // 10, synth.DNA)
        if k % 5 == 0:   # an indel that makes the block grow
            at = int(rng.integers(0, max(1, len(q) - 1))); ln = int(rng.integers(20, 200))
            q = np.concatenate([q[:at], synth.rand_str(rng, ln, synth.DNA), q[at:]])
        t = int(rng.integers(0, 300))
        lists.append((np.concatenate([q, synth.rand_str(rng, t, synth.DNA)]).astype(np.uint8).tobytes(),
                      np.concatenate([r, synth.rand_str(rng, t, synth.DNA)]).astype(np.uint8).tobytes()))
    return synth.PairSet.from_lists(lists)


@pytest.mark.parametrize("mode", [("x_drop",), ("trace", "x_drop"), ("trace",)])
def test_percent_len_ranges_per_pair(hip, oracle, mode):
    rng = np.random.default_rng(41 + len(mode))
    pairs = mixed_pairs(rng, 160, 400, 30000)
    m = 0
    for name in mode:
        m |= {"trace": hip.TRACE, "x_drop": hip.X_DROP}[name]
    if "trace" in mode:
        m |= hip.CIGAR_EQ
    xd = 80 if "x_drop" in mode else 0
    b = hip.SizedBatchAligner(NUC, (-5, -1), xd, m, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, percent=(0.01, 0.1))
    b.run()
    res = b.results()
    assert not res["status"].any()
    cls = b.classes()
    assert len(cls) >= 5 and sum(c[2] for c in cls) == len(pairs)           # 32 / 64 / 128 / 256 / 512-cell starts all occur
    runs, off = b.cigars(res["cigar_len"]) if "trace" in mode else (None, None)
    for p in range(len(pairs)):
        q, r = pairs.query(p), pairs.reference(p)
        ln = max(len(q), len(r))
        size = (hip.percent_len(ln, 0.01), hip.percent_len(ln, 0.1))
        ref = oracle.align(NUC, q, r, (-5, -1), size, xd, mode, cigar_eq=True)
        got = (int(res["score"][p]), int(res["query_idx"][p]), int(res["reference_idx"][p]), int(res["cells"][p]))
        assert got == (ref["score"], ref["query_idx"], ref["reference_idx"], ref["cells"]), (p, len(q), len(r), size, got, ref)
        if "trace" in mode:
            assert cigar_runs_to_string(runs[int(off[p]): int(off[p + 1])]) == ref["cigar"], (p, size)
    b.close()


def test_explicit_ranges_and_caller_order(hip, oracle):
    """Ranges given pair by pair (not in any order), proteins, global alignment: the bins do not disturb the caller's order."""
    rng = np.random.default_rng(5)
    lists, sizes = [], []
    for k in range(90):
        L = int(rng.integers(20, 700))
        r = synth.rand_str(rng, L, synth.AMINO); q = synth.mutate(rng, r, L // 5, synth.AMINO)
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
        sizes.append([(32, 32), (32, 256), (64, 128), (16, 64), (128, 128)][int(rng.integers(0, 5))])
    pairs = synth.PairSet.from_lists(lists)
    b = hip.SizedBatchAligner(S.BLOSUM62, (-11, -1), 0, hip.TRACE, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, sizes=np.array(sizes))
    b.run()
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for p in range(len(pairs)):
        ref = oracle.align(S.BLOSUM62, pairs.query(p), pairs.reference(p), (-11, -1), sizes[p], 0, ("trace",))
        assert (int(res["score"][p]), int(res["cells"][p])) == (ref["score"], ref["cells"]), (p, sizes[p])
        assert cigar_runs_to_string(runs[int(off[p]): int(off[p + 1])]) == ref["cigar"], p
    b.close()


def test_ranges_whose_launches_wait_inside_run_one_after_the_other(hip, oracle):
    """Three ranges that take k_multi with traceback (slot donation: a launch's idle waves wait until all of its fill waves have been counted), together larger
    than the device: launched all at once they each held a part of the device and waited for workgroups the others' waiting waves kept out (round 6: seconds
    instead of milliseconds, every wait ended by a time-out); ba_sized_batch_run launches such ranges one after the other beside the rest."""
    ranges = [(256, 2048), (256, 4096), (512, 4096), (32, 256)]
    sets = [synth.make_pairs(1100 if k < 3 else 3000, (4000, 5200) if k < 3 else (300, 900), (300, 500) if k < 3 else (20, 80), 100, synth.DNA, seed=500 + k, workers=8) for k in range(4)]
    lists, sizes = [], []
    for k, ps in enumerate(sets):
        for p in range(len(ps)):
            lists.append((ps.query(p), ps.reference(p))); sizes.append(ranges[k])
    order = np.random.default_rng(3).permutation(len(lists))
    lists = [lists[i] for i in order]; sizes = [sizes[i] for i in order]
    pairs = synth.PairSet.from_lists(lists)
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    b = hip.SizedBatchAligner(NUC, (-5, -1), 100, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, sizes=np.array(sizes))
    cls = b.classes()
    assert sum(1 for c in cls if c[3] == 1) >= 3, cls          # three ranges in k_multi
    b.run()
    ms = b.run()
    assert ms < 1500, ms
    res = b.results()
    assert not res["status"].any()
    runs, off = b.cigars(res["cigar_len"])
    for rg in ranges:
        idx = np.array([p for p in range(len(pairs)) if tuple(sizes[p]) == rg])
        sub = synth.PairSet.from_lists([lists[p] for p in idx])
        ref = oracle.batch_align(NUC, sub.pool, sub.q_off, sub.q_len, sub.r_off, sub.r_len, (-5, -1), rg, 100, ("trace", "x_drop"), cigar_eq=True, threads=16)
        assert np.array_equal(res["score"][idx], ref["scores"]) and np.array_equal(res["cells"][idx].sum(), ref["cells"]) and np.array_equal(res["cigar_len"][idx], ref["cig_len"]), rg
        for t, p in enumerate(idx[:200]):
            want = ref["cig_ops"][int(ref["cig_off"][t]): int(ref["cig_off"][t]) + int(ref["cig_len"][t])]
            assert np.array_equal(runs[int(off[p]): int(off[p + 1])], want), (rg, p)
    b.close()
