"""GPU parity of the multi-pair kernel (k_multi, ba_multi.hpp): batches that start at 128 cells run four pairs per wave while a
pair's block is 128 cells and hand a pair to the same wave's solo mode (the per-pair driver) for everything else. Every pair is
compared with the oracle: score, end positions, computed cells, CIGAR runs."""
import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from tests.test_gpu_parity import NUC, compare

pytestmark = pytest.mark.gpu

MODES = [(), ("x_drop",), ("trace",), ("trace", "x_drop")]


@pytest.fixture
def force_multi(devlib, monkeypatch):
    monkeypatch.setenv("BA_FORCE_MULTI", "1")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("size", [(128, 128), (128, 256), (128, 1024), (128, 2048)])
def test_multi_dna(hip, oracle, force_multi, mode, size):
    """Indels of 20 .. 200 bases force grows, checkpoint restores and shrinks: pairs move between slot and solo mode many times."""
    pairs = synth.make_pairs(150, (800, 3000), (50, 300), 100, synth.DNA, seed=900 + size[1], indels=3, indel_len=(20, 200))
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, mode)
    assert res["cells"].max() > 0


@pytest.mark.parametrize("mode", MODES)
def test_multi_ragged_and_short(hip, oracle, force_multi, mode):
    """Pairs shorter than a block, empty sequences and one-sided pairs share waves with ordinary ones."""
    rng = np.random.default_rng(11)
    lists = [(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A", b"A"), (b"A" * 127, b"A" * 129), (b"ACGT" * 40, b"ACGT" * 300)]
    for _ in range(60):
        n = int(rng.integers(0, 1200))
        a = synth.rand_str(rng, n, synth.DNA)
        b = synth.mutate(rng, a, int(rng.integers(0, 1 + n // 8)), synth.DNA) if n else a
        lists.append((a.tobytes(), b.tobytes()))
    pairs = synth.PairSet.from_lists(lists)
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 60, mode)


@pytest.mark.parametrize("mode", MODES)
def test_multi_protein(hip, oracle, force_multi, mode):
    pairs = synth.make_pairs(200, (100, 1500), (0, 300), 0, synth.AMINO, seed=77)
    compare(hip, oracle, pairs, S.BLOSUM62, (-11, -1), (128, 512), 50, mode)


SPECIAL_MODES = [("trace", "local_start"), ("trace", "x_drop", "local_start"), ("local_start",), ("x_drop", "local_start"), ("trace", "free_query_start_gaps"),
                 ("trace", "x_drop", "free_query_start_gaps"), ("free_query_start_gaps",)]


def _flanked_pairs(n, seed, core=(300, 3000), flank=400):
    """Related cores behind unrelated flanks (what LOCAL_START skips), queries cut out of longer references (what FREE_QUERY_START_GAPS skips),
    ordinary related pairs."""
    rng = np.random.default_rng(seed)
    lists = []
    for k in range(n):
        c = synth.rand_str(rng, int(rng.integers(core[0], core[1])), synth.DNA)
        other = synth.mutate(rng, c, int(rng.integers(0, 1 + len(c) // 10)), synth.DNA)
        if k % 3 == 0:
            q = np.concatenate([synth.rand_str(rng, int(rng.integers(0, flank)), synth.DNA), c])
            r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, flank)), synth.DNA), other])
        elif k % 3 == 1:
            q = c
            r = np.concatenate([synth.rand_str(rng, int(rng.integers(0, 2 * flank)), synth.DNA), other, synth.rand_str(rng, int(rng.integers(0, flank)), synth.DNA)])
        else:
            q, r = c, other
        lists.append((q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()))
    return synth.PairSet.from_lists(lists)


@pytest.mark.parametrize("mode", SPECIAL_MODES)
@pytest.mark.parametrize("size", [(128, 128), (128, 1024)])
def test_multi_local_and_free_start(hip, oracle, force_multi, mode, size):
    """Round 5: the slots take LOCAL_START / FREE_QUERY_START_GAPS steps too (k_multi's special instantiations): every cell at least the relative
    zero, a zero mask of one bit per cell behind a slot rectangle's trace words, the traceback lanes' early stops (scan_block.rs:1130-1136,
    1184-1187, 1597-1611)."""
    m = 0
    for k in mode:
        m |= {"trace": hip.TRACE, "x_drop": hip.X_DROP, "local_start": hip.LOCAL_START, "free_query_start_gaps": hip.FREE_QUERY_START_GAPS}[k]
    pairs = _flanked_pairs(240, 50 + size[1])
    b = hip.BatchAligner(NUC, (-5, -1), size, 100, m, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi"
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, mode)
    pairs = synth.make_pairs(150, (800, 3000), (50, 300), 100, synth.DNA, seed=51 + size[1], indels=3, indel_len=(20, 200))
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, mode)
    pairs = synth.make_pairs(150, (100, 1500), (0, 300), 0, synth.AMINO, seed=78)
    compare(hip, oracle, pairs, S.BLOSUM62, (-11, -1), size, 50, mode)


def test_multi_special_modes_with_traceback_waves(hip, oracle, force_multi, monkeypatch):
    """... with the in-launch hand-off to traceback waves (records with the zero-mask bits) and recycled trace slots."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    pairs = _flanked_pairs(1500, 7, core=(1000, 4000), flank=600)
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 100, ("trace", "x_drop", "local_start"))
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 100, ("trace", "free_query_start_gaps"))


def test_multi_bytes(hip, oracle, force_multi):
    pairs = synth.make_pairs(100, (100, 900), (0, 60), 5, np.frombuffer(b"abcdefghij\x01\xff", np.uint8), seed=5)
    compare(hip, oracle, pairs, S.BYTES1, (-2, -1), (128, 256), 0, ())
    compare(hip, oracle, pairs, S.BYTES1, (-2, -1), (128, 256), 0, ("trace",))


def test_multi_config3_shape_with_traceback_waves(hip, oracle, force_multi, monkeypatch):
    """Config-3 shaped pairs with the in-launch hand-off to traceback waves and recycled trace slots."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    pairs = synth.make_pairs(600, (3000, 10000), (300, 1000), 500, synth.DNA, seed=4321)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), (128, 1024), 100, ("trace", "x_drop"))
    assert (res["query_idx"] > 2000).all()


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ("trace",)])
@pytest.mark.parametrize("n,wgs", [(600, "1"), (2500, "1"), (9000, "2")])
def test_multi_slots_change_waves_at_the_end_of_the_batch(hip, oracle, force_multi, monkeypatch, n, wgs, mode):
    """The end of a k_multi batch (round 4): waves that go solo for the last time offer their other slots, waves whose slot has emptied
    refill it from an offer, waves with nothing left run an offered slot to its end out of the other wave's arena region. Few pairs per
    wave make most of the launch such an end; 600 pairs run on 75 workgroups -- a wave count that is not a multiple of 64 (the scan over
    the waves' offer words once ran into the counters behind them there). Every pair is compared with the oracle."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", wgs)
    pairs = synth.make_pairs(n, (1500, 6000), (100, 600), 300, synth.DNA, seed=77 + n, indels=1, indel_len=(20, 200))
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 100, mode, threads=16)


@pytest.mark.parametrize("mode", MODES)
def test_multi_at_production_threshold(hip, oracle, mode):
    """No forcing: a batch above the size from which the library itself picks the multi-pair kernel."""
    pairs = synth.make_pairs(20000, (300, 900), (20, 90), 60, synth.DNA, seed=2024)
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 256), 80, mode, threads=16)


# ---- round 6: slots of 256 cells (two pairs per wave, 32 lanes each): DNA batches that start at 256 cells -- percent_len of reads above
# 12.8 kbp (lib.rs:109-111, examples/nanopore_bench_global.rs:144-171)
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("size", [(256, 512), (256, 1024), (256, 2048)])
def test_multi256_dna(hip, oracle, force_multi, mode, size):
    """Indels of 20 .. 400 bases force grows, checkpoint restores and shrinks: pairs move between their 256-cell slot and solo mode many times."""
    pairs = synth.make_pairs(150, (1500, 6000), (100, 600), 200, synth.DNA, seed=1900 + size[1], indels=3, indel_len=(20, 400))
    b = hip.BatchAligner(NUC, (-5, -1), size, 100, sum({"trace": hip.TRACE, "x_drop": hip.X_DROP}[k] for k in mode), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi"
    b.close()
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, mode)
    assert res["cells"].max() > 0


@pytest.mark.parametrize("mode", MODES)
def test_multi256_ragged_and_short(hip, oracle, force_multi, mode):
    """Pairs shorter than a block, empty sequences and one-sided pairs share waves with ordinary ones."""
    rng = np.random.default_rng(12)
    lists = [(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A", b"A"), (b"A" * 255, b"A" * 257), (b"ACGT" * 70, b"ACGT" * 500)]
    for _ in range(60):
        n = int(rng.integers(0, 2400))
        a = synth.rand_str(rng, n, synth.DNA)
        b = synth.mutate(rng, a, int(rng.integers(0, 1 + n // 8)), synth.DNA) if n else a
        lists.append((a.tobytes(), b.tobytes()))
    pairs = synth.PairSet.from_lists(lists)
    compare(hip, oracle, pairs, NUC, (-5, -1), (256, 1024), 60, mode)


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ("trace",), ("x_drop",)])
def test_multi256_long_reads_with_traceback_waves(hip, oracle, force_multi, monkeypatch, mode):
    """13 kbp-shaped pairs (what percent_len starts at 256 cells) with the in-launch hand-off to traceback waves, recycled trace slots and the
    slots changing waves at the end of the batch."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    pairs = synth.make_pairs(700, (6000, 14000), (500, 1400), 500, synth.DNA, seed=4322)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), (256, 2048), 100, mode)
    assert (res["query_idx"] > 4000).all()


def test_multi256_release_library_takes_it_from_the_threshold(hip, oracle):
    """The release library itself picks the 256-cell slots for a batch of long reads (no development switch)."""
    pairs = synth.make_pairs(2100, (3200, 3600), (300, 350), 100, synth.DNA, seed=77, workers=8)
    b = hip.BatchAligner(NUC, (-5, -1), (256, 512), 100, hip.TRACE | hip.X_DROP | hip.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi", b.info()
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), (256, 512), 100, ("trace", "x_drop"))


# ---- round 6: k_multi's launch geometries (four-wave workgroups at three / two waves per SIMD: batches of about one round -- ba_host.cpp batch_build)
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("geom", [2, 3])
@pytest.mark.parametrize("size", [(128, 512), (128, 1024)])
def test_multi_geometries(hip, oracle, force_multi, monkeypatch, mode, geom, size):
    """The same kernel compiled for 256 / 168 registers in workgroups of four waves: indels force grows, restores and shrinks, so pairs change
    between slot and solo mode many times (the solo driver is what the register budget changes)."""
    monkeypatch.setenv("BA_MQ_GEOM", str(geom))
    pairs = synth.make_pairs(150, (800, 3000), (50, 300), 100, synth.DNA, seed=1900 + size[1] + geom, indels=3, indel_len=(20, 200))
    b = hip.BatchAligner(NUC, (-5, -1), size, 100, 0, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi" and b.info()["geometry"] == geom, b.info()
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, mode)


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ("trace",)])
@pytest.mark.parametrize("geom,n,wgs", [(2, 700, "1"), (3, 2500, "1"), (3, 6000, "3"), (2, 6000, "2")])
def test_multi_geometries_with_traceback_waves(hip, oracle, force_multi, monkeypatch, geom, n, wgs, mode):
    """... with the in-launch hand-off to traceback waves (one per six workgroups), recycled trace slots, and slots changing waves at the end of the batch."""
    monkeypatch.setenv("BA_MQ_GEOM", str(geom))
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", wgs)
    pairs = synth.make_pairs(n, (1500, 6000), (100, 600), 300, synth.DNA, seed=177 + n + geom, indels=1, indel_len=(20, 200))
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 100, mode, threads=16)


@pytest.mark.parametrize("n,geom,mode", [(6400, 2, ("trace", "x_drop")), (9800, 3, ("trace", "x_drop")), (8000, 2, ("x_drop",)), (11000, 3, ("x_drop",)), (11000, 3, ())])
def test_multi_geometry_release_library_by_batch_size(hip, oracle, n, geom, mode):
    """No development switch: the release library picks the geometry whose slots the batch fills about once (MI355X: 8192 slots at two waves per SIMD,
    12288 at three; score-only batches: no more workgroups than the batch fills with four pairs per wave), and every pair still matches the oracle."""
    pairs = synth.make_pairs(n, (1500, 1700), (100, 170), 60, synth.DNA, seed=4000 + n, workers=8)
    bits = (hip.TRACE | hip.CIGAR_EQ if "trace" in mode else 0) | (hip.X_DROP if "x_drop" in mode else 0)
    b = hip.BatchAligner(NUC, (-5, -1), (128, 512), 100, bits, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    info = b.info()
    b.close()
    waves = 1024 * geom if "trace" in mode else min(1024 * geom, 4 * ((n + 15) // 16))
    assert info["kernel"] == "k_multi" and info["geometry"] == geom and info["grid"] == waves, info
    compare(hip, oracle, pairs, NUC, (-5, -1), (128, 512), 100, mode, threads=16)


# ---- round 6: one slot of 512 cells per wave (DNA batches that start at 512 cells: percent_len 1 % of reads above 25.6 kbp, lib.rs:109-111)
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("size", [(512, 1024), (512, 2048), (512, 4096)])
def test_multi512_dna(hip, oracle, force_multi, mode, size):
    """Indels of 50 .. 900 bases force grows, checkpoint restores and shrinks: pairs move between their 512-cell slot and solo mode many times
    (512..4096: launched in the 2048-cell class, pairs that grow past it re-run in the row-tiled one)."""
    pairs = synth.make_pairs(100, (3000, 12000), (200, 1200), 300, synth.DNA, seed=2900 + size[1], indels=3, indel_len=(50, 900))
    b = hip.BatchAligner(NUC, (-5, -1), size, 100, sum({"trace": hip.TRACE, "x_drop": hip.X_DROP}[k] for k in mode), pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi"
    b.close()
    res = compare(hip, oracle, pairs, NUC, (-5, -1), size, 100, mode)
    assert res["cells"].max() > 0


@pytest.mark.parametrize("mode", MODES)
def test_multi512_ragged_and_short(hip, oracle, force_multi, mode):
    """Pairs shorter than a block, empty sequences and one-sided pairs beside ordinary ones."""
    rng = np.random.default_rng(13)
    lists = [(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A", b"A"), (b"A" * 511, b"A" * 513), (b"ACGT" * 140, b"ACGT" * 1000)]
    for _ in range(50):
        n = int(rng.integers(0, 5000))
        a = synth.rand_str(rng, n, synth.DNA)
        b = synth.mutate(rng, a, int(rng.integers(0, 1 + n // 8)), synth.DNA) if n else a
        lists.append((a.tobytes(), b.tobytes()))
    pairs = synth.PairSet.from_lists(lists)
    compare(hip, oracle, pairs, NUC, (-5, -1), (512, 2048), 60, mode)


@pytest.mark.parametrize("mode", [("trace", "x_drop"), ("trace",), ("x_drop",)])
def test_multi512_long_reads_with_traceback_waves(hip, oracle, force_multi, monkeypatch, mode):
    """32 kbp-shaped pairs with the in-launch hand-off to traceback waves, recycled trace slots and the slot changing waves at the end of the batch."""
    monkeypatch.setenv("BA_FORCE_TB", "1")
    monkeypatch.setenv("BA_WGS_PER_CU", "1")
    pairs = synth.make_pairs(500, (12000, 30000), (1000, 3000), 500, synth.DNA, seed=4323, workers=8)
    res = compare(hip, oracle, pairs, NUC, (-5, -1), (512, 4096), 100, mode, threads=16)
    assert (res["query_idx"] > 8000).all()


def test_multi512_release_library_takes_it_from_the_threshold(hip, oracle):
    """The release library itself picks the 512-cell slot for a few hundred long reads (no development switch), here through the class bet of 512..4096."""
    pairs = synth.make_pairs(300, (5000, 9000), (400, 900), 200, synth.DNA, seed=78, indels=1, indel_len=(50, 3000), workers=8)
    b = hip.BatchAligner(NUC, (-5, -1), (512, 4096), 100, hip.TRACE | hip.X_DROP | hip.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    assert b.info()["kernel"] == "k_multi", b.info()
    b.close()
    compare(hip, oracle, pairs, NUC, (-5, -1), (512, 4096), 100, ("trace", "x_drop"), threads=16)


def test_multi512_batch_that_loses_the_class_bet(hip, oracle):
    """A 512..4096 batch in the 512-cell slots (launched in the 2048-cell class) of which more than an eighth grows past 2048 cells: those pairs are re-run in
    the row-tiled class, and the batch's next run is the row-tiled class's own -- per-pair kernel, nothing re-run, identical results."""
    quiet = synth.make_pairs(216, (5000, 7000), (300, 600), 100, synth.DNA, seed=81, workers=8)
    growers = synth.make_pairs(44, (11000, 13000), (300, 600), 100, synth.DNA, seed=82, indels=1, indel_len=(3000, 5000), workers=8)
    both = synth.PairSet.from_lists([(quiet.query(p), quiet.reference(p)) for p in range(len(quiet))] + [(growers.query(p), growers.reference(p)) for p in range(len(growers))])
    b = hip.BatchAligner(NUC, (-5, -1), (512, 4096), 0, hip.TRACE | hip.CIGAR_EQ, both.pool, both.q_off, both.q_len, both.r_off, both.r_len)
    assert b.info()["kernel"] == "k_multi", b.info()
    b.run()
    first, n_again = b.results(), b.retried()
    assert n_again * 8 > len(both), n_again
    b.run()
    second = b.results()
    # (what it still re-runs are growers whose trace stack outgrew a slot sized by the expected stack: fewer than grew past the class)
    assert b.retried() < n_again and b.info()["kernel"] == "k_align" and b.info()["lds_bytes_per_wave"] < 4096, (b.retried(), n_again, b.info())
    for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"):
        assert np.array_equal(first[k], second[k]), k
    b.close()
    compare(hip, oracle, both, NUC, (-5, -1), (512, 4096), 0, ("trace",), threads=16)
