"""ba_multibatch_* over REAL devices: runs wherever more than one GPU is visible (the library's own 1 -> 8 launcher has otherwise only
ever run its slices on one device), skips on a one-GPU box. One list of pairs, min(device_count, 8) devices, every pair compared with the
oracle; and `bench.py --multibatch` as a subprocess prints a line whose n_gpus is the device count."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from block_aligner_amd import synth
from tests.test_gpu_parity import NUC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multibatch_over_every_visible_device(hip, oracle):
    ndev = min(hip.device_count(), 8)
    if ndev < 2:
        pytest.skip("one GPU visible: the multi-device launcher needs at least two (tests/test_gpu_parity.py runs its slices on one device)")
    pairs = synth.make_pairs(6000, (200, 3000), (10, 300), 60, synth.DNA, seed=515, indels=1, indel_len=(10, 150))
    mode = hip.TRACE | hip.X_DROP | hip.CIGAR_EQ
    m = hip.MultiBatchAligner(NUC, (-5, -1), (128, 1024), 80, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, list(range(ndev)))
    assert len(m.parts()) == ndev + 1
    for _ in range(2):
        assert m.run() > 0
    got = m.results()
    runs, off = m.cigars(got["cigar_len"])
    m.close()
    ref = oracle.batch_align(NUC, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (128, 1024), 80, ("trace", "x_drop"), cigar_eq=True, threads=16)
    assert not got["status"].any()
    assert np.array_equal(got["score"], ref["scores"]) and np.array_equal(got["query_idx"], ref["query_idx"]) and np.array_equal(got["reference_idx"], ref["reference_idx"])
    assert int(got["cells"].sum()) == ref["cells"] and np.array_equal(got["cigar_len"], ref["cig_len"])
    ln = ref["cig_len"].astype(np.int64)
    start = np.repeat(ref["cig_off"].astype(np.int64), ln)
    within = np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
    assert np.array_equal(runs[: int(off[len(pairs)])], ref["cig_ops"][start + within])


def test_bench_multibatch_line(hip):
    """The one-process bench form on every visible device (a small batch: this checks the harness, not the rate)."""
    ndev = min(hip.device_count(), 8)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--multibatch", "--gpus", str(ndev), "--pairs", "3000", "--len", "2000", "--edits", "200",
                          "--tail", "100", "--steps", "2", "--warmup", "1", "--gen-workers", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == ndev and line["value"] > 0 and line["scaling"] == "weak" and line["config"]["pairs_total"] == 3000 * ndev
    assert len(line["config"]["slice_bounds"]) == ndev + 1
