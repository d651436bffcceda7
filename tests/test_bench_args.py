"""bench.py's argument handling (CPU): `--gpus N` without torch.distributed.run must not refuse to run (exit code 2) -- it takes the
one-process multi-GPU launcher (ba_multibatch_*), which then fails loudly here because no device is visible (there is no CPU fallback)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=600)


def test_gpus_n_without_a_launcher_takes_the_multibatch_path(hip):
    if hip.device_count() >= 2:
        pytest.skip("two devices are visible: the GPU tests run this path for real")
    r = run_bench("--gpus", "2", "--pairs", "8", "--len", "300", "--edits", "10", "--tail", "10", "--steps", "1", "--warmup", "0", "--gen-workers", "1")
    assert r.returncode != 2, r.stderr[-400:]
    assert "library's multi-GPU launcher" in r.stderr
    assert "HIP devices visible" in r.stderr and r.returncode != 0      # fails loudly, after the argument handling, for lack of devices
    assert "launch with torch.distributed.run" not in r.stderr


def test_world_size_mismatch_is_reported_not_fatal_for_one_gpu(hip):
    """--gpus 1 under no launcher is the plain form; the parser accepts every documented flag."""
    r = run_bench("--help")
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--strong", "--multibatch", "--pairs"):
        assert flag in r.stdout
