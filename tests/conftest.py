import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """AVX2-intrinsic CPU restatement of the reference (test infrastructure, oracle/)."""
    from oracle.oracle_py import Oracle, build
    build()
    return Oracle("avx2")


@pytest.fixture(scope="session")
def oracle_scalar():
    from oracle.oracle_py import Oracle, build
    build()
    return Oracle("scalar")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hip():
    """The product binding; building happens in __graft_entry__.build(), never here on the GPU box."""
    from block_aligner_amd import hip as H
    if not os.path.exists(H.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    H.lib()
    return H


@pytest.fixture
def devlib(hip):
    """Tests that force a code path with the BA_* development switches: those exist only in the development build of the library
    (lib/libblock_aligner_hip_dev.so, ba_host.cpp with -DBA_DEV); the release library reads no environment variables."""
    if not os.path.exists(hip.DEV_LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    release = hip.LIB_PATH
    release_id = hip.lib().ba_build_id()
    hip.use_library(hip.DEV_LIB_PATH)
    assert hip.lib().ba_dev_build() == 1
    assert hip.lib().ba_build_id() == release_id, "libblock_aligner_hip_dev.so is from another build than the release library: make -C block_aligner_amd/csrc"
    yield hip
    import gc
    gc.collect()          # objects of the test are freed by the library that made them
    hip.use_library(release)
