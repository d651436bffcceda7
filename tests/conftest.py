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
