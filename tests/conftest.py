import gzip
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with gzip.open(os.path.join(HERE, "golden", "golden.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_k17():
    """G14: the reference's outputs at k = 17 / 21 (tests/golden/gen_golden_k17.py)"""
    with gzip.open(os.path.join(HERE, "golden", "golden_k17.json.gz"), "rt") as f:
        return json.load(f)["G14_shapes_k17_k21"]


@pytest.fixture(scope="session")
def toy():
    from toygenome import make_toy_genome
    return make_toy_genome(seed=7)


@pytest.fixture(scope="session")
def oracle_ctx():
    """Oracle-backed stand-in for the GPU context: lets the host layer
    (subphaser_amd/*.py) be tested on CPU against the golden vectors."""
    from oracle_ctx import OracleContext
    return OracleContext()


@pytest.fixture(scope="session")
def gpu_ctx():
    """The real thing: libsubphaser_hip.so on cuda:0.  No fallback."""
    from subphaser_amd import _native
    ctx = _native.Context(0)
    yield ctx
    ctx.close()
