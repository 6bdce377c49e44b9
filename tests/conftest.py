import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def syn_weights():
    from fisr_amd.weights import synthetic_weights
    return synthetic_weights(2020)


@pytest.fixture(scope="session")
def syn_blob(syn_weights):
    import c_oracle
    c_oracle.build()
    return c_oracle.pack_blob(syn_weights)
