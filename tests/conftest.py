import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import bn254_oracle
    return bn254_oracle


@pytest.fixture(scope="session")
def cc():
    """the product package; building is __graft_entry__.build()'s job, importing must just work"""
    import crescent_credentials_amd
    return crescent_credentials_amd


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
