import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _class_codes_on_small_matrices(monkeypatch):
    """The product builds the class codes only when its cost models expect them to pay (csrc/em_api.hip: em_codes_pay,
    csrc/codes.hip: wgs_codes_pay_for_scoring) -- never for matrices of test size.  The suite wants the coded kernels exercised by
    every test that fits or scores with shared columns, so the models are switched to "always" here; the tests of the models
    themselves (tests/test_gpu_codes.py) remove these settings again."""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_SWEEPS", "0")
    monkeypatch.setenv("WGSASSIGN_SCORE_CODES_ALWAYS", "1")
    monkeypatch.setenv("WGSASSIGN_CODES_ALLOC_WAIT_MS", "-1")     # ... and wait for the codes' memory however long its hipMalloc takes


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
