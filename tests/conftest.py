"""Shared fixtures.  `-m "not gpu"` runs here on CPU; `-m gpu` runs on a real MI355X."""

import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def lib_built():
    """Build libbhcore.so if hipcc is around and the library is stale; return its path."""
    from biahub_amd import build

    if build.needs_build():
        build.build(verbose=False)
    return build.LIB


@pytest.fixture(scope="session")
def deskew_cases():
    z = np.load(GOLDEN / "deskew_cases.npz")
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


@pytest.fixture(scope="session")
def helpers_golden():
    return json.load(open(GOLDEN / "helpers.json"))


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test selected but no GPU is visible (this suite never falls back to CPU)")
    return torch.device("cuda", 0)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / scale)
