import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_gmm(C, D, seed=0, spread=2.0):
    """Synthetic UBM per SURVEY.md 8(d): means ~ N(0, spread^2), var ~ LogNormal(0, 0.5),
    weights Dirichlet(1) floored at 1e-4."""
    rng = np.random.default_rng(seed)
    mean = rng.normal(0.0, spread, (C, D))
    var = np.exp(rng.normal(0.0, 0.5, (C, D)))
    w = rng.dirichlet(np.ones(C))
    w = np.maximum(w, 1e-4)
    w /= w.sum()
    return w, mean, 1.0 / var


def make_frames(w, mean, covinv, T, seed=1, dtype=np.float32):
    """x = mu_c + sqrt(var_c) N(0,1), component ~ weights; float32 like SPro features."""
    rng = np.random.default_rng(seed)
    C, D = mean.shape
    comp = rng.choice(C, size=T, p=w)
    x = mean[comp] + rng.normal(size=(T, D)) / np.sqrt(covinv[comp])
    return np.ascontiguousarray(x.astype(dtype))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
