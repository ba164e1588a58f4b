import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
    return load


def csr_from(g, prefix):
    from scipy import sparse
    n = len(g[prefix + '_indptr']) - 1
    return sparse.csr_matrix((g[prefix + '_data'], g[prefix + '_indices'], g[prefix + '_indptr']), shape=(n, n))


def blobs(n, d, C, seed, scale, labels=None):
    """Same synthetic generator as tests/golden/make_golden.py."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(C, d)) * scale
    if labels is None:
        labels = rng.integers(0, C, size=n)
    X = centers[labels] + rng.normal(size=(n, d))
    return X, labels.astype(np.int64)
