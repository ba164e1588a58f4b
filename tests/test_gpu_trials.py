"""Batched trials: ssl.ssl_trials (reference ssl.py:292-396) with the trials'
right-hand sides stacked as column groups of one device solve.  Bar: every trial's result is
BIT-IDENTICAL to fitting it alone (which the parity tests pin to the reference), iteration counts
included, and the results file is the same text."""
import os
import numpy as np
import pytest
from conftest import csr_from, blobs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gl():
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    _hip.require_device()
    return gl


def _blob_graph(gl, n, C, seed):
    X, labels = blobs(n, 12, C, seed, 1.6)
    return gl.weightmatrix.knn(X, 8), labels


@pytest.mark.parametrize('n,C,ntrials', [(600, 2, 9), (3000, 10, 7), (1500, 3, 30)])
def test_cg_groups_equal_separate_solves(gl, n, C, ntrials):
    """7 trials x 10 classes = 70 columns: two reducer workgroups, one system straddling them."""
    from graphlearning_amd import ssl as glssl
    W, labels = _blob_graph(gl, n, C, 5)
    model = gl.ssl.poisson(W)                    # default solver: conjugate gradient
    trials = [gl.trainsets.generate(labels, rate=1 + (t % 4), seed=t) for t in range(ntrials)]
    alone, iters = [], []
    for ti in trials:
        alone.append(model.fit(ti, labels[ti]).copy())
        iters.append(model.num_iter)
    assert len(set(iters)) > 1                   # the systems really stop at different iterations
    together = model._fit_batch([(ti, labels[ti]) for ti in trials])
    assert together is not None and len(together) == ntrials
    assert model.num_iter == iters
    for a, b in zip(alone, together):
        assert np.array_equal(a, b)


def test_cg_groups_golden_twomoons(gl, golden):
    """The golden two-moons trial inside a batch of other trials still reproduces the reference."""
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    labels = g['labels']
    model = gl.ssl.poisson(W)
    trials = [g['train_ind']] + [gl.trainsets.generate(labels, rate=3, seed=s) for s in (1, 2, 3)]
    together = model._fit_batch([(ti, labels[ti]) for ti in trials])
    assert model.num_iter[0] == int(g['poisson_cg_iters'])
    alone = model.fit(g['train_ind'], labels[g['train_ind']])
    assert np.array_equal(together[0], alone)
    assert np.array_equal(together[0], g['poisson_cg_prob'])


def test_ssl_trials_batched_file_equals_sequential(gl, tmp_path, monkeypatch):
    from graphlearning_amd import ssl as glssl
    W, labels = _blob_graph(gl, 2500, 5, 9)
    trainsets = gl.trainsets.generate(labels, rate=np.array([[1], [2], [4]]), num_trials=4, seed=3)
    monkeypatch.setattr(glssl, 'results_dir', str(tmp_path))
    rows = {}
    for tag, batched in (('b_', True), ('s_', False)):
        for priors in (None, gl.utils.class_priors(labels)):
            model = gl.ssl.poisson(W, class_priors=priors)
            if not batched:
                model._trial_batch_size = lambda labels: 1
            model.ssl_trials(trainsets, labels, tag=tag)
            with open(os.path.join(str(tmp_path), tag + model.get_accuracy_filename())) as f:
                rows[(tag, priors is None)] = f.read()
            assert len(rows[(tag, priors is None)].splitlines()) == len(trainsets) + 1
    assert rows[('b_', True)] == rows[('s_', True)]
    assert rows[('b_', False)] == rows[('s_', False)]
