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


@pytest.mark.parametrize('norm', ['combinatorial', 'randomwalk', 'normalized'])
def test_laplace_trials_on_the_full_operator(gl, golden, norm):
    """ssl.laplace over several training sets: one uploaded operator, Dirichlet rows held at zero
    (glx_cg_groups_masked).  The golden trial inside the batch reproduces the reference (iterates
    and iteration count of the sub-matrix solve), every trial equals its single fit."""
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    labels = g['labels']
    model = gl.ssl.laplace(W, normalization=norm, reduce='exact')
    trials = [g['train_ind']] + [gl.trainsets.generate(labels, rate=r, seed=s) for r, s in ((2, 1), (7, 2), (3, 3))]
    together = model._fit_batch([(ti, labels[ti]) for ti in trials])
    iters = list(model.num_iter)
    assert iters[0] == int(g['laplace_%s_iters' % norm])
    assert np.array_equal(together[0], g['laplace_%s_prob' % norm])
    for j, ti in enumerate(trials):
        alone = model.fit(ti, labels[ti])
        assert model.num_iter == iters[j]
        assert np.array_equal(alone, together[j])


def test_laplace_full_operator_equals_submatrix_solve(gl):
    """The same fits through the reference's literal route (sub-matrix per training set; what the
    reweighted variants still use) on a 10-class graph: identical iterates and iteration counts."""
    from graphlearning_amd import _hip
    from scipy import sparse
    W, labels = _blob_graph(gl, 4000, 10, 12)
    model = gl.ssl.laplace(W, tau=0.01, reduce='exact')
    for seed in (0, 1):
        ti = gl.trainsets.generate(labels, rate=2, seed=seed)
        u = model.fit(ti, labels[ti])
        n = W.shape[0]
        L = sparse.spdiags(model.tau, 0, n, n) + gl.graph(W).laplacian()
        F = gl.utils.labels_to_onehot(labels[ti], 10)
        idx = np.full((n,), True)
        idx[ti] = False
        b = (-L[:, ti] * F)[idx, :]
        A = L[idx, :][:, idx]
        M = sparse.spdiags(1 / np.sqrt(A.diagonal() + 1e-10), 0, A.shape[0], A.shape[0]).tocsr()
        dev = _hip.DeviceGraph(M * A * M, keep_order=True)
        v, it, _ = dev.cg(np.ascontiguousarray(M * b), tol=model.tol)
        dev.close()
        ref = np.zeros((n, 10))
        ref[idx, :] = M * v
        ref[ti, :] = F
        assert it == model.num_iter
        assert np.array_equal(u, ref)


def test_randomwalk_trials_stacked_equal_single(gl, golden):
    g = golden('g7_next_rows.npz')
    W = csr_from(g, 'W')
    labels = g['labels']
    model = gl.ssl.randomwalk(W, reduce='exact')
    trials = [g['train_ind']] + [gl.trainsets.generate(labels, rate=r, seed=s) for r, s in ((1, 4), (6, 5))]
    together = model._fit_batch([(ti, labels[ti]) for ti in trials])
    iters = list(model.num_iter)
    assert iters[0] == int(g['randomwalk_iters']) and np.array_equal(together[0], g['randomwalk_prob'])
    for j, ti in enumerate(trials):
        alone = model.fit(ti, labels[ti])
        assert model.num_iter == iters[j] and np.array_equal(alone, together[j])


def test_tree_mode_history_mirror_survives_a_change_of_layout(gl):
    """The tolerance-mode CG polls a page-locked mirror of the residual history whose rows carry a 'not yet written' marker in
    their last slot.  The slot's position depends on the number of stacked systems: a stacked solve (4 systems) followed by a
    single one (and by a longer history) on the SAME operator must not read the earlier solve's residuals as rows already
    written (early stop, wrong iteration count).  Iteration counts and iterates against reduce='exact'."""
    from graphlearning_amd import _hip
    X, labels = blobs(3000, 10, 4, 31, 1.5)
    W = gl.weightmatrix.knn(X, 9)
    model = gl.ssl.laplace(W, reduce='tree')
    exact = gl.ssl.laplace(W, reduce='exact')
    sets = [gl.trainsets.generate(labels, rate=2, seed=s) for s in range(4)]
    for rep in range(3):
        together = model._fit_batch([(t, labels[t]) for t in sets])                      # 4 systems: stride 5
        its4 = list(model.num_iter)
        ref4 = exact._fit_batch([(t, labels[t]) for t in sets])
        assert its4 == list(exact.num_iter)
        for a, b in zip(together, ref4):
            assert np.max(np.abs(a - b)) <= 1e-9
        for t in sets[:2]:                                                               # 1 system on the same operator: stride 2
            u = model.fit(t, labels[t])
            it1 = model.num_iter
            ue = exact.fit(t, labels[t])
            assert it1 == exact.num_iter, (rep, it1, exact.num_iter)
            assert np.max(np.abs(u - ue)) <= 1e-9
        two = model._fit_batch([(t, labels[t]) for t in sets[:2]])                       # stride 3
        assert list(model.num_iter) == list(exact._fit_batch([(t, labels[t]) for t in sets[:2]]) and exact.num_iter)
    # the solver object directly, with max_iter changing between solves on one DeviceGraph
    A = sparse_spd(400, 7)
    G = _hip.DeviceGraph(A, keep_order=True)
    rng = np.random.default_rng(0)
    B4 = rng.normal(size=(400, 8))
    for max_iter, cols in ((500, 8), (40, 2), (500, 2), (7, 8), (500, 4)):
        xt, it_t, _ = G.cg_groups(np.ascontiguousarray(B4[:, :cols]), 2, tol=1e-9, max_iter=max_iter, reduce='tree')
        xe, it_e, _ = G.cg_groups(np.ascontiguousarray(B4[:, :cols]), 2, tol=1e-9, max_iter=max_iter, reduce='exact')
        assert list(it_t) == list(it_e), (max_iter, cols, it_t, it_e)
        assert np.max(np.abs(xt - xe)) <= 1e-7 * max(1.0, np.max(np.abs(xe)))
    G.close()


def sparse_spd(n, seed):
    from scipy import sparse
    rng = np.random.default_rng(seed)
    A = sparse.random(n, n, density=0.02, random_state=rng, format='csr')
    A = A + A.T
    return sparse.csr_matrix(A + sparse.identity(n) * (np.abs(A).sum(axis=1).max() + 1.0))
