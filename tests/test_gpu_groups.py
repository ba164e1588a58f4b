"""Stacked gradient-descent trials (SURVEY section 8 f-1 for the sweep): several training sets of
ssl.poisson(solver='gradient_descent') as column groups of ONE sweep (glx_sweep_groups; reference ssl.py:292-396 runs
ssl.py:631-670 once per training set).  Bar: every trial's iterate u_T and its sweep count T are BIT-IDENTICAL to fitting it
alone and to the oracle's restatement of the reference loop -- whatever the other trials of the batch do (they stop at other
sweeps) --, the float32 branch within the north star's 1e-5 with the same T and labels, and ssl_trials writes the same file."""
import os
import numpy as np
import pytest
from scipy import sparse
from conftest import csr_from, blobs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gl():
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    _hip.require_device()
    return gl


@pytest.fixture(scope='module')
def orc():
    from oracle import gl_oracle
    return gl_oracle


def _fetch(r):
    return np.array(r.fetch()) if hasattr(r, 'fetch') else np.asarray(r)


def test_groups_golden_twomoons(gl, orc, golden):
    """The golden trial (T = 409) inside a batch of other training sets of other sizes: its iterate is the reference's."""
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    labels = g['labels']
    model = gl.ssl.poisson(W, solver='gradient_descent')
    trials = [g['train_ind']] + [gl.trainsets.generate(labels, rate=r, seed=s) for r, s in ((1, 1), (3, 2), (10, 3), (40, 4))]
    together = model._fit_batch([(ti, labels[ti]) for ti in trials])
    assert together is not None and len(together) == len(trials)
    Ts = list(model.num_iter)
    assert Ts[0] == int(g['poisson_gd_T'])
    assert np.array_equal(together[0], g['poisson_gd_prob'])
    assert len(set(Ts)) > 1                                    # the trials really stop at different sweeps
    for j, ti in enumerate(trials):
        u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], return_T=True)
        assert Ts[j] == T_ref, j
        assert np.array_equal(together[j], u_ref), j
        alone = gl.ssl.poisson(W, solver='gradient_descent')
        assert np.array_equal(alone.fit(ti, labels[ti]), together[j]) and alone.num_iter == Ts[j]


@pytest.mark.parametrize('n,C,k,ntrials,iters', [(600, 2, 8, 2, (50, 1000)), (900, 3, 6, 5, (0, 400)), (3000, 10, 10, 8, (50, 1000)),
                                                 (1500, 5, 12, 7, (10, 60)), (700, 4, 7, 3, (20, 20)), (2500, 10, 9, 4, (3, 1000)),
                                                 (1200, 16, 8, 3, (50, 1000)), (800, 6, 5, 10, (50, 300))])
def test_groups_equal_single_fits_and_oracle(gl, orc, n, C, k, ntrials, iters):
    """Every width of the stacked record (2 .. 10 trials x 2 .. 16 classes: 8 / 16 / 32 / 64 lanes per row), min_iter = 0 (the test in
    front of the first sweep), max_iter reached by some trials and not by others, min_iter = max_iter."""
    X, labels = blobs(n, 10, C, 11 + n, 1.4)
    W = gl.weightmatrix.knn(X, k)
    min_iter, max_iter = iters
    model = gl.ssl.poisson(W, solver='gradient_descent', min_iter=min_iter, max_iter=max_iter)
    trials = [gl.trainsets.generate(labels, rate=1 + (t * 7) % 5, seed=t) for t in range(ntrials)]
    together = model._fit_batch([(ti, labels[ti]) for ti in trials])
    assert together is not None
    Ts = list(model.num_iter)
    single = gl.ssl.poisson(W, solver='gradient_descent', min_iter=min_iter, max_iter=max_iter)
    for j, ti in enumerate(trials):
        u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], min_iter=min_iter, max_iter=max_iter, return_T=True)
        assert Ts[j] == T_ref, (j, Ts, T_ref)
        assert np.array_equal(together[j], u_ref), j
        u1 = single.fit(ti, labels[ti])
        assert single.num_iter == T_ref and np.array_equal(u1, together[j]), j


def test_groups_directed_graph_and_device_decision(gl, orc):
    """A directed (not symmetrised) kNN graph goes through the general operator build; the label decision of a stacked trial
    runs on the device (GroupView.project) and equals the host decision on the fetched iterate."""
    X, labels = blobs(1800, 8, 4, 77, 1.5)
    W = gl.weightmatrix.knn(X, 9, symmetrize=False)
    model = gl.ssl.poisson(W, solver='gradient_descent')
    trials = [gl.trainsets.generate(labels, rate=2, seed=t) for t in range(6)]
    res = model._fit_batch_device([(ti, labels[ti]) for ti in trials])
    assert res is not None
    for j, ti in enumerate(trials):
        with np.errstate(all='ignore'):
            u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], return_T=True)
        assert model.num_iter[j] == T_ref
        model._set_result(res[j])
        model.fitted = True
        pred = model.predict()                               # on the device state of group j
        assert np.array_equal(pred, orc.predict(u_ref))
        assert np.array_equal(_fetch(res[j]), u_ref, equal_nan=True)


def test_groups_float32_branch(gl, orc):
    """use_cuda=True: float32 state, fp64 stop values two to a lane: same T, iterates within 1e-5, same labels as the fp64 fit."""
    X, labels = blobs(2500, 12, 10, 5, 1.6)
    W = gl.weightmatrix.knn(X, 10)
    trials = [gl.trainsets.generate(labels, rate=1 + t % 3, seed=t) for t in range(7)]
    m32 = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=True)
    together = m32._fit_batch([(ti, labels[ti]) for ti in trials])
    assert together is not None
    for j, ti in enumerate(trials):
        u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], return_T=True)
        assert together[j].dtype == np.float32 and m32.num_iter[j] == T_ref
        assert np.max(np.abs(together[j] - u_ref)) <= 1e-5 * max(1.0, np.max(np.abs(u_ref)))
        single = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=True)
        assert np.array_equal(single.fit(ti, labels[ti]), together[j])      # the same float32 arithmetic, stacked or not


def test_groups_reused_across_batches_and_partial_batch(gl, orc):
    """The prepared stacked sweep serves batch after batch (new training sets replace the old rows of their group) and a last
    batch with fewer trials than groups (the idle groups stay idle)."""
    X, labels = blobs(2000, 10, 5, 21, 1.5)
    W = gl.weightmatrix.knn(X, 8)
    model = gl.ssl.poisson(W, solver='gradient_descent')
    all_trials = [gl.trainsets.generate(labels, rate=1 + t % 4, seed=100 + t) for t in range(19)]
    B = model._trial_batch_size(labels)
    assert B >= 2
    for pos in range(0, len(all_trials), B):
        group = all_trials[pos:pos + B]
        if len(group) < 2:
            break
        res = model._fit_batch([(ti, labels[ti]) for ti in group])
        for j, ti in enumerate(group):
            u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], return_T=True)
            assert model.num_iter[j] == T_ref and np.array_equal(res[j], u_ref), (pos, j)


def test_groups_unstackable_batches_fall_back(gl):
    X, labels = blobs(900, 6, 3, 8, 1.5)
    W = gl.weightmatrix.knn(X, 7)
    model = gl.ssl.poisson(W, solver='gradient_descent')
    ti = gl.trainsets.generate(labels, rate=2, seed=0)
    two_classes = ti[labels[ti] < 2]
    assert model._fit_batch([(ti, labels[ti]), (two_classes, labels[two_classes])]) is None        # different class counts
    rep = np.concatenate([ti, ti[:1]])
    assert model._fit_batch([(ti, labels[ti]), (rep, labels[rep])]) is None                        # a repeated labelled row


def test_ssl_trials_gd_file_equals_sequential(gl, tmp_path, monkeypatch):
    from graphlearning_amd import ssl as glssl
    X, labels = blobs(2500, 12, 5, 9, 1.6)
    W = gl.weightmatrix.knn(X, 8)
    trainsets = gl.trainsets.generate(labels, rate=np.array([[1], [2], [4]]), num_trials=5, seed=3)
    monkeypatch.setattr(glssl, 'results_dir', str(tmp_path))
    rows = {}
    for tag, batched in (('b_', True), ('s_', False)):
        for priors in (None, gl.utils.class_priors(labels)):
            model = gl.ssl.poisson(W, class_priors=priors, solver='gradient_descent')
            if not batched:
                model._trial_batch_size = lambda labels: 1
            else:
                assert model._trial_batch_size(labels) >= 2
            model.ssl_trials(trainsets, labels, tag=tag)
            with open(os.path.join(str(tmp_path), tag + model.get_accuracy_filename())) as f:
                rows[(tag, priors is None)] = f.read()
            assert len(rows[(tag, priors is None)].splitlines()) == len(trainsets) + 1
    assert rows[('b_', True)] == rows[('s_', True)]
    assert rows[('b_', False)] == rows[('s_', False)]
