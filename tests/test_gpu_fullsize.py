"""GPU, BASELINE.json's full sizes: config 2 (70000 x 10 classes, k=10) and config 3 (60000,
k=20, Laplace CG) regenerated from seeds and checked against the checksums the reference
produced in the build container (tests/golden/g4_large_meta.json), plus size-independent
properties (symmetry, zero diagonal, row-stochastic P, conservation of sum_i deg_i u_i)."""
import hashlib
import json
import os
import time
import numpy as np
import pytest
from scipy import sparse
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.fixture(scope='module')
def gl():
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    _hip.require_device()
    return gl


# Both modes of weightmatrix.knn's Gaussian weights (see tests/test_gpu_fuzz.py): 'host_exp' = this host's numpy exp, the bits of the
# reference's recorded runs; 'device_exp' = the product's default (correctly rounded exp on the device).  In the default mode the
# recorded checksums of the reference's runs are held to the north star's contract (structure identical, T / Laplace iteration
# counts / labels equal, sampled iterates within 1e-5), and the stages whose reference answer hangs on rounding noise -- the
# singular Poisson CG system, PoissonMBO's thresholding -- are held bit for bit against the oracle run on the SAME matrix instead.
@pytest.fixture(scope='module', params=['host_exp', 'device_exp'])
def mode(request):
    old = os.environ.get('GLX_HOST_EXP')
    if request.param == 'host_exp':
        os.environ['GLX_HOST_EXP'] = '1'
    else:
        os.environ.pop('GLX_HOST_EXP', None)
    yield request.param
    if old is None:
        os.environ.pop('GLX_HOST_EXP', None)
    else:
        os.environ['GLX_HOST_EXP'] = old


def _record(line):
    os.makedirs(os.path.join(os.path.dirname(GOLDEN), '..', 'gpurun_out'), exist_ok=True)
    with open(os.path.join(os.path.dirname(GOLDEN), '..', 'gpurun_out', 'default_mode_deviations.txt'), 'a') as f:
        f.write(line + '\n')


@pytest.fixture(scope='module')
def meta():
    return json.load(open(os.path.join(GOLDEN, 'g4_large_meta.json')))


@pytest.fixture(scope='module')
def config2(gl, mode):
    labels = np.load(os.path.join(GOLDEN, 'MNIST_labels.npz'))['labels'].astype(np.int64)
    rng = np.random.default_rng(0)
    centers = rng.normal(size=(10, 20)) * 2.0
    X = centers[labels] + rng.normal(size=(70000, 20))
    J, D = gl.weightmatrix.knnsearch(X, 11)
    W = gl.weightmatrix.knn(None, 10, knn_data=(J, D))
    return dict(labels=labels, X=X, J=J, D=D, W=W, train_ind=gl.trainsets.generate(labels, rate=1, seed=0), mode=mode)


def test_config2_graph(gl, meta, config2):
    m = meta['config2']
    J, D, W = config2['J'], config2['D'], config2['W']
    assert sha(J.astype(np.int64)) == m['J_sha']                 # identical neighbour lists, 70000 x 11
    assert abs(D.sum() - m['D_sum']) < 1e-6
    assert W.nnz == m['nnz'] and np.diff(W.indptr).max() == m['row_nnz_max']
    assert sha(W.indices.astype(np.int32)) == m['W_indices_sha']
    assert abs(W.data.sum() - m['W_data_sum']) < 1e-8
    assert (abs(W - W.T) > 0).nnz == 0 and W.diagonal().sum() == 0 and W.data.min() > 0
    assert np.all(np.diff(D, axis=1) >= 0) and np.all(D[:, 0] == 0) and np.all(J[:, 0] == np.arange(70000))
    assert list(config2['train_ind']) == m['train_ind']


def _sample_err(u, smp):
    """max |u[i, c] - reference| over the entries of the REFERENCE's run recorded in g4_large_meta.json (make_golden.g4_samples)."""
    rows, cols, ref = np.asarray(smp['rows']), np.asarray(smp['cols']), np.asarray(smp['values'])
    return float(np.max(np.abs(np.asarray(u, dtype=np.float64)[rows, cols] - ref)))


def test_config2_poisson_gd(gl, meta, config2):
    m = meta['config2']
    W, ti, labels = config2['W'], config2['train_ind'], config2['labels']
    model = gl.ssl.poisson(W, solver='gradient_descent')
    u = model.fit(ti, labels[ti])
    pred = model.predict()
    assert model.num_iter == m['T'] == 50
    assert sha(pred.astype(np.int64)) == m['pred_sha']            # identical predicted labels, all 70000
    assert abs(np.abs(u).sum() - m['prob_abs_sum']) < 1e-8 * m['prob_abs_sum']
    assert gl.ssl.ssl_accuracy(pred, labels, ti) == m['accuracy']
    # element-wise, end to end (search, weights, operator, 50 sweeps all on the device) against 1000 sampled entries of the
    # reference's own iterate: the north star asks for 1e-5, the device-built graph differs from the reference's by <= 1e-12
    # in its distances, and the sweeps add nothing to that
    assert _sample_err(u, m['u_samples']) <= 1e-9
    # conservation: sum_i deg_i u_i stays 0 (the source has zero column sums)
    deg = np.asarray(W.sum(axis=1)).ravel()
    assert np.max(np.abs(deg @ u)) < 1e-9
    # fp32 device path of the reference (use_cuda=True): within 1e-5, same labels
    m32 = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=True)
    u32 = m32.fit(ti, labels[ti])
    assert u32.dtype == np.float32 and m32.num_iter == 50
    assert np.max(np.abs(u32 - u)) < 1e-5
    assert _sample_err(u32, m['u_samples']) <= 1e-5                 # the fp32 path against the reference's fp64 run
    assert np.array_equal(m32.predict(), pred)


def test_config2_poisson_cg(gl, meta, config2):
    m = meta['config2']
    W, ti, labels = config2['W'], config2['train_ind'], config2['labels']
    model = gl.ssl.poisson(W)
    t0 = time.perf_counter()
    u = model.fit(ti, labels[ti])
    print('poisson CG 70k: %d iterations, %.3f s' % (model.num_iter, time.perf_counter() - t0))
    if config2['mode'] == 'device_exp':
        # the singular system stopped at 1e-3: one-ulp weights move the iteration count (tests/test_gpu_weights.py) -- bit for bit
        # against the oracle on the SAME matrix; the distance to the reference's recorded run is recorded, not asserted
        from oracle import gl_oracle as orc
        u_ref, it_ref = orc.poisson_cg(W, ti, labels[ti], return_iters=True)
        assert model.num_iter == it_ref and np.array_equal(u, u_ref)
        _record('config 2 Poisson CG, default mode: %d iterations (the reference on its own W: %d), labels equal to the reference run: %s, '
                'sampled |du| %.3e of max |u| %.3e (the ten clusters are separate components: the singular system\'s per-component null-space '
                'part grows to 1e13 in the reference\'s own run and hangs on the last bits of W; it is constant per component and column pattern, labels do not see it)'
                % (it_ref, m['cg_iters'], sha(model.predict().astype(np.int64)) == m['cg_pred_sha'], _sample_err(u, m['cg_u_samples']),
                   float(np.max(np.abs(m['cg_u_samples']['values'])))))
        return
    assert model.num_iter == m['cg_iters'] == 140
    assert sha(model.predict().astype(np.int64)) == m['cg_pred_sha']
    assert abs(np.abs(u).sum() - m['cg_prob_abs_sum']) <= 1e-9 * m['cg_prob_abs_sum']
    assert _sample_err(u, m['cg_u_samples']) <= 1e-5               # 140 iterations of a singular system, element-wise


def _check_config5(gl, m, W, labels, ti, mode='host_exp'):
    pri = gl.utils.class_priors(labels)
    model = gl.ssl.poisson_mbo(W, pri, solver='gradient_descent', Ns=40, mu=1, T=20)
    t0 = time.perf_counter()
    prob = model.fit(ti, labels[ti])
    pred = model.predict()
    print('poisson_mbo 70k: %.3f s, accuracy %.2f' % (time.perf_counter() - t0, gl.ssl.ssl_accuracy(pred, labels, ti)))
    if mode == 'device_exp':
        # thresholding is discontinuous in the weights: bit for bit against the oracle's run on the SAME matrix (the reference's
        # loop, ~13 s on the host); the distance to the reference's recorded run is recorded
        from oracle import gl_oracle as orc
        u_ref, lab_ref, w_ref = orc.poisson_mbo_fit(W, ti, labels[ti], pri, solver='gradient_descent', Ns=40, mu=1, T=20)
        assert np.array_equal(prob, u_ref) and np.array_equal(pred, lab_ref)
        assert np.array_equal(np.asarray(model.weights), np.asarray(w_ref))
        _record('config 5 (%d labelled), default mode: class sizes %s (the reference on its own W: %s), accuracy %.4f (%.4f), labels hash equal: %s'
                % (len(ti), [int(c) for c in np.bincount(pred, minlength=10)], m['class_sizes'], gl.ssl.ssl_accuracy(pred, labels, ti), m['accuracy'],
                   sha(pred.astype(np.int64)) == m['pred_sha']))
        assert abs(gl.ssl.ssl_accuracy(pred, labels, ti) - m['accuracy']) <= 0.05
        return
    # pinned to the reference's run in the build container (tests/golden/make_golden.py g4_config5)
    assert sha(pred.astype(np.int64)) == m['pred_sha']
    assert sha(np.ascontiguousarray(prob, dtype=np.float64)) == m['prob_sha'] and np.abs(prob).sum() == m['prob_abs_sum']
    assert [float(w) for w in model.weights] == m['weights']
    assert float(model.class_priors_error) == m['class_priors_error']
    assert [int(c) for c in np.bincount(pred, minlength=10)] == m['class_sizes']
    assert gl.ssl.ssl_accuracy(pred, labels, ti) == m['accuracy']
    # properties: one-hot state, class sizes within the projection's tolerance of the priors
    assert prob.shape == (70000, 10) and set(np.unique(prob)) <= {0.0, 1.0}
    assert np.max(np.abs(np.bincount(pred, minlength=10) / 70000 - pri)) <= max(model.class_priors_error, 1e-3) + 1e-12


def test_config5_poisson_mbo_pinned(gl, meta, config2):
    """Config 5 (ssl.poisson_mbo on the config-2 graph, reference ssl.py:774-839): labels, one-hot state, volume
    weights and class-priors error equal the reference's."""
    _check_config5(gl, meta['config5'], config2['W'], config2['labels'], config2['train_ind'], config2['mode'])


def test_config5_hard_overlapping_blobs_pinned(gl, meta, mode):
    """The same pipeline on overlapping blobs (accuracy 92 %): the volume-constrained projection moves the class
    weights away from 1 over many steps inside every one of the 21 projections -- all pinned to the reference."""
    m = meta['config5_hard']
    labels = np.load(os.path.join(GOLDEN, 'MNIST_labels.npz'))['labels'].astype(np.int64)
    rng = np.random.default_rng(5)
    centers = rng.normal(size=(10, 20)) * 0.8
    X = centers[labels] + rng.normal(size=(70000, 20))
    J, D = gl.weightmatrix.knnsearch(X, 11)
    assert sha(J.astype(np.int64)) == m['J_sha']
    W = gl.weightmatrix.knn(None, 10, knn_data=(J, D))
    assert W.nnz == m['nnz'] and sha(W.indices.astype(np.int32)) == m['W_indices_sha']
    ti = gl.trainsets.generate(labels, rate=2, seed=3)
    assert any(abs(w - 1.0) > 1e-3 for w in m['weights'])
    _check_config5(gl, m, W, labels, ti, mode)


def test_config3_laplace(gl, meta, mode):
    m = meta['config3']
    labels = np.load(os.path.join(GOLDEN, 'cifar_labels.npz'))['labels'].astype(np.int64)
    rng = np.random.default_rng(1)
    centers = rng.normal(size=(10, 32)) * 1.2
    X = centers[labels] + rng.normal(size=(60000, 32))
    J, D = gl.weightmatrix.knnsearch(X, 21)
    assert sha(J.astype(np.int64)) == m['J_sha']
    W = gl.weightmatrix.knn(None, 20, knn_data=(J, D))
    assert W.nnz == m['nnz'] and np.diff(W.indptr).max() == m['row_nnz_max'] == 660
    assert sha(W.indices.astype(np.int32)) == m['W_indices_sha']
    ti = gl.trainsets.generate(labels, rate=10, seed=0)
    model = gl.ssl.laplace(W, reduce='exact')
    t0 = time.perf_counter()
    u = model.fit(ti, labels[ti])
    print('laplace CG 60k: %d iterations, %.3f s' % (model.num_iter, time.perf_counter() - t0))
    assert model.num_iter == m['cg_iters'] == 54
    assert sha(model.predict().astype(np.int64)) == m['pred_sha']
    assert abs(np.abs(u).sum() - m['prob_abs_sum']) < 1e-8 * m['prob_abs_sum']
    assert _sample_err(u, m['u_samples']) <= 1e-5                   # element-wise against the reference's run
    assert np.array_equal(u[ti], np.eye(10)[labels[ti]])           # labelled rows are exactly one-hot
    assert u.min() > -1e-9 and u.max() < 1 + 1e-9                   # harmonic extension: maximum principle
    # tolerance mode (reduce='tree'): same labels, iterates within the north star's 1e-5
    fast = gl.ssl.laplace(W, reduce='tree')
    fast.fit(ti, labels[ti])
    t0 = time.perf_counter()
    ut = fast.fit(ti, labels[ti])
    print('laplace CG 60k, tree reductions: %d iterations, %.3f s' % (fast.num_iter, time.perf_counter() - t0))
    assert np.max(np.abs(ut - u)) <= 1e-5 and sha(fast.predict().astype(np.int64)) == m['pred_sha']
    assert abs(fast.num_iter - m['cg_iters']) <= 2


def test_config2_published_mnist_trainsets(gl, golden, config2):
    """the first published MNIST train sets (LabelPermutations/MNIST_permutations.npz,
    label rates 1..5 per class) through the HIP sweep vs the oracle run here on the host: bit-identical
    iterates, stop iteration and labels at n = 70000."""
    from oracle import gl_oracle as orc
    g = golden('g6_helpers.npz')
    W, labels = config2['W'], config2['labels']
    model = gl.ssl.poisson(W, solver='gradient_descent')
    for i in (0, 4, 9):
        ti = g['mnist_perm_%d' % i]
        u = model.fit(ti, labels[ti])
        u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], return_T=True)
        assert model.num_iter == T_ref
        assert np.array_equal(u, u_ref), i
        assert np.array_equal(model.predict(), orc.predict(u_ref))


def test_config2_batched_cg_trials_equal_single_fits(gl, golden, meta, config2):
    """SURVEY 8f-1 at n = 70000: five published MNIST train sets (label rates 1..5) plus the config-2
    train set as column groups of ONE conjugate-gradient solve: every trial's iterate and iteration
    count equal the fit on its own (which test_config2_poisson_cg pins to the reference: 140 iterations)."""
    g = golden('g6_helpers.npz')
    W, labels = config2['W'], config2['labels']
    model = gl.ssl.poisson(W)
    trials = [config2['train_ind']] + [g['mnist_perm_%d' % i] for i in (0, 2, 4, 6, 8)]
    t0 = time.perf_counter()
    together = model._fit_batch([(ti, labels[ti]) for ti in trials])
    iters = list(model.num_iter)
    print('6 stacked Poisson CG trials at 70k: iterations %s, %.3f s' % (iters, time.perf_counter() - t0))
    assert iters[0] == meta['config2']['cg_iters']
    for j in (0, 3, 5):
        alone = model.fit(trials[j], labels[trials[j]])
        assert model.num_iter == iters[j]
        assert np.array_equal(alone, together[j]), j


def test_config2_stacked_gd_trials_equal_single_fits_and_oracle(gl, golden, config2):
    """SURVEY 8 f-1 for the sweep at n = 70000: six published MNIST train sets (label rates 1..5) as column
    groups of stacked sweeps (glx_sweep_groups; ssl.ssl_trials' batches): every trial's iterate, its sweep count T and its labels
    equal the fit on its own and the oracle's run of the reference loop, bit for bit."""
    from oracle import gl_oracle as orc
    g = golden('g6_helpers.npz')
    W, labels = config2['W'], config2['labels']
    trials = [g['mnist_perm_%d' % i] for i in (0, 2, 4, 6, 8, 9)]
    model = gl.ssl.poisson(W, solver='gradient_descent')
    B = model._trial_batch_size(labels)
    assert B >= 2
    single = gl.ssl.poisson(W, solver='gradient_descent')
    t0 = time.perf_counter()
    for pos in range(0, len(trials), B):
        group = trials[pos:pos + B]
        res = model._fit_batch_device([(ti, labels[ti]) for ti in group])
        assert res is not None and len(res) == len(group)
        Ts = list(model.num_iter)
        for j, ti in enumerate(group):
            model._set_result(res[j])
            model.fitted = True
            pred = model.predict()                                  # decided on the stacked device state
            u = np.array(res[j].fetch())
            u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], return_T=True)
            assert Ts[j] == T_ref and np.array_equal(u, u_ref), (pos, j)
            assert np.array_equal(pred, orc.predict(u_ref)), (pos, j)
            u1 = single.fit(ti, labels[ti])
            assert single.num_iter == T_ref and np.array_equal(u1, u), (pos, j)
    print('6 stacked GD trials at 70k in batches of %d (incl. the oracle runs): %.2f s' % (B, time.perf_counter() - t0))
