"""GPU parity tests added in round 2 (through the C-ABI): the kNN cache format, conjgrad with an
initial iterate, the NaN rule of the stop test, the float32 label decision of the use_cuda branch, the tolerance
mode of the SPD solves, the verbose per-sweep contract, launch accounting."""
import os
import numpy as np
import pytest
from scipy import sparse
from conftest import csr_from, blobs, GOLDEN

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


# ---- f-2: kNN cache ./knn_data/<name>_<metric>.npz (reference weightmatrix.py:414-427, 431-467, :126-127) --------
def test_knn_cache_roundtrip_and_reference_file(gl, golden, tmp_path, monkeypatch):
    g = golden('g10_knn_cache.npz')
    X = g['X']
    monkeypatch.setattr(gl.weightmatrix, 'knn_dir', str(tmp_path / 'knn_data'))
    J, D = gl.weightmatrix.knnsearch(X, 8, dataset='GlxToy', metric='Raw')       # names are lower-cased (:418)
    path = tmp_path / 'knn_data' / 'glxtoy_raw.npz'
    assert path.exists()
    ours = np.load(path)
    ref = np.load(os.path.join(GOLDEN, 'knn_data', 'glxtoy_raw.npz'))            # written by the REFERENCE's knnsearch
    assert sorted(ours.files) == sorted(ref.files) == ['D', 'J']
    assert ours['J'].shape == ref['J'].shape == (300, 8) and ours['D'].dtype == ref['D'].dtype == np.float64
    assert ours['J'].dtype.kind == ref['J'].dtype.kind == 'i'
    assert np.array_equal(ours['J'], ref['J'])
    assert np.max(np.abs(ours['D'] - ref['D'])) <= 1e-12
    # what knnsearch returned is what it stored, and load_knn_data returns it
    assert np.array_equal(J, ours['J']) and np.array_equal(D, ours['D'])
    J2, D2 = gl.weightmatrix.load_knn_data('GLXTOY')
    assert np.array_equal(J2, J) and np.array_equal(D2, D)
    # string path of knn(): the weight matrix from our file == the reference's from its file
    W = gl.weightmatrix.knn('glxtoy', 7)
    Wr = csr_from(g, 'W_k7')
    assert np.array_equal(W.indptr, Wr.indptr) and np.array_equal(W.indices, Wr.indices)
    assert np.max(np.abs(W.data - Wr.data)) <= 1e-12
    # and from the REFERENCE's file, through our loader: bit-identical weights, fewer neighbours than stored
    monkeypatch.setattr(gl.weightmatrix, 'knn_dir', os.path.join(GOLDEN, 'knn_data'))
    for name, k, kern, key in (('glxtoy', 7, 'gaussian', 'W_k7'), ('GlxToy', 5, 'uniform', 'W_k5_uniform')):
        W = gl.weightmatrix.knn(name, k, kernel=kern)
        Wr = csr_from(g, key)
        assert np.array_equal(W.indptr, Wr.indptr) and np.array_equal(W.indices, Wr.indices)
        assert np.array_equal(W.data, Wr.data), name


# ---- utils.conjgrad with x0 (reference utils.py:510-530) ---------------------------------------------------------
def test_conjgrad_with_initial_iterate_vs_oracle(gl, orc):
    rng = np.random.default_rng(11)
    n = 2500
    A = sparse.random(n, n, density=0.004, random_state=6, format='csr')
    A = sparse.csr_matrix(A + A.T + sparse.identity(n) * 5.0)
    b = rng.normal(size=(n, 5))
    x0 = rng.normal(size=(n, 5))
    x_ref, it_ref, err_ref = orc.conjgrad(A, b, x0=x0, tol=1e-9, return_iters=True)
    x0_copy = x0.copy()
    x, it, err = gl.utils.conjgrad(A, b, x0=x0, tol=1e-9, return_info=True)
    assert np.array_equal(x0, x0_copy)                      # `x = x0.copy()`: the caller's array is untouched
    assert it == it_ref and err == err_ref
    assert np.array_equal(x, x_ref)                         # x accumulates from x0 with the reference's roundings
    # 1-D right-hand side with x0 (numpy's pairwise reductions)
    x1_ref, it1_ref, _ = orc.conjgrad(A, b[:, 0].copy(), x0=x0[:, 0].copy(), tol=1e-9, return_iters=True)
    x1, it1, _ = gl.utils.conjgrad(A, b[:, 0].copy(), x0=x0[:, 0].copy(), tol=1e-9, return_info=True)
    assert it1 == it1_ref and np.array_equal(x1, x1_ref)
    # x0 = exact solution: r0 = 0 ... the loop still runs once in the reference (err starts at 1)
    xs = x_ref.copy()
    a_ref, ita_ref, _ = orc.conjgrad(A, A @ xs, x0=xs, tol=1e-3, return_iters=True)
    a, ita, _ = gl.utils.conjgrad(A, A @ xs, x0=xs, tol=1e-3, return_info=True)
    assert ita == ita_ref and np.array_equal(a, a_ref, equal_nan=True)


# ---- stop test with NaN / an isolated vertex (ssl.py:667: `np.max(|v - vinf|) > 1/n` is False on NaN) -----------
def test_isolated_vertex_and_nan_stop_rule(gl, golden, orc):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian').tolil()
    iso = 17
    W[iso, :] = 0
    W[:, iso] = 0
    W = sparse.csr_matrix(W)
    W.eliminate_zeros()
    ti = g['train_ind']
    assert iso not in set(ti.tolist())
    lab = g['labels']
    with np.errstate(all='ignore'):
        u_ref, T_ref = orc.poisson_gd(W, ti, lab[ti], return_T=True)
        m = gl.ssl.poisson(W, solver='gradient_descent')
        u = m.fit(ti, lab[ti])
    assert m.num_iter == T_ref
    assert np.array_equal(u, u_ref, equal_nan=True)
    assert np.all(np.isnan(u_ref[iso]))                     # D^-1 is inf there: 0 * inf
    # a NaN in the stop vector itself ends the loop at min_iter in the reference (nan > 1/n is False): same here
    from graphlearning_amd import _hip
    n = W.shape[0]
    P = sparse.csr_matrix(sparse.identity(n) * 0.5)
    dev = _hip.DeviceGraph(P)
    w0 = np.full(n, 1.0 / n)
    w0[3] = np.nan
    deg = np.ones(n)
    vinf = np.full(n, 1.0 / n)
    _, T = dev.poisson_sweep(np.zeros((n, 2)), w0, deg, vinf, min_iter=7, max_iter=200)
    assert T == 7
    dev.close()


# ---- label decision on a float32 state (reference use_cuda branch: self.prob stays float32, ssl.py:256-257) -----
def test_fp32_predict_rounds_like_numpy_float32(gl, golden, orc):
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    ti, lab = g['train_ind'], g['labels']
    m = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=True)
    u32 = m.fit(ti, lab[ti])
    assert u32.dtype == np.float32
    assert np.array_equal(m.predict(), orc.predict(u32))            # numpy on the float32 array
    # near-ties: float32 rounding of (p - min)/max decides; class weights are fp64
    rng = np.random.default_rng(2)
    prob = rng.normal(size=(20000, 6)).astype(np.float32)
    prob[:, 1] = prob[:, 0] + rng.integers(-1, 2, size=20000).astype(np.float32) * np.float32(1e-7)
    from graphlearning_amd import _hip
    w = np.array([1.0, 1.0 + 1e-9, 0.97, 1.02, 1.0, 0.99])
    lab32, _, _, _ = _hip.argmax_project(prob, None, w, max_steps=0)
    assert np.array_equal(lab32, orc.predict(prob, w))
    lab64, _, _, _ = _hip.argmax_project(prob.astype(np.float64), None, w, max_steps=0)
    assert np.array_equal(lab64, orc.predict(prob.astype(np.float64), w))
    # the volume projection on a float32 prob follows numpy's float32 scores too
    pri = np.array([0.3, 0.1, 0.2, 0.15, 0.15, 0.1])
    l_ref, w_ref, e_ref, it_ref = orc.volume_label_projection(prob, pri, 1)
    l, wq, e, it = _hip.argmax_project(prob, pri, None, max_steps=10000)
    assert it == it_ref and np.array_equal(wq, w_ref) and e == e_ref and np.array_equal(l, l_ref)


# ---- tolerance mode of the SPD solves (reduce='tree') ------------------------------------------------------------
@pytest.mark.parametrize('norm', ['combinatorial', 'normalized'])
def test_tree_reductions_laplace_twomoons(gl, golden, norm):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    m = gl.ssl.laplace(W, normalization=norm, reduce='tree')
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert np.max(np.abs(u - g['laplace_%s_prob' % norm])) <= 1e-5        # north-star tolerance on the iterates
    assert np.array_equal(m.predict(), g['laplace_%s_pred' % norm])      # identical labels
    assert abs(m.num_iter - int(g['laplace_%s_iters' % norm])) <= 2


def test_tree_reductions_blobs5000_and_randomwalk(gl, golden, orc):
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    ti, lab = g['train_ind'], g['labels']
    exact = gl.ssl.laplace(W, reduce='exact')
    ue = exact.fit(ti, lab[ti])
    tree = gl.ssl.laplace(W, reduce='tree')
    ut = tree.fit(ti, lab[ti])
    assert np.array_equal(ue, orc.laplace_fit(W, ti, lab[ti]))
    assert np.max(np.abs(ut - ue)) <= 1e-5 and np.array_equal(tree.predict(), exact.predict())
    assert np.array_equal(ut[ti], ue[ti])                                 # labelled rows are exactly one-hot either way
    # stacked trials in tolerance mode: each trial within 1e-5 of its exact fit
    sets = [gl.trainsets.generate(lab, rate=r, seed=s) for r, s in ((2, 1), (3, 2), (5, 3))]
    outs = tree._fit_batch([(t, lab[t]) for t in sets])
    for t, o in zip(sets, outs):
        assert np.max(np.abs(o - exact.fit(t, lab[t]))) <= 1e-5
    rw_e = gl.ssl.randomwalk(W, reduce='exact')
    rw_t = gl.ssl.randomwalk(W, reduce='tree')
    a, b = rw_e.fit(ti, lab[ti]), rw_t.fit(ti, lab[ti])
    assert np.max(np.abs(a - b)) <= 1e-5 and np.array_equal(rw_e.predict(), rw_t.predict())
    # a reweighted Laplace fit: graph.reweight('poisson') (singular system) keeps the reference-order reductions, the
    # Dirichlet solve on the reweighted graph (SPD) takes the tolerance mode
    le = gl.ssl.laplace(W, reweighting='poisson', reduce='exact')
    lt = gl.ssl.laplace(W, reweighting='poisson', reduce='tree')
    a, b = le.fit(ti, lab[ti]), lt.fit(ti, lab[ti])
    assert np.max(np.abs(a - b)) <= 1e-5 and np.array_equal(le.predict(), lt.predict())
    with pytest.raises(Exception):
        gl.ssl.laplace(W, reduce='fastest').fit(ti, lab[ti])


# ---- verbose contract: one '%d,Accuracy = %.2f' line per sweep of the CPU loop (ssl.py:672-677) ------------------
def test_all_labels_prints_every_sweep(gl, golden, orc, capsys):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    ti, lab = g['train_ind'], g['labels']
    s = orc.poisson_gd_setup(W, ti, lab[ti])
    n = W.shape[0]
    u = np.zeros((n, s['k']))
    v = s['v0']
    T = 0
    want = []
    while (T < 50 or np.max(np.absolute(v - s['vinf'])) > 1 / n) and T < 1000:
        u = s['Db'] + s['P'] * u
        v = s['RW'] * v
        T += 1
        want.append('%d,Accuracy = %.2f' % (T, orc.ssl_accuracy(orc.predict(u), lab, ti)))
    m = gl.ssl.poisson(W, solver='gradient_descent')
    capsys.readouterr()
    got_u = m.fit(ti, lab[ti], all_labels=lab)
    lines = [l for l in capsys.readouterr().out.splitlines() if 'Accuracy' in l]
    assert lines == want and len(lines) == T == int(g['poisson_gd_T'])
    assert np.array_equal(got_u, g['poisson_gd_prob'])


def test_sweep_launch_count_is_sweeps_run(gl, golden):
    """glx_sweep_launches counts the sweeps that ran, not the kernels of a tail chunk that exit at once."""
    from graphlearning_amd import _hip
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    ti, lab = g['train_ind'], g['labels']
    m = gl.ssl.poisson(W, solver='gradient_descent')
    dev, aux = m._operators()
    src, k = gl.ssl._poisson_source(W.shape[0], ti, lab[ti])
    v0 = np.zeros(W.shape[0]); v0[ti] = 1; v0 /= v0.sum()
    sw = _hip.Sweep(dev, k, 50, 1000, True)
    sw.set_problem(aux['D'] * src, v0 / aux['deg'], aux['deg'], aux['vinf'])
    l0 = sw.launches()
    T, _ = sw.run()
    assert T == int(g['poisson_gd_T']) == 409 and sw.launches() - l0 == T
    T2, _ = sw.run()
    assert T2 == T and sw.launches() - l0 == 2 * T
    sw.close()


def test_new_entry_points_reject_bad_arguments(gl, golden):
    """Status codes instead of crashes (include/glx.h conventions) for the entry points added in round 2."""
    from graphlearning_amd import _hip
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    n = W.shape[0]
    m = gl.ssl.poisson(W, solver='gradient_descent')
    dev, aux = m._operators()
    sw = _hip.Sweep(dev, 2, 50, 1000, True)
    with pytest.raises(_hip.GlxError):                       # graph vectors first
        sw.set_problem_rows(np.array([1, 2]), np.zeros((2, 2)), np.zeros(2), 0.0)
    sw.set_vectors(aux['deg'], aux['vinf'])
    with pytest.raises(_hip.GlxError):                       # row out of range
        sw.set_problem_rows(np.array([1, n]), np.zeros((2, 2)), np.zeros(2), 0.0)
    with pytest.raises(_hip.GlxError):                       # shape mismatch caught at the boundary
        sw.set_problem_rows(np.array([1, 2]), np.zeros((3, 2)), np.zeros(2), 0.0)
    sw.set_problem_rows(np.zeros(0, dtype=np.int64), np.zeros((0, 2)), np.zeros(0), 1.0)   # an empty training set is legal: u stays 0
    T, _ = sw.run()
    assert T >= 50 and not np.any(sw.fetch())
    sw.close()
    # distributed sweep object
    comm = _hip.Comm(1, 0, None, 0)
    P = sparse.csr_matrix(sparse.identity(8) * 0.5)
    with pytest.raises(_hip.GlxError):                       # receive counts must add up to the halo
        _hip.DistSweep(comm, sparse.csr_matrix((8, 10)), 0, [0], np.zeros(0, np.int32), [1], 8, 2)
    with pytest.raises(_hip.GlxError):                       # a send row must be a boundary row
        _hip.DistSweep(comm, P, 2, [1], np.array([5], np.int32), [0], 8, 2)
    ds = _hip.DistSweep(comm, P, 0, [0], np.zeros(0, np.int32), [0], 8, 2)
    with pytest.raises(_hip.GlxError):                       # run before the problem is set
        ds.run(5, 10, 8, 0.0)
    ds.set_problem(None, np.full(8, 0.125), np.ones(8), np.full(8, 0.125))
    with pytest.raises(_hip.GlxError):
        ds.run(5, 10, 0, 0.0)                                # check_every >= 1
    T, _ = ds.run(5, 10, 3, 0.0)
    assert T == 5 and not np.any(ds.fetch())                 # w halves every sweep: |deg*w - vinf| = 0.125 - ... > 1/8? no: stops at min_iter
    ds.close()
    comm.close()
    # label decision: dtype code
    with pytest.raises(_hip.GlxError):
        _hip.check(_hip.load().glx_argmax_project_t(None, 7, 4, 2, None, None, None, None, None, 0, 1, 0), 'glx_argmax_project_t')
    # cg flags: x0 with the wrong shape is caught in Python, unknown reduce mode too
    A = _hip.DeviceGraph(sparse.identity(16, format='csr') * 2.0)
    with pytest.raises(Exception):
        A.cg(np.ones((16, 2)), x0=np.ones((16, 3)))
    with pytest.raises(_hip.GlxError):
        A.cg(np.ones((16, 2)), reduce='approximately')
    x, it, err = A.cg(np.ones((16, 2)), x0=np.full((16, 2), 0.5), tol=1e-12)    # B is the residual r0 = b - A@x0: x = x0 + A^-1 r0
    assert np.allclose(x, 1.0) and it >= 1
    A.close()


# ---- the stop test: fused column deg*(P w) vs the reference's own recurrence v <- RW v (ssl.py:644, 667-669) -----
def _reference_stop_values(orc, W, ti, lab, count):
    s = orc.poisson_gd_setup(W, ti, lab[ti])
    v, out = s['v0'], []
    for _ in range(count):
        out.append(np.max(np.absolute(v - s['vinf'])))
        v = s['RW'] * v
    return np.array(out)


def _stop_cases(golden):
    g = golden('g1_twomoons.npz')
    yield 'twomoons', csr_from(g, 'W_gaussian'), g['train_ind'], g['labels']
    yield 'twomoons_directed', csr_from(g, 'W_gaussian_nosym'), g['train_ind'], g['labels']
    X, lab = blobs(3000, 12, 6, 5, 2.0)
    import graphlearning_amd as gl
    W = gl.weightmatrix.knn(X, 8)
    ti = gl.trainsets.generate(lab, rate=2, seed=3)
    yield 'blobs3000', W, ti, lab


def test_fused_stop_values_track_the_reference_recurrence(gl, golden, orc):
    worst = 0.0
    for name, W, ti, lab in _stop_cases(golden):
        m = gl.ssl.poisson(W, solver='gradient_descent')
        m.fit(ti, lab[ti])
        _, aux = m._operators()
        first, vals = aux['sweep'].stop_values()
        n = W.shape[0]
        capped = m.num_iter == 1000                      # the directed graph never meets the test: max_iter ends the loop,
        assert first == 50 and len(vals) == m.num_iter - 50 + (0 if capped else 1), name   # its value is never compared
        ref = _reference_stop_values(orc, W, ti, lab, m.num_iter + 1)[first:first + len(vals)]
        rel = np.max(np.abs(vals - ref) / ref)
        worst = max(worst, rel)
        if capped:
            assert np.all(vals > 1 / n), name
        else:
            assert np.all(vals[:-1] > 1 / n) and vals[-1] <= 1 / n, name
        assert m.stop_settled is None, name              # nothing within STOP_BAND of 1/n: the usual case
    assert worst <= 1e-12, worst                         # STOP_BAND (1e-9) is >= 1000x the rounding difference
    print('fused vs reference stop values: max relative difference %.2e' % worst)


def test_stop_value_inside_the_band_is_settled_by_the_reference_recurrence(gl, golden, orc, monkeypatch):
    from graphlearning_amd import ssl as ssl_mod
    monkeypatch.setattr(ssl_mod, 'STOP_BAND', 1e30)      # every fit takes the settling path
    for name, W, ti, lab in _stop_cases(golden):
        for kw in ({}, {'min_iter': 0}, {'min_iter': 0, 'max_iter': 37}, {'min_iter': 60, 'max_iter': 60}):
            u_ref, T_ref = orc.poisson_gd(W, ti, lab[ti], return_T=True, **kw)
            m = gl.ssl.poisson(W, solver='gradient_descent', **kw)
            u = m.fit(ti, lab[ti])
            assert m.num_iter == T_ref, (name, kw)
            assert np.array_equal(u, u_ref), (name, kw)
            if kw.get('min_iter', 50) < kw.get('max_iter', 1000):
                assert m.stop_settled == (T_ref, T_ref), (name, kw, m.stop_settled)
                assert m._exact_stop_iteration(ti) == orc.poisson_gd_iterations(W, ti, **kw)
            else:
                assert m.stop_settled is None            # min_iter >= max_iter: the test decides nothing
    # the device SpMV of the recurrence adds in csc_matvec's order: every iterate bit for bit
    name, W, ti, lab = next(_stop_cases(golden))
    s = orc.poisson_gd_setup(W, ti, lab[ti])
    from graphlearning_amd import _hip
    RW = sparse.csr_matrix(s['RW'])
    RW.sort_indices()
    rw = _hip.DeviceGraph(RW, keep_order=True)
    v = s['v0']
    vd = s['v0']
    for _ in range(30):
        v = s['RW'] * v
        vd = rw.spmm_bias(vd)
        assert np.array_equal(v, vd)
    rw.close()


def test_disagreeing_stop_iteration_reruns_exactly_that_many_sweeps(gl, golden, orc, monkeypatch):
    from graphlearning_amd import ssl as ssl_mod
    monkeypatch.setattr(ssl_mod, 'STOP_BAND', 1e30)
    g = golden('g1_twomoons.npz')
    W, ti, lab = csr_from(g, 'W_gaussian'), g['train_ind'], g['labels']
    for use_cuda in (False, True):
        m = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=use_cuda)
        T_true = orc.poisson_gd_iterations(W, ti)
        monkeypatch.setattr(m, '_exact_stop_iteration', lambda train_ind: T_true - 3)   # pretend the roundings disagreed
        u = m.fit(ti, lab[ti])
        assert m.num_iter == T_true - 3 and m.stop_settled == (T_true, T_true - 3)
        u_ref = orc.poisson_gd(W, ti, lab[ti], min_iter=T_true - 3, max_iter=T_true - 3)
        if use_cuda:
            assert u.dtype == np.float32 and np.allclose(u, u_ref, atol=1e-5)
        else:
            assert np.array_equal(u, u_ref)
        assert np.array_equal(m.predict(), np.argmax(u_ref, axis=1))


# ---- repeated builds and fits release what they take -----------------------------------------------------------------
@pytest.mark.skipif('PYTEST_XDIST_WORKER' in os.environ, reason='reads the DEVICE-wide memory in use: other xdist workers on the same GPU move it')
def test_no_device_memory_growth_over_repeated_builds_and_fits(gl):
    """New graph, new models, every learner, a dozen rounds: what the pools cache after the first rounds is all that stays
    on the device (work-buffer pool, page-locked result pool, the per-device list of idle streams)."""
    import gc
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')                     # the runtime libglx is linked against (already loaded)
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 6, size=8000)
    lab[:6] = np.arange(6)
    centres = rng.normal(size=(6, 12)) * 2.5

    def in_use():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return total.value - free.value

    marks = []
    for r in range(12):
        X = centres[lab] + rng.normal(size=(8000, 12))
        W = gl.weightmatrix.knn(X, 9)
        ti = gl.trainsets.generate(lab, rate=2, seed=r)
        for model in (gl.ssl.poisson(W, solver='gradient_descent'), gl.ssl.poisson(W), gl.ssl.laplace(W, reduce='exact'),
                      gl.ssl.poisson_mbo(W, gl.utils.class_priors(lab), solver='gradient_descent', T=2), gl.ssl.randomwalk(W, reduce='exact')):
            model.fit_predict(ti, lab[ti])
        del model, W
        gc.collect()
        marks.append(in_use())
    assert max(marks[4:]) - marks[3] <= 8 << 20, [m >> 20 for m in marks]


def test_operator_cache_follows_in_place_edits_of_W(golden):
    """mutating W.data in place between two fits must not reuse the device operator of the old values (the
    reference rebuilds P in every fit, ssl.py:615-644): the second fit equals a fresh model's fit on the edited matrix -- for the
    gradient-descent and the CG solver of ssl.poisson, ssl.laplace and ssl.randomwalk -- while ssl_trials, inside which W cannot
    change, fingerprints the graph once."""
    import graphlearning_amd as gl
    from graphlearning_amd import _hip, utils as glutils
    from conftest import csr_from
    _hip.require_device()
    g = golden('g1_twomoons.npz')
    ti, lab = g['train_ind'], g['labels']
    makers = [lambda W: gl.ssl.poisson(W, solver='gradient_descent'), lambda W: gl.ssl.poisson(W), lambda W: gl.ssl.laplace(W, reduce='exact'),
              lambda W: gl.ssl.randomwalk(W, reduce='exact')]
    for mk in makers:
        W = sparse_csr(csr_from(g, 'W_gaussian'))
        model = mk(W)
        u0 = np.array(model.fit(ti, lab[ti]))
        # a symmetric in-place edit: scale the weights of vertex 0's edges (both directions)
        idx = np.flatnonzero((np.repeat(np.arange(W.shape[0]), np.diff(W.indptr)) == 0) | (W.indices == 0))
        W.data[idx] *= 0.25
        u1 = np.array(model.fit(ti, lab[ti]))
        fresh = np.array(mk(sparse_csr(W.copy())).fit(ti, lab[ti]))
        assert np.array_equal(u1, fresh), model.name
        assert not np.array_equal(u1, u0), model.name
    # ssl_trials fingerprints once per loop (the per-trial rate does not pay for the hash)
    calls = []
    real = glutils.matrix_fingerprint
    try:
        glutils.matrix_fingerprint = lambda M: (calls.append(1), real(M))[1]
        m = gl.ssl.poisson(sparse_csr(csr_from(g, 'W_gaussian')), solver='gradient_descent')
        m.ssl_trials([ti, ti[::-1].copy(), ti], lab, save_results=False)
    finally:
        glutils.matrix_fingerprint = real
    assert len(calls) == 1, calls


def sparse_csr(W):
    from scipy import sparse
    return sparse.csr_matrix(W)


def test_speculative_fit_leaves_no_trace_when_the_matrix_was_edited(golden, monkeypatch):
    """a fit on the matrix OBJECT of the previous fit starts on the cached operators while the content is hashed
    again; when the hash says the matrix was edited in place the fit is repeated -- from the model state the first attempt
    found (PoissonMBO carries its volume weights from fit to fit, like the reference), not from what the discarded run left."""
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    from conftest import csr_from
    _hip.require_device()
    g = golden('g1_twomoons.npz')
    ti, lab = g['train_ind'], g['labels']
    pri = g['class_priors']

    def sequence(speculate):
        W = sparse_csr(csr_from(g, 'W_gaussian'))
        model = gl.ssl.poisson_mbo(W, pri, solver='gradient_descent', Ns=12, T=4)
        if not speculate:
            monkeypatch.setattr(model, '_speculate_key', lambda: None)
            monkeypatch.setattr(model.poisson_model, '_speculate_key', lambda: None)
        out = [np.array(model.fit(ti, lab[ti])), np.array(model.weights, dtype=np.float64)]
        idx = np.flatnonzero((np.repeat(np.arange(W.shape[0]), np.diff(W.indptr)) < 40) | (W.indices < 40))
        W.data[idx] *= 0.05                                    # symmetric in-place edit
        out += [np.array(model.fit(ti, lab[ti])), np.array(model.weights, dtype=np.float64), float(model.class_priors_error)]
        return out
    a, b = sequence(True), sequence(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_reduce_auto_is_the_tolerance_mode_until_a_solve_runs_long(gl, golden, monkeypatch):
    """reduce='auto' (ssl._solve): the tolerance mode's answer while the solve stays short, the reference-order answer -- bit for bit --
    once it takes more than AUTO_TREE_MAX_ITER iterations (there the reordered sums may move the stopping iteration) or produces a
    non-finite iterate."""
    from graphlearning_amd import ssl as glssl
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    ti, lab = g['train_ind'], g['labels']
    monkeypatch.setattr(glssl, 'AUTO_TREE_MAX_ITER', 10 ** 6)        # the mechanism, whatever the thresholds of the day
    monkeypatch.setattr(glssl, 'AUTO_STOP_BAND', 0.0)
    exact, tree, auto = gl.ssl.laplace(W, reduce='exact'), gl.ssl.laplace(W, reduce='tree'), gl.ssl.laplace(W, reduce='auto')
    ue, ut, ua = exact.fit(ti, lab[ti]), tree.fit(ti, lab[ti]), auto.fit(ti, lab[ti])
    assert np.array_equal(ua, ut) and auto.num_iter == tree.num_iter and np.max(np.abs(ua - ue)) <= 1e-5
    # the same solve with the bound below its iteration count: handed back to the exact mode
    monkeypatch.setattr(glssl, 'AUTO_TREE_MAX_ITER', max(1, exact.num_iter // 2))
    ub = auto.fit(ti, lab[ti])
    assert np.array_equal(ub, ue) and auto.num_iter == exact.num_iter
    rw_e, rw_a = gl.ssl.randomwalk(W, reduce='exact'), gl.ssl.randomwalk(W, reduce='auto')
    assert np.array_equal(rw_a.fit(ti, lab[ti]), rw_e.fit(ti, lab[ti]))          # (bound still lowered: the exact answer)
    monkeypatch.setattr(glssl, 'AUTO_TREE_MAX_ITER', 10 ** 6)
    assert np.max(np.abs(rw_a.fit(ti, lab[ti]) - rw_e.fit(ti, lab[ti]))) <= 1e-5
    # a stop decision inside the band goes back too (here: a band so wide that every decision is inside it)
    monkeypatch.setattr(glssl, 'AUTO_STOP_BAND', 10.0)
    ub = auto.fit(ti, lab[ti])
    assert np.array_equal(ub, ue) and auto.num_iter == exact.num_iter
    monkeypatch.setattr(glssl, 'AUTO_STOP_BAND', 0.0)
    # a non-finite tolerance-mode result goes back as well (simulated: the runner poisons the tolerance-mode answer)
    calls = []

    def run(mode):
        calls.append(mode)
        x = np.ones((3, 2))
        if mode == 'tree':
            x[1, 1] = np.nan
        return x, np.array([5]), np.array([0.0])
    out = glssl._solve(run, 'auto')
    assert calls == ['tree', 'exact'] and np.isfinite(out[0]).all()
