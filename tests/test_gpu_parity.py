"""GPU parity tests: the HIP path (through the C-ABI, via graphlearning_amd) against the
committed golden vectors of the reference and against the oracle on seeded inputs.
Bars: bit-exact for labels / iteration counts / fp64 sweep iterates; 1e-5 (north star)
or tighter, as stated per test, for CG iterates."""
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


def test_spmm_bias_bitexact_vs_scipy(gl, orc):
    from graphlearning_amd import _hip
    rng = np.random.default_rng(0)
    for n, dens, C in [(1, 1.0, 3), (7, 0.5, 1), (300, 0.05, 10), (2000, 0.01, 13), (513, 0.03, 40)]:
        A = sparse.random(n, n, density=dens, random_state=1, format='csr', dtype=np.float64)
        u = rng.normal(size=(n, C))
        Db = rng.normal(size=(n, C))
        G = _hip.DeviceGraph(A)
        got = G.spmm_bias(u, Db, iters=1)
        assert np.array_equal(got, Db + A * u), (n, C)
        got3 = G.spmm_bias(u, None, iters=3)
        assert np.array_equal(got3, A * (A * (A * u))), (n, C)
        G.close()


def test_spmm_unsorted_rows_and_hubs(gl):
    """Entry order inside a row is the accumulation order: unsorted / duplicate-free rows with
    one hub row of ~n entries must match scipy's sequential sum bit for bit."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(3)
    n = 700
    A = sparse.random(n, n, density=0.02, random_state=2, format='lil', dtype=np.float64)
    A[5, :] = rng.normal(size=n)          # hub
    A = sparse.csr_matrix(A)
    # shuffle entries inside each row
    for i in range(n):
        s, e = A.indptr[i], A.indptr[i + 1]
        perm = rng.permutation(e - s)
        A.indices[s:e] = A.indices[s:e][perm]
        A.data[s:e] = A.data[s:e][perm]
    A.has_sorted_indices = False
    u = rng.normal(size=(n, 10))
    G = _hip.DeviceGraph(A)
    assert np.array_equal(G.spmm_bias(u), A * u)
    G.close()


def test_spmm_fp32(gl):
    from graphlearning_amd import _hip
    rng = np.random.default_rng(1)
    A = sparse.random(1000, 1000, density=0.02, random_state=4, format='csr', dtype=np.float64)
    u = rng.normal(size=(1000, 10)).astype(np.float32)
    G = _hip.DeviceGraph(A, dtype=np.float32)
    got = G.spmm_bias(u)
    A32 = A.astype(np.float32)
    assert got.dtype == np.float32
    assert np.array_equal(got, A32 * u)     # scipy fp32 csr_matvecs: same order, same roundings
    G.close()


def test_twomoons_poisson_gd_golden(gl, golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    train_ind = g['train_ind']
    labels = g['labels']
    m = gl.ssl.poisson(W, solver='gradient_descent')
    u = m.fit(train_ind, labels[train_ind])
    assert m.num_iter == int(g['poisson_gd_T']) == 409
    assert np.array_equal(u, g['poisson_gd_prob'])          # bit-identical fp64 iterates
    assert np.array_equal(m.predict(), g['poisson_gd_pred'])
    assert gl.ssl.ssl_accuracy(m.predict(), labels, train_ind) == float(g['accuracy_poisson_gd'])


def test_twomoons_poisson_gd_directed_golden(gl, golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian_nosym')
    m = gl.ssl.poisson(W, solver='gradient_descent')
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert m.num_iter == int(g['poisson_gd_directed_T'])
    assert np.array_equal(u, g['poisson_gd_directed_prob'])


def test_twomoons_poisson_gd_fp32(gl, golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    m = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=True)
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert u.dtype == np.float32
    assert m.num_iter == 409
    assert np.max(np.abs(u - g['poisson_gd_prob'])) < 1e-5   # north-star tolerance
    assert np.array_equal(m.predict(), g['poisson_gd_pred'])


def test_twomoons_poisson_cg_golden(gl, golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    m = gl.ssl.poisson(W)
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert m.num_iter == int(g['poisson_cg_iters'])
    assert np.array_equal(u, g['poisson_cg_prob'])          # reference-order reductions: bit-identical CG
    assert np.array_equal(m.predict(), g['poisson_cg_pred'])


@pytest.mark.parametrize('norm', ['combinatorial', 'randomwalk', 'normalized'])
def test_twomoons_laplace_golden(gl, golden, norm):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    m = gl.ssl.laplace(W, normalization=norm, reduce='exact')
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert m.num_iter == int(g['laplace_%s_iters' % norm])
    assert np.array_equal(u, g['laplace_%s_prob' % norm])
    assert np.array_equal(m.predict(), g['laplace_%s_pred' % norm])


def test_twomoons_laplace_tau_meanshift(gl, golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    m = gl.ssl.laplace(W, tau=0.01, mean_shift=True, reduce='exact')
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert np.array_equal(u, g['laplace_tau_ms_prob'])


@pytest.mark.parametrize('solver', ['gradient_descent', 'conjugate_gradient'])
def test_twomoons_poisson_mbo_golden(gl, golden, solver):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    m = gl.ssl.poisson_mbo(W, g['class_priors'], solver=solver)
    pred = m.fit_predict(g['train_ind'], g['labels'][g['train_ind']])
    assert np.array_equal(m.prob, g['poisson_mbo_%s_prob' % solver])
    assert np.array_equal(pred, g['poisson_mbo_%s_pred' % solver])
    assert np.allclose(m.weights, g['poisson_mbo_%s_weights' % solver], rtol=0, atol=1e-12)


def test_projection_golden(gl, golden):
    from graphlearning_amd import _hip
    g = golden('g5_projection.npz')
    lab0, _, _, _ = _hip.argmax_project(g['prob'], None, None, max_steps=0)
    assert np.array_equal(lab0, g['pred_plain'])
    lab1, w1, err1, it1 = _hip.argmax_project(g['prob'], g['priors'], None, max_steps=10000)
    assert it1 == int(g['iters_1'])
    assert np.array_equal(w1, g['weights_1'])            # bit-identical weights
    assert np.array_equal(lab1, g['labels_1'])
    assert err1 == float(g['err_1'])
    lab2, w2, _, _ = _hip.argmax_project(g['prob'], g['priors'], w1, max_steps=10000)
    assert np.array_equal(w2, g['weights_2'])
    assert np.array_equal(lab2, g['labels_2'])


def test_blobs5000_golden(gl, golden):
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    ti, lab = g['train_ind'], g['labels']
    m = gl.ssl.poisson(W, solver='gradient_descent')
    u = m.fit(ti, lab[ti])
    assert m.num_iter == int(g['poisson_gd_T'])
    assert np.array_equal(u, g['poisson_gd_prob'])
    assert np.array_equal(m.predict(), g['poisson_gd_pred'])
    m = gl.ssl.poisson(W)
    u = m.fit(ti, lab[ti])
    assert m.num_iter == int(g['poisson_cg_iters'])
    assert np.array_equal(u, g['poisson_cg_prob'])
    assert np.array_equal(m.predict(), g['poisson_cg_pred'])
    m = gl.ssl.laplace(W, reduce='exact')
    u = m.fit(ti, lab[ti])
    assert m.num_iter == int(g['laplace_iters'])
    assert np.array_equal(u, g['laplace_prob'])
    assert np.array_equal(m.predict(), g['laplace_pred'])
    m = gl.ssl.poisson_mbo(W, g['class_priors'], solver='gradient_descent')
    pred = m.fit_predict(ti, lab[ti])
    assert np.array_equal(pred, g['poisson_mbo_pred'])
    assert np.array_equal(m.prob, g['poisson_mbo_prob'])


def test_conjgrad_vs_oracle(gl, orc):
    rng = np.random.default_rng(7)
    n = 3000
    A = sparse.random(n, n, density=0.003, random_state=5, format='csr')
    A = A + A.T + sparse.identity(n) * 4.0
    b = rng.normal(size=(n, 6))
    x_ref, it_ref, err_ref = orc.conjgrad(sparse.csr_matrix(A), b, tol=1e-8, return_iters=True)
    x, it, err = gl.utils.conjgrad(A, b, tol=1e-8, return_info=True)
    assert it == it_ref
    assert np.array_equal(x, x_ref)
    assert err == err_ref


def test_errors(gl):
    from graphlearning_amd import _hip
    A = sparse.identity(4, format='csr')
    G = _hip.DeviceGraph(A)
    with pytest.raises(_hip.GlxError):
        G.spmm_bias(np.zeros((5, 2)))
    with pytest.raises(_hip.GlxError):
        _hip.DeviceGraph(A, dtype=np.int32)
    G.close()
    with pytest.raises(SystemExit):
        gl.ssl.poisson(A, solver='nope')


def test_spmm_shapes_and_edge_cases(gl):
    """Ragged and degenerate operators: empty rows, empty matrix, rectangular (halo columns),
    wide label matrices (G = 8 / 16 / 32 lanes per row), fp32 -- all bit-identical to scipy."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(11)
    # rows without entries and an all-zero operator
    A = sparse.random(300, 300, density=0.01, random_state=7, format='csr')
    assert (np.diff(A.indptr) == 0).sum() > 5
    u = rng.normal(size=(300, 10))
    G = _hip.DeviceGraph(A)
    assert np.array_equal(G.spmm_bias(u), A * u)
    G.close()
    Z = sparse.csr_matrix((50, 50))
    G = _hip.DeviceGraph(Z)
    assert np.array_equal(G.spmm_bias(u[:50], u[:50]), u[:50])     # Db + 0
    G.close()
    # rectangular: 200 rows, 500 columns
    R = sparse.random(200, 500, density=0.03, random_state=8, format='csr')
    x = rng.normal(size=(500, 7))
    G = _hip.DeviceGraph(R)
    assert np.array_equal(G.spmm_bias(x), R * x)
    G.close()
    # wide operands
    B = sparse.random(5000, 5000, density=0.002, random_state=9, format='csr') + sparse.identity(5000, format='csr')
    for C in (1, 4, 13, 28, 29, 60, 100):
        x = rng.normal(size=(5000, C))
        G = _hip.DeviceGraph(B)
        assert np.array_equal(G.spmm_bias(x), B * x), C
        G.close()
        x32 = x.astype(np.float32)
        G = _hip.DeviceGraph(B, dtype=np.float32)
        assert np.array_equal(G.spmm_bias(x32), B.astype(np.float32) * x32), C
        G.close()
    # a 1-D operand is a single column
    G = _hip.DeviceGraph(B)
    v = rng.normal(size=5000)
    assert np.array_equal(G.spmm_bias(v), B * v)
    G.close()
    with pytest.raises(_hip.GlxError):
        _hip.DeviceGraph(B).spmm_bias(rng.normal(size=(5000, 300)))      # beyond 256 columns


def test_poisson_sweep_many_classes_and_long_rows(gl, orc):
    """C = 30 classes (G = 8 lanes/row path with the stop column) and hub rows (S = 4 / 16 split
    rows) through the full Poisson sweep, bit-identical to the oracle."""
    rng = np.random.default_rng(5)
    n, C = 6000, 30
    lab = rng.integers(0, C, size=n)
    X = rng.normal(size=(C, 12))[lab] * 2.0 + rng.normal(size=(n, 12))
    W = gl.weightmatrix.knn(X, 12)
    W = W.tolil()
    hub = rng.choice(n, size=400, replace=False)          # make vertex 7 a hub with ~400 neighbours
    for j in hub:
        if j != 7:
            W[7, j] = 0.3
            W[j, 7] = 0.3
    W = sparse.csr_matrix(W)
    assert np.diff(W.indptr).max() > 300
    ti = orc.trainsets_generate(lab, rate=2, seed=3)
    u_ref, T_ref = orc.poisson_gd(W, ti, lab[ti], return_T=True)
    m = gl.ssl.poisson(W, solver='gradient_descent')
    u = m.fit(ti, lab[ti])
    assert m.num_iter == T_ref
    assert np.array_equal(u, u_ref)
    assert np.array_equal(m.predict(), orc.predict(u_ref))
    # C = 10 on the same hub graph exercises S = 16 in the G = 4 plan
    lab10 = lab % 10
    ti = orc.trainsets_generate(lab10, rate=2, seed=4)
    u_ref, T_ref = orc.poisson_gd(W, ti, lab10[ti], return_T=True)
    m = gl.ssl.poisson(W, solver='gradient_descent')
    assert np.array_equal(m.fit(ti, lab10[ti]), u_ref) and m.num_iter == T_ref


def test_ssl_trials_csv(gl, golden, tmp_path, monkeypatch):
    """ssl_trials writes the reference's CSV format; accuracies equal the oracle's per trial."""
    from oracle import gl_oracle as orc
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    labels = g['labels']
    monkeypatch.setattr(gl.ssl, 'results_dir', str(tmp_path / 'results'))
    sets = orc.trainsets_generate(labels, rate=np.array([[1], [2]]), num_trials=2, seed=5)
    m = gl.ssl.poisson(W, solver='gradient_descent')
    m.ssl_trials(sets, labels, tag='t_')
    lines = open(tmp_path / 'results' / 't__poisson_accuracy.csv').read().strip().split('\n')
    assert lines[0] == 'Number of labels,Accuracy' and len(lines) == 5
    for ts, line in zip(sets, lines[1:]):
        acc = orc.ssl_accuracy(orc.predict(orc.poisson_gd(W, ts, labels[ts])), labels, ts)
        assert line == '%d' % len(ts) + ',%.2f' % acc
    num_train, mean, std, nt = m.trials_statistics(tag='t_')
    assert list(num_train) == [10.0, 20.0] and nt == 2 and mean.shape == (2, 1)
    m.ssl_trials(sets, labels, tag='t_')          # exists -> aborts without touching the file
    assert len(open(tmp_path / 'results' / 't__poisson_accuracy.csv').read().strip().split('\n')) == 5
    mp = gl.ssl.poisson(W, class_priors=g['class_priors'], solver='gradient_descent')
    mp.ssl_trials(sets[:1], labels, tag='p_')
    hdr = open(tmp_path / 'results' / 'p__poisson_classpriors_accuracy.csv').readline().strip()
    assert hdr == 'Number of labels,Accuracy,Accuracy with class priors,Class priors error'


def test_next_rows_reweight_randomwalk_golden(gl, golden):
    """graph.reweight ('poisson' = a 1-D CG with numpy's pairwise reductions, 'wnll'),
    laplace reweightings and ssl.randomwalk -- bit-identical to the reference goldens."""
    g = golden('g7_next_rows.npz')
    W = csr_from(g, 'W')
    ti, lab = g['train_ind'], g['labels']
    A = csr_from(g, 'cg1d_A')
    x = gl.utils.conjgrad(A, g['cg1d_rhs'], tol=1e-9)
    assert x.shape == (500,) and np.array_equal(x, g['cg1d_x'])
    G = gl.graph.graph(W)
    for method, norm in [('poisson', 'combinatorial'), ('poisson', 'normalized'), ('wnll', 'combinatorial')]:
        tag = method + '_' + norm
        Wr = sparse.csr_matrix(G.reweight(ti, method=method, normalization=norm))
        Wg = csr_from(g, 'Wr_' + tag)
        assert np.array_equal(Wr.indices, Wg.indices) and np.array_equal(Wr.data, Wg.data), tag
        m = gl.ssl.laplace(W, reweighting=method, normalization=norm, reduce='exact')
        u = m.fit(ti, lab[ti])
        assert np.array_equal(u, g['laplace_' + tag + '_prob']), tag
        assert np.array_equal(m.predict(), g['laplace_' + tag + '_pred'])
    m = gl.ssl.randomwalk(W, reduce='exact')
    u = m.fit(ti, lab[ti])
    assert m.num_iter == int(g['randomwalk_iters'])
    assert np.array_equal(u, g['randomwalk_prob']) and np.array_equal(m.predict(), g['randomwalk_pred'])
    with pytest.raises(SystemExit):        # reference graph.py:450-451: `properly` needs the features
        G.reweight(ti, method='properly')


def test_reweight_properly_golden(gl, golden):
    """graph.reweight(method='properly') (reference graph.py:448-462: gamma = 1 + (r / distance to the nearest labelled point)^alpha) and
    ssl.laplace(reweighting='properly'): the nearest-labelled distances computed all pairs on the GPU are cKDTree's bit for bit, so the
    reweighted matrix and the fit are the reference's (goldens written by tests/golden/make_golden.py g11_properly)."""
    from graphlearning_amd import _hip
    from scipy import spatial
    g = golden('g11_properly.npz')
    for tag in ('blobs', 'moons'):
        X, lab, ti = g[tag + '_X'], g[tag + '_labels'], g[tag + '_train_ind']
        W = csr_from(g, tag + '_W')
        D = _hip.nearest_dist(X, ti)
        assert np.array_equal(D, spatial.cKDTree(X[ti]).query(X)[0]), tag
        G = gl.graph.graph(W)
        for ptag, kw in (('default', {}), ('p2', dict(alpha=3, zeta=1e5, r=0.5))):
            Wr = sparse.csr_matrix(G.reweight(ti, method='properly', X=X, **kw))
            Wg = csr_from(g, tag + '_Wr_' + ptag)
            assert np.array_equal(Wr.indptr, Wg.indptr) and np.array_equal(Wr.indices, Wg.indices) and np.array_equal(Wr.data, Wg.data), (tag, ptag)
        m = gl.ssl.laplace(W, X=X, reweighting='properly', reduce='exact')
        u = m.fit(ti, lab[ti])
        assert np.array_equal(u, g[tag + '_laplace_prob']), tag
        assert np.array_equal(m.predict(), g[tag + '_laplace_pred']), tag
        m2 = gl.ssl.laplace(W, X=X, reweighting='properly')            # the default mode: within 1e-5, same labels
        u2 = m2.fit(ti, lab[ti])
        assert np.max(np.abs(u2 - u)) <= 1e-5 and np.array_equal(m2.predict(), g[tag + '_laplace_pred']), tag


def test_prepared_sweep_reuse(gl, golden):
    """A prepared glx_sweep reused across problems (different train sets, with and without bias):
    captured launch graphs must not leak state from one problem into the next."""
    from graphlearning_amd import _hip
    from oracle import gl_oracle as orc
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    labels = g['labels']
    m = gl.ssl.poisson(W, solver='gradient_descent')
    dev, aux = m._operators()
    n = W.shape[0]
    sw = _hip.Sweep(dev, 10, min_iter=50, max_iter=1000, use_hipgraph=True)
    for seed in (1, 2, 3):
        ti = orc.trainsets_generate(labels, rate=2, seed=seed)
        src, k = gl.ssl._poisson_source(n, ti, labels[ti])
        v0 = np.zeros(n); v0[ti] = 1; v0 /= v0.sum()
        sw.set_problem(aux['D'] * src, v0 / aux['deg'], aux['deg'], aux['vinf'])
        for _ in range(2):                      # replaying the captured graph gives the same answer
            T, _ = sw.run()
            u_ref, T_ref = orc.poisson_gd(W, ti, labels[ti], return_T=True)
            assert T == T_ref and np.array_equal(sw.fetch(), u_ref), seed
    sw.close()
    heat = _hip.Sweep(dev, 10, min_iter=0, max_iter=0, use_hipgraph=True)
    rng = np.random.default_rng(0)
    u0 = rng.normal(size=(n, 10)); Db = rng.normal(size=(n, 10))
    P = aux['D'] * W.T
    for bias in (None, Db, None):
        heat.set_state(u0, bias)
        heat.iterate(4)
        ref = u0
        for _ in range(4):
            ref = P * ref if bias is None else P * ref + bias
        assert np.array_equal(heat.fetch(), ref)
    heat.close()


def test_page_rank_golden(gl, golden):
    """graph.page_rank on the device: vectors and sweep counts bit-identical to the
    reference, symmetric and directed graph, default and custom teleportation vector."""
    g = golden('g8_pagerank.npz')
    for tag in ('sym', 'dir'):
        G = gl.graph(csr_from(g, 'W_' + tag))
        u = G.page_rank()
        assert G.page_rank_iters == int(g['pr_' + tag + '_iters'])
        assert np.array_equal(u, g['pr_' + tag])
        u = G.page_rank(alpha=0.5, v=g['pr_' + tag + '_v'], tol=1e-8)
        assert G.page_rank_iters == int(g['pr_' + tag + '_tele_iters'])
        assert np.array_equal(u, g['pr_' + tag + '_tele'])


def test_page_rank_midsize_vs_oracle(gl, orc):
    X, labels = blobs(6000, 10, 6, 4, 2.0)
    W = gl.weightmatrix.knn(X, 12)
    G = gl.graph(W)
    u = G.page_rank(alpha=0.9, tol=1e-12)
    uo, it = orc.page_rank(W, alpha=0.9, tol=1e-12, return_iters=True)
    assert G.page_rank_iters == it
    assert np.array_equal(u, uo)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_sweep_project_equals_host_projection(gl, dtype):
    """glx_sweep_project_iterate with iters = 0 (decision on the device-resident sweep state) against glx_argmax_project on the
    fetched array (which the g5 golden pins to the reference): labels, weights, error, step count; then
    the state has become onehot(labels)."""
    from graphlearning_amd import _hip
    from scipy import sparse
    rng = np.random.default_rng(11)
    n, C = 3000, 5
    prob = (rng.normal(size=(n, C)) + np.array([0.8, 0.0, -0.3, 0.2, 0.1])).astype(dtype)
    priors = np.array([0.1, 0.3, 0.2, 0.25, 0.15])
    G = _hip.DeviceGraph(sparse.identity(n, format='csr'), dtype=dtype)
    S = _hip.Sweep(G, C, min_iter=0, max_iter=0, use_hipgraph=False)
    S.set_state(prob, None)
    ref = _hip.argmax_project(prob.astype(np.float64), priors, np.ones(C), max_steps=10000)
    labels, w, err, steps = S.project(priors, np.ones(C), max_steps=10000, to_onehot=True)
    assert steps == ref[3] and steps > 1
    assert np.array_equal(labels, ref[0]) and np.array_equal(w, ref[1]) and err == ref[2]
    onehot = S.fetch()
    assert onehot.dtype == dtype and np.array_equal(onehot, np.eye(C, dtype=dtype)[labels])
    # plain predict with given weights, labels not requested
    S.set_state(prob, None)
    none, w2, _, _ = S.project(None, w, max_steps=0, want_labels=False)
    assert none is None and np.array_equal(w2, w)
    lab2, _, _, _ = S.project(None, w, max_steps=0)
    assert np.array_equal(lab2, _hip.argmax_project(prob.astype(np.float64), None, w, max_steps=0)[0])
    S.close(); G.close()


def test_poisson_mbo_fp32_device_path(gl, golden):
    """use_cuda=True (the reference's fp32 device variant, ssl.py:807-823): same labels as the fp64 fit
    on two-moons, one-hot result returned as float64 like the reference's labels_to_onehot."""
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    ti = g['train_ind']
    tl = g['labels'][ti]
    m = gl.ssl.poisson_mbo(W, g['class_priors'], solver='gradient_descent', use_cuda=True)
    u = m.fit(ti, tl)
    assert u.dtype == np.float64 and set(np.unique(u)) <= {0.0, 1.0} and np.all(u.sum(axis=1) == 1)
    assert np.array_equal(m.predict(), g['poisson_mbo_gradient_descent_pred'])


def test_plaplace_jacobi_golden(gl, golden, orc):
    """graph.plaplace(fast=False) on the device against the compiled reference's outputs.
    The order in which a vertex's terms are summed comes from np.argsort's unstable default sort of
    the vertex column (reference graph.py:73), which may differ between hosts: bit-exact when this
    host's order equals the one the golden run saw, 1e-12 otherwise; the stopping iteration is exact."""
    g = golden('g9_plaplace.npz')
    W = csr_from(g, 'W')
    G = gl.graph(W)
    for tag in ('p10', 'p3', 'T57', 'T200'):
        p, tol, T, it_ref = g[tag + '_params']
        u = G.plaplace(g['bdy'], g['bdy_val'], p, tol=tol, max_num_it=T, fast=False)
        assert abs(G.plaplace_iters - int(it_ref)) <= (0 if tag.startswith('T') else 1)
        uo = orc.plaplace_jacobi(W, g['bdy'], g['bdy_val'], p, tol=tol, max_num_it=T)
        assert np.array_equal(u, uo)                      # same host, same entry order: bit-identical to the oracle
        assert np.max(np.abs(u - g[tag + '_u'])) <= 1e-12
    with pytest.raises(NotImplementedError):
        G.plaplace(g['bdy'], g['bdy_val'], 10)            # fast=True: sequential Gauss-Seidel in the reference


def test_reweight_poisson_midsize_vs_oracle(gl, orc):
    """1-D conjgrad (numpy's pairwise-summed reductions, parallel on the device) on a graph whose
    summation tree is several levels deep: the reweighted matrix equals the oracle's bit for bit."""
    X, labels = blobs(9000, 6, 5, 8, 2.2)
    W = gl.weightmatrix.knn(X, 9)
    ti = gl.trainsets.generate(labels, rate=4, seed=1)
    for norm in ('combinatorial', 'normalized'):
        Wr = gl.graph(W).reweight(ti, method='poisson', normalization=norm)
        Wo = orc.reweight(W, ti, method='poisson', normalization=norm)
        assert (Wr != Wo).nnz == 0
        assert np.array_equal(sparse.csr_matrix(Wr).data, sparse.csr_matrix(Wo).data)


def test_plaplace_jacobi_midsize_vs_oracle(gl, orc):
    """A larger graph (several workgroups, hub-free kNN), odd and even iteration caps and a run to the
    stop test: bit-identical to the oracle's C restatement of lp_iterate_main."""
    rng = np.random.default_rng(4)
    X = rng.random((9000, 2))
    W = gl.weightmatrix.knn(X, 10)
    x, y = X[:, 0], X[:, 1]
    bdy = (x < 0.03) | (x > 0.97) | (y < 0.03) | (y > 0.97)
    val = np.sin(3 * x) + y ** 2
    G = gl.graph(W)
    for p, tol, T in [(6.0, 1e-1, 301), (20.0, 1e-1, 400), (3.0, 0.5, 1e6)]:
        u = G.plaplace(bdy, val[bdy], p, tol=tol, max_num_it=T, fast=False)
        uo, it = orc.plaplace_jacobi(W, bdy, val[bdy], p, tol=tol, max_num_it=T, return_iters=True)
        assert G.plaplace_iters == it
        assert np.array_equal(u, uo), (p, tol, T)
