"""glx_graph_create_resident + glx_graph_set_row_transform: the operator of a fresh ssl.poisson fit is built from W's own arrays on
the device -- degrees there, P = D^-1 W^T formed while the sliced-ELL image is filled -- and must be the operator scipy's
`D * W.transpose()` writes down, entry for entry (reference graphlearning/ssl.py:615-617, 634-635; graph.py:108-122)."""
import numpy as np
import pytest
from scipy import sparse
from conftest import csr_from, blobs

pytestmark = pytest.mark.gpu


def _symmetric_knn_graph(n, d, k, seed):
    from oracle import gl_oracle as orc
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(6, d))[rng.integers(0, 6, size=n)] * 2.0 + rng.normal(size=(n, d))
    return orc.knn(X, k)


@pytest.mark.parametrize('n,dtype', [(700, np.float64), (6000, np.float64), (6000, np.float32)])
def test_resident_operator_equals_the_host_built_one(n, dtype):
    from graphlearning_amd import _hip, ssl as gssl
    _hip.require_device()
    W = _symmetric_knn_graph(n, 12, 9, 3 + n)
    W = sparse.csr_matrix((W.data.astype(np.float64), W.indices.astype(np.int32), W.indptr.astype(np.int32)), shape=W.shape)
    P, deg, dinv = gssl._poisson_operator_symmetric(W)                  # the host form: reversed rows times 1 / degree
    D = sparse.spdiags((W * np.ones(n)) ** (-1), 0, n, n).tocsr()
    Pref = D * W.transpose()                                             # the reference's expression
    assert np.array_equal(Pref.indices, P.indices) and np.array_equal(Pref.data, P.data)
    dev, sums = _hip.DeviceGraph.resident(W, dtype=dtype, want_row_sums=True)
    assert np.array_equal(sums, W * np.ones(n)) and np.array_equal(sums, deg)
    dev.set_row_transform(sums ** (-1), reverse_rows=True)
    host = _hip.DeviceGraph(P, dtype=dtype)
    rng = np.random.default_rng(1)
    u = rng.normal(size=(n, 10)).astype(dtype)
    Db = rng.normal(size=(n, 10)).astype(dtype)
    a = dev.spmm_bias(u, Db, iters=3)
    b = host.spmm_bias(u, Db, iters=3)
    assert np.array_equal(a, b)                                          # same entries in the same order: bit-identical products
    if dtype == np.float64:
        ref = u.copy()
        for _ in range(3):
            ref = Db + P * ref
        assert np.array_equal(a, ref)
    # the plain resident form (no transform) is the matrix itself, and its own locality pass works without a host pattern
    dev2, _ = _hip.DeviceGraph.resident(W, dtype=dtype)
    host2 = _hip.DeviceGraph(W, dtype=dtype)
    assert np.array_equal(dev2.spmm_bias(u), host2.spmm_bias(u))
    assert np.array_equal(np.sort(dev2.order()), np.arange(n))
    for g in (dev, host, dev2, host2):
        g.close()


def test_resident_operator_rejects_bad_input():
    from graphlearning_amd import _hip
    _hip.require_device()
    W = _symmetric_knn_graph(500, 5, 6, 9).tocsr()
    bad = sparse.csr_matrix((W.data.copy(), W.indices.copy(), W.indptr.copy()), shape=W.shape)
    bad.indices[7] = 500                                                 # out of range: found by the device-side check
    with pytest.raises(_hip.GlxError):
        _hip.DeviceGraph.resident(bad)
    dev, _ = _hip.DeviceGraph.resident(W)
    dev.spmm_bias(np.ones((500, 3)))
    with pytest.raises(_hip.GlxError):
        dev.set_row_transform(np.ones(500), reverse_rows=True)           # only before the operator is first used
    dev.close()
    host = _hip.DeviceGraph(W)
    with pytest.raises(_hip.GlxError):
        host.set_row_transform(np.ones(500))                             # not a resident graph
    host.close()


def test_fresh_fit_through_the_resident_path_matches_the_goldens(golden):
    """ssl.poisson(gradient_descent) on a stamped weightmatrix.knn matrix takes the resident path: T and iterates of the n = 5000
    golden run, bit for bit; an unstamped copy of the same matrix (the general, transposing build) gives the same."""
    import graphlearning_amd as gl
    from graphlearning_amd import utils
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    W._glx_sym = utils.symmetric_fingerprint(W)
    ti, lab = g['train_ind'], g['labels']
    m = gl.ssl.poisson(W, solver='gradient_descent')
    u = m.fit(ti, lab[ti])
    assert m.num_iter == int(g['poisson_gd_T']) and np.array_equal(u, g['poisson_gd_prob'])
    m2 = gl.ssl.poisson(csr_from(g, 'W'), solver='gradient_descent')
    assert np.array_equal(m2.fit(ti, lab[ti]), u) and m2.num_iter == m.num_iter


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_heat_state_from_labels_equals_the_dense_form(dtype):
    """glx_sweep_set_state_labels (PoissonMBO's start: u = onehot(labels) formed on the device, the bias from its m nonzero rows)
    against glx_sweep_set_state with the two dense arrays the reference builds (ssl.py:798, 805): the same state after 0, 1 and 7
    sweeps, bit for bit; duplicate rows and out-of-range rows are the caller's to avoid / are refused."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(5)
    n, k = 3000, 4
    X, lab = blobs(n, 6, k, 3, 2.0)
    from oracle import gl_oracle as orc
    W = orc.knn(X, 8)
    deg = np.asarray(W.sum(axis=1)).ravel()
    dt = 1.0 / deg.max()
    from scipy import sparse
    P = sparse.csr_matrix(sparse.identity(n) - dt * (sparse.spdiags(deg, 0, n, n) - W))
    labels = rng.integers(0, k, size=n).astype(np.int64)
    rows = rng.choice(n, size=37, replace=False).astype(np.int64)
    Db_rows = rng.normal(size=(37, k)).astype(dtype)
    Db = np.zeros((n, k), dtype=dtype)
    Db[rows] = Db_rows
    u0 = np.zeros((n, k), dtype=dtype)
    u0[np.arange(n), labels] = 1
    dev = _hip.DeviceGraph(P, dtype=dtype)
    a = _hip.Sweep(dev, k, min_iter=0, max_iter=0, use_hipgraph=True)
    b = _hip.Sweep(dev, k, min_iter=0, max_iter=0, use_hipgraph=True)
    a.set_state(u0, Db)
    b.set_state_labels(labels, rows, Db_rows)
    assert np.array_equal(a.fetch(), b.fetch())
    for iters in (1, 7):
        a.iterate(iters); b.iterate(iters)
        ua, ub = a.fetch(), b.fetch()
        assert np.array_equal(ua, ub) and np.isfinite(ua).all()
    # a second problem on the same object: the previous bias rows must be gone
    rows2 = rows[:5]
    b.set_state_labels(labels, rows2, Db_rows[:5])
    Db2 = np.zeros((n, k), dtype=dtype); Db2[rows2] = Db_rows[:5]
    a.set_state(u0, Db2)
    a.iterate(3); b.iterate(3)
    assert np.array_equal(a.fetch(), b.fetch())
    # no bias at all
    b.set_state_labels(labels, rows[:0], Db_rows[:0]); a.set_state(u0, None)
    a.iterate(2); b.iterate(2)
    assert np.array_equal(a.fetch(), b.fetch())
    with pytest.raises(_hip.GlxError):
        b.set_state_labels(labels, np.array([n]), Db_rows[:1])
    a.close(); b.close(); dev.close()
