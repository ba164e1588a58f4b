"""The block form of the reference-order reduction chains of utils.conjgrad (graphlearning/utils.py:524,527) on the device
(csrc/cg_seqsum.hip, arithmetic in csrc/seqsum_exact.h) against the row-by-row chain kernel and the oracle: same iterates, same
iteration counts, same residual norms, bit for bit -- on small systems (forced), on systems large enough to take it by default, on
the singular Poisson system whose products cancel (the case the block form likes least), with Dirichlet rows, stacked systems that
stop at different iterations, fp32 operators, and breakdowns (nan)."""
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


@pytest.fixture
def form(monkeypatch):
    from graphlearning_amd import _hip

    def set_form(name):
        monkeypatch.setattr(_hip, 'CG_EXACT_FORM', name)
    return set_form


def _spd(n, seed, density=None, shift=4.0):
    A = sparse.random(n, n, density=density or min(1.0, 8.0 / n), random_state=seed, format='csr')
    return sparse.csr_matrix(A + A.T + sparse.identity(n) * shift)


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize('n,C', [(40, 2), (65, 3), (500, 4), (1500, 7), (3000, 10), (4097, 13), (12000, 5)])
def test_block_form_equals_chain_and_oracle(gl, orc, form, n, C):
    from graphlearning_amd import _hip
    rng = np.random.default_rng(n + C)
    A = _spd(n, n)
    b = rng.normal(size=(n, C)) * np.exp(rng.normal(size=C) * 3)
    x_ref, it_ref, err_ref = orc.conjgrad(A, b, tol=1e-9, return_iters=True)
    G = _hip.DeviceGraph(A)
    out = {}
    for name in ('chain', 'blocks'):
        form(name)
        out[name] = G.cg(b, tol=1e-9)
        st = G.last_block_stats()
        assert (st == (-1, -1, -1)) == (name == 'chain'), (name, st)
        if name == 'blocks':
            assert sum(st) > 0 and st[0] + st[1] > 0, st      # blocks were applied as integers, not only row by row
    G.close()
    for name in ('chain', 'blocks'):
        x, it, err = out[name]
        assert it == it_ref, (name, it, it_ref)
        assert err == err_ref, (name, err, err_ref)
        assert np.array_equal(x, x_ref), name


@pytest.mark.parametrize('n,C,max_iter', [(700, 3, 100000), (9000, 10, 100000), (9000, 4, 13), (9000, 4, 16), (9000, 4, 3), (300, 2, 0)])
def test_captured_chunks_equal_eager_launches(gl, orc, monkeypatch, n, C, max_iter):
    """Round 6: the reference-order solve replays chunks of 8 iterations from captured launch sequences and launches chunk k + 1 before
    it has looked at chunk k (iterations past convergence must leave everything as it is).  Same bits as the launch-by-launch form
    (GLX_CG_EAGER) and as the oracle: iterates, iteration counts, residual norms -- for solves that converge inside a chunk, that are cut
    by max_iter inside a chunk (partial chunks run eagerly), at a chunk boundary, and for repeated solves on one operator (the replay of
    sequences captured by an earlier solve; another tolerance and another right-hand side width re-capture)."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(n + C + max_iter)
    A = _spd(n, n + 7)
    G = _hip.DeviceGraph(A)
    for rep, tol in enumerate((1e-9, 1e-9, 1e-6)):
        b = rng.normal(size=(n, C)) * np.exp(rng.normal(size=C) * 2)
        x_ref, it_ref, err_ref = orc.conjgrad(A, b, tol=tol, max_iter=max_iter, return_iters=True)
        got = {}
        for eager in (False, True):
            monkeypatch.setattr(_hip, 'CG_EXACT_EAGER', eager)
            got[eager] = G.cg(b, tol=tol, max_iter=max_iter)
        for eager in (False, True):
            x, it, err = got[eager]
            assert it == it_ref and err == err_ref and np.array_equal(x, x_ref), (rep, eager, it, it_ref, err, err_ref)
    G.close()


def test_block_form_with_the_prefix_scan_of_many_groups(gl, orc, form):
    """From 1024 groups of 1024 rows on, the approximate prefix in front of a group comes from a scan pass (ss_scan_kernel) instead of every
    workgroup adding all group sums in front of it (quadratic in the number of groups).  1.1 M rows, a few iterations: the same
    bits as the chain form and the oracle."""
    from graphlearning_amd import _hip
    n, C = 1100000, 2
    rng = np.random.default_rng(5)
    i = np.arange(n)
    A = sparse.csr_matrix(sparse.diags([np.full(n, 4.0), np.full(n - 1, -1.0), np.full(n - 1, -1.0), np.full(n - 997, 0.5), np.full(n - 997, 0.5)],
                                       [0, 1, -1, 997, -997]))
    b = rng.normal(size=(n, C)) * np.array([1.0, 1e-3])
    x_ref, it_ref, err_ref = orc.conjgrad(A, b, tol=1e-12, max_iter=4, return_iters=True)
    G = _hip.DeviceGraph(A)
    for name in ('blocks', 'chain'):
        form(name)
        x, it, err = G.cg(b, tol=1e-12, max_iter=4)
        assert it == it_ref and err == err_ref and np.array_equal(x, x_ref), name
    G.close()


def test_single_column_2d_right_hand_side_reduces_pairwise(gl, orc):
    """an (n,1) right-hand side: numpy's `np.sum(..., axis=0)` of one column is a contiguous run, summed pairwise like the 1-D case"""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(3)
    for n in (40, 700, 9000):
        A = _spd(n, n + 1)
        b = rng.normal(size=(n, 1))
        x_ref, it_ref, err_ref = orc.conjgrad(A, b, tol=1e-9, return_iters=True)
        G = _hip.DeviceGraph(A)
        x, it, err = G.cg(b, tol=1e-9)
        G.close()
        assert it == it_ref and err == err_ref and np.array_equal(x, x_ref), n


def test_goldens_in_block_form(gl, golden, form):
    form('blocks')
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    m = gl.ssl.poisson(W)
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert m.num_iter == int(g['poisson_cg_iters'])
    assert np.array_equal(u, g['poisson_cg_prob'])
    m = gl.ssl.laplace(W, reduce='exact')
    u = m.fit(g['train_ind'], g['labels'][g['train_ind']])
    assert np.array_equal(u, g['laplace_combinatorial_prob'])
    g3 = golden('g3_blobs5000.npz')
    W3 = csr_from(g3, 'W')
    m3 = gl.ssl.poisson(W3)
    u3 = m3.fit(g3['train_ind'], g3['labels'][g3['train_ind']])
    assert m3.num_iter == int(g3['poisson_cg_iters'])
    assert np.array_equal(u3, g3['poisson_cg_prob'])


def test_singular_poisson_system_with_cancelling_products(gl, orc, form):
    """separate clusters: the singular system's iterates grow along the null space, p*Ap changes sign from row to row and the
    running sums wander through binades -- many records, many row-by-row blocks, the same bits; n >= 8192 takes the block form
    by default"""
    from graphlearning_amd import _hip
    X, labels = blobs(9000, 20, 10, 3, 2.0)
    W = orc.knn(X, 10)
    ti = orc.trainsets_generate(labels, rate=1, seed=0)
    u_ref, it_ref = orc.poisson_cg(W, ti, labels[ti], return_iters=True)
    res = {}
    for name in (None, 'chain', 'blocks'):
        form(name)
        m = gl.ssl.poisson(W)
        res[name] = (m.fit(ti, labels[ti]).copy(), m.num_iter)
    for name, (u, it) in res.items():
        assert it == it_ref, (name, it, it_ref)
        assert np.array_equal(u, u_ref), name


def test_default_takes_the_block_form_from_8192_rows(gl, form):
    from graphlearning_amd import _hip
    rng = np.random.default_rng(5)
    form(None)
    for n, blocks in [(8191, False), (8192, True)]:
        G = _hip.DeviceGraph(_spd(n, 1))
        G.cg(rng.normal(size=(n, 2)), tol=1e-6)
        assert (G.last_block_stats() != (-1, -1, -1)) == blocks, (n, G.last_block_stats())
        G.close()


def test_stacked_systems_dirichlet_rows_and_early_stops(gl, form):
    """columns in groups with their own stop tests (frozen columns skip their chains), rows held at zero (zero products: blocks
    of zeros are no-ops for any state), 9 columns padded to 12"""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(11)
    n = 9000
    A = _spd(n, 2)
    B = rng.normal(size=(n, 9)) * np.array([1, 1, 1, 1e-3, 1e-3, 1e-3, 1e3, 1e3, 1e3])
    masks = [rng.choice(n, size=k, replace=False) for k in (0, 700, 4000)]
    G = _hip.DeviceGraph(A)
    out = {}
    for name in ('chain', 'blocks'):
        form(name)
        out[name] = (G.cg_groups(B, 3, tol=1e-7), G.cg_groups(B, 3, tol=1e-7, masks=masks))
    G.close()
    for k in range(2):
        for q in range(3):
            assert _same(out['chain'][k][q], out['blocks'][k][q]), (k, q)
    assert len(set(out['blocks'][0][1].tolist())) > 1          # the systems did stop at different iterations


def test_fp32_operator_and_breakdown(gl, form):
    from graphlearning_amd import _hip
    rng = np.random.default_rng(13)
    n = 10000
    A = _spd(n, 3)
    b = rng.normal(size=(n, 4))
    b[:, 2] = 0.0                                               # rsold = 0: alpha = 0/0, the column turns to nan as in the reference
    res = {}
    for dt in (np.float32, np.float64):
        G = _hip.DeviceGraph(A, dtype=dt)
        for name in ('chain', 'blocks'):
            form(name)
            res[(dt, name)] = G.cg(b.astype(dt), tol=1e-6, max_iter=60)
        G.close()
        for q in range(3):
            assert _same(res[(dt, 'chain')][q], res[(dt, 'blocks')][q]), (dt, q)
