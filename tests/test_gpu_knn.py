"""GPU parity of the exact kNN search and the kNN weight matrix against the reference's
golden vectors (cKDTree results captured by tests/golden/make_golden.py)."""
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


@pytest.mark.parametrize('flt', ['bf16x3', 'bf16x3_cat', 'bf16x3_blocks', 'f32'])
@pytest.mark.parametrize('tag', ['d20', 'd64', 'd3'])
def test_knnsearch_golden(gl, golden, tag, flt, monkeypatch):
    """The candidate filters -- split-bf16 operands on the bf16 matrix cores (default for d <= 128; for 17 <= d <= 21 as ONE
    contraction over concatenated operands [rh|rh|rl].[qh|ql|qh], for d <= 20 with |r|^2 folded into it as well; knn_options(concat=1)
    without the fold, concat=0 in blocks of 16 features) and the fp32-input MFMA kernel -- end in the same exact answer: the cKDTree
    lists of the reference."""
    from graphlearning_amd import _hip
    g = golden('g2_knn.npz')
    X, J, D = g['X_' + tag], g['J_' + tag], g['D_' + tag]
    concat = {'bf16x3': None, 'bf16x3_cat': 1, 'bf16x3_blocks': 0, 'f32': None}[flt]
    with _hip.knn_options(filter='f32' if flt == 'f32' else 'bf16', concat=concat):
        ind, dist = gl.weightmatrix.knnsearch(X, 11)
    st = _hip.knn_stats()
    assert st['filter'] == flt.split('_')[0] and st['fallback_rows'] <= 20
    assert st['concatenated'] == (2 if flt == 'bf16x3' else 1 if flt == 'bf16x3_cat' else 0) * (tag == 'd20')
    assert ind.dtype == np.int64 and dist.dtype == np.float64
    assert np.array_equal(ind, J)                      # identical neighbour sets and order
    assert np.array_equal(dist[:, 0], np.zeros(len(X)))  # self distance exactly 0
    assert np.max(np.abs(dist - D)) <= 1e-12           # fp64 direct-difference distances
    from graphlearning_amd import _hip
    print(tag, _hip.knn_stats())


def test_knnsearch_brute_and_angular_golden(gl, golden):
    g = golden('g2_knn.npz')
    X = g['X_d20'][:400]
    ind, dist = gl.weightmatrix.knnsearch(X, 11, method='brute')
    assert np.array_equal(ind, g['Jb_brute400'])
    assert np.max(np.abs(dist - g['Db_brute400'])) <= 1e-12
    ind, dist = gl.weightmatrix.knnsearch(X, 8, method='kdtree', similarity='angular')
    assert np.array_equal(ind, g['J_angular400'])
    assert np.max(np.abs(dist - g['D_angular400'])) <= 1e-12


def test_knnsearch_edge_cases(gl):
    from graphlearning_amd import _hip
    rng = np.random.default_rng(0)
    # tiny inputs, k == n, ragged tile (n not a multiple of 128), duplicates
    for n, d, k in [(1, 3, 1), (5, 2, 5), (129, 7, 4), (300, 33, 27), (200, 130, 3), (400, 10, 45), (150, 30, 60),
                    # feature-blocked variant (d > 130, or k > 28 with d > 34)
                    (500, 131, 5), (700, 200, 11), (1000, 784, 11), (300, 300, 40), (150, 64, 60), (257, 1000, 20)]:
        X = rng.normal(size=(n, d))
        ind, dist = _hip.knn_bruteforce(X, k)
        D2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
        ref = np.argsort(D2, axis=1, kind='stable')[:, :k]
        assert np.array_equal(ind, ref), (n, d, k)
        assert np.allclose(dist, np.sqrt(np.take_along_axis(D2, ref, 1)), rtol=0, atol=1e-12)
    X = np.repeat(rng.normal(size=(40, 5)), 3, axis=0)       # every point three times
    ind, dist = _hip.knn_bruteforce(X, 4)
    assert np.all(dist[:, :3] == 0)
    assert np.all(np.sort(ind[:, :3], axis=1) == (np.arange(120) // 3 * 3)[:, None] + np.arange(3))
    # far from the origin and on very different scales per feature: centring + the fp32 filter's error bound
    for d in (20, 200):
        Y = rng.normal(size=(900, d)) * np.logspace(-2, 2, d) + 1e6
        ind, dist = _hip.knn_bruteforce(Y, 8)
        Yc = Y - Y.mean(axis=0)
        D2 = ((Yc[:, None, :] - Yc[None, :, :]) ** 2).sum(-1)
        ref = np.argsort(D2, axis=1, kind='stable')[:, :8]
        assert np.array_equal(ind, ref), d
    Z = np.repeat(rng.normal(size=(50, 300)), 4, axis=0)      # duplicates through the feature-blocked variant
    ind, dist = _hip.knn_bruteforce(Z, 5)
    assert np.all(dist[:, :4] == 0) and np.all(np.sort(ind[:, :4], axis=1) == (np.arange(200) // 4 * 4)[:, None] + np.arange(4))
    with pytest.raises(_hip.GlxError):
        _hip.knn_bruteforce(X, 1000)
    with pytest.raises(_hip.GlxError):          # k (incl. self) above 60
        _hip.knn_bruteforce(rng.normal(size=(150, 8)), 61)
    with pytest.raises(SystemExit):
        gl.weightmatrix.knnsearch(X, 3, similarity='manhattan')


def test_knn_range_matches_full(gl):
    from graphlearning_amd import _hip
    X, _ = blobs(1000, 16, 5, 3, 2.0)
    ind, dist = _hip.knn_bruteforce(X, 9)
    ind2, dist2 = _hip.knn_bruteforce(X, 9, query_range=(300, 811))
    assert np.array_equal(ind2, ind[300:811]) and np.array_equal(dist2, dist[300:811])


@pytest.mark.parametrize('kernel', ['gaussian', 'uniform', 'symgaussian', 'distance', 'singular'])
def test_knn_weightmatrix_golden(gl, golden, kernel):
    g = golden('g1_twomoons.npz')
    W = gl.weightmatrix.knn(g['X'], 10, kernel=kernel)
    Wg = csr_from(g, 'W_' + kernel)
    assert W.dtype == np.float64 and W.format == 'csr'
    assert np.array_equal(W.indptr, Wg.indptr) and np.array_equal(W.indices, Wg.indices)
    assert np.max(np.abs(W.data - Wg.data)) <= 1e-12
    assert (abs(W - W.T) > 1e-15).nnz == 0 or kernel == 'symgaussian'
    assert W.diagonal().sum() == 0


def test_knn_weightmatrix_nosym_and_injected(gl, golden):
    g = golden('g1_twomoons.npz')
    W = gl.weightmatrix.knn(g['X'], 10, symmetrize=False)
    Wg = csr_from(g, 'W_gaussian_nosym')
    assert np.array_equal(W.indices, Wg.indices) and np.max(np.abs(W.data - Wg.data)) <= 1e-12
    # knn_data injection (reference weightmatrix.py:122-123): the golden kNN data through the
    # device assembly reproduces the golden matrices -- structure exactly, values exactly where no
    # exp is involved and within 2 ulp of numpy's exp otherwise
    kd = (g['knn_ind'], g['knn_dist'])
    for kernel in ['gaussian', 'uniform', 'symgaussian', 'distance', 'singular']:
        W2 = gl.weightmatrix.knn(None, 10, kernel=kernel, knn_data=(kd[0], kd[1].copy()))
        Wg = csr_from(g, 'W_' + kernel)
        assert W2.format == 'csr' and W2.dtype == np.float64 and W2.indices.dtype == np.int32
        assert np.array_equal(W2.indptr, Wg.indptr) and np.array_equal(W2.indices, Wg.indices), kernel
        assert np.array_equal(W2.data, Wg.data), kernel            # bit-identical weight matrix
    W3 = gl.weightmatrix.knn(None, 10, symmetrize=False, knn_data=kd)
    assert np.array_equal(W3.data, csr_from(g, 'W_gaussian_nosym').data)
    # the default: the correctly rounded exp on the device -- same structure, every weight within one ulp of the golden host's numpy
    old = os.environ.pop('GLX_HOST_EXP', None)
    try:
        for kernel in ['gaussian', 'symgaussian']:
            W4 = gl.weightmatrix.knn(None, 10, kernel=kernel, knn_data=kd)
            Wg = csr_from(g, 'W_' + kernel)
            assert np.array_equal(W4.indices, Wg.indices)
            # (a stored weight is the mean of two exponentials: two ulps at most; symgaussian's rule subtracts and carries a few more)
            assert np.max(np.abs(W4.data.view(np.int64) - Wg.data.view(np.int64))) <= (2 if kernel == 'gaussian' else 8), kernel
    finally:
        if old is not None:
            os.environ['GLX_HOST_EXP'] = old
    # k is clamped to the columns available (reference weightmatrix.py:135)
    W5 = gl.weightmatrix.knn(None, 50, knn_data=kd)
    assert np.array_equal(W5.indices, csr_from(g, 'W_gaussian').indices)
    # user eta overrides the kernel (host callable, device assembly)
    We = gl.weightmatrix.knn(None, 10, eta=lambda t: np.exp(-4 * t), knn_data=kd)
    assert np.allclose(We.data, csr_from(g, 'W_gaussian').data, rtol=1e-14, atol=0)
    assert np.array_equal(We.indices, csr_from(g, 'W_gaussian').indices)


def test_knn_to_csr_duplicates_and_asymmetry(gl):
    """Hand-made knn data: duplicate neighbours in a row (summed like COO->CSR), a one-directional
    edge (carries w/2), self loops (removed), against the oracle's scipy assembly."""
    from oracle import gl_oracle as orc
    ind = np.array([[0, 1, 1], [1, 2, 0], [2, 2, 3], [3, 0, 1]], dtype=np.int64)
    dist = np.array([[0.0, 0.5, 0.7], [0.0, 0.2, 0.9], [0.0, 0.0, 0.4], [0.0, 1.0, 1.5]])
    for kernel in ['gaussian', 'uniform', 'distance', 'singular', 'symgaussian']:
        for symmetrize in (True, False):
            W = gl.weightmatrix.knn(None, 2, kernel=kernel, symmetrize=symmetrize, knn_data=(ind, dist.copy()))
            Wo = orc.knn_weights(ind, dist.copy(), 2, kernel=kernel, symmetrize=symmetrize)
            assert np.array_equal(W.indptr, Wo.indptr) and np.array_equal(W.indices, Wo.indices), (kernel, symmetrize)
            assert np.allclose(W.data, Wo.data, rtol=1e-15, atol=0), (kernel, symmetrize)


def test_knn_to_csr_hub_vertex(gl):
    """Vertices that are many rows' neighbour (more forward + reverse entries than the 1024 one wavefront merges in
    LDS) are merged by a workgroup each in a global scratch (merge_hub_kernel); hubs of 1030, 2100 and 5990 reverse
    neighbours, with duplicate entries in and towards the hubs: identical to the oracle's scipy assembly."""
    from oracle import gl_oracle as orc
    rng = np.random.default_rng(2)
    n, k = 6000, 5
    ind = np.empty((n, k), dtype=np.int64)
    ind[:, 0] = np.arange(n)
    ind[:, 1] = 0                                   # hub 0: everyone's neighbour
    ind[:, 2] = (np.arange(n) + 1) % n
    ind[:, 3] = (np.arange(n) + 7) % n
    ind[:, 4] = (np.arange(n) + 13) % n
    ind[:2100, 3] = 11                              # hub 11: 2100 reverse neighbours
    ind[3000:4030, 4] = 4000                        # hub 4000: 1030
    ind[0, 1] = 5
    ind[100:140, 2] = 0                             # rows that list hub 0 twice
    ind[0, 2:5] = [11, 11, 4000]                    # a hub that lists another hub twice
    dist = np.sort(rng.random((n, k)), axis=1)
    dist[:, 0] = 0
    for kernel, symmetrize in [('gaussian', True), ('distance', True), ('symgaussian', True), ('singular', True), ('uniform', False)]:
        W = gl.weightmatrix.knn(None, 4, kernel=kernel, symmetrize=symmetrize, knn_data=(ind, dist.copy()))
        Wo = orc.knn_weights(ind, dist.copy(), 4, kernel=kernel, symmetrize=symmetrize)
        assert np.array_equal(W.indptr, Wo.indptr) and np.array_equal(W.indices, Wo.indices), kernel
        if kernel == 'symgaussian':
            assert np.allclose(W.data, Wo.data, rtol=1e-15, atol=0), kernel     # exp on the device: as in the test above
        else:
            assert np.array_equal(W.data, Wo.data), kernel
        if symmetrize:
            assert np.diff(W.indptr)[[0, 11, 4000]].min() > 1024
    assert np.diff(W.indptr).max() <= 5 and np.diff(Wo.indptr).max() <= 5    # unsymmetrised: k entries per row


def test_blobs5000_knn_graph_golden(gl, golden):
    g = golden('g3_blobs5000.npz')
    ind, dist = gl.weightmatrix.knnsearch(g['X'], 11)
    assert np.array_equal(ind, g['knn_ind'])
    assert np.max(np.abs(dist - g['knn_dist'])) <= 1e-12
    W = gl.weightmatrix.knn(g['X'], 10)
    Wg = csr_from(g, 'W')
    assert np.array_equal(W.indptr, Wg.indptr) and np.array_equal(W.indices, Wg.indices)
    assert np.max(np.abs(W.data - Wg.data)) <= 1e-12


def test_knn_filter_soundness_on_near_duplicates_and_offsets(gl):
    """The split-bf16 filter is only as good as its error bound: clusters of near-duplicate points (spread 1e-6 around
    centres of norm ~10, on top of a common offset of 1e3) are far below what the filter can resolve, so the acceptance test
    must hand those rows to the exact fp64 fallback -- and the answer must still be the exact one (cKDTree order)."""
    from oracle import gl_oracle as orc
    from graphlearning_amd import _hip
    rng = np.random.default_rng(12)
    centres = rng.normal(size=(300, 24)) * 2.0
    X = np.repeat(centres, 10, axis=0) + rng.normal(size=(3000, 24)) * 1e-6 + 1e3
    J_ref, D_ref = orc.knnsearch(X, 11)
    for flt in ('bf16', 'f32'):
        with _hip.knn_options(filter=flt):
            J, D = gl.weightmatrix.knnsearch(X, 11)
        st = _hip.knn_stats()
        # the 10 members of a cluster are each other's nearest neighbours; ties are broken by index like cKDTree's sort
        assert np.array_equal(np.sort(J[:, :10], axis=1), np.sort(J_ref[:, :10], axis=1)), flt
        assert np.max(np.abs(D - D_ref)) <= 1e-9 * 1e3                 # distances of order 1e-6 computed at offset 1e3: cancellation in fp64 itself
        assert np.array_equal(J[:, 10], J_ref[:, 10]) or np.max(np.abs(D[:, 10] - D_ref[:, 10])) <= 1e-9
        print(flt, 'fallback rows', st['fallback_rows'])
    # exact duplicates: distance 0 ties resolved by index, self first
    Y = np.repeat(rng.normal(size=(50, 7)), 4, axis=0)
    J, D = gl.weightmatrix.knnsearch(Y, 4)
    assert np.all(D == 0) and np.array_equal(np.sort(J, axis=1), (np.arange(200) // 4 * 4)[:, None] + np.arange(4)[None, :])


def test_symmetric_stamp_only_on_bitwise_symmetric_graphs(gl):
    """weightmatrix.knn stamps its output as symmetric (ssl.poisson then builds D^-1 W^T without a transpose) only where the
    symmetrisation rule is symmetric bit for bit: (a+b)/2 and the element-wise max are, the symgaussian rule is not."""
    from graphlearning_amd import utils as glutils
    rng = np.random.default_rng(21)
    X = rng.normal(size=(3000, 2))
    asym = 0
    for kernel in ['gaussian', 'uniform', 'distance', 'singular', 'symgaussian']:
        W = gl.weightmatrix.knn(X, 7, kernel=kernel)
        diff = (W != W.T).nnz
        if kernel == 'symgaussian':
            assert not glutils.known_symmetric(W)
            asym = diff
        else:
            assert glutils.known_symmetric(W), kernel
            assert diff == 0, kernel
    assert asym > 0          # fl(fl(a+b)-a) != b somewhere: the reason symgaussian graphs are not stamped
    assert not glutils.known_symmetric(gl.weightmatrix.knn(X, 7, symmetrize=False))


def test_search_on_data_sorted_by_locality(gl):
    """Data whose index order follows its geometry.  (1) Sorted by class / by a coarse locality order: a query's neighbours
    are neighbours in index; the ref ranges of the candidate lists are interleaved tiles, so they still spread over all the
    lists (a contiguous range per list sent 29 % of such rows to the exact fallback).  (2) Tight groups stored one after another: all k neighbours
    lie among the 12 rows around the query, 8 of them in the same short list -- the search notices (acceptance test of the
    re-rank) and repeats itself with the long lists instead of scanning row by row.
    Exact answer (cKDTree's) in both cases."""
    from oracle import gl_oracle as orc
    from graphlearning_amd import _hip, dist_build
    rng = np.random.default_rng(33)
    lab = np.sort(rng.integers(0, 10, size=20000))
    X = rng.normal(size=(10, 24))[lab] * 3.0 + rng.normal(size=(20000, 24))
    X = np.ascontiguousarray(X[dist_build.coarse_locality_order(X, ncells=32, seed=1)])
    J, D = gl.weightmatrix.knnsearch(X, 11)
    st = _hip.knn_stats()
    Jo, Do = orc.knnsearch(X, 11)
    assert np.array_equal(J, Jo) and np.max(np.abs(D - Do)) <= 1e-12
    assert st['escalated_rows'] == 0 and st['fallback_rows'] <= 20, st
    # (2) 2000 tight groups of 12 points, stored group after group: the 12 nearest of every point (itself included) are 12
    # consecutive rows, 8 of which share one half-wavefront list of 8 entries
    centres = rng.normal(size=(2000, 8)) * 3.0
    G = np.repeat(centres, 12, axis=0) + rng.normal(size=(24000, 8)) * 0.2
    for k in (12, 8, 5):
        J, D = gl.weightmatrix.knnsearch(G, k)
        st = _hip.knn_stats()
        Jo, Do = orc.knnsearch(G, k)
        assert np.array_equal(J, Jo) and np.max(np.abs(D - Do)) <= 1e-12, k
        print('groups of 12, k=%d: %d rows escalated, %d fallback rows, lists of %d' % (k, st['escalated_rows'], st['fallback_rows'], st['KP']))
        if k == 12:
            assert st['escalated_rows'] > 1000 and st['KP'] == 16 and st['fallback_rows'] <= 24, st   # the repeat ran with the long lists and needed no row scans


def _cell_order(X, ncells, seed=0):
    from graphlearning_amd import dist_build
    perm, starts = dist_build.coarse_locality_order(X, ncells=ncells, seed=seed, return_cells=True)
    return np.ascontiguousarray(X[perm]), np.asarray(starts, dtype=np.int64)


@pytest.mark.parametrize('case', ['blobs64', 'blobs20', 'uniform8', 'tiny_cells', 'one_cell', 'empty_cells', 'k26', 'subrange'])
def test_knn_cells_identical_to_all_pairs(gl, case, monkeypatch):
    """glx_knn_cells_range (per query block only the cells that can hold a neighbour) returns the lists of the all-pairs search
    bit for bit -- clustered data (most cells skipped), uniform data (few skipped), cells smaller than a tile, one cell, empty
    cells, longer lists, a query sub-range (the rank-local share of a sharded search)."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(11)
    k, qr = 11, None
    if case == 'blobs64':
        X, _ = blobs(60000, 64, 10, 3, 4.0)
        X, starts = _cell_order(X, 64)
    elif case == 'blobs20':
        X, _ = blobs(50000, 20, 6, 5, 6.0)
        X, starts = _cell_order(X, 48)
    elif case == 'uniform8':
        X, starts = _cell_order(rng.random((40000, 8)), 32)
    elif case == 'tiny_cells':
        X, _ = blobs(20000, 16, 5, 7, 5.0)
        X, starts = _cell_order(X, 1500)
    elif case == 'one_cell':
        X, _ = blobs(20000, 32, 4, 9, 4.0)
        starts = np.zeros(1, dtype=np.int64)
    elif case == 'empty_cells':
        X, _ = blobs(30000, 24, 6, 2, 5.0)
        X, st = _cell_order(X, 24)
        starts = np.sort(np.concatenate([st, st[5:9], [len(X), len(X)]])).astype(np.int64)     # repeated starts = empty cells
    elif case == 'k26':
        X, _ = blobs(40000, 32, 8, 4, 5.0)
        X, starts = _cell_order(X, 40)
        k = 26
    else:
        X, _ = blobs(50000, 48, 7, 6, 5.0)
        X, starts = _cell_order(X, 56)
        qr = (12345, 31000)
    J0, D0 = _hip.knn_bruteforce(X, k, query_range=qr)
    J0, D0 = np.array(J0), np.array(D0)
    J1, D1 = _hip.knn_bruteforce(X, k, query_range=qr, cell_starts=starts)
    st = _hip.knn_stats()
    assert np.array_equal(J0, J1) and np.array_equal(D0, D1)
    if case not in ('one_cell',):
        assert st['cells'] == len(starts)
    if case in ('blobs64', 'blobs20', 'k26', 'subrange'):
        assert st['visited_share'] < 0.5, st          # well-separated clusters: most (block, cell) pairs are skipped


@pytest.mark.parametrize('case', ['blobs64', 'dup_ties', 'uniform', 'small_forced', 'k26_d32'])
def test_knn_clustered_identical_to_all_pairs(gl, case, monkeypatch):
    """glx_knn_clustered (cells formed by the library, rows reordered on the device, caller's indices and rows out) returns the
    lists of the all-pairs search bit for bit -- including which of several equidistant points makes the list (duplicated points:
    the lower caller index wins in both)."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(3)
    k, m = 11, 32
    if case == 'blobs64':
        X, _ = blobs(70000, 64, 10, 3, 4.0)
    elif case == 'dup_ties':
        base, _ = blobs(9000, 12, 6, 5, 5.0)
        X = base[rng.integers(0, len(base), size=40000)]          # every point about 4 times: ties at distance 0 and beyond
    elif case == 'uniform':
        X = rng.random((50000, 10))
    elif case == 'small_forced':
        X, _ = blobs(3000, 8, 3, 1, 4.0)
        m = 7
    else:
        X, _ = blobs(45000, 32, 9, 8, 5.0)
        k = 26
    J0, D0 = _hip.knn_bruteforce(X, k, clustered=0)
    assert _hip.knn_stats()['cells'] == 0
    J0, D0 = np.array(J0), np.array(D0)
    J1, D1 = _hip.knn_bruteforce(X, k, clustered=m)
    st = _hip.knn_stats()
    assert st['cells'] == m
    assert np.array_equal(J0, J1) and np.array_equal(D0, D1)
    if case in ('blobs64', 'k26_d32'):
        assert st['visited_share'] < 0.5, st
    assert _hip.auto_cells(70000, 20) == 0 and _hip.auto_cells(1000000, 64) == 122 and _hip.auto_cells(10 ** 7, 64) == 256


def test_clustered_search_hands_its_cell_order_to_the_operator(gl, monkeypatch):
    """From 2^17 rows on weightmatrix.knn searches with cells it forms itself and leaves their (chained) order on the matrix; ssl.poisson
    gives it to the device operator instead of the library's own pass over the graph.  The order is a permutation, and nothing
    about the results depends on it: the fit equals the one on a matrix without the order, bit for bit."""
    X, lab = blobs(140000, 16, 6, 11, 5.0)
    W = gl.weightmatrix.knn(X, 10)
    order = getattr(W, '_glx_order', None)
    assert order is not None and sorted(order.tolist()) == list(range(len(X)))
    ti = gl.trainsets.generate(lab, rate=3, seed=1)
    m1 = gl.ssl.poisson(W, solver='gradient_descent')
    u1 = m1.fit(ti, lab[ti])
    assert m1._operators()[0].info()['renumbered'] == 1
    monkeypatch.setenv('GLX_KNN_ORDER', '0')
    W2 = gl.weightmatrix.knn(X, 10)
    assert getattr(W2, '_glx_order', None) is None
    assert np.array_equal(W.indptr, W2.indptr) and np.array_equal(W.indices, W2.indices) and np.array_equal(W.data, W2.data)
    m2 = gl.ssl.poisson(W2, solver='gradient_descent')
    u2 = m2.fit(ti, lab[ti])
    assert m1.num_iter == m2.num_iter and np.array_equal(u1, u2)
    assert not np.array_equal(m1._operators()[0].order(), m2._operators()[0].order())     # (two different vertex orders were in use)


@pytest.mark.gpu
@pytest.mark.parametrize('k,nsplit,short', [(8, 4, '1'), (11, 8, '1'), (21, 8, '1'), (40, 8, '1'), (11, 2, '0'), (28, 4, '0'), (60, 8, '0')])
def test_rerank_every_candidate_width(gl, monkeypatch, k, nsplit, short):
    """The re-rank ranks a query's candidates in registers (one to eight per lane: 64 .. 512 candidates) or, beyond that, in LDS
    (1024: 16 lists of 64): every width against cKDTree's lists -- ties by (distance, index) included, the data has duplicates.
    knn_options(nsplit, lists) pick the number and the length of the lists (reference weightmatrix.py:297-429)."""
    from graphlearning_amd import _hip
    from oracle import gl_oracle as orc
    rng = np.random.default_rng(900 + k)
    n, d = 5000, 24
    X = rng.normal(size=(8, d))[rng.integers(0, 8, size=n)] * 2.0 + rng.normal(size=(n, d))
    X[n - 200:] = X[:200]                                     # exact duplicates: ties at distance 0 and beyond
    with _hip.knn_options(nsplit=nsplit, lists='short' if short == '1' else 'long'):
        J, D = gl.weightmatrix.knnsearch(X, k)
    st = _hip.knn_stats()
    Jo, Do = orc.knnsearch(X, k)
    Jo, Do = Jo.reshape(n, -1), Do.reshape(n, -1)
    assert np.max(np.abs(D - Do)) <= 1e-12 * max(1.0, float(np.max(Do)))
    # duplicates make cKDTree's order among equal distances arbitrary: the lists agree as sets wherever the k-th distance is not tied
    for i in np.flatnonzero(np.any(J != Jo, axis=1)):
        assert np.array_equal(np.sort(D[i]), np.sort(Do[i]))
        untied = D[i] < D[i, -1]
        assert set(J[i][untied]) == set(Jo[i][Do[i] < Do[i, -1]]), i
    # ours: ties in ascending index order
    same = (D[:, 1:] == D[:, :-1])
    assert np.all(J[:, 1:][same] > J[:, :-1][same])
    ncand = int(abs(st['KP'])) * 2 * int(st['nsplit'])
    assert ncand in (64, 128, 256, 512, 1024), (st['KP'], st['nsplit'])


@pytest.mark.gpu
@pytest.mark.parametrize('kernel', ['gaussian', 'uniform', 'distance'])
def test_result_objects_give_the_same_matrix(gl, kernel, device_exp):
    """weightmatrix.knn keeps the neighbour lists of its own search on the device in a result object (glx_knn_search /
    glx_knn_result_to_csr; the lists never visit the host): the matrix equals the one built from lists that did, and the object
    can be consumed repeatedly and released (reference weightmatrix.py:119-187)."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(77)
    X = rng.normal(size=(6, 12))[rng.integers(0, 6, size=9000)] * 2.0 + rng.normal(size=(9000, 12))
    W1 = gl.weightmatrix.knn(X, 12, kernel=kernel)
    J, D = gl.weightmatrix.knnsearch(X, 13)
    W0 = gl.weightmatrix.knn(X, 12, kernel=kernel, knn_data=(J, D))
    assert np.array_equal(W1.indptr, W0.indptr) and np.array_equal(W1.indices, W0.indices) and np.array_equal(W1.data, W0.data)
    res = _hip.KnnResult(X, 13, want_order=True)
    J2, D2 = res.lists()
    assert np.array_equal(J2, J) and np.array_equal(D2, D)
    order = res.order()
    assert order is not None and np.array_equal(np.sort(order), np.arange(9000))
    sym = 2 if kernel in ('uniform', 'distance') else 1
    for _ in range(2):          # the object is borrowed by the assembly, not consumed
        W2 = res.to_csr(13, kernel=kernel, sym=sym)
        assert np.array_equal(W2.indptr, W1.indptr) and np.array_equal(W2.indices, W1.indices) and np.array_equal(W2.data, W1.data)
    res.close()
    res.close()                 # idempotent
    with pytest.raises(_hip.GlxError):
        _hip.KnnResult(np.full((10, 3), np.nan), 3)      # non-finite input: refused, no object is left behind


@pytest.mark.gpu
def test_result_objects_are_independent_across_threads(gl):
    """Two threads building different graphs at once: each assembly reads the result object of ITS search (no state is shared
    between searches), the matrices equal the ones built one after the other."""
    import threading
    rng = np.random.default_rng(78)
    Xs = [rng.normal(size=(5, 10))[rng.integers(0, 5, size=n)] * 2.0 + rng.normal(size=(n, 10)) for n in (8000, 8000, 9000)]
    serial = [gl.weightmatrix.knn(X, 10) for X in Xs]
    out, errs = {}, []

    def work(t):
        try:
            for rep in range(4):
                for j, X in enumerate(Xs):
                    out[(t, rep, j)] = gl.weightmatrix.knn(X, 10)
        except BaseException as exc:      # noqa: BLE001
            errs.append(exc)
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for (t, rep, j), W in out.items():
        S = serial[j]
        assert np.array_equal(W.indptr, S.indptr) and np.array_equal(W.indices, S.indices) and np.array_equal(W.data, S.data), (t, rep, j)


@pytest.mark.parametrize('kernel', ['gaussian', 'symgaussian', 'uniform'])
def test_out_of_range_neighbour_index_is_an_error_not_a_fault(gl, kernel, device_exp):
    """User-supplied knn_data with a neighbour index outside [0, n): GlxError('... out of range'), for every kernel -- the
    symgaussian weights read the neighbour's k-th distance and must not do so out of bounds first.  In the
    product's default mode (device exp: the `device_exp` fixture)."""
    from graphlearning_amd import _hip
    rng = np.random.default_rng(3)
    X = rng.normal(size=(300, 4))
    J, D = gl.weightmatrix.knnsearch(X, 8)
    for bad in (10**9, -5, 300):
        J2 = J.copy()
        J2[17, 3] = bad
        with pytest.raises(_hip.GlxError, match='out of range'):
            gl.weightmatrix.knn(None, 7, kernel=kernel, knn_data=(J2, D.copy()))
    W = gl.weightmatrix.knn(None, 7, kernel=kernel, knn_data=(J, D.copy()))      # the device is still fine afterwards
    assert W.shape == (300, 300)
