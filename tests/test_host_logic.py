"""CPU: the host-side mirror of the reference interface (graphlearning_amd) -- helpers,
graph calculus, kNN weight assembly, constructor/attribute contract, error behaviour --
against the golden vectors and the oracle; and the C-ABI library: it loads and exports
every symbol include/glx.h declares (no compute calls without a GPU)."""
import os
import re
import numpy as np
import pytest
from scipy import sparse
from conftest import csr_from, ROOT
import graphlearning_amd as gl
from graphlearning_amd import _hip
from oracle import gl_oracle as orc


def test_package_never_imports_oracle():
    import ast
    pkg = os.path.join(ROOT, 'graphlearning_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or '']
                assert not any(n.split('.')[0] == 'oracle' for n in names), fn


def test_helpers_match_golden(golden):
    g = golden('g6_helpers.npz')
    labels = np.load(os.path.join(ROOT, 'tests', 'golden', 'MNIST_labels.npz'))['labels']
    assert np.array_equal(gl.trainsets.generate(labels, rate=1, seed=0), g['gen_rate1_seed0'])
    assert np.array_equal(gl.trainsets.generate(labels, rate=3, seed=7), g['gen_rate3_seed7'])
    assert np.array_equal(np.stack(gl.trainsets.generate(labels[:5000], rate=2, num_trials=3, seed=4)), g['gen_multi'])
    assert np.array_equal(gl.trainsets.generate(labels[:5000], rate=0.01, seed=9), g['gen_frac'])
    assert np.array_equal(gl.utils.class_priors(labels), g['priors'])
    assert np.array_equal(gl.utils.labels_to_onehot(np.array([2, 0, 1, 1]), 3), g['onehot_small'])
    assert gl.utils.labels_to_onehot(np.array([0, 5]), 2).shape == (2, 6)      # width grows to max label + 1
    with pytest.raises(SystemExit):
        gl.trainsets.generate(labels, rate='x')


def test_knn_argument_errors_need_no_gpu(golden):
    g = golden('g1_twomoons.npz')
    with pytest.raises(SystemExit):
        gl.weightmatrix.knn(None, 10, kernel='nope', knn_data=(g['knn_ind'], g['knn_dist']))
    with pytest.raises(SystemExit):
        gl.weightmatrix.knnsearch(g['X'], 3, method='faiss')
    with pytest.raises(SystemExit):
        gl.weightmatrix.knnsearch(g['X'], 3, similarity='hamming')


def test_graph_calculus_matches_oracle(golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    G = gl.graph.graph(W)
    assert G.num_nodes == 500
    assert np.array_equal(G.degree_vector(), orc.degree_vector(W))
    for p in (1, -1, -0.5):
        assert (G.degree_matrix(p) != orc.degree_matrix(W, p)).nnz == 0
    for norm in ['combinatorial', 'randomwalk', 'normalized']:
        A, B = G.laplacian(norm), orc.laplacian(W, norm)
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data)
    with pytest.raises(SystemExit):
        G.laplacian('bogus')
    assert gl.graph.graph(np.eye(3)).weight_matrix.format == 'csr'


def test_learner_contract():
    W = sparse.identity(6, format='csr')
    m = gl.ssl.poisson(W)
    for attr in ['prob', 'fitted', 'name', 'accuracy_filename', 'graph', 'weights', 'class_priors',
                 'class_priors_error', 'requires_eig', 'onevsrest', 'similarity']:
        assert hasattr(m, attr)
    assert m.name == 'Poisson Learning' and m.get_accuracy_filename() == '_poisson_accuracy.csv'
    assert m.solver == 'conjugate_gradient' and m.min_iter == 50 and m.max_iter == 1000 and m.tol == 1e-3
    assert gl.ssl.poisson(W, p=2).solver == 'spectral'
    mm = gl.ssl.poisson_mbo(W, np.array([1.0, 3.0]))
    assert np.allclose(mm.class_priors, [0.25, 0.75]) and mm.get_accuracy_filename().endswith('_classpriors_accuracy.csv')
    assert mm.accuracy_filename == '_poisson_mbo_Ns_40_mu_1.00_T_20'
    ml = gl.ssl.laplace(W, normalization='normalized', tau=0.5, mean_shift=True)
    assert ml.accuracy_filename == '_laplace_normalized_meanshift_tau_0.500'
    assert gl.ssl.laplace(gl.graph.graph(W)).graph.num_nodes == 6         # a graph object is accepted as is
    with pytest.raises(SystemExit):
        gl.ssl.poisson(W, solver='bogus')
    with pytest.raises(SystemExit):
        m.predict()                                                       # not fitted yet
    with pytest.raises(SystemExit):
        gl.ssl.poisson(None).fit(np.array([0]), np.array([0]))
    assert gl.ssl.ssl_accuracy(np.array([0, 1, 1, 0]), np.array([0, 1, 0, -1]), np.array([0])) == 50.0


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the solvers must fail loudly, never fall back."""
    try:
        n = _hip.device_count()
    except _hip.GlxError:
        n = 0
    if n > 0:
        pytest.skip('GPU present')
    W = sparse.identity(6, format='csr') + sparse.diags([1.0] * 5, 1) + sparse.diags([1.0] * 5, -1)
    with pytest.raises(_hip.GlxError):
        gl.ssl.poisson(W, solver='gradient_descent').fit(np.array([0, 5]), np.array([0, 1]))
    with pytest.raises(_hip.GlxError):
        gl.weightmatrix.knnsearch(np.random.rand(10, 3), 3)
    with pytest.raises(_hip.GlxError):
        gl.weightmatrix.knn(None, 2, knn_data=(np.zeros((4, 3), dtype=np.int64), np.zeros((4, 3))))
    for model in (gl.ssl.poisson(W), gl.ssl.laplace(W), gl.ssl.randomwalk(W), gl.ssl.poisson_mbo(W, np.array([0.5, 0.5]))):
        with pytest.raises(_hip.GlxError):
            model.fit(np.array([0, 5]), np.array([0, 1]))
    with pytest.raises(_hip.GlxError):
        gl.ssl.laplace(W).ssl_trials([np.array([0, 5]), np.array([1, 4])], np.array([0, 0, 0, 1, 1, 1]), save_results=False)
    with pytest.raises(_hip.GlxError):
        gl.graph(W).page_rank()
    with pytest.raises(_hip.GlxError):
        gl.graph(W).plaplace(np.array([0, 5]), np.array([0.0, 1.0]), 4, fast=False)


def test_plaplace_default_is_refused_not_emulated():
    """The reference's default fast=True is a sequential Gauss-Seidel sweep: no silent substitute."""
    W = sparse.identity(6, format='csr') + sparse.diags([1.0] * 5, 1) + sparse.diags([1.0] * 5, -1)
    with pytest.raises(NotImplementedError):
        gl.graph(W).plaplace(np.array([0, 5]), np.array([0.0, 1.0]), 4)
    assert gl.graph is gl.graph.graph                       # gl.graph(W) like the reference, gl.graph.graph(W) still works


def _declared(header):
    hdr = open(os.path.join(ROOT, 'include', header)).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return set(re.findall(r'\b(glx_[a-z0-9_]+)\s*\(', hdr))


def test_cabi_exports_every_declared_symbol():
    """Every symbol of include/glx.h (the drop-in boundary: at most 61 functions -- round 4's cap of 60 plus glx_nearest_dist, the one entry point
    graph.reweight(method='properly') added in round 6) and of include/glx_experimental.h
    (laboratory hooks, host helpers, the device-pointer calls of the fallback engine) is exported by libglx.so, and the ctypes
    binding covers exactly that surface."""
    core, lab = _declared('glx.h'), _declared('glx_experimental.h')
    assert 20 <= len(core) <= 61, len(core)
    assert not (core & lab), core & lab
    for name in ('glx_knn_set_options', 'glx_dist_sweep_begin', 'glx_dist_sweep_boundary', 'glx_dist_sweep_get_send', 'glx_dist_sweep_put_halo',
                 'glx_dist_sweep_interior', 'glx_graph_order', 'glx_dist_sweep_time_parts'):
        assert name in lab, name
    declared = core | lab
    lib = _hip.load()
    for sym in sorted(declared):
        assert getattr(lib, sym) is not None, sym
    # and the Python binding covers exactly the declared surface
    assert declared == set(_hip.EXPORTED_SYMBOLS), declared ^ set(_hip.EXPORTED_SYMBOLS)
    assert lib.glx_version() >= 100


def test_laplace_rhs_fast_path_equals_scipy_expression():
    """ssl._neg_columns_times == `-L[:, cols] * F` bit for bit (several labelled neighbours of one
    class in a row, unsorted `cols`, explicit zeros), and falls back on non-canonical input."""
    from graphlearning_amd import ssl as glssl
    rng = np.random.default_rng(0)
    for n, dens, m, k in [(300, 0.08, 60, 3), (1000, 0.02, 200, 10), (50, 0.5, 30, 2)]:
        A = sparse.random(n, n, density=dens, random_state=int(rng.integers(1 << 30)), format='csr')
        L = sparse.csr_matrix(sparse.diags(np.asarray(A.sum(axis=1)).ravel()) - A - A.T)
        L.sum_duplicates()
        L.sort_indices()
        cols = rng.permutation(n)[:m]
        F = np.eye(k)[rng.integers(0, k, m)]
        ref = -L[:, cols] * F
        got = glssl._neg_columns_times(L, L.tocsc(), cols, F)
        assert np.array_equal(got, ref)
    # unsorted rows: the literal expression is used
    L2 = L.copy()
    L2.indices[L2.indptr[0]:L2.indptr[1]] = L2.indices[L2.indptr[0]:L2.indptr[1]][::-1]
    L2.data[L2.indptr[0]:L2.indptr[1]] = L2.data[L2.indptr[0]:L2.indptr[1]][::-1]
    L2.has_sorted_indices = False
    assert np.array_equal(glssl._neg_columns_times(L2, L2.tocsc(), cols, F), -L2[:, cols] * F)


def test_laplace_rhs_rows_of_the_library_equal_the_scipy_expression():
    """_hip.host_neg_columns_rows (glx_host_neg_columns_rows, what ssl.laplace.fit sends up) == the rows of `M * (-L[:, cols] * F)`
    (reference ssl.py:1236, 1249) bit for bit: rows ascending, the labelled rows left out, every other row of the product zero; the
    numpy form of the same sums (ssl._neg_columns_times_rows) agrees; duplicate columns are handed back (None)."""
    from graphlearning_amd import ssl as glssl
    rng = np.random.default_rng(1)
    for n, dens, m, k in [(300, 0.08, 60, 3), (1000, 0.02, 200, 10), (50, 0.5, 30, 2), (400, 0.01, 1, 4)]:
        A = sparse.random(n, n, density=dens, random_state=int(rng.integers(1 << 30)), format='csr')
        L = sparse.csr_matrix(sparse.diags(np.asarray(A.sum(axis=1)).ravel() + 0.5) - A - A.T)
        L.sum_duplicates()
        L.sort_indices()
        cols = rng.permutation(n)[:m]
        F = np.eye(k)[rng.integers(0, k, m)] + (rng.normal(size=(m, k)) if n == 50 else 0.0)
        Mv = 1 / np.sqrt(L.diagonal() + 1e-10)
        ref = Mv[:, None] * (-L[:, cols] * F)
        rows, vals = _hip.host_neg_columns_rows(L.tocsc(), cols, F, row_scale=Mv)
        assert rows.dtype == np.int32 and np.all(np.diff(rows) > 0) and not np.isin(rows, cols).any()
        assert np.array_equal(vals, ref[rows])
        rest = np.ones(n, dtype=bool)
        rest[rows] = False
        rest[cols] = False
        assert not ref[rest].any()
        r2, b2 = glssl._neg_columns_times_rows(L, L.tocsc(), cols, F)
        keep = ~np.isin(r2, cols)
        assert np.array_equal(r2[keep], rows) and np.array_equal(Mv[r2[keep], None] * b2[keep], vals)
        r3, v3 = _hip.host_neg_columns_rows(L.tocsc(), cols, F)          # no scaling
        assert np.array_equal(r3, rows) and np.array_equal(v3, (-L[:, cols] * F)[rows])
    assert _hip.host_neg_columns_rows(L.tocsc(), np.array([3, 5, 3]), np.eye(4)[[0, 1, 2]]) is None
    with pytest.raises(_hip.GlxError):
        _hip.load().glx_host_neg_columns_rows  # the symbol exists ...
        _hip.check(_hip.load().glx_host_neg_columns_rows(n, None, None, None, 0, None, None, 1, None, 0, None, None, None), 'null arguments')


def test_load_knn_data_reads_the_reference_cache_file(golden, monkeypatch):
    """./knn_data/<dataset>_<metric>.npz with keys J, D (reference weightmatrix.py:416-427, 451-465).
    tests/golden/knn_data/glxtoy_raw.npz was written by the reference's own knnsearch(dataset='GlxToy')."""
    from graphlearning_amd import weightmatrix as wm
    g = golden('g10_knn_cache.npz')
    monkeypatch.setattr(wm, 'knn_dir', os.path.join(ROOT, 'tests', 'golden', 'knn_data'))
    for name, metric in (('glxtoy', 'raw'), ('GLXtoy', 'RAW')):       # not case-sensitive (:418, :452)
        J, D = wm.load_knn_data(name, metric=metric)
        assert J.shape == (300, 8) and np.array_equal(J, g['J']) and np.array_equal(D, g['D'])
    with pytest.raises(SystemExit):
        wm.load_knn_data('no_such_dataset')


def test_symmetric_poisson_operator_equals_scipy_expressions(golden):
    """ssl._poisson_operator_symmetric writes down D^-1 W^T of a symmetric W without transposing: indptr, indices (descending
    inside a row, the order scipy's csr product leaves) and data must equal the reference's own expressions
    (ssl.py:615-617, 634-635) entry for entry; the stamp weightmatrix.knn puts on its output is invalidated by edits."""
    from graphlearning_amd import ssl as glssl, utils as glutils, graph as glgraph
    from oracle import gl_oracle as orc
    from conftest import csr_from
    cases = [csr_from(golden('g1_twomoons.npz'), 'W_gaussian'), csr_from(golden('g1_twomoons.npz'), 'W_uniform'), csr_from(golden('g3_blobs5000.npz'), 'W')]
    R = sparse.random(300, 300, density=0.03, random_state=3, format='csr')
    R = (R + R.T).tocsr()
    R.setdiag(0)
    R.eliminate_zeros()
    cases.append(R)
    for W in cases:
        n = W.shape[0]
        Wz = W - sparse.spdiags(W.diagonal(), 0, n, n)
        D = sparse.spdiags(orc.degree_vector(Wz) ** (-1), 0, n, n).tocsr()
        P_ref = sparse.csr_matrix(D * Wz.transpose())
        P, deg, dinv = glssl._poisson_operator_symmetric(sparse.csr_matrix(W))
        assert np.array_equal(P.indptr, P_ref.indptr) and np.array_equal(P.indices, P_ref.indices)
        assert np.array_equal(P.data, P_ref.data)
        assert np.array_equal(deg, orc.degree_vector(Wz)) and np.array_equal(dinv, D.diagonal())
    # the stamp: valid on the stamped arrays and through graph(), gone after an in-place edit or a structural one
    W = sparse.csr_matrix(cases[0])
    W._glx_sym = glutils.symmetric_fingerprint(W)
    assert glutils.known_symmetric(W) and glutils.known_symmetric(glgraph.graph(W).weight_matrix)
    W.data[3] *= 1.5
    assert not glutils.known_symmetric(W)
    W2 = sparse.csr_matrix(cases[0])
    W2._glx_sym = glutils.symmetric_fingerprint(W2)
    W3 = W2 + sparse.identity(W2.shape[0])
    assert not glutils.known_symmetric(W3) and not glutils.known_symmetric(sparse.csr_matrix(cases[0]))


def test_matrix_fingerprint_sees_in_place_edits(golden):
    """The device-resident operators of the learners are keyed by utils.matrix_fingerprint (ADVICE / VERDICT r02: an id(W) key
    reused a stale operator after an in-place edit of W.data; the reference rebuilds its operator in every fit, ssl.py:615-644):
    equal content -> equal key (also for a copy); ANY in-place edit -- one value by one ulp, two values swapped (which leaves every
    sum unchanged), an index -- gives another key."""
    from graphlearning_amd import utils as glutils
    from conftest import csr_from
    W = sparse.csr_matrix(csr_from(golden('g1_twomoons.npz'), 'W_gaussian'))
    f0 = glutils.matrix_fingerprint(W)
    assert f0 == glutils.matrix_fingerprint(W.copy()) and f0[0] == W.shape and f0[1] == W.nnz
    W.data[17] = np.nextafter(W.data[17], 1.0)
    f1 = glutils.matrix_fingerprint(W)
    assert f1 != f0
    a, b = W.data[5], W.data[9]
    assert a != b
    W.data[5], W.data[9] = b, a                      # sums of the data are unchanged
    f2 = glutils.matrix_fingerprint(W)
    assert f2 != f1
    W.indices[0], W.indices[1] = W.indices[1], W.indices[0]
    assert glutils.matrix_fingerprint(W) != f2
    # W[i, j] = v on an existing entry is an in-place edit of W.data
    W2 = sparse.csr_matrix(csr_from(golden('g1_twomoons.npz'), 'W_gaussian'))
    g0 = glutils.matrix_fingerprint(W2)
    i, j = 0, int(W2.indices[0])
    W2[i, j] = 0.123
    assert glutils.matrix_fingerprint(W2) != g0


def test_gaussian_weight_blocks_and_auto_cells(monkeypatch):
    """Host pieces of weightmatrix.knn added in round 3: numpy's exp over row blocks on host threads gives the bits of the one-call
    expression (reference weightmatrix.py:144-150), and the policy that decides when the library forms cells for the search."""
    from graphlearning_amd import weightmatrix as wm, _hip
    rng = np.random.default_rng(5)
    n, k = 50000, 11
    d = np.sort(rng.random((n, k)), axis=1)
    d[:, 0] = 0.0
    J = rng.integers(0, n, size=(n, k))
    D = d * d
    ref_g = np.exp(-4 * D / D[:, k - 1][:, None])
    eps = d[:, k - 1]
    ref_s = np.exp(-4 * d * d / eps[:, None] / eps[J])
    out_g, out_s = np.empty((n, k)), np.empty((n, k))

    def rows_g(lo, hi):
        Db = d[lo:hi] * d[lo:hi]
        np.exp(-4 * Db / Db[:, k - 1][:, None], out=out_g[lo:hi])

    def rows_s(lo, hi):
        np.exp(-4 * d[lo:hi] * d[lo:hi] / eps[lo:hi, None] / eps[J[lo:hi]], out=out_s[lo:hi])
    wm._row_blocks(rows_g, n)
    wm._row_blocks(rows_s, n)
    assert np.array_equal(out_g, ref_g) and np.array_equal(out_s, ref_s)
    monkeypatch.setenv('GLX_HOST_THREADS', '1')
    out_g[:] = 0
    wm._row_blocks(rows_g, n)
    assert np.array_equal(out_g, ref_g)
    monkeypatch.delenv('GLX_KNN_CLUSTERED', raising=False)
    assert _hip.auto_cells(70000, 20) == 0 and _hip.auto_cells(131072, 64) == 64 and _hip.auto_cells(10 ** 6, 64) == 122
    assert _hip.auto_cells(10 ** 7, 64) == 256 and _hip.auto_cells(10 ** 6, 200) == 0
    monkeypatch.setenv('GLX_KNN_CLUSTERED', '0')
    assert _hip.auto_cells(10 ** 6, 64) == 0
    monkeypatch.setenv('GLX_KNN_CLUSTERED', '48')
    assert _hip.auto_cells(1000, 3) == 48


def test_order_cells_policy(monkeypatch):
    from graphlearning_amd import _hip
    monkeypatch.delenv('GLX_KNN_ORDER', raising=False)
    assert _hip.auto_order_cells(70000, 20) == 128 and _hip.auto_order_cells(5000, 8) == 78 and _hip.auto_order_cells(4000, 8) == 0
    assert _hip.auto_order_cells(1 << 17, 20) == 0 and _hip.auto_order_cells(70000, 200) == 0
    monkeypatch.setenv('GLX_KNN_ORDER', '0')
    assert _hip.auto_order_cells(70000, 20) == 0


def test_exact_mode_form_switch_maps_to_the_cabi_flags(monkeypatch):
    """_hip.CG_EXACT_FORM (tests / measurements: pytest --cg-form) -> the GLX_CG_BLOCKS / GLX_CG_CHAIN bits of include/glx.h; the
    tolerance mode ignores it; anything else is refused"""
    import re
    from graphlearning_amd import _hip
    hdr = open(os.path.join(ROOT, 'include', 'glx.h')).read()
    flag = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define (GLX_CG_[A-Z0-9]+) (\d+)', hdr)}
    assert (flag['GLX_CG_NP1D'], flag['GLX_CG_TREE'], flag['GLX_CG_X0'], flag['GLX_CG_BLOCKS'], flag['GLX_CG_CHAIN']) == \
        (_hip.GLX_CG_NP1D, _hip.GLX_CG_TREE, _hip.GLX_CG_X0, _hip.GLX_CG_BLOCKS, _hip.GLX_CG_CHAIN)
    assert len(set(flag.values())) == len(flag) and all(v & (v - 1) == 0 for v in flag.values())      # distinct single bits
    for form, want in ((None, 0), ('blocks', flag['GLX_CG_BLOCKS']), ('chain', flag['GLX_CG_CHAIN'])):
        monkeypatch.setattr(_hip, 'CG_EXACT_FORM', form)
        assert _hip._reduce_flag('exact') == want
        assert _hip._reduce_flag('tree') == flag['GLX_CG_TREE']
    monkeypatch.setattr(_hip, 'CG_EXACT_FORM', 'fast')
    with pytest.raises(_hip.GlxError):
        _hip._reduce_flag('exact')


def test_captured_launch_sequences_hold_no_memset_node():
    """Round 6 (profiles/r06_graph_memset_probe.txt): on the HIP runtime bundled with PyTorch a memset NODE of a replayed launch graph writes the
    value of the process's last eager hipMemset.  Whatever the library enqueues between hipStreamBeginCapture and hipStreamEndCapture therefore
    clears memory with a kernel (glx_zero_async).  A source check: the functions that are captured contain no hipMemset call."""
    csrc = os.path.join(ROOT, 'graphlearning_amd', 'csrc')

    def body(fname, start, end):
        text = open(os.path.join(csrc, fname)).read()
        i = text.index(start)
        return text[i:text.index(end, i + len(start))]

    captured = {
        'solver.hip: enqueue_head': body('solver.hip', 'static int enqueue_head(glx_sweep* s)', 'extern "C" int glx_sweep_run'),
        'solver.hip: launch_sweep': body('solver.hip', 'static int launch_sweep(', 'static int enqueue_head('),
        'groups.hip: grp_launch_sweep + grp_enqueue_head': body('groups.hip', 'static int grp_launch_sweep(', 'static double grp_err_value('),
        'cg.hip: enqueue_iteration + enqueue_chunk': body('cg.hip', 'auto enqueue_iteration = ', '// Full chunks are replayed from a captured launch sequence'),
        'cg_fused.hip: enqueue_chunk': body('cg_fused.hip', 'auto enqueue_chunk = ', 'hipStreamBeginCapture'),
        'dist.hip: enqueue_reset .. enqueue_sweep helpers': body('dist.hip', 'static int enqueue_reset(', 'static int ensure_ring('),
    }
    for name, text in captured.items():
        code = '\n'.join(line.split('//')[0] for line in text.splitlines())
        assert 'hipMemset' not in code, name
    # the lambdas handed to run_captured (dist.hip): from the call to the closing of the lambda
    text = open(os.path.join(csrc, 'dist.hip')).read()
    for m in re.finditer(r'run_captured\(s, \{[^}]*\}, \[&\]\(\) -> int \{', text):
        lam = text[m.end():text.index('});', m.end())]
        assert 'hipMemset' not in '\n'.join(line.split('//')[0] for line in lam.splitlines()), lam[:200]
    assert 'glx_zero_async' in captured['solver.hip: enqueue_head'] and 'glx_zero_async' in captured['groups.hip: grp_launch_sweep + grp_enqueue_head']


def test_block_census_agrees_with_the_walk():
    """tests/seqsum_census.py's classifier (ss_host_census) and the lane-for-lane restatement of the device walk (ss_host_walk) count the same
    blocks on the same column: plain + empty = the walk's plain, record, row by row."""
    import ctypes, subprocess, tempfile
    lib = os.path.join(tempfile.mkdtemp(), 'libss_census.so')
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', lib, os.path.join(ROOT, 'tests', 'seqsum_host.cpp')], check=True)
    L = ctypes.CDLL(lib)
    L.ss_host_walk.restype = ctypes.c_double
    rng = np.random.default_rng(5)
    for x in (rng.normal(size=40000) ** 2, rng.normal(size=40000) * rng.normal(size=40000) + 0.05, np.concatenate([np.zeros(3000), rng.random(20000)])):
        x = np.ascontiguousarray(x)
        out = (ctypes.c_int64 * 8)()
        L.ss_host_census(ctypes.c_void_p(x.ctypes.data), ctypes.c_int64(x.size), ctypes.c_int64(1), out, 0)
        st = (ctypes.c_int64 * 3)()
        L.ss_host_walk(ctypes.c_void_p(x.ctypes.data), ctypes.c_int64(x.size), ctypes.c_double(0.0), st)
        c = list(out)
        assert c[0] + c[1] == st[0] and c[2] == st[1] and c[3] + c[4] + c[5] == st[2], (c, list(st))
