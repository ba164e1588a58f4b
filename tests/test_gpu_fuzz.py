"""Randomised end-to-end parity (through the C-ABI): random point clouds, graph parameters and training sets; every
stage of the path -- search, weight matrix, Poisson (both solvers), Laplace, PoissonMBO, the label decision -- is held
against the oracle bit for bit.  The goldens pin a few shapes; this walks many (sizes 40 .. 1500, 2 .. 6 classes,
3 .. 15 neighbours, all kernels, directed and symmetrised graphs, every normalisation)."""
import numpy as np
import pytest
from scipy import sparse
from scipy.sparse import csgraph

pytestmark = pytest.mark.gpu

# GLX_FUZZ_SCALE=10 runs ten times as many random cases of every kind (a soak run; the default is what CI runs)
_SCALE = int(__import__('os').environ.get('GLX_FUZZ_SCALE', '1'))
# GLX_FUZZ_BASE=b shifts every seed range by b: concurrent soak sessions walk DIFFERENT cases (scripts/soak.sh gives session i base i * 10^5)
_BASE = int(__import__('os').environ.get('GLX_FUZZ_BASE', '0'))


def _seeds(k):
    return range(_BASE, _BASE + k * _SCALE)


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


# The two modes of weightmatrix.knn's Gaussian weights: 'host_exp' (GLX_HOST_EXP=1: numpy's exp on this host -- the bits of the
# reference run here, what the golden vectors hold) and 'device_exp' (the PRODUCT'S DEFAULT: the correctly rounded exp on the
# device).  In the default mode W differs from this host's reference by at most one ulp per exponential, so the contract has two
# halves: (1) every solver stage is bit-identical to the oracle applied to the SAME W (nothing but the exp differs from the
# reference's arithmetic), (2) against the reference's own W the north star's contract -- structure identical, weights within an
# ulp (two for a symmetrised sum), T equal, labels equal, iterates within 1e-5 -- with every deviation RECORDED with its margin
# (gpurun_out/default_mode_deviations.txt) and allowed only where the reference's own answer hangs on a rounding: a stop value or a
# residual within 1e-9 of its threshold, a label whose two best scores tie to 1e-9.  ssl.poisson's default CG solves a SINGULAR
# system to 1e-3: its iteration count is decided by rounding noise (tests/test_gpu_weights.py: 140 vs 462 at config 2 for one-ulp
# weights) -- for that stage only half (1) is asserted and the difference to the reference's run is recorded.
@pytest.fixture(params=['host_exp', 'device_exp'])
def mode(request):
    import os
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


def _ulps(a, b):
    return np.abs(np.ascontiguousarray(a, dtype=np.float64).view(np.int64) - np.ascontiguousarray(b, dtype=np.float64).view(np.int64))


def _record(line):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(root, 'gpurun_out', 'default_mode_deviations.txt'), 'a') as f:
        f.write(line + '\n')


def _north_star(tag, what, u, u_ref, pred=None, pred_ref=None, count=None, count_ref=None, near=None, strict_count=True, assert_du=True):
    """Half (2) of the default-mode contract against the run on the reference's own W.  `near`: how close (relative) the
    reference's run came to the threshold that decides `count` -- a deviation of the count is allowed (and recorded) only
    when the decision hung on a rounding."""
    scale = max(1.0, float(np.nanmax(np.abs(u_ref)))) if np.size(u_ref) else 1.0
    du = float(np.nanmax(np.abs(np.asarray(u, dtype=np.float64) - u_ref))) if np.size(u_ref) else 0.0
    if count is not None and count != count_ref:
        _record('%s | %s: count %s vs the reference-W run %s (reference run within %.3e of its threshold), max |du| %.3e'
                % (tag, what, count, count_ref, -1.0 if near is None else near, du))
        if strict_count:
            assert near is not None and near <= 1e-9, (tag, what, count, count_ref, near)
        return
    if not assert_du:
        if du > 1e-5 * scale:
            _record('%s | %s: same count %s, max |du| %.3e (max |u| %.3e)' % (tag, what, count, du, scale))
        return
    assert du <= 1e-5 * scale, (tag, what, du)
    if pred is not None and not np.array_equal(pred, pred_ref):
        rows = np.flatnonzero(np.asarray(pred) != np.asarray(pred_ref))
        srt = np.sort(np.asarray(u_ref, dtype=np.float64)[rows], axis=1)
        gap = float(np.max(srt[:, -1] - srt[:, -2]))
        _record('%s | %s: %d labels differ from the reference-W run; largest gap between their two best scores %.3e (max |u| %.3e)'
                % (tag, what, len(rows), gap, scale))
        assert gap <= 1e-9 * scale, (tag, what, len(rows), gap)


def _gd_stop_margin(orc, W, train_ind, min_iter=50, max_iter=1000):
    """How close (relative to 1/n) the reference's stop values max|v_t - vinf| (ssl.py:667) come to the threshold 1/n."""
    n = W.shape[0]
    W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
    D = orc.degree_matrix(W, p=-1)
    deg = orc.degree_vector(W)
    v = np.zeros(n)
    v[train_ind] = 1
    v = v / np.sum(v)
    vinf = deg / np.sum(deg)
    RW = W.transpose() * D
    T, best = 0, np.inf
    while T < max_iter:
        e = np.max(np.absolute(v - vinf))
        if T >= min_iter:
            best = min(best, abs(e * n - 1.0))
            if not (e > 1 / n):
                break
        v = RW * v
        T += 1
    return float(best)


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(40, 1500))
    d = int(rng.choice([2, 3, 5, 8, 20, 40]))
    C = int(rng.integers(2, 7))
    k = int(rng.integers(3, 16))
    lab = rng.integers(0, C, size=n)
    lab[:C] = np.arange(C)                                  # every class occurs
    centres = rng.normal(size=(C, d)) * rng.uniform(0.5, 3.0)
    X = centres[lab] + rng.normal(size=(n, d))
    kernel = str(rng.choice(['gaussian', 'gaussian', 'uniform', 'distance', 'singular', 'symgaussian']))
    symmetrize = bool(rng.random() < 0.8)
    per_class = int(rng.integers(1, 4))
    ti = np.concatenate([rng.choice(np.flatnonzero(lab == c), size=min(per_class, int(np.sum(lab == c))), replace=False)
                         for c in range(C)])
    return dict(n=n, d=d, C=C, k=k, X=X, lab=lab.astype(np.int64), kernel=kernel, symmetrize=symmetrize, ti=ti, rng=rng)


@pytest.mark.parametrize('seed', _seeds(24))
def test_random_pipeline_matches_the_oracle(gl, orc, seed, mode):
    c = _case(seed)
    X, lab, ti, k = c['X'], c['lab'], c['ti'], c['k']
    tag = '%s seed %d: n=%d d=%d C=%d k=%d %s sym=%s' % (mode, seed, c['n'], c['d'], c['C'], k, c['kernel'], c['symmetrize'])
    # a-1: the search (k + 1 columns incl. self), cKDTree's lists
    J, D = gl.weightmatrix.knnsearch(X, k + 1)
    Jo, Do = orc.knnsearch(X, k + 1)
    assert np.array_equal(J, Jo), tag
    assert np.max(np.abs(D - Do)) <= 1e-12, tag
    # a-2: the weight matrix from the SAME kNN data
    W = gl.weightmatrix.knn(None, k, kernel=c['kernel'], symmetrize=c['symmetrize'], knn_data=(Jo, Do.copy()))
    Wo = orc.knn_weights(Jo, Do.copy(), k, kernel=c['kernel'], symmetrize=c['symmetrize'])
    assert np.array_equal(W.indptr, Wo.indptr) and np.array_equal(W.indices, Wo.indices), tag
    device = mode == 'device_exp' and c['kernel'] in ('gaussian', 'symgaussian')
    if not device:
        assert np.array_equal(W.data, Wo.data), tag
    else:
        # one ulp per exponential; (a + b) / 2 of two of them at most two; symgaussian's rule subtracts (fl(fl(a+b)-a)): a few ulps
        # of the smaller entry (tests/test_gpu_weights.py)
        most = 1 if not c['symmetrize'] else (2 if c['kernel'] == 'gaussian' else 8)
        assert int(_ulps(W.data, Wo.data).max(initial=0)) <= most, (tag, int(_ulps(W.data, Wo.data).max()))
    if (not c['symmetrize'] or c['kernel'] in ('distance', 'singular')) and not device:
        # directed graphs / unbounded weights: the sweep is the part of the path defined for them
        W = Wo
    Wsame = sparse.csr_matrix(W) if device else Wo              # the matrix the oracle is handed for half (1)
    with np.errstate(all='ignore'):
        # a-3: gradient descent incl. the stop test
        u_ref, T_ref = orc.poisson_gd(Wsame, ti, lab[ti], return_T=True)
        m = gl.ssl.poisson(W, solver='gradient_descent')
        u = m.fit(ti, lab[ti])
        assert m.num_iter == T_ref, tag
        assert np.array_equal(u, u_ref, equal_nan=True), tag
        pred = m.predict()
        assert np.array_equal(pred, orc.predict(u_ref)), tag
        if device:
            uo, To = orc.poisson_gd(Wo, ti, lab[ti], return_T=True)
            near = _gd_stop_margin(orc, Wo, ti) if To != T_ref else None
            _north_star(tag, 'poisson GD', u, uo, pred, orc.predict(uo), T_ref, To, near)
        if not c['symmetrize'] or c['kernel'] in ('distance', 'singular'):
            return
        # a-4: conjugate gradient on the singular system
        u_ref, it_ref = orc.poisson_cg(Wsame, ti, lab[ti], return_iters=True)
        m = gl.ssl.poisson(W)
        u = m.fit(ti, lab[ti])
        assert m.num_iter == it_ref, tag
        assert np.array_equal(u, u_ref, equal_nan=True), tag
        if device:
            uo, ito = orc.poisson_cg(Wo, ti, lab[ti], return_iters=True)
            _north_star(tag, 'poisson CG (singular system: count and null-space component hang on rounding noise)', u, uo, None, None, it_ref,
                        ito, None, strict_count=False, assert_du=False)
        # a-5: Laplace, a random normalisation / tau / mean shift
        norm = str(c['rng'].choice(['combinatorial', 'randomwalk', 'normalized']))
        tau = float(c['rng'].choice([0.0, 0.0, 0.01]))
        shift = bool(c['rng'].random() < 0.3)
        u_ref, it_ref = orc.laplace_fit(Wsame, ti, lab[ti], normalization=norm, tau=tau, mean_shift=shift, return_iters=True)
        m = gl.ssl.laplace(W, normalization=norm, tau=tau, mean_shift=shift, reduce='exact')
        u = m.fit(ti, lab[ti])
        assert m.num_iter == it_ref, (tag, norm, tau, shift)
        assert np.array_equal(u, u_ref, equal_nan=True), (tag, norm, tau, shift)
        if device:
            uo, ito = orc.laplace_fit(Wo, ti, lab[ti], normalization=norm, tau=tau, mean_shift=shift, return_iters=True)
            # The system is DEFINITE iff tau > 0 or every connected component of W holds a labelled vertex; otherwise the harmonic extension
            # is not unique on the unlabelled components, CG's answer there is whatever rounding leaves (seed 458 at 20 x the seeds: k = 3
            # in two dimensions, five of six components without a label -- same count, the unlabelled components 2e-3 apart) and the
            # reference's own result hangs on the last bits of W: the rule of the singular Poisson system, recorded, not asserted.
            ncomp, comp = csgraph.connected_components(Wo, directed=False)
            definite = tau > 0 or len(np.unique(comp[ti])) == ncomp
            if not np.all(np.isfinite(uo)):
                # the reference's own solve BROKE DOWN on its W: a column whose residual is exactly zero makes utils.conjgrad divide
                # 0 / 0 (utils.py:524), the NaN column never satisfies the stop and the other columns iterate on rounding noise for
                # thousands of steps (seed 214 at 9 x the seeds: six separate components, mean shift -- 3752 iterations on the
                # reference's W, 4522 on the device's).  Half (1) above held bit for bit, NaN column included; a count decided by noise
                # is recorded, not asserted
                _record('%s | laplace %s: the reference-W run broke down (non-finite iterate, %d iterations); %d iterations here, finite columns within %.3e'
                        % (tag, norm, ito, it_ref, float(np.nanmax(np.abs(u - uo))) if np.any(np.isfinite(u - uo)) else float('nan')))
            elif max(ito, it_ref) >= 100000:
                # the reference's own solve did NOT CONVERGE on its W (utils.conjgrad's max_iter = 1e5 ran out: the Jacobi-scaled
                # random-walk Laplacian is not symmetric, CG may stagnate -- seed 100068: 837 vertices, one label per class, 100 000
                # iterations; one-ulp changes of the weights move the reference's own answer by 0.1).  Half (1) above held bit for bit
                _record('%s | laplace %s: the reference-W run did not converge (%d iterations; %d here), max |du| %.3e'
                        % (tag, norm, ito, it_ref, float(np.nanmax(np.abs(u - uo)))))
            elif not definite:
                _north_star(tag, 'laplace %s (singular: %d of %d components without a label)' % (norm, ncomp - len(np.unique(comp[ti])), ncomp),
                            u, uo, None, None, it_ref, ito, None, strict_count=False, assert_du=False)
            # (an SPD system stopped at 1e-5: a one-ulp weight may move the stop by an iteration; the iterates then differ by ~tol)
            elif abs(it_ref - ito) <= 1 and it_ref != ito:
                _record('%s | laplace %s: %d iterations vs the reference-W run %d, max |du| %.3e' % (tag, norm, it_ref, ito, float(np.nanmax(np.abs(u - uo)))))
                assert np.nanmax(np.abs(u - uo)) <= 1e-4 * max(1.0, np.nanmax(np.abs(uo))), (tag, norm)
            else:
                _north_star(tag, 'laplace ' + norm, u, uo, m.predict(), orc.predict(uo), it_ref, ito, None)
        # a-6 / a-7: PoissonMBO with the volume constraint (short schedule)
        priors = orc.class_priors(lab)
        u_ref, lab_ref, w_ref = orc.poisson_mbo_fit(Wsame, ti, lab[ti], priors, solver='gradient_descent', Ns=12, T=4)
        m = gl.ssl.poisson_mbo(W, priors, solver='gradient_descent', Ns=12, T=4)
        pred = m.fit_predict(ti, lab[ti])
        assert np.array_equal(m.prob, u_ref), tag
        assert np.array_equal(pred, lab_ref), tag
        assert np.array_equal(np.asarray(m.weights), np.asarray(w_ref)), tag
        if device:
            # (thresholding is discontinuous: the comparison with the reference-W run is a count of labels, recorded when not zero)
            _, lab_o, _ = orc.poisson_mbo_fit(Wo, ti, lab[ti], priors, solver='gradient_descent', Ns=12, T=4)
            nd = int(np.sum(np.asarray(pred) != np.asarray(lab_o)))
            if nd:
                _record('%s | poisson_mbo: %d of %d labels differ from the reference-W run' % (tag, nd, c['n']))
            assert nd <= max(2, c['n'] // 100), (tag, nd)


@pytest.mark.parametrize('seed', _seeds(12))
def test_random_trials_and_comparison_methods_match_the_oracle(gl, orc, seed, mode):
    """Stacked trials (several training sets as column groups of one solve), the float32 branch, and the comparison
    methods of SURVEY 8 f-3 on random symmetric graphs."""
    c = _case(100 + seed)
    X, lab, k, rng = c['X'], c['lab'], c['k'], c['rng']
    C = c['C']
    Jo, Do = orc.knnsearch(X, k + 1)
    Wo = orc.knn_weights(Jo, Do.copy(), k)
    W = gl.weightmatrix.knn(X, k)
    assert np.array_equal(W.indices, Wo.indices) and np.max(np.abs(W.data - Wo.data)) <= 1e-12
    W = gl.weightmatrix.knn(None, k, knn_data=(Jo, Do.copy()))
    if mode == 'host_exp':
        assert np.array_equal(W.data, Wo.data)
    else:
        # the product's default: weights within two ulps of this host's reference ((a + b) / 2 of two exponentials); every solver
        # below is then held against the oracle on the SAME matrix
        assert np.array_equal(W.indices, Wo.indices) and int(_ulps(W.data, Wo.data).max(initial=0)) <= 2
        Wo = sparse.csr_matrix(W)
    tag = '%s seed %d: n=%d d=%d C=%d k=%d' % (mode, seed, c['n'], c['d'], C, k)
    sets = []
    for _ in range(int(rng.integers(2, 6))):
        per_class = int(rng.integers(1, 4))
        sets.append(np.concatenate([rng.choice(np.flatnonzero(lab == cc), size=min(per_class, int(np.sum(lab == cc))), replace=False)
                                    for cc in range(C)]))
    with np.errstate(all='ignore'):
        # f-1: each stacked trial equals its own solve (iterates and iteration count)
        m = gl.ssl.poisson(W)
        probs = m._fit_batch([(t, lab[t]) for t in sets])
        for j, t in enumerate(sets):
            u_ref, it_ref = orc.poisson_cg(Wo, t, lab[t], return_iters=True)
            assert m.num_iter[j] == it_ref, (tag, j)
            assert np.array_equal(probs[j], u_ref, equal_nan=True), (tag, j)
        same_size = [t for t in sets if len(t) == len(sets[0])]
        if len(same_size) > 1:
            m = gl.ssl.laplace(W, reduce='exact')
            probs = m._fit_batch([(t, lab[t]) for t in same_size])
            if probs is not None:
                for j, t in enumerate(same_size):
                    assert np.array_equal(probs[j], orc.laplace_fit(Wo, t, lab[t]), equal_nan=True), (tag, j)
        ti = sets[0]
        # the use_cuda branch: float32 state, same T, iterates within the north star's 1e-5, same labels
        u_ref, T_ref = orc.poisson_gd(Wo, ti, lab[ti], return_T=True)
        m = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=True)
        u = m.fit(ti, lab[ti])
        assert u.dtype == np.float32 and m.num_iter == T_ref, tag
        assert np.max(np.abs(u - u_ref)) <= 1e-5 * max(1.0, np.max(np.abs(u_ref))), tag     # T sweeps of float32 rounding scale with |u|
        # f-3: random walk, reweighted Laplace, page rank
        u_ref, it_ref = orc.randomwalk_fit(Wo, ti, lab[ti], return_iters=True)
        m = gl.ssl.randomwalk(W, reduce='exact')
        u = m.fit(ti, lab[ti])
        assert m.num_iter == it_ref and np.array_equal(u, u_ref), tag
        for rw in ('poisson', 'wnll'):
            u_ref = orc.laplace_reweighted_fit(Wo, ti, lab[ti], rw)
            u = gl.ssl.laplace(W, reweighting=rw, reduce='exact').fit(ti, lab[ti])
            assert np.array_equal(u, u_ref, equal_nan=True), (tag, rw)
        G = gl.graph(W)
        pr_ref, it_ref = orc.page_rank(Wo, return_iters=True)
        pr = G.page_rank()
        assert G.page_rank_iters == it_ref and np.array_equal(pr, pr_ref), tag


@pytest.mark.parametrize('seed', _seeds(8))
def test_random_plaplace_jacobi_matches_the_oracle(gl, orc, seed):
    """graph.plaplace(fast=False) on random graphs, boundary sets, exponents and iteration caps: iterates and the
    stopping iteration equal to the C restatement of lp_iterate_main."""
    c = _case(200 + seed)
    rng = c['rng']
    W = gl.weightmatrix.knn(c['X'], c['k'], kernel=str(rng.choice(['gaussian', 'uniform'])))
    n = c['n']
    m = int(rng.integers(2, max(3, n // 20)))
    bdy = rng.choice(n, size=m, replace=False)
    val = rng.normal(size=m)
    p = float(rng.choice([2.5, 3.0, 6.0, 10.0, 40.0]))
    tol = float(rng.choice([1e-1, 1e-2, 1e-4]))
    T = int(rng.choice([1, 7, 150, 100000]))
    G = gl.graph(W)
    u = G.plaplace(bdy, val, p, tol=tol, max_num_it=T, fast=False)
    uo, it = orc.plaplace_jacobi(W, bdy, val, p, tol=tol, max_num_it=T, return_iters=True)
    assert G.plaplace_iters == it, (seed, n, p, tol, T)
    assert np.array_equal(u, uo, equal_nan=True), (seed, n, p, tol, T)      # (p < 3 can send the reference iteration to NaN: same NaNs)


def _virtual_ranks_sweep(gdist, _hip, prob, order, bounds, min_iter, max_iter, dtype=np.float64, gather=False):
    """Every rank's glx_dist_sweep object in ONE process, the packed boundary records moved between them by numpy (the
    all-to-all-v of the real transport): the stepwise protocol of graphlearning_amd.dist.run_stepwise."""
    P, C = prob['P'], prob['k']
    n = P.shape[0]
    world = len(bounds) - 1
    plans = [(gdist.GatherPlan if gather else gdist.RankPlan)(P, order, bounds, r) for r in range(world)]
    comms = [_hip.Comm(world, r, None) for r in range(world)]
    dss = [gdist.glx_dist_sweep(comms[r], plans[r], C, dtype=dtype) for r in range(world)]
    try:
        for r, ds in enumerate(dss):
            own = plans[r].own
            ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])

        def exchange(next_iterate):
            if plans[0].global_halo == 0:
                return
            if gather:       # the all-gather form: every rank's whole block (cap records) to every other rank, in rank order
                blocks = [ds.get_send() for ds in dss]
                for r, ds in enumerate(dss):
                    ds.put_halo(np.concatenate([b for q, b in enumerate(blocks) if q != r], axis=0), next_iterate)
                return
            sends = [ds.get_send() for ds in dss]
            offs = [np.concatenate([[0], np.cumsum(plans[s].send_counts)]) for s in range(world)]
            for r, ds in enumerate(dss):
                parts = [sends[s][offs[s][r]:offs[s][r + 1]] for s in range(world)]
                assert [len(p) for p in parts] == list(plans[r].recv_counts)
                ds.put_halo(np.concatenate(parts, axis=0) if plans[r].n_halo else np.zeros((0, ds.lay['ld']), dtype=dtype), next_iterate)

        for ds in dss:
            ds.begin()
        exchange(False)
        thresh = 1.0 / n
        err_T = float(np.max(np.abs(prob['deg'] * prob['w0'] - prob['vinf'])))
        T = 0
        while T < max_iter:
            if T >= min_iter and not (err_T > thresh):
                break
            want = (T + 1) >= min_iter
            for ds in dss:
                ds.boundary(want)
            exchange(True)
            errs = [ds.interior(want) for ds in dss]
            if want:
                err_T = max(errs)
            T += 1
        u = np.zeros((n, C), dtype=dtype)
        for r, ds in enumerate(dss):
            u[plans[r].own] = ds.fetch()
        halo = sum(p.n_halo for p in plans)
    finally:
        for ds in dss:
            ds.close()
        for cm in comms:
            cm.close()
    return u, T, halo


@pytest.mark.parametrize('seed', _seeds(14))
def test_random_vertex_partitions_match_the_oracle(gl, orc, seed):
    """The rank-local pieces of the sharded sweep (boundary rows | pack | halo | interior rows, per-rank maxima of the stop
    column) on random graphs -- symmetric and directed -- cut into 2..6 vertex blocks in three ways (the library's locality
    order with cuts in the gaps, the same order in equal blocks, the caller's order in equal blocks): iterates and T equal
    to the single-process reference for every partition."""
    from graphlearning_amd import dist as gdist, _hip
    c = _case(300 + seed)
    rng, lab, ti, k = c['rng'], c['lab'], c['ti'], c['k']
    Jo, Do = orc.knnsearch(c['X'], k + 1)
    sym = bool(rng.random() < 0.7)
    Wo = orc.knn_weights(Jo, Do.copy(), k, symmetrize=sym)
    min_iter, max_iter = [(50, 1000), (0, 80), (10, 10), (3, 300)][int(rng.integers(0, 4))]
    world = int(rng.integers(2, 7))
    how = ['cut', 'even', 'natural'][int(rng.integers(0, 3))]
    with np.errstate(all='ignore'):
        u_ref, T_ref = orc.poisson_gd(Wo, ti, lab[ti], min_iter=min_iter, max_iter=max_iter, return_T=True)
        prob = gdist.poisson_problem(Wo, ti, lab[ti])
        n = c['n']
        order = np.arange(n) if how == 'natural' else gdist.locality_order(prob['P'])
        bounds = gdist.cut_bounds(prob['P'], order, world) if how == 'cut' else gdist.block_bounds(n, world)
        u, T, halo = _virtual_ranks_sweep(gdist, _hip, prob, order, bounds, min_iter, max_iter)
        tag = 'seed %d: n=%d k=%d sym=%s world=%d %s iters=(%d,%d) halo rows %d' % (seed, n, k, sym, world, how, min_iter, max_iter, halo)
        assert T == T_ref, tag
        assert np.array_equal(u, u_ref, equal_nan=True), tag
        if seed % 2 == 0:      # the all-gather form of the exchange (GLX_DIST_FORM_GATHER): the same iterates
            ug, Tg, _ = _virtual_ranks_sweep(gdist, _hip, prob, order, bounds, min_iter, max_iter, gather=True)
            assert Tg == T_ref and np.array_equal(ug, u_ref, equal_nan=True), tag + ' (all-gather form)'
        u32, T32, _ = _virtual_ranks_sweep(gdist, _hip, prob, order, bounds, min_iter, max_iter, dtype=np.float32, gather=(seed % 4 == 1))
        assert T32 == T_ref and u32.dtype == np.float32, tag
        assert np.nanmax(np.abs(u32 - u_ref)) <= 1e-5 * max(1.0, np.nanmax(np.abs(u_ref))), tag


def _dump_knn_mismatch(seed, n, d, k, style, sim, J, D, J2, D2, Jo, Do, st1, st2, X, _hip):
    import json
    import os
    rows = np.flatnonzero(np.any(J != J2, axis=1) | np.any(D != D2, axis=1))
    scale = max(1.0, float(np.max(Do)))
    rec = {'seed': int(seed), 'pid': os.getpid(), 'n': n, 'd': d, 'k': k, 'style': style, 'sim': sim, 'nrows': int(len(rows)), 'rows': rows[:32].tolist(),
           'plain_equals_ckdtree': bool(np.array_equal(J, Jo) and np.max(np.abs(D - Do)) <= 1e-12 * scale),
           'ordered_equals_ckdtree': bool(np.array_equal(J2, Jo) and np.max(np.abs(D2 - Do)) <= 1e-12 * scale),
           'stats_plain': {a: (b if isinstance(b, str) else float(b)) for a, b in st1.items()},
           'stats_ordered': {a: (b if isinstance(b, str) else float(b)) for a, b in st2.items()}, 'detail': [], 'again': [],
           'ablate': os.environ.get('GLX_TEST_ABLATE', ''), 'launch_blocking': os.environ.get('HIP_LAUNCH_BLOCKING', ''), 'debug_counters': _hip.debug_counters()}
    for i in rows[:6]:
        rec['detail'].append({'row': int(i), 'J_plain': J[i].tolist(), 'J_ordered': J2[i].tolist(), 'J_ckdtree': Jo[i].tolist(),
                              'D_plain': D[i].tolist(), 'D_ordered': D2[i].tolist(), 'D_ckdtree': Do[i].tolist()})
    for rep in range(3):       # the same two searches again, at once, in the same process
        Ja, Da = _hip.knn_bruteforce(X, k, similarity=sim)
        Jb, Db = _hip.knn_bruteforce(X, k, similarity=sim, want_order=True)
        rec['again'].append({'plain_equals_ckdtree': bool(np.array_equal(Ja, Jo)), 'ordered_equals_ckdtree': bool(np.array_equal(Jb, Jo)),
                             'plain_equals_first_plain': bool(np.array_equal(Ja, J) and np.array_equal(Da, D)),
                             'ordered_equals_first_ordered': bool(np.array_equal(Jb, J2) and np.array_equal(Db, D2))})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(root, 'gpurun_out', 'knn_mismatch.jsonl'), 'a') as f:
        f.write(json.dumps(rec) + '\n')
        f.flush()
        os.fsync(f.fileno())


@pytest.mark.parametrize('seed', _seeds(30))
def test_random_knn_searches_match_ckdtree(gl, orc, seed):
    """The exact search over a wide range of shapes -- n = 2 .. 6000, d = 1 .. 300 (every feature-block count of the bf16
    filter and the blocked fp32 kernel beyond d = 128), k = 1 .. 60 (every list length), clustered / isotropic / offset /
    badly scaled data, euclidean and angular -- against cKDTree's lists."""
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([2, 3, 17, 64, 65, 129, 500, 1500, 3000, 6000]))
    d = int(rng.choice([1, 2, 3, 7, 16, 17, 32, 33, 50, 64, 65, 96, 97, 128, 129, 200, 300]))
    k = int(min(n, rng.choice([1, 2, 5, 11, 12, 13, 21, 28, 29, 40, 60])))
    style = int(rng.integers(0, 4))
    if style == 0:
        X = rng.normal(size=(n, d))
    elif style == 1:
        C = int(rng.integers(2, 12))
        X = rng.normal(size=(C, d))[rng.integers(0, C, size=n)] * 3.0 + rng.normal(size=(n, d))
    elif style == 2:
        X = rng.normal(size=(n, d)) + 50.0                                    # far from the origin: the filter must centre
    else:
        X = rng.normal(size=(n, d)) * np.exp(rng.normal(size=(1, d)) * 2.0)   # features of very different scale
    sim = 'angular' if rng.random() < 0.2 and d > 1 else 'euclidean'
    J, D = gl.weightmatrix.knnsearch(X, k, similarity=sim)
    # the search weightmatrix.knn runs (rows put into the order of chained cells first, where the size calls for it): the same lists
    from graphlearning_amd import _hip
    st1 = _hip.knn_stats()
    J2, D2 = _hip.knn_bruteforce(X, k, similarity=sim, want_order=True)
    st2 = _hip.knn_stats()
    Jo, Do = orc.knnsearch(X, k, similarity=sim)
    Jo, Do = Jo.reshape(n, -1), Do.reshape(n, -1)          # (cKDTree drops the axis for k = 1)
    if not (np.array_equal(J2, J) and np.array_equal(D2, D)):
        # (what the round-4 flake turned out to be: two searches of the same data that differ once in ~18 000 cases under 12 processes on
        # one GPU -- the evidence goes to gpurun_out/knn_mismatch.jsonl before the assertion: which rows, which of the two is cKDTree's)
        _dump_knn_mismatch(seed, n, d, k, style, sim, np.array(J), np.array(D), np.array(J2), np.array(D2), Jo, Do, st1, st2, X, _hip)
    assert np.array_equal(J2, J) and np.array_equal(D2, D), 'seed %d: the reordered search differs' % seed
    tag = 'seed %d: n=%d d=%d k=%d style=%d %s' % (seed, n, d, k, style, sim)
    assert J.dtype == np.int64 and D.dtype == np.float64 and J.shape == (n, k), tag
    scale = max(1.0, float(np.max(Do)))
    assert np.max(np.abs(D - Do)) <= 1e-12 * scale, tag
    if not np.array_equal(J, Jo):
        # the only admissible difference: two refs at the same distance (to the last bit) in the other order
        bad = np.flatnonzero(np.any(J != Jo, axis=1))
        for i in bad:
            assert np.array_equal(np.sort(J[i]), np.sort(Jo[i])) or np.abs(D[i, -1] - Do[i, -1]) <= 1e-12 * scale, (tag, i)
            cols = np.flatnonzero(J[i] != Jo[i])
            assert np.all(np.abs(Do[i, cols] - D[i, cols]) <= 1e-12 * scale), (tag, i)
        assert len(bad) <= max(1, n // 1000), (tag, len(bad))


@pytest.mark.parametrize('seed', _seeds(16))
def test_random_clustered_searches_match_all_pairs(gl, seed):
    """glx_knn_clustered / glx_knn_cells_range over random shapes -- n = 300 .. 40 000, d = 1 .. 128, k = 1 .. 60, 2 .. 300 cells,
    clustered / isotropic / offset / duplicated data, whole set or a query sub-range with the caller's cells -- against the all-pairs
    search: identical lists and distances, bit for bit."""
    from graphlearning_amd import _hip, dist_build
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([300, 1000, 4097, 12000, 40000]))
    d = int(rng.choice([1, 2, 3, 8, 16, 17, 20, 21, 32, 48, 64, 100, 128]))
    k = int(rng.choice([1, 2, 11, 12, 13, 28, 29, 60]))
    m = int(min(n // 4, rng.choice([2, 3, 16, 64, 300])))
    style = int(rng.integers(0, 4))
    if style == 0:
        X = rng.normal(size=(n, d))
    elif style == 1:
        C = int(rng.integers(2, 12))
        X = rng.normal(size=(C, d))[rng.integers(0, C, size=n)] * 5.0 + rng.normal(size=(n, d))
    elif style == 2:
        X = rng.normal(size=(n, d)) * 0.01 + 1000.0                           # far from the origin, tiny distances
    else:
        base = rng.normal(size=(max(n // 5, k + 1), d))
        X = base[rng.integers(0, len(base), size=n)]                          # every point several times: ties everywhere
    tag = 'seed %d: n=%d d=%d k=%d cells=%d style=%d' % (seed, n, d, k, m, style)
    J0, D0 = _hip.knn_bruteforce(X, k, clustered=0)
    J0, D0 = np.array(J0), np.array(D0)
    J1, D1 = _hip.knn_bruteforce(X, k, clustered=m)
    assert np.array_equal(J0, J1) and np.array_equal(D0, D1), tag
    if seed % 2 == 0:        # the caller's own cells and a query sub-range
        perm, starts = dist_build.coarse_locality_order(X, ncells=m, seed=seed, return_cells=True)
        Xp = np.ascontiguousarray(X[perm])
        q0, q1 = sorted(int(v) for v in rng.integers(0, n + 1, size=2))
        if q0 == q1:
            q0, q1 = 0, n
        Ja, Da = _hip.knn_bruteforce(Xp, k, query_range=(q0, q1))
        Ja, Da = np.array(Ja), np.array(Da)
        Jb, Db = _hip.knn_bruteforce(Xp, k, query_range=(q0, q1), cell_starts=starts)
        assert np.array_equal(Ja, Jb) and np.array_equal(Da, Db), tag + ' range %d:%d' % (q0, q1)
