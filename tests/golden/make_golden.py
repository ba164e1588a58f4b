#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING THE REFERENCE.

Run in the build container only (the reference never travels to the GPU box):

    cd /root/repo && PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        MPLBACKEND=Agg python3 tests/golden/make_golden.py [--large]

Versions the vectors were captured with: Python 3.10.12, numpy 2.2.6,
scipy 1.15.3, scikit-learn 1.7.2, reference graphlearning 1.7.5.

Every file holds inputs + the reference's outputs (data only; no reference
source).  Iteration counts (T, CG iterations) are not returned by the
reference API; they are recorded from the oracle after asserting that the
oracle's iterates are bit-identical to the reference's for that case.
"""
import os
import sys
import json
import hashlib
import argparse
import numpy as np
from scipy import sparse
import sklearn.datasets as skd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import graphlearning as gl                      # the REFERENCE (PYTHONPATH=/root/reference)
from oracle import gl_oracle as orc             # our restatement, cross-checked below

assert gl.__file__.startswith('/root/reference'), gl.__file__


def csr_parts(W, prefix):
    W = sparse.csr_matrix(W)
    return {prefix + '_indptr': W.indptr.astype(np.int32), prefix + '_indices': W.indices.astype(np.int32),
            prefix + '_data': W.data.astype(np.float64)}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def blobs(n, d, C, seed, scale, labels=None):
    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(C, d)) * scale
    if labels is None:
        labels = rng.integers(0, C, size=n)
    X = centers[labels] + rng.normal(size=(n, d))
    return X, labels.astype(np.int64)


def g1_twomoons():
    X, labels = skd.make_moons(n_samples=500, noise=0.1, random_state=0)
    out = {'X': X, 'labels': labels.astype(np.int64)}
    J, D = gl.weightmatrix.knnsearch(X, 11, method='kdtree')
    out['knn_ind'], out['knn_dist'] = J.astype(np.int64), D
    for kernel in ['gaussian', 'uniform', 'symgaussian', 'distance', 'singular']:
        W = gl.weightmatrix.knn(X, 10, kernel=kernel, knn_data=(J.copy(), D.copy()))
        out.update(csr_parts(W, 'W_' + kernel))
    Wd = gl.weightmatrix.knn(X, 10, symmetrize=False, knn_data=(J.copy(), D.copy()))
    out.update(csr_parts(Wd, 'W_gaussian_nosym'))
    W = gl.weightmatrix.knn(X, 10, knn_data=(J.copy(), D.copy()))
    train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
    train_labels = labels[train_ind]
    out['train_ind'] = np.asarray(train_ind, dtype=np.int64)

    m = gl.ssl.poisson(W, solver='gradient_descent')
    out['poisson_gd_prob'] = m.fit(train_ind, train_labels)
    out['poisson_gd_pred'] = m.predict()
    u, T = orc.poisson_gd(W, train_ind, train_labels, return_T=True)
    assert np.array_equal(u, out['poisson_gd_prob'])
    out['poisson_gd_T'] = np.int64(T)

    m = gl.ssl.poisson(W)
    out['poisson_cg_prob'] = m.fit(train_ind, train_labels)
    out['poisson_cg_pred'] = m.predict()
    u, it = orc.poisson_cg(W, train_ind, train_labels, return_iters=True)
    assert np.array_equal(u, out['poisson_cg_prob'])
    out['poisson_cg_iters'] = np.int64(it)

    # directed (non-symmetrized) graph through the sweep: exercises P = D^-1 W^T
    m = gl.ssl.poisson(Wd, solver='gradient_descent')
    out['poisson_gd_directed_prob'] = m.fit(train_ind, train_labels)
    u, T = orc.poisson_gd(Wd, train_ind, train_labels, return_T=True)
    assert np.array_equal(u, out['poisson_gd_directed_prob'])
    out['poisson_gd_directed_T'] = np.int64(T)

    for norm in ['combinatorial', 'randomwalk', 'normalized']:
        m = gl.ssl.laplace(W, normalization=norm)
        out['laplace_%s_prob' % norm] = m.fit(train_ind, train_labels)
        out['laplace_%s_pred' % norm] = m.predict()
        u, it = orc.laplace_fit(W, train_ind, train_labels, normalization=norm, return_iters=True)
        assert np.array_equal(u, out['laplace_%s_prob' % norm])
        out['laplace_%s_iters' % norm] = np.int64(it)
    m = gl.ssl.laplace(W, tau=0.01, mean_shift=True)
    out['laplace_tau_ms_prob'] = m.fit(train_ind, train_labels)

    priors = gl.utils.class_priors(labels)
    out['class_priors'] = priors
    for solver in ['gradient_descent', 'conjugate_gradient']:
        m = gl.ssl.poisson_mbo(W, priors, solver=solver)
        pred = m.fit_predict(train_ind, train_labels)
        out['poisson_mbo_%s_prob' % solver] = m.prob
        out['poisson_mbo_%s_pred' % solver] = pred
        out['poisson_mbo_%s_weights' % solver] = np.asarray(m.weights, dtype=float)
    out['accuracy_poisson_gd'] = np.float64(gl.ssl.ssl_accuracy(out['poisson_gd_pred'], labels, train_ind))
    np.savez_compressed(os.path.join(HERE, 'g1_twomoons.npz'), **out)
    print('g1: T=%d cg=%d' % (out['poisson_gd_T'], out['poisson_cg_iters']))


def g2_knn():
    out = {}
    for tag, (n, d, seed) in {'d20': (2000, 20, 10), 'd64': (2000, 64, 11), 'd3': (1500, 3, 12)}.items():
        X, _ = blobs(n, d, 10, seed, 2.0)
        J, D = gl.weightmatrix.knnsearch(X, 11, method='kdtree')
        out['X_' + tag], out['J_' + tag], out['D_' + tag] = X, J.astype(np.int64), D
    X = out['X_d20'][:400]
    J, D = gl.weightmatrix.knnsearch(X, 11, method='brute')
    out['Jb_brute400'], out['Db_brute400'] = J.astype(np.int64), D
    J, D = gl.weightmatrix.knnsearch(X, 8, method='kdtree', similarity='angular')
    out['J_angular400'], out['D_angular400'] = J.astype(np.int64), D
    np.savez_compressed(os.path.join(HERE, 'g2_knn.npz'), **out)
    print('g2 done')


def g3_mid():
    n, d, C, k = 5000, 20, 10, 10
    X, labels = blobs(n, d, C, 3, 2.0)
    out = {'X': X, 'labels': labels}
    J, D = gl.weightmatrix.knnsearch(X, k + 1, method='kdtree')
    out['knn_ind'], out['knn_dist'] = J.astype(np.int64), D
    W = gl.weightmatrix.knn(X, k, knn_data=(J, D))
    out.update(csr_parts(W, 'W'))
    train_ind = gl.trainsets.generate(labels, rate=2, seed=1)
    train_labels = labels[train_ind]
    out['train_ind'] = np.asarray(train_ind, dtype=np.int64)
    m = gl.ssl.poisson(W, solver='gradient_descent')
    out['poisson_gd_prob'] = m.fit(train_ind, train_labels)
    out['poisson_gd_pred'] = m.predict()
    u, T = orc.poisson_gd(W, train_ind, train_labels, return_T=True)
    assert np.array_equal(u, out['poisson_gd_prob'])
    out['poisson_gd_T'] = np.int64(T)
    m = gl.ssl.poisson(W)
    out['poisson_cg_prob'] = m.fit(train_ind, train_labels)
    out['poisson_cg_pred'] = m.predict()
    u, it = orc.poisson_cg(W, train_ind, train_labels, return_iters=True)
    assert np.array_equal(u, out['poisson_cg_prob'])
    out['poisson_cg_iters'] = np.int64(it)
    m = gl.ssl.laplace(W)
    out['laplace_prob'] = m.fit(train_ind, train_labels)
    out['laplace_pred'] = m.predict()
    u, it = orc.laplace_fit(W, train_ind, train_labels, return_iters=True)
    assert np.array_equal(u, out['laplace_prob'])
    out['laplace_iters'] = np.int64(it)
    priors = gl.utils.class_priors(labels)
    m = gl.ssl.poisson_mbo(W, priors, solver='gradient_descent')
    out['poisson_mbo_pred'] = m.fit_predict(train_ind, train_labels)
    out['poisson_mbo_prob'] = m.prob
    out['poisson_mbo_weights'] = np.asarray(m.weights, dtype=float)
    out['class_priors'] = priors
    np.savez_compressed(os.path.join(HERE, 'g3_blobs5000.npz'), **out)
    print('g3: T=%d cg=%d lap=%d' % (out['poisson_gd_T'], out['poisson_cg_iters'], out['laplace_iters']))


def g5_projection():
    rng = np.random.default_rng(5)
    n, C = 2000, 5
    lab = rng.choice(C, size=n, p=[0.4, 0.25, 0.2, 0.1, 0.05])
    prob = rng.normal(size=(n, C)) * 0.6
    prob[np.arange(n), lab] += 1.0
    prob[:, 0] += 0.4                      # bias so the first call needs several steps
    priors = gl.utils.class_priors(lab)
    W = sparse.identity(n, format='csr')
    m = gl.ssl.poisson(W, class_priors=priors)
    m.prob = prob.copy()
    m.fitted = True
    out = {'prob': prob, 'priors': priors, 'pred_plain': m.predict(ignore_class_priors=True)}
    out['labels_1'] = m.volume_label_projection()
    out['weights_1'] = np.array(m.weights, dtype=float)
    out['err_1'] = np.float64(m.class_priors_error)
    out['labels_2'] = m.volume_label_projection()          # warm start from weights_1
    out['weights_2'] = np.array(m.weights, dtype=float)
    _, w, e, it = orc.volume_label_projection(prob, priors, 1)
    assert np.array_equal(w, out['weights_1'])
    out['iters_1'] = np.int64(it)
    np.savez_compressed(os.path.join(HERE, 'g5_projection.npz'), **out)
    print('g5: iters=%d err=%g' % (it, e))


def g6_helpers():
    labels = np.load('/root/reference/Data/MNIST_labels.npz')['labels']
    out = {'gen_rate1_seed0': np.asarray(gl.trainsets.generate(labels, rate=1, seed=0), dtype=np.int64),
           'gen_rate3_seed7': np.asarray(gl.trainsets.generate(labels, rate=3, seed=7), dtype=np.int64),
           'priors': gl.utils.class_priors(labels),
           'onehot_small': gl.utils.labels_to_onehot(np.array([2, 0, 1, 1]), 3)}
    multi = gl.trainsets.generate(labels[:5000], rate=2, num_trials=3, seed=4)
    out['gen_multi'] = np.stack([np.asarray(t, dtype=np.int64) for t in multi])
    out['gen_frac'] = np.asarray(gl.trainsets.generate(labels[:5000], rate=0.01, seed=9), dtype=np.int64)
    perm = np.load('/root/reference/LabelPermutations/MNIST_permutations.npz', allow_pickle=True)['perm']
    for i in range(10):
        out['mnist_perm_%d' % i] = np.asarray(perm[i], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'g6_helpers.npz'), **out)
    print('g6 done')


def g7_next_rows():
    """SURVEY 8f-3 rows on two-moons: laplace reweightings (graph.reweight) and ssl.randomwalk."""
    X, labels = skd.make_moons(n_samples=500, noise=0.1, random_state=0)
    W = gl.weightmatrix.knn(X, 10, knn_data=gl.weightmatrix.knnsearch(X, 11, method='kdtree'))
    train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
    tl = labels[train_ind]
    out = {'train_ind': np.asarray(train_ind, dtype=np.int64), 'labels': labels.astype(np.int64)}
    out.update(csr_parts(W, 'W'))
    G = gl.graph(W)
    for method, norm in [('poisson', 'combinatorial'), ('poisson', 'normalized'), ('wnll', 'combinatorial')]:
        Wr = G.reweight(train_ind, method=method, normalization=norm)
        tag = method + '_' + norm
        out.update(csr_parts(Wr, 'Wr_' + tag))
        Wo = orc.reweight(W, train_ind, method=method, normalization=norm)
        assert np.array_equal(sparse.csr_matrix(Wr).data, sparse.csr_matrix(Wo).data)
        mdl = gl.ssl.laplace(W, reweighting=method, normalization=norm)
        out['laplace_' + tag + '_prob'] = mdl.fit(train_ind, tl)
        out['laplace_' + tag + '_pred'] = mdl.predict()
        assert np.array_equal(orc.laplace_reweighted_fit(W, train_ind, tl, method, norm), out['laplace_' + tag + '_prob'])
    mdl = gl.ssl.randomwalk(W)
    out['randomwalk_prob'] = mdl.fit(train_ind, tl)
    out['randomwalk_pred'] = mdl.predict()
    u, it = orc.randomwalk_fit(W, train_ind, tl, return_iters=True)
    assert np.array_equal(u, out['randomwalk_prob'])
    out['randomwalk_iters'] = np.int64(it)
    # a 1-D conjgrad solve on its own (numpy's pairwise-summed reductions)
    L = G.laplacian() + sparse.identity(500) * 0.05
    f = np.sin(np.arange(500) * 0.1)
    out['cg1d_rhs'] = f
    out.update(csr_parts(L, 'cg1d_A'))
    out['cg1d_x'] = gl.utils.conjgrad(L, f, tol=1e-9)
    np.savez_compressed(os.path.join(HERE, 'g7_next_rows.npz'), **out)
    print('g7 done, randomwalk iters', it)


def g8_pagerank():
    """graph.page_rank (graph.py:1371-1412) on two-moons and on a directed (unsymmetrised) graph."""
    X, labels = skd.make_moons(n_samples=500, noise=0.1, random_state=0)
    knn_data = gl.weightmatrix.knnsearch(X, 11, method='kdtree')
    out = {}
    for tag, sym in (('sym', True), ('dir', False)):
        W = gl.weightmatrix.knn(X, 10, knn_data=knn_data, symmetrize=sym)
        out.update(csr_parts(W, 'W_' + tag))
        G = gl.graph(W)
        out['pr_' + tag] = G.page_rank()
        u, it = orc.page_rank(W, return_iters=True)
        assert np.array_equal(u, out['pr_' + tag])
        out['pr_' + tag + '_iters'] = np.int64(it)
        v = np.zeros(500)
        v[[3, 77, 400]] = 1 / 3
        out['pr_' + tag + '_v'] = v
        out['pr_' + tag + '_tele'] = G.page_rank(alpha=0.5, v=v, tol=1e-8)
        u, it = orc.page_rank(W, alpha=0.5, v=v, tol=1e-8, return_iters=True)
        assert np.array_equal(u, out['pr_' + tag + '_tele'])
        out['pr_' + tag + '_tele_iters'] = np.int64(it)
    np.savez_compressed(os.path.join(HERE, 'g8_pagerank.npz'), **out)
    print('g8 done, iterations', out['pr_sym_iters'], out['pr_dir_iters'], out['pr_sym_tele_iters'])


def g9_plaplace():
    """graph.plaplace(fast=False) (graph.py:1262-1278).  The reference's C extension is not
    built in /root/reference (read-only), so its lp_iterate_main is compiled as is by oracle/Makefile into
    oracle/_ref/liblp_ref.so and called here on the REFERENCE graph object's own (I, J, V) arrays with the
    reference's own set-up expressions; outputs = what graph.plaplace would return."""
    import ctypes
    ref = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'oracle', '_ref', 'liblp_ref.so'))
    f = getattr(ref, '_Z15lp_iterate_mainPdS_PiS0_S_S0_S_didbiii')      # lp_iterate_main(double*,double*,int*,int*,double*,int*,double*,double,int,double,bool,int,int,int)
    f.restype = None
    vp = ctypes.c_void_p
    rng = np.random.default_rng(5)
    X = rng.random((1500, 2))
    W = gl.weightmatrix.knn(X, 8, knn_data=gl.weightmatrix.knnsearch(X, 9, method='kdtree'))
    G = gl.graph(W)
    x, y = X[:, 0], X[:, 1]
    bdy = (x < 0.06) | (x > 0.94) | (y < 0.06) | (y > 0.94)
    g = x ** 2 - y ** 2
    out = {'X': X, 'bdy': bdy, 'bdy_val': g[bdy]}
    out.update(csr_parts(W, 'W'))
    n = G.num_nodes
    for tag, p, tol, T in [('p10', 10.0, 1e-1, 1e6), ('p3', 3.0, 1e-2, 1e6), ('T57', 2.5, 1e-9, 57), ('T200', 50.0, 1e-3, 200)]:
        bdy_set, bdy_val = gl.utils._boundary_handling(bdy, g[bdy])
        uu = np.max(bdy_val) * np.ones((n,))
        ul = np.min(bdy_val) * np.ones((n,))
        uu[bdy_set] = bdy_val
        ul[bdy_set] = bdy_val
        uu = np.ascontiguousarray(uu, dtype=np.float64)
        ul = np.ascontiguousarray(ul, dtype=np.float64)
        bdy_set = np.ascontiguousarray(bdy_set, dtype=np.int32)
        bdy_val = np.ascontiguousarray(bdy_val, dtype=np.float64)
        f(uu.ctypes.data_as(vp), ul.ctypes.data_as(vp), G.J.ctypes.data_as(vp), G.I.ctypes.data_as(vp), G.V.ctypes.data_as(vp),
          bdy_set.ctypes.data_as(vp), bdy_val.ctypes.data_as(vp), ctypes.c_double(p), ctypes.c_int(int(T)), ctypes.c_double(tol),
          ctypes.c_bool(False), ctypes.c_int(n), ctypes.c_int(len(G.V)), ctypes.c_int(len(bdy_set)))
        u, it, ouu, oul = orc.plaplace_jacobi(W, bdy, g[bdy], p, tol=tol, max_num_it=T, return_iters=True, return_bounds=True)
        assert np.array_equal(ouu, uu) and np.array_equal(oul, ul) and np.array_equal(u, (uu + ul) / 2)
        out[tag + '_u'] = (uu + ul) / 2
        out[tag + '_uu'] = uu
        out[tag + '_ul'] = ul
        out[tag + '_params'] = np.array([p, tol, T, it], dtype=np.float64)
        print('g9', tag, 'stopped at', it)
    np.savez_compressed(os.path.join(HERE, 'g9_plaplace.npz'), **out)
    print('g9 done')


def g10_knn_cache():
    """f-2: the kNN cache file the reference's knnsearch(dataset=...) writes (weightmatrix.py:414-427) and the weight
    matrix its string path `knn('name', k)` (weightmatrix.py:126-127, 431-467) builds from it."""
    import tempfile, shutil
    rng = np.random.default_rng(10)
    X = rng.normal(size=(300, 5))
    tmp = tempfile.mkdtemp()
    old = gl.weightmatrix.knn_dir
    gl.weightmatrix.knn_dir = os.path.join(tmp, 'knn_data')
    try:
        J, D = gl.weightmatrix.knnsearch(X, 8, method='kdtree', dataset='GlxToy', metric='Raw')
        path = os.path.join(gl.weightmatrix.knn_dir, 'glxtoy_raw.npz')
        assert os.path.exists(path)
        W = gl.weightmatrix.knn('glxtoy', 7)
        Wu = gl.weightmatrix.knn('GLXTOY', 5, kernel='uniform')      # fewer neighbours than the file holds
        os.makedirs(os.path.join(HERE, 'knn_data'), exist_ok=True)
        shutil.copy(path, os.path.join(HERE, 'knn_data', 'glxtoy_raw.npz'))
    finally:
        gl.weightmatrix.knn_dir = old
        shutil.rmtree(tmp)
    out = {'X': X, 'J': J.astype(np.int64), 'D': D}
    out.update(csr_parts(W, 'W_k7'))
    out.update(csr_parts(Wu, 'W_k5_uniform'))
    np.savez_compressed(os.path.join(HERE, 'g10_knn_cache.npz'), **out)
    print('g10 done', W.nnz, Wu.nnz)


def g11_properly():
    """graph.reweight(method='properly') (reference graph.py:448-462) and ssl.laplace(reweighting='properly'): blobs in eight dimensions and
    two-moons, the default parameters and a second set."""
    out = {}
    for tag, (X, labels) in (('blobs', blobs(3000, 8, 4, 11, 2.0)), ('moons', skd.make_moons(n_samples=500, noise=0.1, random_state=0))):
        X = np.ascontiguousarray(X, dtype=np.float64)
        labels = np.asarray(labels, dtype=np.int64)
        W = gl.weightmatrix.knn(X, 10, knn_data=gl.weightmatrix.knnsearch(X, 11, method='kdtree'))
        train_ind = gl.trainsets.generate(labels, rate=3, seed=1)
        out[tag + '_X'] = X
        out[tag + '_labels'] = labels
        out[tag + '_train_ind'] = np.asarray(train_ind, dtype=np.int64)
        out.update(csr_parts(W, tag + '_W'))
        G = gl.graph(W)
        for ptag, kw in (('default', {}), ('p2', dict(alpha=3, zeta=1e5, r=0.5))):
            Wr = sparse.csr_matrix(G.reweight(train_ind, method='properly', X=X, **kw))
            out.update(csr_parts(Wr, tag + '_Wr_' + ptag))
            Wo = sparse.csr_matrix(orc.reweight(W, train_ind, method='properly', X=X, **kw))
            assert np.array_equal(Wr.data, Wo.data) and np.array_equal(Wr.indices, Wo.indices)
        mdl = gl.ssl.laplace(W, X=X, reweighting='properly')
        out[tag + '_laplace_prob'] = mdl.fit(train_ind, labels[train_ind])
        out[tag + '_laplace_pred'] = mdl.predict()
        assert np.array_equal(orc.laplace_reweighted_fit(W, train_ind, labels[train_ind], 'properly', X=X), out[tag + '_laplace_prob'])
    np.savez_compressed(os.path.join(HERE, 'g11_properly.npz'), **out)
    print('g11 done')


def g4_large():
    """Config 2 (70k) and config 3 (60k): checksums only; the graphs are
    regenerated from seeds by the oracle on the GPU box."""
    meta = {}
    labels = np.load('/root/reference/Data/MNIST_labels.npz')['labels'].astype(np.int64)
    rng = np.random.default_rng(0)
    centers = rng.normal(size=(10, 20)) * 2.0
    X = centers[labels] + rng.normal(size=(70000, 20))
    J, D = gl.weightmatrix.knnsearch(X, 11, method='kdtree')
    W = gl.weightmatrix.knn(X, 10, knn_data=(J, D))
    train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
    m = gl.ssl.poisson(W, solver='gradient_descent')
    prob = m.fit(train_ind, labels[train_ind])
    u, T = orc.poisson_gd(W, train_ind, labels[train_ind], return_T=True)
    assert np.array_equal(u, prob)
    mc = gl.ssl.poisson(W)
    probc = mc.fit(train_ind, labels[train_ind])
    uc, itc = orc.poisson_cg(W, train_ind, labels[train_ind], return_iters=True)
    assert np.array_equal(uc, probc)
    meta['config2'] = dict(n=70000, d=20, k=10, nnz=int(W.nnz), row_nnz_max=int(np.diff(W.indptr).max()),
                           J_sha=sha(J.astype(np.int64)), D_sum=float(D.sum()),
                           W_indices_sha=sha(W.indices.astype(np.int32)), W_data_sum=float(W.data.sum()),
                           train_ind=[int(t) for t in train_ind], T=int(T), prob_abs_sum=float(np.abs(prob).sum()),
                           pred_sha=sha(m.predict().astype(np.int64)),
                           accuracy=float(gl.ssl.ssl_accuracy(m.predict(), labels, train_ind)),
                           cg_iters=int(itc), cg_prob_abs_sum=float(np.abs(probc).sum()),
                           cg_pred_sha=sha(mc.predict().astype(np.int64)))
    print('config2', meta['config2'])
    meta['config5'] = _config5(W, labels, train_ind)
    print('config5', meta['config5'])
    clabels = np.load('/root/reference/Data/cifar_labels.npz')['labels'].astype(np.int64)
    rng = np.random.default_rng(1)
    centers = rng.normal(size=(10, 32)) * 1.2
    X = centers[clabels] + rng.normal(size=(60000, 32))
    J, D = gl.weightmatrix.knnsearch(X, 21, method='kdtree')
    W = gl.weightmatrix.knn(X, 20, knn_data=(J, D))
    train_ind = gl.trainsets.generate(clabels, rate=10, seed=0)
    ml = gl.ssl.laplace(W)
    prob = ml.fit(train_ind, clabels[train_ind])
    ul, itl = orc.laplace_fit(W, train_ind, clabels[train_ind], return_iters=True)
    assert np.array_equal(ul, prob)
    meta['config3'] = dict(n=60000, d=32, k=20, nnz=int(W.nnz), row_nnz_max=int(np.diff(W.indptr).max()),
                           J_sha=sha(J.astype(np.int64)), W_indices_sha=sha(W.indices.astype(np.int32)),
                           W_data_sum=float(W.data.sum()), cg_iters=int(itl),
                           prob_abs_sum=float(np.abs(prob).sum()), pred_sha=sha(ml.predict().astype(np.int64)),
                           accuracy=float(gl.ssl.ssl_accuracy(ml.predict(), clabels, train_ind)))
    print('config3', meta['config3'])
    with open(os.path.join(HERE, 'g4_large_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)


def g4_samples():
    """Adds `u_samples` to the config-2 and config-3 entries of g4_large_meta.json: 1000 sampled entries (row, column, value)
    of the REFERENCE's iterates at full size -- Poisson gradient descent and Poisson CG at 70k, Laplace at 60k -- so that
    the GPU tests bound the end-to-end iterates element-wise (north star: within 1e-5), not only by checksums."""
    path = os.path.join(HERE, 'g4_large_meta.json')
    meta = json.load(open(path))
    pick = np.random.default_rng(12345)

    def sample(prob):
        rows = pick.integers(0, prob.shape[0], size=1000)
        cols = pick.integers(0, prob.shape[1], size=1000)
        return dict(rows=[int(r) for r in rows], cols=[int(c) for c in cols], values=[float(v) for v in prob[rows, cols]])
    labels = np.load('/root/reference/Data/MNIST_labels.npz')['labels'].astype(np.int64)
    rng = np.random.default_rng(0)
    centers = rng.normal(size=(10, 20)) * 2.0
    X = centers[labels] + rng.normal(size=(70000, 20))
    J, D = gl.weightmatrix.knnsearch(X, 11, method='kdtree')
    assert sha(J.astype(np.int64)) == meta['config2']['J_sha']
    W = gl.weightmatrix.knn(X, 10, knn_data=(J, D))
    train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
    prob = gl.ssl.poisson(W, solver='gradient_descent').fit(train_ind, labels[train_ind])
    assert abs(float(np.abs(prob).sum()) - meta['config2']['prob_abs_sum']) == 0.0
    meta['config2']['u_samples'] = sample(prob)
    probc = gl.ssl.poisson(W).fit(train_ind, labels[train_ind])
    assert float(np.abs(probc).sum()) == meta['config2']['cg_prob_abs_sum']
    meta['config2']['cg_u_samples'] = sample(probc)
    print('config2 samples recorded; max |u| of the sample %.3e' % np.max(np.abs(meta['config2']['u_samples']['values'])))
    clabels = np.load('/root/reference/Data/cifar_labels.npz')['labels'].astype(np.int64)
    rng = np.random.default_rng(1)
    centers = rng.normal(size=(10, 32)) * 1.2
    X = centers[clabels] + rng.normal(size=(60000, 32))
    J, D = gl.weightmatrix.knnsearch(X, 21, method='kdtree')
    assert sha(J.astype(np.int64)) == meta['config3']['J_sha']
    W = gl.weightmatrix.knn(X, 20, knn_data=(J, D))
    train_ind = gl.trainsets.generate(clabels, rate=10, seed=0)
    prob = gl.ssl.laplace(W).fit(train_ind, clabels[train_ind])
    assert float(np.abs(prob).sum()) == meta['config3']['prob_abs_sum']
    meta['config3']['u_samples'] = sample(prob)
    print('config3 samples recorded')
    with open(path, 'w') as f:
        json.dump(meta, f, indent=1)


def _config5(W, labels, train_ind):
    """Config 5: PoissonMBO on the config-2 graph (reference ssl.py:774-839), 851 SpMMs + 21 volume projections."""
    pri = gl.utils.class_priors(labels)
    mm = gl.ssl.poisson_mbo(W, pri, solver='gradient_descent', Ns=40, mu=1, T=20)
    prob = mm.fit(train_ind, labels[train_ind])
    pred = mm.predict()
    up, lp, wp = orc.poisson_mbo_fit(W, train_ind, labels[train_ind], pri, solver='gradient_descent')
    assert np.array_equal(up, prob) and np.array_equal(wp, mm.weights) and np.array_equal(lp, pred)
    return dict(pred_sha=sha(pred.astype(np.int64)), prob_sha=sha(prob.astype(np.float64)), prob_abs_sum=float(np.abs(prob).sum()),
                weights=[float(w) for w in mm.weights], class_priors_error=float(mm.class_priors_error),
                class_sizes=[int(c) for c in np.bincount(pred, minlength=10)],
                accuracy=float(gl.ssl.ssl_accuracy(pred, labels, train_ind)))


def g4_config5():
    """Adds / refreshes the config-5 entry of g4_large_meta.json without redoing configs 2 and 3."""
    path = os.path.join(HERE, 'g4_large_meta.json')
    meta = json.load(open(path))
    labels = np.load('/root/reference/Data/MNIST_labels.npz')['labels'].astype(np.int64)
    rng = np.random.default_rng(0)
    centers = rng.normal(size=(10, 20)) * 2.0
    X = centers[labels] + rng.normal(size=(70000, 20))
    J, D = gl.weightmatrix.knnsearch(X, 11, method='kdtree')
    assert sha(J.astype(np.int64)) == meta['config2']['J_sha']
    W = gl.weightmatrix.knn(X, 10, knn_data=(J, D))
    train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
    meta['config5'] = _config5(W, labels, train_ind)
    print('config5', meta['config5'])
    # the same pipeline on overlapping blobs (centers * 0.8): the volume projection has real work to do
    # (weights move away from 1, several hundred projection steps) -- the synthetic config-2 blobs are separable
    rng = np.random.default_rng(5)
    centers = rng.normal(size=(10, 20)) * 0.8
    X = centers[labels] + rng.normal(size=(70000, 20))
    J, D = gl.weightmatrix.knnsearch(X, 11, method='kdtree')
    W = gl.weightmatrix.knn(X, 10, knn_data=(J, D))
    train_ind = gl.trainsets.generate(labels, rate=2, seed=3)
    hard = _config5(W, labels, train_ind)
    hard.update(J_sha=sha(J.astype(np.int64)), W_indices_sha=sha(W.indices.astype(np.int32)), nnz=int(W.nnz),
                generator='default_rng(5): centers = normal((10,20))*0.8; X = centers[labels] + normal((70000,20)); '
                          'trainsets.generate(labels, rate=2, seed=3)')
    meta['config5_hard'] = hard
    print('config5_hard', hard)
    with open(path, 'w') as f:
        json.dump(meta, f, indent=1)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--large', action='store_true', help='also regenerate the 70k/60k checksum file (minutes)')
    ap.add_argument('--only', default=None, help='run a single generator, e.g. g8_pagerank')
    args = ap.parse_args()
    if args.only:
        globals()[args.only]()
        sys.exit(0)
    g1_twomoons()
    g2_knn()
    g3_mid()
    g5_projection()
    g6_helpers()
    g7_next_rows()
    g8_pagerank()
    g9_plaplace()
    g10_knn_cache()
    g11_properly()
    if args.large:
        g4_large()
