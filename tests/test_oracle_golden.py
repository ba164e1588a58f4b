"""CPU: the oracle (oracle/gl_oracle.py) against the golden vectors captured from the
reference (tests/golden/make_golden.py).  This is what pins the oracle: bit-exact for
every integer output, CSR structure and fp64 iterate (the oracle calls the same scipy
kernels in the same order as the reference)."""
import json
import os
import numpy as np
import pytest
from conftest import csr_from, GOLDEN
from oracle import gl_oracle as orc


def test_g1_knn_and_weights(golden):
    g = golden('g1_twomoons.npz')
    X = g['X']
    J, D = orc.knnsearch(X, 11, method='kdtree')
    assert np.array_equal(J, g['knn_ind']) and np.array_equal(D, g['knn_dist'])
    for kernel in ['gaussian', 'uniform', 'symgaussian', 'distance', 'singular']:
        W = orc.knn_weights(g['knn_ind'], g['knn_dist'], 10, kernel=kernel)
        Wg = csr_from(g, 'W_' + kernel)
        assert np.array_equal(W.indptr, Wg.indptr), kernel
        assert np.array_equal(W.indices, Wg.indices), kernel
        assert np.array_equal(W.data, Wg.data), kernel
    W = orc.knn_weights(g['knn_ind'], g['knn_dist'], 10, symmetrize=False)
    Wg = csr_from(g, 'W_gaussian_nosym')
    assert np.array_equal(W.indices, Wg.indices) and np.array_equal(W.data, Wg.data)
    assert Wg.nnz == 5000 and csr_from(g, 'W_gaussian').nnz == 6144


def test_g1_trainset_and_helpers(golden):
    g = golden('g1_twomoons.npz')
    ti = orc.trainsets_generate(g['labels'], rate=5, seed=0)
    assert np.array_equal(ti, g['train_ind'])
    assert list(ti) == [268, 354, 290, 267, 216, 330, 213, 441, 476, 186]
    assert np.array_equal(orc.class_priors(g['labels']), g['class_priors'])
    acc = orc.ssl_accuracy(g['poisson_gd_pred'], g['labels'], g['train_ind'])
    assert acc == float(g['accuracy_poisson_gd'])


def test_g1_poisson(golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    ti, lab = g['train_ind'], g['labels']
    u, T = orc.poisson_gd(W, ti, lab[ti], return_T=True)
    assert T == int(g['poisson_gd_T']) == 409
    assert np.array_equal(u, g['poisson_gd_prob'])
    assert np.array_equal(orc.predict(u), g['poisson_gd_pred'])
    assert orc.poisson_gd_iterations(W, ti) == 409
    u, it = orc.poisson_cg(W, ti, lab[ti], return_iters=True)
    assert it == int(g['poisson_cg_iters'])
    assert np.array_equal(u, g['poisson_cg_prob'])
    assert np.array_equal(orc.predict(u), g['poisson_cg_pred'])
    Wd = csr_from(g, 'W_gaussian_nosym')
    u, T = orc.poisson_gd(Wd, ti, lab[ti], return_T=True)
    assert T == int(g['poisson_gd_directed_T']) and np.array_equal(u, g['poisson_gd_directed_prob'])


def test_g1_laplace(golden):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    ti, lab = g['train_ind'], g['labels']
    for norm in ['combinatorial', 'randomwalk', 'normalized']:
        u, it = orc.laplace_fit(W, ti, lab[ti], normalization=norm, return_iters=True)
        assert it == int(g['laplace_%s_iters' % norm])
        assert np.array_equal(u, g['laplace_%s_prob' % norm])
        assert np.array_equal(orc.predict(u), g['laplace_%s_pred' % norm])
    u = orc.laplace_fit(W, ti, lab[ti], tau=0.01, mean_shift=True)
    assert np.array_equal(u, g['laplace_tau_ms_prob'])


@pytest.mark.parametrize('solver', ['gradient_descent', 'conjugate_gradient'])
def test_g1_poisson_mbo(golden, solver):
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    ti, lab = g['train_ind'], g['labels']
    u, pred, w = orc.poisson_mbo_fit(W, ti, lab[ti], g['class_priors'], solver=solver)
    assert np.array_equal(u, g['poisson_mbo_%s_prob' % solver])
    assert np.array_equal(pred, g['poisson_mbo_%s_pred' % solver])
    assert np.array_equal(w, g['poisson_mbo_%s_weights' % solver])


def test_g2_knn(golden):
    g = golden('g2_knn.npz')
    for tag in ['d20', 'd64', 'd3']:
        J, D = orc.knnsearch(g['X_' + tag], 11, method='kdtree')
        assert np.array_equal(J, g['J_' + tag]) and np.array_equal(D, g['D_' + tag])
        # the direct-difference formula is the same quantity to rounding
        assert np.max(np.abs(orc.knn_exact_dist(g['X_' + tag], J) - D)) < 1e-12
    X = g['X_d20'][:400]
    J, D = orc.knnsearch(X, 11, method='brute')
    assert np.array_equal(J, g['Jb_brute400']) and np.array_equal(D, g['Db_brute400'])
    J, D = orc.knnsearch(X, 8, method='kdtree', similarity='angular')
    assert np.array_equal(J, g['J_angular400']) and np.array_equal(D, g['D_angular400'])


def test_g3_blobs5000(golden):
    g = golden('g3_blobs5000.npz')
    W = orc.knn_weights(g['knn_ind'], g['knn_dist'], 10)
    Wg = csr_from(g, 'W')
    assert np.array_equal(W.indices, Wg.indices) and np.array_equal(W.data, Wg.data)
    ti, lab = g['train_ind'], g['labels']
    assert np.array_equal(orc.trainsets_generate(lab, rate=2, seed=1), ti)
    u, T = orc.poisson_gd(Wg, ti, lab[ti], return_T=True)
    assert T == int(g['poisson_gd_T']) and np.array_equal(u, g['poisson_gd_prob'])
    u, it = orc.poisson_cg(Wg, ti, lab[ti], return_iters=True)
    assert it == int(g['poisson_cg_iters']) and np.array_equal(u, g['poisson_cg_prob'])
    u, it = orc.laplace_fit(Wg, ti, lab[ti], return_iters=True)
    assert it == int(g['laplace_iters']) and np.array_equal(u, g['laplace_prob'])
    u, pred, w = orc.poisson_mbo_fit(Wg, ti, lab[ti], g['class_priors'], solver='gradient_descent')
    assert np.array_equal(pred, g['poisson_mbo_pred']) and np.array_equal(u, g['poisson_mbo_prob'])
    assert np.array_equal(w, g['poisson_mbo_weights'])


def test_g5_projection(golden):
    g = golden('g5_projection.npz')
    assert np.array_equal(orc.predict(g['prob']), g['pred_plain'])
    lab, w, err, it = orc.volume_label_projection(g['prob'], g['priors'], 1)
    assert it == int(g['iters_1']) and it > 1
    assert np.array_equal(w, g['weights_1']) and np.array_equal(lab, g['labels_1']) and err == float(g['err_1'])
    lab2, w2, _, _ = orc.volume_label_projection(g['prob'], g['priors'], w)
    assert np.array_equal(w2, g['weights_2']) and np.array_equal(lab2, g['labels_2'])


def test_g6_helpers(golden):
    g = golden('g6_helpers.npz')
    labels = np.load(os.path.join(GOLDEN, 'MNIST_labels.npz'))['labels']
    assert labels.shape == (70000,)
    assert np.array_equal(orc.trainsets_generate(labels, rate=1, seed=0), g['gen_rate1_seed0'])
    assert np.array_equal(orc.trainsets_generate(labels, rate=3, seed=7), g['gen_rate3_seed7'])
    multi = orc.trainsets_generate(labels[:5000], rate=2, num_trials=3, seed=4)
    assert np.array_equal(np.stack(multi), g['gen_multi'])
    assert np.array_equal(orc.trainsets_generate(labels[:5000], rate=0.01, seed=9), g['gen_frac'])
    assert np.array_equal(orc.class_priors(labels), g['priors'])
    assert np.array_equal(orc.labels_to_onehot(np.array([2, 0, 1, 1]), 3), g['onehot_small'])
    # the published MNIST train sets (LabelPermutations/MNIST_permutations.npz, first 10) are valid index sets
    for i in range(10):
        p = g['mnist_perm_%d' % i]
        assert p.min() >= 0 and p.max() < 70000 and len(np.unique(p)) == len(p)


def test_g4_meta_is_consistent():
    meta = json.load(open(os.path.join(GOLDEN, 'g4_large_meta.json')))
    c2, c3 = meta['config2'], meta['config3']
    assert c2['n'] == 70000 and c2['nnz'] == 1136022 and c2['T'] == 50 and c2['row_nnz_max'] == 138
    assert c3['n'] == 60000 and c3['nnz'] == 1992536 and c3['row_nnz_max'] == 660


def test_g7_next_rows(golden):
    g = golden('g7_next_rows.npz')
    W = csr_from(g, 'W')
    ti, lab = g['train_ind'], g['labels']
    for method, norm in [('poisson', 'combinatorial'), ('poisson', 'normalized'), ('wnll', 'combinatorial')]:
        tag = method + '_' + norm
        Wr = orc.reweight(W, ti, method=method, normalization=norm)
        assert np.array_equal(Wr.tocsr().data, g['Wr_' + tag + '_data'])
        assert np.array_equal(orc.laplace_reweighted_fit(W, ti, lab[ti], method, norm), g['laplace_' + tag + '_prob'])
    u, it = orc.randomwalk_fit(W, ti, lab[ti], return_iters=True)
    assert it == int(g['randomwalk_iters']) and np.array_equal(u, g['randomwalk_prob'])
    assert np.array_equal(orc.conjgrad(csr_from(g, 'cg1d_A'), g['cg1d_rhs'], tol=1e-9), g['cg1d_x'])


def test_g8_pagerank(golden):
    """graph.page_rank (graph.py:1371-1412): the oracle reproduces the reference's vectors and
    iteration counts bit for bit, symmetric and directed graph, default and custom teleportation."""
    g = golden('g8_pagerank.npz')
    for tag in ('sym', 'dir'):
        W = csr_from(g, 'W_' + tag)
        u, it = orc.page_rank(W, return_iters=True)
        assert it == int(g['pr_' + tag + '_iters']) and np.array_equal(u, g['pr_' + tag])
        u, it = orc.page_rank(W, alpha=0.5, v=g['pr_' + tag + '_v'], tol=1e-8, return_iters=True)
        assert it == int(g['pr_' + tag + '_tele_iters']) and np.array_equal(u, g['pr_' + tag + '_tele'])


def test_g9_plaplace(golden):
    """graph.plaplace(fast=False): the C restatement of lp_iterate_main (oracle/csr_ref.c) reproduces
    the compiled reference's barriers bit for bit, including which iterate its swapped pointers leave
    in the caller's arrays (T = 57 odd, T = 200 even) and the stopping iteration."""
    g = golden('g9_plaplace.npz')
    W = csr_from(g, 'W')
    for tag in ('p10', 'p3', 'T57', 'T200'):
        p, tol, T, it_ref = g[tag + '_params']
        u, it, uu, ul = orc.plaplace_jacobi(W, g['bdy'], g['bdy_val'], p, tol=tol, max_num_it=T, return_iters=True, return_bounds=True)
        assert it == int(it_ref)
        assert np.array_equal(uu, g[tag + '_uu']) and np.array_equal(ul, g[tag + '_ul']) and np.array_equal(u, g[tag + '_u'])


def test_g11_properly(golden):
    from scipy import sparse
    """graph.reweight(method='properly') (graph.py:448-462) and ssl.laplace(reweighting='properly'): the oracle reproduces the reference's
    reweighted matrices (default parameters and a second set) and the fit, bit for bit."""
    g = golden('g11_properly.npz')
    for tag in ('blobs', 'moons'):
        X, lab, ti = g[tag + '_X'], g[tag + '_labels'], g[tag + '_train_ind']
        W = csr_from(g, tag + '_W')
        for ptag, kw in (('default', {}), ('p2', dict(alpha=3, zeta=1e5, r=0.5))):
            Wr = sparse.csr_matrix(orc.reweight(W, ti, method='properly', X=X, **kw))
            Wg = csr_from(g, tag + '_Wr_' + ptag)
            assert np.array_equal(Wr.indices, Wg.indices) and np.array_equal(Wr.data, Wg.data), (tag, ptag)
        assert np.array_equal(orc.laplace_reweighted_fit(W, ti, lab[ti], 'properly', X=X), g[tag + '_laplace_prob']), tag
