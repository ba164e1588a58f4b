"""CPU oracle for the kNN-graph + Poisson/Laplace label-propagation hot path.

TEST INFRASTRUCTURE ONLY.  This module is a numpy/scipy restatement of the
reference algorithm (jwcalder/GraphLearning v1.7.5) for the path named by
BASELINE.json:north_star.  It may be imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- as
the checker, never as the product.  The product (``graphlearning_amd``) runs
its hot loops in HIP and raises when the HIP library is missing.

Parity status: PINNED.  Every function below is checked bit-for-bit (integer
outputs, CSR structure, fp64 iterates produced by the same scipy kernels) or to
1e-12 against golden vectors captured by importing the reference in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py``).

The arithmetic of the reference's hot loops lives in un-vendored third-party
code: scipy 1.15.3 ``scipy.sparse._sparsetools.csr_matvecs`` / ``csc_matvec``
(sparse x dense products) and ``scipy.spatial.cKDTree.query`` (exact kNN).  The
oracle calls the same scipy entry points in the same order as the reference,
so its iterates are bit-identical to the reference's on the same inputs.
``oracle/csr_ref.c`` restates the two scipy product loops in plain C.

Reference citations are ``file:line`` relative to /root/reference/.
"""
import os
import numpy as np
from scipy import sparse, spatial


# ----------------------------------------------------------------------------
# bookkeeping helpers (graphlearning/utils.py, trainsets.py, ssl.py)
# ----------------------------------------------------------------------------
def labels_to_onehot(labels, k):
    """utils.labels_to_onehot, graphlearning/utils.py:536-572 (standardize=False).
    Width is max(max(labels)+1, k) (utils.py:558)."""
    labels = np.asarray(labels)
    n = labels.shape[0]
    width = max(int(np.max(labels)) + 1, k)
    out = np.zeros((n, width))
    out[np.arange(n), labels.astype(int)] = 1
    return out


def class_priors(labels):
    """utils.class_priors, graphlearning/utils.py:117-142 (negative labels ignored)."""
    labels = np.asarray(labels)
    classes = np.unique(labels)
    classes = classes[classes >= 0]
    total = np.sum(labels >= 0)
    return np.array([np.sum(labels == c) / total for c in classes], dtype=float)


def ssl_accuracy(pred_labels, true_labels, train_ind):
    """ssl.ssl_accuracy, graphlearning/ssl.py:1795-1834: percent correct over
    nodes outside train_ind whose true label is >= 0."""
    keep = np.ones(len(pred_labels), dtype=bool)
    keep[np.asarray(train_ind)] = False
    p = np.asarray(pred_labels)[keep]
    t = np.asarray(true_labels)[keep]
    known = t >= 0
    return 100 * np.mean(p[known] == t[known])


def trainsets_generate(labels, rate=1, num_trials=1, seed=None):
    """trainsets.generate, graphlearning/trainsets.py:47-156 (mask=None,
    dataset=None).  Seeds the *global* numpy RNG (trainsets.py:89-90) and draws
    per class, in class order, with np.random.choice(n, size, p, replace=False)
    (trainsets.py:121-128)."""
    labels = np.asarray(labels)
    if seed is not None:
        np.random.seed(seed)
    classes = np.unique(labels)
    per_class = np.bincount(labels)
    nc = len(classes)
    n = len(labels)
    if type(rate) == int:
        counts = (np.ones(nc)[None, :] * rate).astype(int)
    elif type(rate) == float:
        counts = (rate * per_class[None, :]).astype(int)
    elif type(rate) == np.ndarray:
        if rate.ndim != 2:
            raise ValueError('rate must be 2-dimensional')
        kind = rate.dtype
        if rate.shape[1] == 1:
            rate = rate @ np.ones((1, nc))
        if np.issubdtype(kind, np.integer):
            counts = rate.astype(int)
        elif np.issubdtype(kind, np.floating):
            counts = (rate * per_class).astype(int)
        else:
            raise ValueError('invalid rate dtype')
    else:
        raise ValueError('invalid rate type')
    sets = []
    for _ in range(num_trials):
        for row in range(counts.shape[0]):
            picked = []
            for j, c in enumerate(classes):
                p = (labels == c).astype(float)
                p = p / np.sum(p)
                picked += np.random.choice(n, size=counts[row, j], p=p, replace=False).tolist()
            sets.append(np.array(picked))
    return sets[0] if len(sets) == 1 else sets


# ----------------------------------------------------------------------------
# a-1  kNN search (graphlearning/weightmatrix.py:297-429)
# ----------------------------------------------------------------------------
def knnsearch(X, k, method='kdtree', similarity='euclidean'):
    """weightmatrix.knnsearch exact branches: 'kdtree' (weightmatrix.py:349-352,
    scipy cKDTree.query) and 'brute' (weightmatrix.py:354-361).  k counts the
    self point.  'angular' = euclidean on row-normalised data (:344-345).
    Returns (ind int64 (n,k), dist float64 (n,k)), rows ascending by distance."""
    if similarity not in ('euclidean', 'angular'):
        raise ValueError('Invalid choice of similarity ' + similarity)
    X = np.asarray(X)
    Y = X / np.linalg.norm(X, axis=1)[:, None] if similarity == 'angular' else X
    n = Y.shape[0]
    if method == 'kdtree':
        dist, ind = spatial.cKDTree(Y).query(Y, k=k)
        return ind, dist
    if method == 'brute':
        ind = np.zeros((n, k), dtype=int)
        dist = np.zeros((n, k))
        for i in range(n):
            d = np.linalg.norm(Y - Y[i, :], axis=1)
            ind[i, :] = np.argsort(d)[:k]
            dist[i, :] = d[ind[i, :]]
        return ind, dist
    raise ValueError('Invalid choice of knnsearch method ' + method)


def knn_exact_dist(X, ind):
    """fp64 direct-difference distances for given neighbour indices:
    sqrt(sum_d (x_i - x_j)^2), the quantity both exact branches above return."""
    X = np.asarray(X, dtype=np.float64)
    diff = X[:, None, :] - X[ind]
    return np.sqrt(np.sum(diff * diff, axis=2))


# ----------------------------------------------------------------------------
# a-2  kNN weight matrix (graphlearning/weightmatrix.py:68-187)
# ----------------------------------------------------------------------------
def sparse_max(A, B):
    """utils.sparse_max, graphlearning/utils.py:263-286."""
    nz = (A + B) > 0
    b_wins = B > A
    a_wins = nz - b_wins
    return A.multiply(a_wins) + B.multiply(b_wins)


def knn_weights(knn_ind, knn_dist, k, kernel='gaussian', symmetrize=True):
    """weightmatrix.knn given knn_data, graphlearning/weightmatrix.py:119-187.
    k excludes self (incremented at :119, clamped to available columns at :135)."""
    k = k + 1
    n = knn_ind.shape[0]
    k = int(np.minimum(knn_ind.shape[1], k))
    J = knn_ind[:, :k]
    Dk = knn_dist[:, :k]
    if kernel == 'uniform':
        w = np.ones_like(Dk)
    elif kernel == 'gaussian':
        sq = Dk * Dk
        eps = sq[:, k - 1]
        w = np.exp(-4 * sq / eps[:, None])
    elif kernel == 'symgaussian':
        eps = Dk[:, k - 1]
        w = np.exp(-4 * Dk * Dk / eps[:, None] / eps[J])
    elif kernel == 'distance':
        w = Dk
    elif kernel == 'singular':
        w = Dk.copy()          # the reference aliases knn_dist here (:153-156); copy keeps the caller's data
        w[Dk == 0] = 1
        w = 1 / w
    else:
        raise ValueError('Invalid choice of kernel: ' + kernel)
    rows = (np.ones((n, k)) * np.arange(n)[:, None]).flatten()   # float64 row ids, as at :171
    W = sparse.coo_matrix((w.flatten(), (rows, J.flatten())), shape=(n, n)).tocsr()
    if symmetrize:
        if kernel in ('distance', 'uniform', 'singular'):
            W = sparse_max(W, W.transpose())
        elif kernel == 'symgaussian':
            W = W + W.T.multiply(W.T > W) - W.multiply(W.T > W)
        else:
            W = (W + W.transpose()) / 2
    W.setdiag(0)
    W.eliminate_zeros()
    return W


def knn(X, k, kernel='gaussian', symmetrize=True, similarity='euclidean', method='kdtree'):
    """weightmatrix.knn end to end with an exact search (weightmatrix.py:68-187)."""
    ind, dist = knnsearch(X, k + 1, method=method, similarity=similarity)
    return knn_weights(ind, dist, k, kernel=kernel, symmetrize=symmetrize)


# ----------------------------------------------------------------------------
# a-8  graph calculus (graphlearning/graph.py:108-122, 210-233, 469-513)
# ----------------------------------------------------------------------------
def degree_vector(W):
    """graph.degree_vector, graphlearning/graph.py:108-122 (row sums W*1)."""
    return W * np.ones(W.shape[0])


def degree_matrix(W, p=1):
    """graph.degree_matrix, graphlearning/graph.py:210-233."""
    n = W.shape[0]
    return sparse.spdiags(degree_vector(W) ** p, 0, n, n).tocsr()


def laplacian(W, normalization='combinatorial'):
    """graph.laplacian, graphlearning/graph.py:469-513."""
    n = W.shape[0]
    I = sparse.identity(n)
    D = degree_matrix(W)
    if normalization == 'combinatorial':
        L = D - W
    elif normalization == 'randomwalk':
        L = I - degree_matrix(W, p=-1) * W
    elif normalization == 'normalized':
        Dh = degree_matrix(W, p=-0.5)
        L = I - Dh * W * Dh
    else:
        raise ValueError('Invalid option for graph Laplacian normalization.')
    return L.tocsr()


# ----------------------------------------------------------------------------
# utils.conjgrad (graphlearning/utils.py:483-532)
# ----------------------------------------------------------------------------
def conjgrad(A, b, x0=None, max_iter=1e5, tol=1e-10, return_iters=False):
    """Multi right-hand-side CG with per-column alpha/beta and the global stop
    sqrt(sum over ALL columns of ||r||^2) <= tol (utils.py:521,528)."""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    p = r.copy()
    rsold = np.sum(r ** 2, axis=0)
    err = 1
    it = 0
    while (err > tol) and (it < max_iter):
        it += 1
        Ap = A @ p
        alpha = rsold / np.sum(p * Ap, axis=0)
        x += alpha * p
        r -= alpha * Ap
        rsnew = np.sum(r ** 2, axis=0)
        err = np.sqrt(np.sum(rsnew))
        p = r + (rsnew / rsold) * p
        rsold = rsnew
    if return_iters:
        return x, it, err
    return x


# ----------------------------------------------------------------------------
# a-7  predict / volume-constrained projection (graphlearning/ssl.py:172-266)
# ----------------------------------------------------------------------------
def predict(prob, weights=1, similarity=True):
    """ssl.predict, graphlearning/ssl.py:230-266: global min/max normalisation,
    then argmax (first index wins ties) of scores*weights."""
    scores = prob - np.min(prob)
    scores = scores / np.max(scores)
    if similarity:
        return np.argmax(scores * weights, axis=1)
    return np.argmin(scores * weights, axis=1)


def volume_label_projection(prob, priors, weights=1, similarity=True):
    """ssl.volume_label_projection, graphlearning/ssl.py:172-209.
    Returns (labels, weights, err, iterations).  `weights` is the persistent
    per-object state (int 1 before the first call, ssl.py:149,191-192)."""
    k = prob.shape[1]
    if type(weights) == int:
        weights = np.ones((k,))
    else:
        weights = np.array(weights, dtype=float)
    dt = 0.1
    if similarity:
        dt *= -1
    it = 0
    err = 1
    while it < 1e4 and err > 1e-3:
        it += 1
        sizes = np.mean(labels_to_onehot(predict(prob, weights, similarity), k), axis=0)
        grad = sizes - priors
        err = np.max(np.absolute(grad))
        weights += dt * grad
        weights = weights / weights[0]
    return predict(prob, weights, similarity), weights, err, it


# ----------------------------------------------------------------------------
# a-3 / a-4  Poisson learning (graphlearning/ssl.py:608-693)
# ----------------------------------------------------------------------------
def poisson_source(n, train_ind, train_labels):
    """Source term b[train] = onehot - mean(onehot) (ssl.py:619-622)."""
    k = len(np.unique(train_labels))
    onehot = labels_to_onehot(train_labels, k)
    src = np.zeros((n, onehot.shape[1]))
    src[train_ind] = onehot - np.mean(onehot, axis=0)
    return src, k


def poisson_gd_setup(W, train_ind, train_labels):
    """Host-side setup of the gradient-descent solver, ssl.py:615-645.
    Returns dict(P, Db, v0, vinf, RW, deg, k).  P = D^-1 W^T (CSR), RW = W^T D^-1."""
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = W - sparse.spdiags(W.diagonal(), 0, n, n)
    W = sparse.csr_matrix(W)
    src, k = poisson_source(n, train_ind, train_labels)
    D = degree_matrix(W, p=-1)
    P = D * W.transpose()
    Db = D * src
    v = np.zeros(n)
    v[train_ind] = 1
    v = v / np.sum(v)
    deg = degree_vector(W)
    vinf = deg / np.sum(deg)
    RW = W.transpose() * D
    return dict(P=P, Db=Db, v0=v, vinf=vinf, RW=RW, deg=deg, k=k, W=W)


def poisson_gd(W, train_ind, train_labels, min_iter=50, max_iter=1000, return_T=False):
    """ssl.poisson._fit, solver='gradient_descent', CPU branch ssl.py:631-670."""
    s = poisson_gd_setup(W, train_ind, train_labels)
    n = W.shape[0]
    P, Db, v, vinf, RW = s['P'], s['Db'], s['v0'], s['vinf'], s['RW']
    u = np.zeros((n, s['k']))
    T = 0
    while (T < min_iter or np.max(np.absolute(v - vinf)) > 1 / n) and (T < max_iter):
        u = Db + P * u
        v = RW * v
        T += 1
    return (u, T) if return_T else u


def poisson_gd_iterations(W, train_ind, min_iter=50, max_iter=1000):
    """Number of sweeps T the stop test of ssl.py:667 yields; depends only on
    (W, train_ind)."""
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
    D = degree_matrix(W, p=-1)
    v = np.zeros(n)
    v[train_ind] = 1
    v = v / np.sum(v)
    deg = degree_vector(W)
    vinf = deg / np.sum(deg)
    RW = W.transpose() * D
    T = 0
    while (T < min_iter or np.max(np.absolute(v - vinf)) > 1 / n) and (T < max_iter):
        v = RW * v
        T += 1
    return T


def poisson_cg(W, train_ind, train_labels, tol=1e-3, return_iters=False):
    """ssl.poisson._fit, solver='conjugate_gradient' (default), ssl.py:624-629."""
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
    src, _ = poisson_source(n, train_ind, train_labels)
    L = laplacian(W, 'normalized')
    D = degree_matrix(W, p=-0.5)
    x, it, err = conjgrad(L, D * src, tol=tol, return_iters=True)
    u = D * x
    return (u, it) if return_iters else u


def poisson_fit(W, train_ind, train_labels, solver='conjugate_gradient', min_iter=50,
                max_iter=1000, tol=1e-3):
    if solver == 'conjugate_gradient':
        return poisson_cg(W, train_ind, train_labels, tol=tol)
    if solver == 'gradient_descent':
        return poisson_gd(W, train_ind, train_labels, min_iter=min_iter, max_iter=max_iter)
    raise ValueError('Invalid Poisson solver')


# ----------------------------------------------------------------------------
# a-5  Laplace learning (graphlearning/ssl.py:1206-1261), reweighting='none', order=1
# ----------------------------------------------------------------------------
def laplace_system(W, train_ind, train_labels, normalization='combinatorial', tau=0):
    """Dirichlet sub-system of ssl.py:1222-1246: returns (MAM, Mb, Mdiag, idx, F, k)."""
    W = sparse.csr_matrix(W)
    n = W.shape[0]
    k = len(np.unique(train_labels))
    tau_vec = np.ones(n) * tau if np.isscalar(tau) else np.asarray(tau, dtype=float)
    L = sparse.spdiags(tau_vec, 0, n, n) + laplacian(W, normalization)
    F = labels_to_onehot(train_labels, k)
    idx = np.full((n,), True, dtype=bool)
    idx[train_ind] = False
    b = -L[:, train_ind] * F
    b = b[idx, :]
    A = L[idx, :]
    A = A[:, idx]
    m = A.shape[0]
    Md = 1 / np.sqrt(A.diagonal() + 1e-10)
    M = sparse.spdiags(Md, 0, m, m).tocsr()
    return M * A * M, M * b, M, idx, F, k


def laplace_fit(W, train_ind, train_labels, normalization='combinatorial', tau=0,
                mean_shift=False, tol=1e-5, return_iters=False):
    """ssl.laplace._fit, ssl.py:1206-1261."""
    n = W.shape[0]
    MAM, Mb, M, idx, F, k = laplace_system(W, train_ind, train_labels, normalization, tau)
    v, it, err = conjgrad(MAM, Mb, tol=tol, return_iters=True)
    v = M * v
    u = np.zeros((n, k))
    u[idx, :] = v
    u[train_ind, :] = F
    if mean_shift:
        u -= np.mean(u, axis=0)
    return (u, it) if return_iters else u


# ----------------------------------------------------------------------------
# "next" rows: graph.reweight, laplace reweightings, ssl.randomwalk
# ----------------------------------------------------------------------------
def reweight(W, idx, method='poisson', normalization='combinatorial', X=None, alpha=2, zeta=1e7, r=0.1):
    """graph.reweight, graphlearning/graph.py:368-466, methods 'poisson' (:413-434), 'wnll'
    (:436-446) and 'properly' (:448-462).  Note the 1-D right-hand side of 'poisson': numpy reduces it with pairwise summation."""
    W = sparse.csr_matrix(W)
    n = W.shape[0]
    if method == 'poisson':
        f = np.zeros(n)
        f[idx] = 1
        if normalization == 'combinatorial':
            f -= np.mean(f)
            L = laplacian(W)
        elif normalization == 'normalized':
            d = degree_vector(W) ** (0.5)
            c = np.sum(d * f) / np.sum(d)
            f -= c
            L = laplacian(W, normalization)
        else:
            raise ValueError('Unsupported normalization ' + normalization + ' for graph.reweight.')
        w = conjgrad(L, f, tol=1e-5)
        w -= np.min(w)
        w += 1e-5
        D = sparse.spdiags(w, 0, n, n).tocsr()
        return D * W * D
    if method == 'wnll':
        m = len(idx)
        a = np.ones((n,))
        a[idx] = n / m
        D = sparse.spdiags(a, 0, n, n).tocsr()
        return D * W + W * D
    if method == 'properly':        # graph.py:448-462: gamma = 1 + (r / dist to the nearest labelled point)^alpha, distances floored at r_zeta
        from scipy import spatial
        rzeta = r / (zeta - 1) ** (1 / alpha)
        Xtree = spatial.cKDTree(X[idx, :])
        D, J = Xtree.query(X)
        D[D < rzeta] = rzeta
        gamma = 1 + (r / D) ** alpha
        D = sparse.spdiags(gamma, 0, n, n).tocsr()
        return D * W + W * D
    raise ValueError('Invalid reweighting method ' + method + '.')


def laplace_reweighted_fit(W, train_ind, train_labels, reweighting, normalization='combinatorial', tol=1e-5, X=None):
    """ssl.laplace._fit with reweighting != 'none', graphlearning/ssl.py:1208-1213."""
    Wr = reweight(W, train_ind, method=reweighting, normalization=normalization, X=X)
    return laplace_fit(Wr, train_ind, train_labels, normalization=normalization, tol=tol)


def randomwalk_fit(W, train_ind, train_labels, alpha=0.95, return_iters=False):
    """ssl.randomwalk._fit, graphlearning/ssl.py:1765-1793."""
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
    L = (1 - alpha) * sparse.identity(n) + alpha * laplacian(W, 'normalized')
    M = sparse.spdiags(1 / np.sqrt(L.diagonal() + 1e-10), 0, n, n).tocsr()
    k = len(np.unique(train_labels))
    Y = np.zeros((n, k))
    Y[train_ind, :] = labels_to_onehot(train_labels, k)
    u, it, err = conjgrad(M * L * M, M * Y, tol=1e-6, return_iters=True)
    u = M * u
    return (u, it) if return_iters else u


def page_rank(W, alpha=0.85, v=None, tol=1e-10, return_iters=False):
    """graph.page_rank, graphlearning/graph.py:1371-1412: power iteration
    u <- alpha*P@u + (1-alpha)*v with P = W^T D^-1, from u = 1/n, `while err > tol` on
    err = max|w - u|.  (`alpha*P@u` parses as (alpha*P)@u: the scaled matrix times u.)"""
    W = sparse.csr_matrix(W)
    n = W.shape[0]
    u = np.ones((n,)) / n
    if v is None:
        v = np.ones((n,)) / n
    D = degree_matrix(W, p=-1)
    P = W.T @ D
    err = tol + 1
    it = 0
    while err > tol:
        w = alpha * P @ u + (1 - alpha) * v
        err = np.max(np.absolute(w - u))
        u = w.copy()
        it += 1
    return (u, it) if return_iters else u


# ----------------------------------------------------------------------------
# f-4  p-Laplace, Jacobi variant (graph.plaplace with fast=False, graphlearning/graph.py:1177-1278)
# ----------------------------------------------------------------------------
def ccode_arrays(W):
    """graph.__ccode_init__, graphlearning/graph.py:69-84: the stored entries of W as
    (vertex, neighbour, weight) sorted by vertex with np.argsort's default (unstable) sort --
    the order inside a vertex's block is whatever that sort leaves, and it is the order in which
    the C code adds a vertex's terms."""
    I, J, V = sparse.find(sparse.csr_matrix(W))
    ind = np.argsort(I)
    I, J, V = I[ind], J[ind], V[ind]
    return (np.ascontiguousarray(I, dtype=np.int32), np.ascontiguousarray(J, dtype=np.int32),
            np.ascontiguousarray(V, dtype=np.float64))


def boundary_handling(bdy_set, bdy_val):
    """utils._boundary_handling, graphlearning/utils.py:144-174."""
    if type(bdy_set) == list:
        bdy_set = np.array(bdy_set)
    if bdy_set.dtype == bool:
        bdy_set = np.where(bdy_set)[0]
    m = len(bdy_set)
    if type(bdy_val) != np.ndarray:
        bdy_val = np.ones((m,)) * bdy_val
    return bdy_set, bdy_val


def _c_lib():
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, '_build', 'libcsr_ref.so')
    if not os.path.exists(path):
        subprocess.run(['make', '-s', '-C', here], check=True)
    return ctypes.CDLL(path)


def plaplace_jacobi(W, bdy_set, bdy_val, p, tol=1e-1, max_num_it=1e6, return_iters=False, return_bounds=False):
    """graph.plaplace(..., fast=False), graphlearning/graph.py:1262-1278: upper / lower barriers
    initialised to max / min of the boundary values, iterated by lp_iterate_main
    (c_code/lp_iterate.cpp:35-125, restated in oracle/csr_ref.c:ref_lp_iterate), u = (uu+ul)/2."""
    import ctypes
    n = W.shape[0]
    I, J, V = ccode_arrays(W)
    bdy_set, bdy_val = boundary_handling(bdy_set, bdy_val)
    uu = np.max(bdy_val) * np.ones((n,))
    ul = np.min(bdy_val) * np.ones((n,))
    uu[bdy_set] = bdy_val
    ul[bdy_set] = bdy_val
    uu = np.ascontiguousarray(uu, dtype=np.float64)
    ul = np.ascontiguousarray(ul, dtype=np.float64)
    bdy_set = np.ascontiguousarray(bdy_set, dtype=np.int32)
    bdy_val = np.ascontiguousarray(bdy_val, dtype=np.float64)
    lib = _c_lib()
    lib.ref_lp_iterate.restype = ctypes.c_int64
    vp = ctypes.c_void_p
    it = lib.ref_lp_iterate(uu.ctypes.data_as(vp), ul.ctypes.data_as(vp), J.ctypes.data_as(vp), I.ctypes.data_as(vp),
                            V.ctypes.data_as(vp), bdy_set.ctypes.data_as(vp), bdy_val.ctypes.data_as(vp), ctypes.c_double(p),
                            ctypes.c_int64(int(max_num_it)), ctypes.c_double(float(tol)), ctypes.c_int64(n),
                            ctypes.c_int64(len(V)), ctypes.c_int64(len(bdy_set)))
    u = (uu + ul) / 2
    out = (u,)
    if return_iters:
        out += (int(it),)
    if return_bounds:
        out += (uu, ul)
    return out if len(out) > 1 else u


# ----------------------------------------------------------------------------
# a-6  PoissonMBO (graphlearning/ssl.py:774-839)
# ----------------------------------------------------------------------------
def poisson_mbo_fit(W, train_ind, train_labels, priors, solver='conjugate_gradient',
                    min_iter=50, max_iter=1000, tol=1e-3, Ns=40, mu=1, T=20):
    """ssl.poisson_mbo._fit followed by the projection ssl.fit applies
    (ssl.py:478-479).  Returns (u, labels, weights): u is the one-hot matrix
    _fit returns; labels = predict() after fit; weights = final volume weights."""
    priors = np.asarray(priors, dtype=float)
    priors = priors / np.sum(priors)
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
    src, k = poisson_source(n, train_ind, train_labels)
    # initial labels: inner poisson model has no class priors -> plain argmax (ssl.py:797-798)
    u0 = poisson_fit(W, train_ind, train_labels, solver=solver, min_iter=min_iter,
                     max_iter=max_iter, tol=tol)
    labels = predict(u0, 1)
    u = labels_to_onehot(labels, k)
    dt = 1 / np.max(degree_vector(W))
    P = sparse.identity(n) - dt * laplacian(W)
    Db = mu * dt * src
    weights = 1
    for _ in range(T):
        for _ in range(Ns):
            u = P * u + Db
        labels, weights, _, _ = volume_label_projection(u, priors, weights)
        u = labels_to_onehot(labels, k)
    # ssl.fit: self.prob = u (one-hot), then one more projection with the warm weights
    labels, weights, err, _ = volume_label_projection(u, priors, weights)
    return u, labels, weights
