"""kNN weight-matrix construction (reference graphlearning/weightmatrix.py: `knn` :68-187,
`knnsearch` :297-429, `load_knn_data` :431-467).  The search is an exact brute-force
tiled pairwise-distance kernel on the GPU (glx_knn_bruteforce); the kernel weights and
the sparse symmetrisation follow the reference operation by operation so the returned
scipy CSR matrix has the identical structure."""
import os
import sys
import numpy as np
from scipy import sparse
from . import utils
from . import _hip

knn_dir = os.path.abspath(os.path.join(os.getcwd(), 'knn_data'))


def knn(data, k, kernel='gaussian', eta=None, symmetrize=True, metric='raw', similarity='euclidean', knn_data=None):
    """kNN weight matrix, same signature and result as reference weightmatrix.py:68-187.
    Returns a scipy CSR (n,n) float64 matrix: symmetric (unless symmetrize=False), zero
    diagonal, canonical format."""
    k += 1                                   # self is counted in knn data (reference :119)
    if knn_data is not None:
        knn_ind, knn_dist = knn_data
    elif type(data) is str:
        knn_ind, knn_dist = load_knn_data(data, metric=metric)
    else:
        knn_ind, knn_dist = knnsearch(data, k, similarity=similarity)
    n = knn_ind.shape[0]
    k = np.minimum(knn_ind.shape[1], k)      # clamp to the columns available (reference :135)
    knn_ind = knn_ind[:, :k]
    knn_dist = knn_dist[:, :k]
    if eta is None:
        if kernel == 'uniform':
            weights = np.ones_like(knn_dist)
        elif kernel == 'gaussian':
            D = knn_dist * knn_dist
            eps = D[:, k - 1]
            weights = np.exp(-4 * D / eps[:, None])
        elif kernel == 'symgaussian':
            eps = knn_dist[:, k - 1]
            weights = np.exp(-4 * knn_dist * knn_dist / eps[:, None] / eps[knn_ind])
        elif kernel == 'distance':
            weights = knn_dist
        elif kernel == 'singular':
            weights = np.array(knn_dist, dtype=float)
            weights[knn_dist == 0] = 1
            weights = 1 / weights
        else:
            sys.exit('Invalid choice of kernel: ' + kernel)
    else:
        D = knn_dist * knn_dist
        eps = D[:, k - 1]
        # the reference divides (n,k) by (n,) here (weightmatrix.py:164), which cannot broadcast;
        # the documented formula eta(|x_i-x_j|^2 / d_k(x_i)^2) is what is computed
        weights = eta(D / eps[:, None])
    knn_ind = knn_ind.flatten()
    weights = weights.flatten()
    self_ind = (np.ones((n, k)) * np.arange(n)[:, None]).flatten()
    W = sparse.coo_matrix((weights, (self_ind, knn_ind)), shape=(n, n)).tocsr()   # duplicates are summed
    if symmetrize:
        if kernel in ['distance', 'uniform', 'singular']:
            W = utils.sparse_max(W, W.transpose())
        elif kernel == 'symgaussian':
            W = W + W.T.multiply(W.T > W) - W.multiply(W.T > W)
        else:
            W = (W + W.transpose()) / 2
    W.setdiag(0)
    W.eliminate_zeros()
    return W


def knnsearch(X, k, method=None, similarity='euclidean', dataset=None, metric='raw', device=0):
    """k nearest neighbours including the self point (reference weightmatrix.py:297-429).
    Every `method` the reference knows ('kdtree', 'brute', 'annoy', None) is served by the
    exact GPU search ('hip'); 'annoy' is approximate in the reference, exact here.
    Returns (knn_ind int64 (n,k), knn_dist float64 (n,k)), rows ascending by distance."""
    if method is None:
        method = 'hip'
    if method not in ['hip', 'kdtree', 'brute', 'annoy']:
        sys.exit('Invalid choice of knnsearch method ' + method)
    if similarity not in ['angular', 'euclidean']:
        sys.exit('Invalid choice of similarity ' + similarity)
    X = np.asarray(X, dtype=np.float64)
    knn_ind, knn_dist = _hip.knn_bruteforce(X, int(k), similarity=similarity, device=device)
    if dataset is not None:                  # npz cache 'J','D' (reference :416-427)
        path = os.path.join(knn_dir, dataset.lower() + '_' + metric.lower() + '.npz')
        if not os.path.exists(knn_dir):
            os.makedirs(knn_dir)
        np.savez_compressed(path, J=knn_ind, D=knn_dist)
    return knn_ind, knn_dist


def load_knn_data(dataset, metric='raw'):
    """Load cached kNN data 'J','D' from ./knn_data (reference weightmatrix.py:431-467).
    There is no download here: a missing file is an error."""
    path = os.path.join(knn_dir, dataset.lower() + '_' + metric.lower() + '.npz')
    if not os.path.exists(path):
        sys.exit('Error: kNN data file ' + path + ' not found (no download in this build).')
    f = np.load(path, allow_pickle=True)
    return f['J'], f['D']
