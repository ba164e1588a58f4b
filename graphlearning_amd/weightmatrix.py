"""kNN weight-matrix construction (reference graphlearning/weightmatrix.py: `knn` :68-187,
`knnsearch` :297-429, `load_knn_data` :431-467).  The search is an exact brute-force
tiled pairwise-distance kernel on the GPU (glx_knn_bruteforce); the kernel weights, the
sparse assembly and the symmetrisation run on the GPU too (glx_knn_search / glx_knn_result_to_csr)
and follow the reference operation by operation, so the returned scipy CSR matrix has the
identical structure and values (the exponential of the Gaussian kernels is correctly rounded
where the reference's is its host's libm; see `knn`)."""
import os
import sys
import numpy as np
from scipy import sparse
from . import utils
from . import _hip

knn_dir = os.path.abspath(os.path.join(os.getcwd(), 'knn_data'))


def knn(data, k, kernel='gaussian', eta=None, symmetrize=True, metric='raw', similarity='euclidean', knn_data=None,
        device=None):
    """kNN weight matrix, same signature and result as reference weightmatrix.py:68-187: a scipy CSR (n,n) float64 matrix,
    symmetric (unless symmetrize=False), zero diagonal, canonical format.

    Searched, weighted, assembled and symmetrised on the device; the lists of the search never visit the host.  The one
    operation on this path whose bits the reference leaves to its host is the exponential of the Gaussian kernels
    (weightmatrix.py:144-150, `np.exp`: glibc's on some hosts, numpy's own SIMD kernel on others).  Here it is the correctly
    rounded exp of csrc/exp_cr.h -- a machine-independent W that agrees with any host's numpy to that host's libm error (about 5 %
    of the weights differ by one ulp from the build container's numpy; tests/test_gpu_weights.py counts them and what they do
    downstream: nothing to the labels or the iteration counts of configs 2 and 3).  GLX_HOST_EXP=1 evaluates numpy's exp on the
    host instead: the W of THIS host's reference, bit for bit (what the golden-vector suite compares against)."""
    # validate before anything is launched (a bad kernel name must not leave work behind on the device)
    if similarity not in ['angular', 'euclidean']:
        sys.exit('Invalid choice of similarity ' + similarity)
    if eta is None and kernel not in ['uniform', 'gaussian', 'symgaussian', 'distance', 'singular']:
        sys.exit('Invalid choice of kernel: ' + kernel)
    k += 1                                   # self is counted in knn data (reference :119)
    # symmetrisation rule (reference :177-183)
    if not symmetrize:
        sym = 0
    elif kernel in ['distance', 'uniform', 'singular']:
        sym = 2
    elif kernel == 'symgaussian':
        sym = 3
    else:
        sym = 1
    host_exp = os.environ.get('GLX_HOST_EXP') == '1' and eta is None and kernel in ('gaussian', 'symgaussian')
    res, order = None, None
    try:
        if knn_data is not None:
            knn_ind, knn_dist = knn_data
        elif type(data) is str:
            knn_ind, knn_dist = load_knn_data(data, metric=metric)
        else:
            # (the result arrays of the assembly are page-locked beside the search when the pool has none of their size yet)
            n_rows = int(np.shape(data)[0])
            cap = n_rows * int(k) * (2 if sym else 1)
            reserve = _hip.pinned_reserve([((n_rows + 1,), np.int32), ((cap,), np.int32), ((cap,), np.float64)]) if cap * 12 <= _hip._PINNED_CSR_MAX else None
            try:
                res = _hip.KnnResult(data, int(k), similarity=similarity, device=device, want_order=True)
            finally:
                if reserve is not None:
                    reserve.join()
            order = res.order()
            knn_ind, knn_dist = (res.lists() if (host_exp or eta is not None) else (None, None))
        if res is not None:
            n, kk = res.n, res.k
        else:
            n, kk = np.shape(knn_dist)
        k = int(np.minimum(kk, k))           # clamp to the columns available (reference :135)
        if eta is not None:
            # user kernel: a Python callable, evaluated on the host; assembly on the device.
            # (the reference divides (n,k) by (n,) here (weightmatrix.py:164), which cannot broadcast;
            # the documented formula eta(|x_i-x_j|^2 / d_k(x_i)^2) is what is computed)
            D = knn_dist[:, :k] * knn_dist[:, :k]
            eps = D[:, k - 1]
            W = _assemble(res, knn_ind, knn_dist, k, 'given', sym, eta(D / eps[:, None]), device)
        elif host_exp:
            W = _assemble(res, knn_ind, knn_dist, k, 'given', sym, _host_gaussian(knn_ind, knn_dist, k, kernel), device)
        else:
            W = _assemble(res, knn_ind, knn_dist, k, kernel, sym, None, device)
    finally:
        if res is not None:
            res.close()
    if order is not None and os.environ.get('GLX_KNN_ORDER', '1') != '0':
        # the search left the order of (chained) cells of feature space behind: a locality order of the vertices for the operators
        # on this graph -- the learners hand it to the device operator instead of the library's pass over the graph (3.7 ms at
        # 70 000 vertices, 0.19 s at 10^6)
        W._glx_order = order
    if symmetrize and kernel != 'symgaussian':
        # symmetric bit for bit ((a+b)/2 = (b+a)/2, and the element-wise max) with an empty diagonal: stamped so that ssl.poisson
        # can write down D^-1 W^T without transposing (utils.known_symmetric re-checks the stamp).  NOT the symgaussian rule:
        # W + W^T*(W^T>W) - W*(W^T>W) leaves fl(fl(a+b)-a) on one side of an edge and b on the other
        W._glx_sym = utils.symmetric_fingerprint(W)
    return W


def _assemble(res, knn_ind, knn_dist, k, kernel, sym, weights, device):
    if res is not None:
        return res.to_csr(k, kernel=kernel, sym=sym, weights=weights)
    return _hip.knn_to_csr(knn_ind, knn_dist, k, kernel=kernel, sym=sym, weights=weights, device=device)


def _host_gaussian(knn_ind, knn_dist, k, kernel):
    """The Gaussian weights with numpy's exp on the host, elementwise over (n,k): the reference's own expressions
    (weightmatrix.py:144-150)."""
    d = np.asarray(knn_dist)[:, :k]
    n = d.shape[0]
    weights = _hip.pinned_empty((n, k), np.float64)          # page-locked: the upload to the assembly runs at PCIe speed
    J = np.asarray(knn_ind)[:, :k] if kernel == 'symgaussian' else None
    eps_all = d[:, k - 1] if kernel == 'symgaussian' else None

    def rows(lo, hi):           # elementwise: any split into row blocks gives the same bits
        if kernel == 'gaussian':
            D = d[lo:hi] * d[lo:hi]
            eps = D[:, k - 1]
            np.exp(-4 * D / eps[:, None], out=weights[lo:hi])
        else:
            np.exp(-4 * d[lo:hi] * d[lo:hi] / eps_all[lo:hi, None] / eps_all[J[lo:hi]], out=weights[lo:hi])
    _row_blocks(rows, n)
    return weights


def _row_blocks(fn, n, min_rows=16384):
    """fn(lo, hi) over row blocks on a few host threads (numpy releases the GIL inside its loops)."""
    nt = int(min(8, os.cpu_count() or 1, n // min_rows))
    if nt <= 1 or os.environ.get('GLX_HOST_THREADS') == '1':
        fn(0, n)
        return
    from concurrent.futures import ThreadPoolExecutor
    cuts = [n * t // nt for t in range(nt + 1)]
    with ThreadPoolExecutor(max_workers=nt) as pool:      # per call: an executor kept across a fork() would hang its child
        for f in [pool.submit(fn, cuts[t], cuts[t + 1]) for t in range(nt)]:
            f.result()


def knnsearch(X, k, method=None, similarity='euclidean', dataset=None, metric='raw', device=None):
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
