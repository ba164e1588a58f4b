"""Helpers on the hot path (reference graphlearning/utils.py), restated; `conjgrad` runs on the GPU."""
import sys
import numpy as np
from scipy import sparse
from . import _hip


def labels_to_onehot(labels, k, standardize=False):
    """One-hot encoding, reference utils.py:536-572.  Width = max(max(labels)+1, k)."""
    labels = np.asarray(labels)
    n = labels.shape[0]
    k = max(int(np.max(labels)) + 1, k)
    if standardize:
        uniq = np.unique(labels)
        k = len(uniq)
        labels = np.searchsorted(uniq, labels)
    labels = labels.astype(int)
    onehot = np.zeros((n, k))
    onehot[np.arange(n), labels] = 1
    return onehot


def class_priors(labels):
    """Fraction of data in each class, negative labels ignored (reference utils.py:117-142)."""
    labels = np.asarray(labels)
    classes = np.unique(labels)
    classes = classes[classes >= 0]
    total = np.sum(labels >= 0)
    priors = np.zeros((len(classes),))
    for i, c in enumerate(classes):
        priors[i] = np.sum(labels == c) / total
    return priors


def sparse_max(A, B):
    """Element-wise max of two non-negative square sparse matrices (reference utils.py:263-286)."""
    nz = (A + B) > 0
    b_wins = B > A
    a_wins = nz - b_wins
    return A.multiply(a_wins) + B.multiply(b_wins)


def conjgrad(A, b, x0=None, max_iter=1e5, tol=1e-10, dtype=np.float64, return_info=False, device=None, reduce='exact'):
    """Multi right-hand-side conjugate gradient, reference utils.py:483-532, on the GPU
    (glx_cg_solve).  Per-column alpha/beta, global stop sqrt(sum_cols ||r||^2) <= tol.
    With x0 the iteration starts like the reference's: `x = x0.copy(); r = b - A@x` (utils.py:510-514,
    the residual formed on the host with the reference's own expression) and x accumulates from x0.
    reduce='tree' selects the tolerance mode of the reductions (include/glx.h GLX_CG_TREE)."""
    G = _hip.DeviceGraph(sparse.csr_matrix(A), dtype=dtype, device=device, keep_order=True)
    try:
        if x0 is None:
            x, it, err = G.cg(np.asarray(b), tol=tol, max_iter=int(max_iter), reduce=reduce)
        else:
            x0 = np.asarray(x0)
            r0 = b - A @ x0
            x, it, err = G.cg(r0, tol=tol, max_iter=int(max_iter), x0=x0, reduce=reduce)
    finally:
        G.close()
    return (x, it, err) if return_info else x


def _boundary_handling(bdy_set, bdy_val):
    """Boundary data in standard form: index array + value array (reference utils.py:144-174)."""
    if type(bdy_set) == list:
        bdy_set = np.array(bdy_set)
    if bdy_set.dtype == bool:
        bdy_set = np.where(bdy_set)[0]
    m = len(bdy_set)
    if type(bdy_val) != np.ndarray:
        bdy_val = np.ones((m,)) * bdy_val
    return bdy_set, bdy_val


def matrix_fingerprint(W):
    """Content fingerprint of a scipy sparse matrix: shape, nnz and a 128-bit hash of its three arrays (the library's threaded
    host hash: ~0.2 ms for the 25 MB of the config-3 graph; hashlib's blake2b when the library is not built).  The
    device-resident operators of the learners are keyed by it, so a matrix whose values were edited IN PLACE between two fits
    is seen as a new graph -- the reference rebuilds its operator on every fit (ssl.py:615-644), an `id(W)` key would silently
    reuse the stale one."""
    arrays = [np.ascontiguousarray(getattr(W, name)) for name in ('data', 'indices', 'indptr') if hasattr(W, name)]
    if not arrays:                                          # a dense matrix or another sparse format
        arrays = [np.ascontiguousarray(sparse.csr_matrix(W).data)]
    digest = _hip.host_fingerprint(arrays)
    if digest is None:                                      # slower, same role
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        for a in arrays:
            h.update(memoryview(a).cast('B'))
        digest = int.from_bytes(h.digest(), 'little')
    return (tuple(W.shape), int(getattr(W, 'nnz', arrays[0].size)), str(arrays[0].dtype), digest)


def symmetric_fingerprint(W):
    """What weightmatrix.knn stamps on a matrix it has just symmetrised (attribute `_glx_sym`), so that later consumers may
    skip forming W^T: the content fingerprint.  Any edit -- of the structure or of a single value in place -- gives another
    fingerprint, and the matrix is then treated as unknown (the general, transposing operator build)."""
    return matrix_fingerprint(W)


def known_symmetric(W, fingerprint=None):
    """Is W a matrix weightmatrix.knn symmetrised, unchanged since?  `fingerprint`: matrix_fingerprint(W) if the caller has it."""
    tag = getattr(W, '_glx_sym', None)
    if tag is None:
        return False
    return tag == (matrix_fingerprint(W) if fingerprint is None else fingerprint)
