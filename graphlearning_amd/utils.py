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


def symmetric_fingerprint(W):
    """What weightmatrix.knn stamps on a matrix it has just symmetrised (attribute `_glx_sym`), so that later consumers may
    skip forming W^T: addresses of the three CSR arrays plus two sums of the data.  A matrix whose structure was edited has
    new arrays, one whose values were edited in place has other sums; anything that does not match is treated as unknown."""
    addr = lambda a: a.__array_interface__['data'][0]
    return (addr(W.data), addr(W.indices), addr(W.indptr), int(W.nnz), float(W.data.sum()), float(W.data[::97].sum()))


def known_symmetric(W):
    tag = getattr(W, '_glx_sym', None)
    return tag is not None and tag == symmetric_fingerprint(W)
