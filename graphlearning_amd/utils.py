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


def conjgrad(A, b, x0=None, max_iter=1e5, tol=1e-10, dtype=np.float64, return_info=False, device=None):
    """Multi right-hand-side conjugate gradient, reference utils.py:483-532, on the GPU
    (glx_cg_multi).  Per-column alpha/beta, global stop sqrt(sum_cols ||r||^2) <= tol."""
    if x0 is not None:
        # x0 != 0: solve for the correction (A d = b - A x0), the reference's r0 = b - A@x0
        b = np.asarray(b, dtype=np.float64)
        r0 = b - A @ x0
        d, it, err = conjgrad(A, r0, None, max_iter, tol, dtype, True, device)
        x = x0 + d
        return (x, it, err) if return_info else x
    G = _hip.DeviceGraph(sparse.csr_matrix(A), dtype=dtype, device=device, keep_order=True)
    try:
        x, it, err = G.cg(np.asarray(b), tol=tol, max_iter=int(max_iter))
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
