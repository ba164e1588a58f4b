"""CPU: the plain-C restatement of scipy's csr_matvecs / csc_matvec (oracle/csr_ref.c) is
bit-identical to scipy on random operators, and the C sweep loop reproduces the golden iterates."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
from scipy import sparse
from conftest import ROOT, csr_from


@pytest.fixture(scope='module')
def lib():
    subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'oracle')], check=True)
    return C.CDLL(os.path.join(ROOT, 'oracle', '_build', 'libcsr_ref.so'))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_csr_matvecs_and_csc_matvec_bitexact(lib):
    rng = np.random.default_rng(0)
    for n, dens, k in [(50, 0.2, 1), (400, 0.03, 10), (1000, 0.01, 7)]:
        A = sparse.random(n, n, density=dens, random_state=3, format='csr')
        X = rng.normal(size=(n, k))
        Y = np.zeros((n, k))
        lib.ref_csr_matvecs(C.c_int64(n), C.c_int64(k), _p(A.indptr.astype(np.int32)), _p(A.indices.astype(np.int32)), _p(A.data), _p(X), _p(Y))
        assert np.array_equal(Y, A * X)
        Ac = A.tocsc()
        x = rng.normal(size=n)
        y = np.zeros(n)
        lib.ref_csc_matvec(C.c_int64(n), _p(Ac.indptr.astype(np.int32)), _p(Ac.indices.astype(np.int32)), _p(Ac.data), _p(x), _p(y))
        assert np.array_equal(y, Ac * x)


def test_c_sweeps_reproduce_golden(lib, golden):
    from oracle import gl_oracle as orc
    g = golden('g1_twomoons.npz')
    W = csr_from(g, 'W_gaussian')
    ti = g['train_ind']
    s = orc.poisson_gd_setup(W, ti, g['labels'][ti])
    P = sparse.csr_matrix(s['P'])
    n, k = W.shape[0], s['k']
    u = np.zeros((n, k))
    tmp = np.zeros((n, k))
    Db = np.ascontiguousarray(s['Db'])
    lib.ref_poisson_sweeps(C.c_int64(n), C.c_int64(k), _p(P.indptr.astype(np.int32)), _p(P.indices.astype(np.int32)), _p(P.data), _p(Db), _p(u), _p(tmp), C.c_int64(409))
    assert np.array_equal(u, g['poisson_gd_prob'])


def test_lp_iterate_restatement_equals_compiled_reference():
    """oracle/_ref/liblp_ref.so is the reference's own lp_iterate.cpp compiled as is (oracle/Makefile,
    present where /root/reference was available at build time): the restatement must agree bit for bit."""
    from oracle import gl_oracle as orc
    path = os.path.join(ROOT, 'oracle', '_ref', 'liblp_ref.so')
    if not os.path.exists(path):
        subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'oracle')], check=True)
    if not os.path.exists(path):
        pytest.skip('compiled reference not available (no /root/reference at build time)')
    f = getattr(C.CDLL(path), '_Z15lp_iterate_mainPdS_PiS0_S_S0_S_didbiii')
    f.restype = None
    rng = np.random.default_rng(1)
    for n, dens, p, tol, T in [(300, 0.03, 4.0, 1e-2, 5000), (800, 0.01, 20.0, 1e-1, 33), (150, 0.08, 2.2, 1e-6, 400)]:
        A = sparse.random(n, n, density=dens, random_state=7, format='csr')
        W = sparse.csr_matrix(A + A.T)
        bdy = rng.permutation(n)[:n // 10]
        val = rng.normal(size=len(bdy))
        u, it, uu, ul = orc.plaplace_jacobi(W, bdy, val, p, tol=tol, max_num_it=T, return_iters=True, return_bounds=True)
        I, J, V = orc.ccode_arrays(W)
        ru = np.max(val) * np.ones(n); rl = np.min(val) * np.ones(n); ru[bdy] = val; rl[bdy] = val
        b32 = np.ascontiguousarray(bdy, dtype=np.int32)
        f(_p(ru), _p(rl), _p(J), _p(I), _p(V), _p(b32), _p(val), C.c_double(p), C.c_int(T), C.c_double(tol), C.c_bool(False),
          C.c_int(n), C.c_int(len(V)), C.c_int(len(b32)))
        assert np.array_equal(ru, uu, equal_nan=True) and np.array_equal(rl, ul, equal_nan=True), (n, p)


def test_openmp_sweeps_equal_scipy(lib):
    """bench.py's all-core CPU baseline (rows of a sweep over OpenMP threads, fused stop column) is the
    same arithmetic as the scipy sweeps: bit-identical u and w after an even and an odd number of sweeps."""
    rng = np.random.default_rng(2)
    n, nc = 3000, 7
    P = sparse.random(n, n, density=0.004, random_state=5, format='csr')
    Db = rng.normal(size=(n, nc))
    w0 = rng.random(n)
    ip, ix = P.indptr.astype(np.int32), P.indices.astype(np.int32)
    for T in (4, 5):
        u, tmp, w, wt = np.zeros((n, nc)), np.zeros((n, nc)), w0.copy(), np.zeros(n)
        th = lib.ref_poisson_sweeps_omp(C.c_int64(n), C.c_int64(7), _p(ip), _p(ix), _p(P.data), _p(Db), _p(u), _p(tmp), _p(w), _p(wt),
                                        C.c_int64(T), C.c_int(3))
        assert th >= 1
        ur, wr = np.zeros((n, 7)), w0.copy()
        for _ in range(T):
            ur = Db + P * ur
            wr = P * wr
        assert np.array_equal(u if T % 2 == 0 else tmp, ur)
        assert np.array_equal(w if T % 2 == 0 else wt, wr)
