"""The block form of numpy's row-after-row reduction chains (graphlearning_amd/csrc/seqsum_exact.h; the reductions are
`np.sum(p * Ap, axis=0)` / `np.sum(r ** 2, axis=0)` of the reference's utils.conjgrad, graphlearning/utils.py:524,527): the header's
scalar arithmetic is compiled for the host and must return the plain chain's bits on every input -- whatever the approximate prefix it
is handed, and through the lane-for-lane restatement of the device walk as well as block by block."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module', params=[(16, 16, 2), (16, 2, 2), (32, 1, 2), (8, 8, 4), (16, 4, 1), (16, 16, 0)],
                ids=lambda p: 'rows%d_merged%d_split%d' % p)
def lib(request, tmp_path_factory):
    sub, q, S = request.param
    out = str(tmp_path_factory.mktemp('seqsum') / ('libss_%d_%d_%d.so' % (sub, q, S)))
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-DSS_SUB=%d' % sub, '-DSS_Q=%d' % q, '-DSS_MAXSPLIT=%d' % S, '-shared', '-fPIC', '-o', out,
                    os.path.join(ROOT, 'tests', 'seqsum_host.cpp')], check=True)
    L = ctypes.CDLL(out)
    for f in (L.ss_host_chain, L.ss_host_blocks, L.ss_host_walk):
        f.restype = ctypes.c_double
    return L


def _bits(v):
    return np.float64(v).tobytes()


def _check(L, x, noise=0.0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    p = x.ctypes.data_as(ctypes.c_void_p)
    n = ctypes.c_int64(x.size)
    want = L.ss_host_chain(p, n)
    # np.cumsum is the same chain; it starts from +0 here as in the device reducers (a column of nothing but -0 sums to +0: cg.hip)
    assert _bits(want) == _bits(np.cumsum(np.concatenate([[0.0], x]))[-1]) or np.isnan(want)
    out = []
    for f in (L.ss_host_blocks, L.ss_host_walk):
        st = (ctypes.c_int64 * 3)()
        got = f(p, n, ctypes.c_double(noise), st)
        assert _bits(got) == _bits(want) or (np.isnan(got) and np.isnan(want)), (f, want, got, list(st))
        out.append(list(st))
    return out


def _cases():
    rng = np.random.default_rng(0)
    n = 70000
    yield 'squares', rng.normal(size=n) ** 2
    yield 'mostly positive products', rng.normal(size=n) * rng.normal(size=n) + 0.3
    yield 'zero-mean walk', rng.normal(size=n)
    yield 'wide dynamic range', np.exp(rng.normal(size=n) * 8)
    yield 'tiny then large', np.concatenate([np.full(1000, 1e-30), rng.random(50000)])
    yield 'leading zeros', np.concatenate([np.zeros(5000), rng.random(30000) ** 2])
    yield 'halves (ties everywhere)', rng.integers(0, 1000, size=n).astype(float) * 0.5
    yield 'powers of two (ties)', 2.0 ** rng.integers(-30, 5, size=n)
    yield 'constant', np.full(n, 0.1)
    yield 'negative sum', -(rng.normal(size=n) ** 2)
    yield 'an inf', np.concatenate([rng.random(1000), [np.inf], rng.random(1000)])
    yield 'a nan', np.concatenate([rng.random(1000), [np.nan], rng.random(1000)])
    yield 'inf minus inf', np.concatenate([rng.random(100), [np.inf], rng.random(100), [-np.inf], rng.random(100)])
    yield 'cancellation', np.concatenate([rng.random(3000), -rng.random(3000) * 1.0001, rng.random(3000)])
    yield 'exact cancellation to zero', np.concatenate([np.arange(1, 2001.0), -np.arange(1, 2001.0)[::-1], rng.random(3000)])
    yield 'subnormals', np.full(5000, 5e-324)
    yield 'near overflow', np.full(3000, 1e305)
    yield 'overflow to inf', np.full(3000, 1e308)
    yield 'Dirichlet rows', (rng.normal(size=n) ** 2) * (rng.random(n) > 0.1)
    yield 'negative zeros', np.concatenate([np.full(100, -0.0), rng.random(500), np.full(100, -0.0)])
    yield 'one ulp steps', np.concatenate([[1.0], np.full(5000, 2.0 ** -53), np.full(5000, 2.0 ** -52)])
    yield 'alternating large', np.tile([1e16, -1e16, 1.0], 2000)
    for m in (0, 1, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4097, 16383, 16384, 16385, 40000):
        yield 'length %d' % m, rng.random(m)


@pytest.mark.parametrize('noise', [0.0, 1e-12, 1e-3, 0.5])
def test_block_form_returns_the_chain(lib, noise):
    """every input, every quality of guess (noise = relative error put on the approximate prefix): the chain's bits"""
    for name, x in _cases():
        _check(lib, x, noise)


def test_random_chains(lib):
    rng = np.random.default_rng(1)
    for t in range(300):
        n = int(rng.integers(1, 6000))
        kind = t % 5
        x = rng.normal(size=n)
        if kind == 1:
            x = x ** 2
        elif kind == 2:
            x = x * np.exp(rng.normal(size=n) * rng.uniform(0, 20))
        elif kind == 3:
            x = np.round(x * 2 ** rng.integers(0, 12)) / 2 ** rng.integers(0, 12)          # few significant bits: ties
        elif kind == 4:
            x = (x ** 2) * (rng.random(n) > rng.random())                                   # zero runs
        _check(lib, x, float(rng.choice([0.0, 1e-9])))


def test_squares_are_almost_all_plain_blocks(lib, request):
    """what makes it fast: a sum of squares (r.r) leaves a handful of blocks to their records and fewer to their rows"""
    if 'rows16_merged16_split2' not in request.node.name:
        pytest.skip('counts are asserted for the shipped block size / split budget')
    rng = np.random.default_rng(2)
    plain, by_record, by_rows = _check(lib, rng.normal(size=70000) ** 2)[1]
    assert by_rows <= 6 and by_record <= 30, (plain, by_record, by_rows)


def test_property_any_finite_or_not_doubles(lib):
    """hypothesis: arbitrary doubles (all exponents, signed zeros, subnormals, inf, nan), arbitrary lengths -- the chain's bits"""
    hyp = pytest.importorskip('hypothesis')
    from hypothesis import given, settings, strategies as st, HealthCheck
    import hypothesis.extra.numpy as hnp

    elems = st.one_of(st.floats(allow_nan=True, allow_infinity=True, width=64),
                      st.floats(min_value=-1e3, max_value=1e3, width=64),
                      st.sampled_from([0.0, -0.0, 0.5, 1.0, 2.0 ** -52, 2.0 ** -53, 1.0 + 2.0 ** -52, 2.0 ** 52, 2.0 ** 53, -2.0 ** 53]))

    @settings(max_examples=300, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
    @given(hnp.arrays(np.float64, st.integers(0, 700), elements=elems), st.sampled_from([0.0, 1e-12, 0.3]))
    def run(x, noise):
        with np.errstate(all='ignore'):
            _check(lib, x, noise)

    run()


def test_block_form_equals_numpys_own_axis0_sum_on_the_reference_cg(lib, golden):
    """the chain being replaced IS numpy's: `np.sum(p * Ap, axis=0)` and `np.sum(r ** 2, axis=0)` of utils.conjgrad
    (graphlearning/utils.py:524,527), evaluated by numpy on the products of the Poisson CG solve of the n = 5000 golden graph
    (reference inputs: tests/golden/g3_blobs5000.npz) -- every column of every iteration through the block form, bit for bit"""
    from conftest import csr_from
    from oracle import gl_oracle as orc
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    n = W.shape[0]
    ti = g['train_ind']
    src, _ = orc.poisson_source(n, ti, g['labels'][ti])
    L = orc.laplacian(W, 'normalized')
    D = orc.degree_matrix(W, p=-0.5)
    b = D * src
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    rsold = np.sum(r ** 2, axis=0)
    err, it, checked = 1.0, 0, 0

    def through_blocks(prod, want):
        nonlocal checked
        for c in range(prod.shape[1]):
            col = np.ascontiguousarray(prod[:, c])
            for f in (lib.ss_host_blocks, lib.ss_host_walk):
                st = (ctypes.c_int64 * 3)()
                got = f(col.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n), ctypes.c_double(0.0), st)
                assert _bits(got) == _bits(want[c]), (it, c, got, want[c])
            checked += 1

    while err > 1e-3 and it < 400:                      # utils.conjgrad, utils.py:521-530
        it += 1
        Ap = L @ p
        pAp = np.sum(p * Ap, axis=0)
        through_blocks(p * Ap, pAp)
        alpha = rsold / pAp
        x += alpha * p
        r -= alpha * Ap
        rsnew = np.sum(r ** 2, axis=0)
        through_blocks(r ** 2, rsnew)
        err = np.sqrt(np.sum(rsnew))
        p = r + (rsnew / rsold) * p
        rsold = rsnew
    assert it == int(g['poisson_cg_iters'])             # the loop above is the reference's solve
    assert checked == 2 * it * b.shape[1]
