"""The off switches of the subsystems that sit between a caller and the kernels: the device work-buffer pool
and idle work sets (_hip.pool_set_enabled), the page-locked result blocks (_hip.PINNED_RESULTS), the speculative fit on the previous
content fingerprint (ssl.SPECULATIVE_FITS).  With each one off, the whole path -- search, weight matrix, every learner, an in-place
edit of the matrix between two fits -- returns the same bits as with everything on."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _walk(gl):
    """A pass over the path that exercises all three subsystems: graphs of several sizes built back to back (pooled work buffers of
    several size classes, recycled page-locked CSR arrays), fits repeated on the same matrix object (speculation), an in-place edit."""
    from conftest import blobs
    out = []
    for seed, n, k in ((1, 700, 7), (2, 2600, 10), (3, 700, 7), (4, 9000, 12)):
        X, lab = blobs(n, 12, 4, seed, 1.6)
        W = gl.weightmatrix.knn(X, k)
        out += [W.indptr.copy(), W.indices.copy(), W.data.copy()]
        ti = gl.trainsets.generate(lab, rate=3, seed=seed)
        for make in (lambda: gl.ssl.poisson(W, solver='gradient_descent'), lambda: gl.ssl.poisson(W), lambda: gl.ssl.laplace(W),
                     lambda: gl.ssl.laplace(W, reduce='exact'), lambda: gl.ssl.poisson_mbo(W, gl.utils.class_priors(lab), Ns=10, T=4)):
            m = make()
            for rep in range(3):                          # the second and third fit find the first one's fingerprint
                ti_r = gl.trainsets.generate(lab, rate=3, seed=seed + rep)
                out.append(np.array(m.fit(ti_r, lab[ti_r]), copy=True))
                out.append(np.array(m.predict(), copy=True))
                out.append(np.array([getattr(m, 'num_iter', -1)]))
        # stacked trials with class priors: the head graph of the stacked sweeps is replayed batch after batch, the weights of the volume
        # projection run through all of them (a stop row that a replay leaves wrong shows as a 51st sweep and other labels: round 6)
        tsets = gl.trainsets.generate(lab, rate=np.array([[1], [2], [4]]), num_trials=4, seed=seed)
        mt = gl.ssl.poisson(W, class_priors=gl.utils.class_priors(lab), solver='gradient_descent')
        out.append(np.array([[float(v) for v in row.split(',')] for row in mt._trial_rows(tsets, lab)]))
        m = gl.ssl.poisson(W, solver='gradient_descent')
        m.fit(ti, lab[ti])
        W.data *= 0.5                                      # edited in place: the next fit must see the new content
        W.data[::7] *= 1.25
        out.append(np.array(m.fit(ti, lab[ti]), copy=True))
        out.append(np.array([m.num_iter]))
    return out


@pytest.fixture(scope='module')
def gl():
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    _hip.require_device()
    return gl


@pytest.fixture(scope='module')
def everything_on(gl):
    return _walk(gl)


def _same(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y, equal_nan=True), i


def test_identical_with_the_device_pool_bypassed(gl, everything_on):
    from graphlearning_amd import _hip
    _hip.pool_set_enabled(False)
    try:
        _same(everything_on, _walk(gl))
    finally:
        _hip.pool_set_enabled(True)
    _same(everything_on, _walk(gl))                        # and after switching it back on


@pytest.mark.parametrize('byte', [0x7f, 0xff, 0x00])
def test_identical_with_a_poisoned_pool(gl, everything_on, byte):
    """Every pooled block is filled with `byte` when it is handed out (_hip.pool_set_poison): nothing may read a work buffer before writing
    it -- and, on the HIP runtime that comes with PyTorch (the suite loads torch first: conftest.py), the eager hipMemset of the fill used to
    change what the memset NODES of replayed launch graphs wrote (scripts/probes/graph_memset_probe.hip): captured sequences zero
    their rows with a kernel now (glx_zero_async)."""
    from graphlearning_amd import _hip
    _hip.pool_set_poison(byte)
    try:
        got = _walk(gl)
    finally:
        session = [a for a in os.environ.get('GLX_TEST_ABLATE', '').split(',') if a.startswith('poison')]      # (an ablation run's own fill)
        _hip.pool_set_poison(int(session[0][6:] or '255') if session else -1)
    _same(everything_on, got)


def test_identical_without_page_locked_result_blocks(gl, everything_on):
    from graphlearning_amd import _hip
    old = _hip.PINNED_RESULTS
    _hip.PINNED_RESULTS = False
    try:
        got = _walk(gl)
    finally:
        _hip.PINNED_RESULTS = old
    _same(everything_on, got)


def test_identical_without_speculative_fits(gl, everything_on):
    from graphlearning_amd import ssl
    old = ssl.SPECULATIVE_FITS
    ssl.SPECULATIVE_FITS = False
    try:
        got = _walk(gl)
    finally:
        ssl.SPECULATIVE_FITS = old
    _same(everything_on, got)


@pytest.mark.parametrize('mode', [1, 2])
def test_identical_however_the_big_transfers_travel(gl, everything_on, mode):
    """_hip.upload_set_mode: 0 (default) = transfers of 128 KB or more through the library's page-locked staging, checked by word sums;
    1 = staged, unchecked; 2 = straight from / into the caller's memory as rounds 1-5 did.  Same results -- and in the default mode the
    walk's transfers were all checked and none differed."""
    from graphlearning_amd import _hip
    if 'pageableupload' in os.environ.get('GLX_TEST_ABLATE', ''):
        pytest.skip('the session runs with direct copies (an ablation run): nothing is checked')
    before = _hip.upload_stats()
    assert before['checked'] > 0 and before['wrong_sums'] == 0 and before['given_up'] == 0, before      # (the module's `everything_on` walk)
    _hip.upload_set_mode(mode)
    try:
        got = _walk(gl)
    finally:
        _hip.upload_set_mode(0)
    _same(everything_on, got)
    assert _hip.upload_stats()['checked'] == before['checked']          # nothing was checked in modes 1 / 2


def test_big_transfers_in_pieces_round_trip(gl):
    """glx_upload / glx_download beyond one piece of the staging area (4 MB pieces from 6 MB on, two halves taking turns), with lengths that are
    not multiples of the piece, through the one entry point that is nothing but upload -> elementwise kernel -> download (glx_exp_cr): a 40 MB
    array gives the same bits as its slices sent one by one, and every transfer was checked."""
    from graphlearning_amd import _hip
    if 'pageableupload' in os.environ.get('GLX_TEST_ABLATE', ''):
        pytest.skip('the session runs with direct copies (an ablation run): nothing is checked')
    rng = np.random.default_rng(0)
    x = rng.normal(size=5_000_011) * 3.0
    before = _hip.upload_stats()
    big = _hip.exp_cr(x)
    after = _hip.upload_stats()
    assert after['checked'] >= before['checked'] + 2 and after['wrong_sums'] == before['wrong_sums'] and after['given_up'] == 0, (before, after)
    parts = np.concatenate([_hip.exp_cr(x[a:a + 9001]) for a in range(0, 300_000, 9001)])           # (72 KB each: direct copies)
    assert np.array_equal(big[:len(parts)], parts)
    tail = _hip.exp_cr(x[-70_001:])                                                                    # 560 KB: one checked piece
    assert np.array_equal(big[-70_001:], tail)
    with np.errstate(over='ignore'):
        assert np.max(np.abs(big - np.exp(x)) / np.exp(x)) < 3e-16
