import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


# The golden vectors were written by the reference on the build container, whose numpy exp is that host's libm.  The suite that
# compares against them bit for bit therefore builds its Gaussian weights with numpy's exp on the host too (GLX_HOST_EXP=1:
# weightmatrix.knn's compatibility mode); the DEFAULT -- the correctly rounded exp on the device -- is covered by
# tests/test_gpu_weights.py and tests/test_exp_cr.py, which clear the variable (the `device_exp` fixture).
os.environ.setdefault('GLX_HOST_EXP', '1')


def pytest_addoption(parser):
    # --cg-form blocks|chain: every reference-order conjugate-gradient solve of the session walks numpy's reduction chains in that form
    # (graphlearning_amd._hip.CG_EXACT_FORM; same bits either way -- the whole parity suite can be run under either)
    parser.addoption('--cg-form', action='store', default=None, choices=['blocks', 'chain'])


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if config.getoption('--cg-form'):
        from graphlearning_amd import _hip
        _hip.CG_EXACT_FORM = config.getoption('--cg-form')
    # the multi-GPU tests need torch's HIP runtime to be the first one loaded in the process (graphlearning_amd.dist checks it):
    # whichever GPU test runs first, torch is already there
    # (GLX_TEST_NO_TORCH=1, a harness variable: leave torch out, so that libglx runs on the SYSTEM HIP runtime instead of the one bundled
    # with PyTorch -- the two differ, profiles/r06_graph_memset_probe.txt; the multi-GPU tests then have to be deselected: -k "not dist")
    mark = config.getoption('-m') or ''
    if 'gpu' in mark and 'not gpu' not in mark and os.environ.get('GLX_TEST_NO_TORCH') != '1':
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


# ---- breadcrumbs (soak runs) ----------------------------------------------------------------------------------------
# GLX_CRUMBS=<directory>: every worker process appends one line per test to <directory>/crumbs.<pid>.log BEFORE the test
# runs (test id, wall time) and one line with the outcome behind it, flushed and fsync'ed -- so that a failure, a crash of a
# worker or a hang inside a soak of thousands of cases names its case, and the sequence of cases the same process ran
# before it (what a failure that depends on the process's history needs: scripts/soak.sh, scripts/replay_crumbs.py).
_CRUMBS = os.environ.get('GLX_CRUMBS')


def _crumb(text):
    if not _CRUMBS:
        return
    import time
    os.makedirs(_CRUMBS, exist_ok=True)
    with open(os.path.join(_CRUMBS, 'crumbs.%d.log' % os.getpid()), 'a') as f:
        f.write('%.3f %s\n' % (time.time(), text))
        f.flush()
        os.fsync(f.fileno())


def pytest_sessionfinish(session, exitstatus):
    # the checked uploads of this process (round 6: the soak's one failure was an upload that arrived with a hole of zeros)
    if _CRUMBS:
        try:
            from graphlearning_amd import _hip
            if _hip.load(required=False) is not None:
                _crumb('UPLOADS %s' % _hip.upload_stats())
        except Exception:
            pass


def pytest_runtest_logstart(nodeid, location):
    _crumb('START %s' % nodeid)


def pytest_runtest_logreport(report):
    if _CRUMBS and (report.when == 'call' or report.outcome != 'passed'):
        _crumb('%s %s %s %.3fs' % (report.outcome.upper(), report.when, report.nodeid, report.duration))
        if report.outcome == 'failed':
            _crumb('TRACEBACK %s\n%s' % (report.nodeid, report.longreprtext))


# ---- ablations (soak runs) ------------------------------------------------------------------------------------------
# GLX_TEST_ABLATE=nopool,nopinned,nospec (any subset): the session runs with the named subsystem switched OFF through the product's
# own API switches (_hip.pool_set_enabled, _hip.PINNED_RESULTS, ssl.SPECULATIVE_FITS).  A test-harness variable, not a product one: the
# library never reads it.  scripts/soak.sh passes it through; a failure rate that moves with a switch names the subsystem.
_ABLATE = [a for a in os.environ.get('GLX_TEST_ABLATE', '').split(',') if a]


@pytest.fixture(scope='session', autouse=True)
def _ablations():
    if _ABLATE:
        unknown = {a for a in _ABLATE if not a.startswith('poison')} - {'nopool', 'nopinned', 'nospec', 'verifyupload', 'pageableupload'}
        assert not unknown, 'GLX_TEST_ABLATE: unknown names %s' % sorted(unknown)
        from graphlearning_amd import _hip, ssl
        if 'nospec' in _ABLATE:
            ssl.SPECULATIVE_FITS = False
        if 'nopinned' in _ABLATE:
            _hip.PINNED_RESULTS = False
        for a in _ABLATE:                       # poison<byte>: every pooled work buffer is filled with that byte when handed out
            if a.startswith('poison') and _hip.load(required=False) is not None:
                _hip.pool_set_poison(int(a[6:] or '255'))
        if ('verifyupload' in _ABLATE or 'pageableupload' in _ABLATE) and _hip.load(required=False) is not None:
            _hip.debug_set(1 if 'verifyupload' in _ABLATE else 0)
            if 'pageableupload' in _ABLATE:
                _hip.upload_set_mode(2)                 # every search checks its device copy of X against the caller's array (stderr + counters)
        if 'nopool' in _ABLATE and _hip.load(required=False) is not None:
            try:
                _hip.pool_set_enabled(False)
            except Exception:
                pass
        _crumb('ABLATE %s' % ','.join(_ABLATE))
    yield


@pytest.fixture
def device_exp():
    """weightmatrix.knn in its default mode inside the test."""
    old = os.environ.pop('GLX_HOST_EXP', None)
    yield
    if old is not None:
        os.environ['GLX_HOST_EXP'] = old


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
    return load


def csr_from(g, prefix):
    from scipy import sparse
    n = len(g[prefix + '_indptr']) - 1
    return sparse.csr_matrix((g[prefix + '_data'], g[prefix + '_indices'], g[prefix + '_indptr']), shape=(n, n))


def blobs(n, d, C, seed, scale, labels=None):
    """Same synthetic generator as tests/golden/make_golden.py."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(C, d)) * scale
    if labels is None:
        labels = rng.integers(0, C, size=n)
    X = centers[labels] + rng.normal(size=(n, d))
    return X, labels.astype(np.int64)


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def run_ranks(cmd, retries=3, **kw):
    """subprocess.run for a `python -m torch.distributed.run ... --master-port P ...` job: the port was free when it was picked,
    but another process may take it before the rendezvous binds it -- on EADDRINUSE the job is started again on a new port."""
    import subprocess
    r = None
    for _ in range(retries):
        r = subprocess.run(cmd, **kw)
        text = (r.stderr or '') + (r.stdout or '') if kw.get('capture_output') or kw.get('stderr') is not None else ''
        if r.returncode != 0 and '--master-port' in cmd and ('EADDRINUSE' in text or 'address already in use' in text.lower()):
            cmd = list(cmd)
            cmd[cmd.index('--master-port') + 1] = str(free_port())
            continue
        return r
    return r
