"""Developer probe (round 6): ssl_trials with class priors, batched against one by one, under a poisoned pool and with some history in the
process -- where do the two first differ?  Records, per trial: the sweeps T, the class weights before and after the
volume projection, its error and a hash of the labels."""
import os, sys, hashlib, contextlib, io, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
os.environ.setdefault('GLX_HOST_EXP', '1')          # the suite's mode (tests/conftest.py)
if os.environ.get('WITH_TORCH', '1') == '1':
    import torch  # noqa: F401  (the suite loads it first: tests/conftest.py)
import graphlearning_amd as gl
from graphlearning_amd import _hip, ssl as glssl
from conftest import blobs

for a in (sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] else []):
    if a == 'nopool': _hip.pool_set_enabled(False)
    if a.startswith('poison'): _hip.pool_set_poison(int(a[6:]))
history = len(sys.argv) > 2 and sys.argv[2] == 'history'
if history:
    import test_gpu_groups as tg
    tg.test_groups_unstackable_batches_fall_back(gl)
if history and os.environ.get('MORE_HISTORY'):
    X0, l0 = blobs(900, 6, 3, 4, 1.5)
    W0 = gl.weightmatrix.knn(X0, 7)
    t0 = gl.trainsets.generate(l0, rate=2, seed=1)
    m0 = gl.ssl.poisson(W0, class_priors=gl.utils.class_priors(l0), solver='gradient_descent')
    with contextlib.redirect_stdout(io.StringIO()):
        glssl.results_dir = tempfile.mkdtemp()
        m0.ssl_trials(gl.trainsets.generate(l0, rate=np.array([[1], [2]]), num_trials=4, seed=3), l0, tag='h_', overwrite=True)
X, labels = blobs(2500, 12, 5, 9, 1.6)
W = gl.weightmatrix.knn(X, 8)
trainsets = gl.trainsets.generate(labels, rate=np.array([[1], [2], [4]]), num_trials=5, seed=3)
glssl.results_dir = tempfile.mkdtemp()
sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
_run = _hip.SweepGroups.run
def run_logged(self, used=None):
    out = _run(self, used)
    u = self.B if used is None else used
    print('   groups run: T', out[0], 'stop values', [(_hip.SweepGroups.stop_values(self, b)[0], ['%.6e' % v for v in _hip.SweepGroups.stop_values(self, b)[1]]) for b in range(u)], '1/n = %.6e' % (1.0 / self.graph.shape[0]), file=sys.stderr)
    return out
_hip.SweepGroups.run = run_logged
logs = {}
labs = {}
Ts = {}
for tag, batched in (('b_', True), ('s_', False)):
    plain = gl.ssl.poisson(W, solver='gradient_descent')         # (the test runs the learner without priors first)
    if not batched:
        plain._trial_batch_size = lambda labels: 1
    with contextlib.redirect_stdout(io.StringIO()):
        plain.ssl_trials(trainsets, labels, tag=tag, overwrite=True)
    model = gl.ssl.poisson(W, class_priors=gl.utils.class_priors(labels), solver='gradient_descent')
    if not batched:
        model._trial_batch_size = lambda labels: 1
    log = logs[tag] = []
    orig = model.volume_label_projection
    def wrapped(model=model, orig=orig, log=log):
        w0 = np.array(model.weights, dtype=float, copy=True) if type(model.weights) != int else None
        lab = orig()
        Ts.setdefault(id(log), []).append(getattr(model, 'num_iter', None))
        log.append(dict(w0=None if w0 is None else w0.tolist(), w1=np.array(model.weights).tolist(),
                        err=float(model.class_priors_error), lab=sha(lab)))
        labs.setdefault(id(log), []).append(np.array(lab, copy=True))
        return lab
    model.volume_label_projection = wrapped
    with contextlib.redirect_stdout(io.StringIO()):
        model.ssl_trials(trainsets, labels, tag=tag, overwrite=True)
nd = 0
for i, (a, b) in enumerate(zip(logs['b_'], logs['s_'])):
    if a != b:
        nd += 1
        if nd <= 3:
            print('trial %d differs:' % i)
            for k in a:
                if a[k] != b[k]: print('   %s: batched %s | one by one %s' % (k, a[k], b[k]))
print('%d trials, %d differ' % (len(logs['b_']), nd))
print('sweeps per trial, batched:', Ts[id(logs['b_'])])
print('sweeps per trial, one by one:', Ts[id(logs['s_'])])
la, lb = labs[id(logs['b_'])], labs[id(logs['s_'])]
for i, (x, y) in enumerate(zip(la, lb)):
    d = np.flatnonzero(x != y)
    if len(d):
        print('trial %d: %d of %d labels differ; first positions %s; batched there %s, one by one %s; batched label range %d..%d' % (
            i, len(d), len(x), d[:12].tolist(), x[d[:12]].tolist(), y[d[:12]].tolist(), x.min(), x.max()))
        break
