"""Every environment variable the package reads (README.md has the table) with the parity core run under it: the golden kNN lists,
the two-moons weight matrix, Poisson gradient descent (iterates + T), Poisson CG and Laplace learning -- bit-identical to the
reference's vectors whatever the setting (reference graphlearning/weightmatrix.py:68-187, ssl.py:608-677, :1206-1261).  One fresh
process per setting: several of the variables are read once, at import or at first use."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CORE = r'''
import os, sys
import numpy as np
from scipy import sparse
sys.path.insert(0, %(root)r)
import graphlearning_amd as gl
from graphlearning_amd import _hip
g1 = dict(np.load(os.path.join(%(root)r, 'tests', 'golden', 'g1_twomoons.npz')))
g2 = dict(np.load(os.path.join(%(root)r, 'tests', 'golden', 'g2_knn.npz')))
J, D = gl.weightmatrix.knnsearch(g2['X_d20'], 11)
assert np.array_equal(J, g2['J_d20']) and np.max(np.abs(D - g2['D_d20'])) <= 1e-12
W = gl.weightmatrix.knn(g1['X'], 10)
assert np.array_equal(W.indptr, g1['W_gaussian_indptr']) and np.array_equal(W.indices, g1['W_gaussian_indices'])
if os.environ.get('GLX_HOST_EXP') == '1':
    assert np.array_equal(W.data, g1['W_gaussian_data'])
else:
    assert np.max(np.abs(W.data - g1['W_gaussian_data'])) <= 1e-15
n = len(g1['labels'])
W = sparse.csr_matrix((g1['W_gaussian_data'], g1['W_gaussian_indices'], g1['W_gaussian_indptr']), shape=(n, n))
ti, lab = g1['train_ind'], g1['labels']
m = gl.ssl.poisson(W, solver='gradient_descent')
u = m.fit(ti, lab[ti])
assert m.num_iter == int(g1['poisson_gd_T']) and np.array_equal(u, g1['poisson_gd_prob'])
assert np.array_equal(gl.ssl.poisson(W).fit(ti, lab[ti]), g1['poisson_cg_prob'])
assert np.array_equal(gl.ssl.laplace(W, reduce='exact').fit(ti, lab[ti]), g1['laplace_combinatorial_prob'])
# a graph above the sizes at which the search forms cells / the operators take the search's order
rng = np.random.default_rng(5)
X = rng.normal(size=(8, 16))[rng.integers(0, 8, size=6000)] * 2.0 + rng.normal(size=(6000, 16))
from oracle import gl_oracle as orc
Jo, Do = orc.knnsearch(X, 11)
Wb = gl.weightmatrix.knn(X, 10)
Jb, Db = gl.weightmatrix.knnsearch(X, 11)
assert np.array_equal(Jb, Jo.reshape(Jb.shape))
lb = rng.integers(0, 8, size=6000)
tb = gl.trainsets.generate(lb, rate=3, seed=1)
mb = gl.ssl.poisson(Wb, solver='gradient_descent')
ub = mb.fit(tb, lb[tb])
uo, To = orc.poisson_gd(Wb, tb, lb[tb], return_T=True)
assert mb.num_iter == To and np.array_equal(ub, uo)
print('parity core ok')
'''


def _golden_keys():
    import numpy as np
    g1 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_twomoons.npz'))
    return set(g1.files)


@pytest.mark.parametrize('name,value', [
    ('GLX_DEVICE', '0'),                 # default device of calls without an explicit one
    ('GLX_TIMING', '1'),                 # stage timers on stderr
    ('GLX_HOST_EXP', '1'),               # numpy's exp on the host for the Gaussian weights (this host's reference bits)
    ('GLX_HOST_EXP', '0'),               # (the default: correctly rounded exp on the device)
    ('GLX_HOST_THREADS', '1'),           # host-side helpers on one thread
    ('GLX_PINNED_MAX_MB', '1'),          # page-locked result pool capped: arrays come from ordinary memory beyond it
    ('GLX_KNN_CLUSTERED', '0'),          # never the cell-pruned search
    ('GLX_KNN_CLUSTERED', '8'),          # the cell-pruned search with 8 cells, whatever the size
    ('GLX_KNN_ORDER', '0'),              # operators order the vertices by their own pass over the graph
])
def test_parity_core_under_environment_variable(name, value, tmp_path):
    need = {'X', 'W_gaussian_indptr', 'poisson_gd_T', 'poisson_gd_prob', 'poisson_cg_prob', 'laplace_combinatorial_prob', 'train_ind', 'labels'}
    assert need <= _golden_keys(), need - _golden_keys()
    env = dict(os.environ)
    env.pop('GLX_HOST_EXP', None)
    env[name] = value
    r = subprocess.run([sys.executable, '-c', CORE % {'root': ROOT}], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and 'parity core ok' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    if name == 'GLX_TIMING':
        assert '[glx]' in r.stderr
