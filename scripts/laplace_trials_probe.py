"""Developer probe: ssl.laplace over many training sets (config-3-shaped graph), batched vs one by one."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lab = np.load(os.path.join(root, 'tests', 'golden', 'cifar_labels.npz'))['labels'][:60000].astype(np.int64)
rng = np.random.default_rng(1)
X = (rng.normal(size=(10, 32)) * 1.2)[lab] + rng.normal(size=(60000, 32))
W = gl.weightmatrix.knn(X, 20)
trainsets = gl.trainsets.generate(lab, rate=np.array([[2], [5], [10]]), num_trials=8, seed=0)   # 24 training sets
for batched in (True, False):
    m = gl.ssl.laplace(W)
    if not batched:
        m._trial_batch_size = lambda labels: 1
    m.fit(trainsets[0], lab[trainsets[0]])        # operator upload
    t0 = time.perf_counter()
    m.ssl_trials(trainsets, lab, save_results=False)
    dt = time.perf_counter() - t0
    print('== laplace 60k k=20, %d trials, batched=%s: %.3f s = %.1f ms per trial (iterations %s)' % (len(trainsets), batched, dt, dt / len(trainsets) * 1e3, m.num_iter))
