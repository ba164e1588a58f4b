"""Developer probe: microseconds per launch of the sweep kernel for the GLX_PERSIST of this process
(blocks per workgroup; 1 = one block per workgroup, the round-2 form) -- config 2 (70 000 vertices), fp64 and fp32, and
optionally the n = 10^6 shard of config 4 (--big, kNN lists cached in --cache).  Run once per GLX_PERSIST value."""
import os, sys, time, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument('--big', type=int, default=0)
ap.add_argument('--cache', default='/tmp/knn_big.npz')
ap.add_argument('--reps', type=int, default=40)
args = ap.parse_args()
tag = 'GLX_PERSIST=%s GLX_SELL_L1/L4=%s GLX_XCD_BALANCE=%s' % (os.environ.get('GLX_PERSIST', '(default)'), '%s/%s' % (os.environ.get('GLX_SELL_L1', 'default'), os.environ.get('GLX_SELL_L4', 'default')),
                                                          os.environ.get('GLX_XCD_BALANCE', '(default)'))


def probe(W, labels, train_ind, name, dtype, reps, T=50):
    n = W.shape[0]
    m = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=(dtype == np.float32), min_iter=T, max_iter=T)
    dev, aux = m._operators()
    src, k = gl.ssl._poisson_source(n, train_ind, labels[train_ind])
    v0 = np.zeros(n); v0[train_ind] = 1; v0 /= v0.sum()
    sw = _hip.Sweep(dev, k, min_iter=T, max_iter=T, use_hipgraph=True)
    sw.set_problem(aux['D'] * src, v0 / aux['deg'], aux['deg'], aux['vinf'])
    for _ in range(3):
        sw.run()
    best, tot = 1e9, 0.0
    for _ in range(reps):
        ms = sw.run()[1]
        tot += ms
        best = min(best, ms)
    print('%s  %-18s %s: %.2f us/launch mean, %.2f best (HIP events over %d x %d launches); slices %d, stored %d' % (
        tag, name, np.dtype(dtype).name, tot * 1e3 / (reps * T), best * 1e3 / T, reps, T, dev.info()['slices'], dev.info()['stored']), flush=True)
    u = sw.fetch()
    sw.close()
    m._cache[1].close()
    return u


labels = bench.load_labels(70000)
W = gl.weightmatrix.knn(bench.make_features(labels), 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
u64 = probe(W, labels, ti, 'config 2 (70k)', np.float64, args.reps)
probe(W, labels, ti, 'config 2 (70k)', np.float32, args.reps)
import hashlib
print('%s  u sha %s' % (tag, hashlib.sha256(np.ascontiguousarray(u64).tobytes()).hexdigest()[:16]))
if args.big:
    n = args.big
    rng = np.random.default_rng(2)
    labels = rng.integers(0, 10, size=n)
    centers = rng.normal(size=(10, 64)) * 4
    if os.path.exists(args.cache):
        f = np.load(args.cache)
        ind, dist = f['J'], f['D']
    else:
        X = centers[labels] + rng.normal(size=(n, 64))
        ind, dist = gl.weightmatrix.knnsearch(X, 11)
        np.savez(args.cache, J=ind, D=dist)
        del X
    W = gl.weightmatrix.knn(None, 10, knn_data=(ind, dist))
    ti = gl.trainsets.generate(labels, rate=5, seed=0)
    u = probe(W, labels, ti, 'n = %d (d=64)' % n, np.float64, max(3, args.reps // 10))
    print('%s  big u sha %s' % (tag, hashlib.sha256(np.ascontiguousarray(u).tobytes()).hexdigest()[:16]))
