"""Developer probe: config-4-shaped run (isotropic blobs, d=64, k=10, C=10, default_rng(2)) at n vertices
on one GPU: exact kNN, weight matrix, T fixed sweeps of ssl.poisson(gradient_descent).

    python scripts/scale_probe.py N [--cache /tmp/knn.npy] [--diag] [--dtype f64|f32|both] [--T 50] [--reps 5]

--cache keeps the kNN lists of the first run in an .npz (uncompressed) so that profiling passes of the same
size skip the search.  --diag prints how local the gathers of the sweep are under the vertex order the
library uses (fraction of stored entries whose endpoints lie within a window of the order; degree skew).
"""
import os, sys, time, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument('n', type=int, nargs='?', default=1000000)
ap.add_argument('--cache', default=None)
ap.add_argument('--diag', action='store_true')
ap.add_argument('--dtype', default='f64')
ap.add_argument('--T', type=int, default=50)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--bound', action='store_true', help='the distinct-line lower bound of the sweep\'s memory traffic (bench.distinct_line_bound)')
args = ap.parse_args()
n = args.n

rng = np.random.default_rng(2)
labels = rng.integers(0, 10, size=n)
centers = rng.normal(size=(10, 64)) * 4
if args.cache and os.path.exists(args.cache):
    f = np.load(args.cache)
    ind, dist = f['J'], f['D']
    print('kNN lists from %s' % args.cache)
else:
    X = centers[labels] + rng.normal(size=(n, 64))
    t0 = time.perf_counter(); ind, dist = gl.weightmatrix.knnsearch(X, 11); t1 = time.perf_counter()
    st = _hip.knn_stats()
    print('knn n=%d d=64: %.2f s wall, tile %.1f ms (%.1f TFLOP/s), rerank %.1f ms, fallback rows %d' % (
        n, t1 - t0, st['tile_ms'], 2.0 * n * n * st['dpa'] / st['tile_ms'] / 1e9, st['rerank_ms'], st['fallback_rows']))
    del X
    if args.cache:
        np.savez(args.cache, J=ind, D=dist)
t0 = time.perf_counter(); W = gl.weightmatrix.knn(None, 10, knn_data=(ind, dist)); t1 = time.perf_counter()
del ind, dist
lens = np.diff(W.indptr)
print('weight matrix: nnz=%d max row %d, %.2f s' % (W.nnz, lens.max(), t1 - t0))
train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
T = args.T

if args.diag:
    srt = np.sort(lens)[::-1].astype(np.float64)
    cum = np.cumsum(srt) / srt.sum()
    for frac in (0.001, 0.003, 0.01, 0.025, 0.1, 0.2):
        print('diag: the %.1f%% highest-degree vertices hold %.1f%% of the stored entries' % (frac * 100, 100 * cum[int(frac * n) - 1]))

for dt in (['f64', 'f32'] if args.dtype == 'both' else [args.dtype]):
    dtype = np.float64 if dt == 'f64' else np.float32
    m = gl.ssl.poisson(W, solver='gradient_descent', min_iter=T, max_iter=T, use_cuda=(dt == 'f32'))
    t0 = time.perf_counter(); dev, aux = m._operators(); t1 = time.perf_counter()
    print('host operator setup + upload: %.2f s' % (t1 - t0))
    src, k = gl.ssl._poisson_source(n, train_ind, labels[train_ind])
    v0 = np.zeros(n); v0[train_ind] = 1; v0 /= v0.sum()
    sw = _hip.Sweep(dev, k, min_iter=T, max_iter=T, use_hipgraph=True)
    sw.set_problem(aux['D'] * src, v0 / aux['deg'], aux['deg'], aux['vinf'])
    sw.run()
    tot = 0.0
    for _ in range(args.reps):
        _, ms = sw.run(); tot += ms
    us = tot * 1e3 / (args.reps * T)
    es = 8 if dt == 'f64' else 4
    ab = bench.algorithmic_bytes(n, W.nnz, 10, es, es)
    print('sweep %s: %.1f us/launch, algorithmic %.1f MB -> %.2f TB/s = %.1f%% of 8 TB/s; %.1f Gedge/s; %s' % (
        dt, us, ab / 1e6, ab / us / 1e6, ab / us / 1e6 / 8 * 100, W.nnz / us / 1e3, dev.info()))
    if args.bound and dt == 'f64':
        recs, rb = bench.distinct_line_bound(W, dev.order(), 128)
        bound = rb + W.nnz * 12 + 4 * (n + 1) + n * 128 + 2 * n * 8
        print('traffic bounds per sweep (fp64): algorithmic %.1f MB | distinct-line bound %.1f MB (%d distinct neighbour records over the 8 XCD ranges = %.2f per vertex, '
              '%.2f of the stored entries) | counters: see the rocprofv3 passes' % (ab / 1e6, bound / 1e6, recs, recs / n, recs / W.nnz))
    u = sw.fetch()
    pred = np.argmax(u, axis=1)
    print('accuracy %.2f%%' % gl.ssl.ssl_accuracy(pred, labels, train_ind))
    if args.diag and dt == 'f64':
        perm = getattr(dev, 'order', lambda: None)()
        if perm is not None:
            pos = np.empty(n, dtype=np.int64); pos[perm] = np.arange(n)
            rows = np.repeat(np.arange(n), lens)
            dpos = np.abs(pos[rows] - pos[W.indices])
            for win in (8192, 32768, 262144, 2097152):
                print('diag: %.1f%% of the stored entries have both endpoints within %d positions of the library order' % (
                    100 * np.mean(dpos < win), win))
            del rows, dpos
    sw.close()
    m._cache[1].close()
