"""Developer probe: the two candidate filters of the exact kNN search (split-bf16 on the bf16 matrix cores vs fp32-input
MFMA) on the shapes of configs 2, 3 and 4: tile-kernel time, fallback rows, identical results."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphlearning_amd import _hip

def blobs(n, d, seed, scale):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 10, size=n)
    return rng.normal(size=(10, d))[lab] * scale + rng.normal(size=(n, d))

cases = [('config 2: n=70000 d=20 k=11', bench.make_features(bench.load_labels(70000)), 11),
         ('config 3: n=60000 d=32 k=21', blobs(60000, 32, 1, 1.2), 21),
         ('d=64 n=300000 k=11', blobs(300000, 64, 2, 4.0), 11),
         ('d=128 n=100000 k=11', blobs(100000, 128, 3, 2.0), 11),
         ('d=5 n=200000 k=11', blobs(200000, 5, 4, 1.0), 11)]
if os.environ.get('PROBE_ONLY_D64'):
    cases = [c for c in cases if 'd=64' in c[0]]
if len(sys.argv) > 1:
    cases.append(('config 4 shard: n=1e6 d=64 k=11', blobs(1000000, 64, 2, 4.0), 11))
for name, X, k in cases:
    res = {}
    for flt in (('bf16',) if os.environ.get('PROBE_ONLY_D64') else ('bf16', 'f32')):
        with _hip.knn_options(filter=flt):
            _hip.knn_bruteforce(X, k)
            t0 = time.perf_counter(); J, D = _hip.knn_bruteforce(X, k); wall = time.perf_counter() - t0
        st = _hip.knn_stats()
        res[flt] = (J, D)
        n, d = X.shape
        print('%-34s %-4s: tile %8.2f ms (%6.1f TFLOP/s on 2 n^2 d), rerank %6.2f ms, fallback rows %5d (%.2f ms), wall %.1f ms, lists %s' % (
            name, flt, st['tile_ms'], 2.0 * n * n * d / st['tile_ms'] / 1e9, st['rerank_ms'], st['fallback_rows'], st['fallback_ms'], wall * 1e3, st['KP']))
    if 'f32' in res:
        print('   identical neighbour lists: %s, identical distances: %s' % (np.array_equal(res['bf16'][0], res['f32'][0]), np.array_equal(res['bf16'][1], res['f32'][1])))
