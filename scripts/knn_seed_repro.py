"""Replay of seeds of tests/test_gpu_fuzz.py::test_random_knn_searches_match_ckdtree: python scripts/knn_seed_repro.py SEED [SEED ...] [--reps N]
Both searches the test compares (weightmatrix.knnsearch and the reordered one) against cKDTree, repeated: is a difference a property of
the input (every repetition) or of timing (some)?  Prints the rows / columns that differ and the distances involved."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip
from scipy.spatial import cKDTree

reps = 20
seeds = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == '--reps':
        reps = int(args.pop(0))
    else:
        seeds.append(int(a))


def case(seed):
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([2, 3, 17, 64, 65, 129, 500, 1500, 3000, 6000]))
    d = int(rng.choice([1, 2, 3, 7, 16, 17, 32, 33, 50, 64, 65, 96, 97, 128, 129, 200, 300]))
    k = int(min(n, rng.choice([1, 2, 5, 11, 12, 13, 21, 28, 29, 40, 60])))
    style = int(rng.integers(0, 4))
    if style == 0:
        X = rng.normal(size=(n, d))
    elif style == 1:
        C = int(rng.integers(2, 12))
        X = rng.normal(size=(C, d))[rng.integers(0, C, size=n)] * 3.0 + rng.normal(size=(n, d))
    elif style == 2:
        X = rng.normal(size=(n, d)) + 50.0
    else:
        X = rng.normal(size=(n, d)) * np.exp(rng.normal(size=(1, d)) * 2.0)
    sim = 'angular' if rng.random() < 0.2 and d > 1 else 'euclidean'
    return X, n, d, k, style, sim


for seed in seeds:
    X, n, d, k, style, sim = case(seed)
    print('seed %d: n=%d d=%d k=%d style=%d %s' % (seed, n, d, k, style, sim), flush=True)
    Y = X / np.linalg.norm(X, axis=1)[:, None] if sim == 'angular' else X
    Do, Jo = cKDTree(Y).query(Y, k=k)
    Jo, Do = Jo.reshape(n, -1), Do.reshape(n, -1)
    bad = {'plain': 0, 'ordered': 0, 'plain_vs_ordered': 0}
    for r in range(reps):
        J, D = gl.weightmatrix.knnsearch(X, k, similarity=sim)
        J, D = np.array(J), np.array(D)
        st1 = _hip.knn_stats()
        J2, D2 = _hip.knn_bruteforce(X, k, similarity=sim, want_order=True)
        J2, D2 = np.array(J2), np.array(D2)
        st2 = _hip.knn_stats()
        for name, (a, b, c, e) in (('plain', (J, D, Jo, Do)), ('ordered', (J2, D2, Jo, Do)), ('plain_vs_ordered', (J, D, J2, D2))):
            if not (np.array_equal(a, c) and np.array_equal(b, e)):
                bad[name] += 1
                if bad[name] <= 2:
                    rows = np.flatnonzero(np.any(a != c, axis=1) | np.any(b != e, axis=1))
                    print('  rep %d %s: %d rows differ; first: row %d' % (r, name, len(rows), rows[0]))
                    i = rows[0]
                    cols = np.flatnonzero((a[i] != c[i]) | (b[i] != e[i]))
                    print('    cols', cols[:8], 'idx', a[i, cols[:8]], 'vs', c[i, cols[:8]])
                    print('    dist', b[i, cols[:8]], 'vs', e[i, cols[:8]], 'diff', (b[i, cols[:8]] - e[i, cols[:8]]))
                    print('    stats plain: filter=%s fallback_rows=%s lists=%s | ordered: filter=%s fallback_rows=%s' % (
                        st1.get('filter'), st1.get('fallback_rows'), st1.get('lists'), st2.get('filter'), st2.get('fallback_rows')))
    print('  of %d repetitions: %s' % (reps, bad), flush=True)
