"""Developer probe: does a vertex order taken from FEATURE space put more of a row's neighbours inside
one L2 window than the library's reverse Cuthill-McKee pass over the graph?  Config-4-shaped data (isotropic blobs, d = 64,
k = 10) at n vertices; for every order: the share of stored entries whose endpoints lie within 32 768 / 262 144 positions
(the records one XCD's 4 MB L2 / all eight hold) and the microseconds of a sweep with that order; `--pmc` runs only the
sweeps of one order (for rocprofv3 counter passes).

    python scripts/order_probe.py N [--cache /tmp/knn.npz] [--orders rcm,kmeans,kdtree,pca,none] [--only ORDER]
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
ap.add_argument('--orders', default='rcm,none,blob,kdtree,kmeans,kmeans_rcm')
ap.add_argument('--T', type=int, default=50)
ap.add_argument('--reps', type=int, default=4)
ap.add_argument('--dtype', default='f64')
args = ap.parse_args()
n = args.n
rng = np.random.default_rng(2)
labels = rng.integers(0, 10, size=n)
centers = rng.normal(size=(10, 64)) * 4
X = centers[labels] + rng.normal(size=(n, 64))
if args.cache and os.path.exists(args.cache):
    f = np.load(args.cache)
    ind, dist = f['J'], f['D']
else:
    ind, dist = gl.weightmatrix.knnsearch(X, 11)
    if args.cache:
        np.savez(args.cache, J=ind, D=dist)
W = gl.weightmatrix.knn(None, 10, knn_data=(ind, dist))
del ind, dist
lens = np.diff(W.indptr)
rows = np.repeat(np.arange(n), lens)
train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
T = args.T


def kdtree_order(X, leaf=2048):
    """Recursive median split on the coordinate of largest spread; leaves in tree order."""
    out = []
    stack = [np.arange(len(X))]
    while stack:
        idx = stack.pop()
        if len(idx) <= leaf:
            out.append(idx)
            continue
        sub = X[idx]
        dim = int(np.argmax(sub.max(axis=0) - sub.min(axis=0)))
        half = len(idx) // 2
        part = np.argpartition(sub[:, dim], half)
        stack.append(idx[part[half:]])
        stack.append(idx[part[:half]])
    return np.concatenate(out)


def kmeans_cells(X, ncells, iters=6, seed=0, sample=200000):
    """Lloyd on a sample; returns the cell of every point."""
    g = np.random.default_rng(seed)
    S = X[g.choice(len(X), size=min(sample, len(X)), replace=False)]
    Cn = S[g.choice(len(S), size=ncells, replace=False)].copy()
    for _ in range(iters):
        d = (S * S).sum(1)[:, None] - 2 * S @ Cn.T + (Cn * Cn).sum(1)[None, :]
        a = d.argmin(1)
        for c in range(ncells):
            m = a == c
            if m.any():
                Cn[c] = S[m].mean(0)
    cell = np.empty(len(X), dtype=np.int64)
    for lo in range(0, len(X), 200000):
        B = X[lo:lo + 200000]
        d = -2 * B @ Cn.T + (Cn * Cn).sum(1)[None, :]
        cell[lo:lo + 200000] = d.argmin(1)
    return cell, Cn


def chain(Cn):
    """Greedy nearest-neighbour chain through the cell centres."""
    left = list(range(len(Cn)))
    cur = left.pop(0)
    out = [cur]
    while left:
        d = ((Cn[left] - Cn[cur]) ** 2).sum(1)
        cur = left.pop(int(d.argmin()))
        out.append(cur)
    return np.array(out)


def make_order(name):
    t0 = time.perf_counter()
    if name == 'rcm':
        return None, 0.0
    if name == 'none':
        o = np.arange(n)
    elif name == 'blob':
        o = np.argsort(labels, kind='stable')
    elif name == 'kdtree':
        o = kdtree_order(X)
    elif name.startswith('kmeans'):
        cell, Cn = kmeans_cells(X, 512)
        rank = np.empty(len(Cn), dtype=np.int64)
        rank[chain(Cn)] = np.arange(len(Cn))
        o = np.argsort(rank[cell], kind='stable')
    elif name == 'pairs':
        # fp32 records are 64 bytes: two per 128-byte line.  Keep the library's breadth-first order but make line-mates of
        # vertices that are neighbours (greedy matching along that order, preferring the unmatched neighbour with the most
        # common neighbours): a row that references both then fetches ONE line for the two of them.
        base = _hip.DeviceGraph(gl.ssl._poisson_operator_symmetric(W)[0], dtype=np.float32)
        rcm = base.order().astype(np.int64)
        base.close()
        mate = np.full(n, -1, dtype=np.int64)
        ip, ix = W.indptr, W.indices
        nbsets = None
        for v in rcm:
            if mate[v] >= 0:
                continue
            nb = ix[ip[v]:ip[v + 1]]
            free = nb[mate[nb] < 0]
            free = free[free != v]
            if len(free) == 0:
                continue
            sv = nb
            best, bc = -1, -1
            for u in free[:8]:
                c = np.intersect1d(sv, ix[ip[u]:ip[u + 1]], assume_unique=True).size
                if c > bc:
                    best, bc = u, c
            mate[v] = best
            mate[best] = v
        out = np.empty(n, dtype=np.int64)
        seen = np.zeros(n, dtype=bool)
        w = 0
        singles = []
        for v in rcm:
            if seen[v]:
                continue
            seen[v] = True
            if mate[v] >= 0:
                out[w] = v; out[w + 1] = mate[v]; seen[mate[v]] = True; w += 2
            else:
                singles.append(v)
        out[w:] = np.array(singles, dtype=np.int64)
        o = out
        print('pairs: %d of %d vertices matched with a neighbour' % (int((mate >= 0).sum()), n), flush=True)
    else:
        raise SystemExit('unknown order ' + name)
    return o.astype(np.int32), time.perf_counter() - t0


for name in args.orders.split(','):
    order, t_order = make_order(name)
    dtype = np.float64 if args.dtype == 'f64' else np.float32
    m = gl.ssl.poisson(W, solver='gradient_descent', min_iter=T, max_iter=T, use_cuda=(args.dtype == 'f32'))
    n_ = W.shape[0]
    P, deg, dinv = gl.ssl._poisson_operator_symmetric(W)
    dev = _hip.DeviceGraph(P, dtype=dtype, order=order)
    src, k = gl.ssl._poisson_source(n, train_ind, labels[train_ind])
    v0 = np.zeros(n); v0[train_ind] = 1; v0 /= v0.sum()
    sw = _hip.Sweep(dev, k, min_iter=T, max_iter=T, use_hipgraph=True)
    from scipy import sparse
    sw.set_problem(sparse.spdiags(dinv, 0, n, n).tocsr() * src, v0 / deg, deg, deg / np.sum(deg))
    sw.run()
    tot = 0.0
    for _ in range(args.reps):
        tot += sw.run()[1]
    us = tot * 1e3 / (args.reps * T)
    perm = dev.order()
    pos = np.empty(n, dtype=np.int64); pos[perm] = np.arange(n)
    dpos = np.abs(pos[rows] - pos[W.indices])
    # distinct 128-byte lines an XCD's share of the rows touches, relative to its own rows (lower = less L2 traffic)
    xcd = (pos[rows] * 8 // n)
    distinct = sum(len(np.unique(W.indices[xcd == x])) for x in range(8)) / n
    if args.dtype == 'f32':
        lines = pos[W.indices] // 2
        key = rows.astype(np.int64) * n + lines
        per_row = len(np.unique(key)) / len(key)
        print('   fp32: distinct 128-byte lines per stored entry within a row: %.3f' % per_row, flush=True)
    print('order %-10s: %7.1f us/sweep (%s), %5.1f%% / %5.1f%% of entries within 32768 / 262144 positions, %.2f distinct records per vertex over the 8 XCD ranges, order built in %.2f s'
          % (name, us, args.dtype, 100 * np.mean(dpos < 32768), 100 * np.mean(dpos < 262144), distinct, t_order), flush=True)
    sw.close()
    dev.close()
