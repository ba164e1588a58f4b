"""Round-3 experiment (closed): how many 128-row blocks of the reference-order column sums of a CG solve could be taken as ONE exact
addition (the running sum stays inside a binade, so fl(s + a) = s + RN_u(a) and the rounded addends add exactly in any order)?
Config-2 graph, Poisson system: r.r 4 % row-by-row blocks, p.Ap up to 95 % (the singular system cancels: sum 7.5e6 of magnitudes
1.3e13); Laplace system (see the variant at the end of DESIGN.md 4.3): both sums 4-13 %.  Bit equality with the plain chain was
checked on every sum (scripts/probes/cg_blocked_sums_experiment.patch holds the device implementation that was measured)."""
import numpy as np, sys, time, pickle, os
sys.path.insert(0, '/root/repo')
import bench
from oracle import gl_oracle as orc
from scipy import sparse
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = orc.knn(X, 10); n = W.shape[0]
ti = orc.trainsets_generate(labels, rate=5, seed=0); tl = labels[ti]
W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
src, _ = orc.poisson_source(n, ti, tl)
L = orc.laplacian(W, 'normalized'); D = orc.degree_matrix(W, p=-0.5)
b = D * src

def classify(a, B):
    """vectorised classification using the TRUE sequential prefix states (np.cumsum is sequential for 1-D? no: use add.accumulate which is sequential)"""
    pre = np.add.accumulate(a)                    # sequential rounding chain (ufunc.accumulate is strictly sequential)
    nb = (len(a) + B - 1) // B
    reasons = {'zero': 0, 'tie': 0, 'window': 0, 'easy': 0}
    for i in range(nb):
        blk = a[i*B:(i+1)*B]
        s = 0.0 if i == 0 else pre[i*B - 1]
        if s == 0.0 or not np.isfinite(s): reasons['zero'] += 1; continue
        m, es = np.frexp(abs(s)); e = es - 1
        t = blk * np.ldexp(1.0, 52 - e); q = np.rint(t)
        M = np.abs(q).sum(); u = np.ldexp(1.0, e - 52)
        if not (M < 2.0**52 and abs(s) - M*u > np.ldexp(1.0, e) and abs(s) + M*u < np.ldexp(1.0, e + 1)): reasons['window'] += 1; continue
        if np.any(np.abs(t - q) == 0.5): reasons['tie'] += 1; continue
        reasons['easy'] += 1
    return reasons

x = np.zeros_like(b); r = b - L @ x; p = r.copy()
rsold = np.sum(r**2, axis=0)
agg = {}
for it in range(1, 200):
    Ap = L @ p; prod = p * Ap
    alpha = rsold / np.sum(prod, axis=0)
    x += alpha * p; r -= alpha * Ap
    rr = r**2
    if it in (2, 10, 30, 60, 100, 140):
        for name, arr in (('pAp', prod[:, 3]), ('rr', rr[:, 3])):
            for B in (128, 32, 16):
                print(it, name, 'B=%d' % B, classify(np.ascontiguousarray(arr), B), 'sum %.3e  sum|.| %.3e' % (arr.sum(), np.abs(arr).sum()), flush=True)
    rsnew = np.sum(rr, axis=0); err = np.sqrt(np.sum(rsnew))
    p = r + (rsnew / rsold) * p; rsold = rsnew
    if err <= 1e-3: break
