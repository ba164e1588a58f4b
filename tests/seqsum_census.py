"""Developer tool (round 6; lives under tests/ because it drives the oracle): a census of the block form of numpy's reduction chains on the
REFERENCE's own products.  Runs the oracle's conjugate gradient on config 3 (ssl.laplace, 60 000 x 10) or config 2 (ssl.poisson's default
solver, 70 000 x 10), keeps p*Ap and r*r of every iteration, and counts per column walk how the host restatement of the device's
quantising pass (tests/seqsum_host.cpp: ss_host_census over csrc/seqsum_exact.h) classifies the 256-row blocks: plain integer blocks,
blocks through a record with splits, blocks that go row by row -- and, with an iteration number, why.
    python tests/seqsum_census.py c3|c2 [iteration to explain]
Round 6: config 3 per column walk 220 plain / 12 by record / 1.3 (r.r) - 2.8 (p.Ap) row by row; config 2's p.Ap 117 / 60 / 94."""
import ctypes, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import gl_oracle as orc
from scipy import sparse


def products(which):
    if which == 'c3':
        lab, X = bench.config3_data()
        W = orc.knn(X, 20)
        ti = orc.trainsets_generate(lab, rate=10, seed=0)
        A, b, M, idx, F, k = orc.laplace_system(W, ti, lab[ti])
        tol, n, scatter = 1e-5, W.shape[0], np.flatnonzero(idx)
    else:
        lab = bench.load_labels(70000)
        X = bench.make_features(lab)
        W = orc.knn(X, 10)
        ti = orc.trainsets_generate(lab, rate=1, seed=0)
        n = W.shape[0]
        W = sparse.csr_matrix(W); W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
        src, _ = orc.poisson_source(n, ti, lab[ti])
        A = orc.laplacian(W, 'normalized'); D = orc.degree_matrix(W, p=-0.5)
        b, tol, scatter = D * src, 1e-3, np.arange(n)
    def full(a):                                       # the device works on all n rows: Dirichlet rows are zeros in the chain
        f = np.zeros((n, b.shape[1]))
        f[scatter] = a
        return f
    x = np.zeros_like(b); r = b - A @ x; p = r.copy(); rsold = np.sum(r ** 2, axis=0)
    it, err = 0, 1
    while err > tol and it < 1e5:                      # utils.conjgrad (reference utils.py:483-532), the oracle's loop with the products kept
        it += 1
        Ap = A @ p
        pAp = p * Ap
        yield 'p.Ap', it, full(pAp)
        alpha = rsold / np.sum(pAp, axis=0)
        x += alpha * p; r -= alpha * Ap
        rr = r ** 2
        yield 'r.r', it, full(rr)
        rsnew = np.sum(rr, axis=0); err = np.sqrt(np.sum(rsnew)); p = r + (rsnew / rsold) * p; rsold = rsnew


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    explain = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    lib = os.path.join(tempfile.mkdtemp(), 'libss_census.so')
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', lib, os.path.join(ROOT, 'tests', 'seqsum_host.cpp')], check=True)
    L = ctypes.CDLL(lib)
    tot, walks = {}, {}
    for kind, it, d in products(which):
        d = np.ascontiguousarray(d)
        n, C = d.shape
        row = np.zeros(8, dtype=np.int64)
        for c in range(C):
            out = (ctypes.c_int64 * 8)()
            v = 1 if it == explain and c < 2 else 0
            if v:
                print(kind, 'iteration', it, 'column', c, flush=True)
            L.ss_host_census(ctypes.c_void_p(d.ctypes.data + 8 * c), ctypes.c_int64(n), ctypes.c_int64(C), out, v)
            row += np.array(list(out))
        tot[kind] = tot.get(kind, 0) + row
        walks[kind] = walks.get(kind, 0) + C
        print('%-5s iteration %3d: plain %5d | empty %4d | by record %4d | record refused %3d | row by row %4d | plain refused %3d' % ((kind, it) + tuple(row[:6])), flush=True)
    for kind in tot:
        print(kind, 'per column walk: plain %.1f, by record %.1f, row by row %.1f' % (tot[kind][0] / walks[kind], tot[kind][2] / walks[kind],
                                                                                  (tot[kind][3] + tot[kind][4] + tot[kind][5]) / walks[kind]))


if __name__ == '__main__':
    main()
