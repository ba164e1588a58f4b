"""Developer probe (library built with GLX_CXXFLAGS=-DGLX_WAVE_PROBE): when does every wavefront of the LAST sweep / SpMM launch
start, finish its chunk loop and finish its stores?  Prints the distribution by slice class (segments per row S, chunks)."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip

which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
lib = _hip.load()
if which == 'c2':
    labels = bench.load_labels(70000); X = bench.make_features(labels)
    W = gl.weightmatrix.knn(X, 10)
    ti = gl.trainsets.generate(labels, rate=1, seed=0)
    m = gl.ssl.poisson(W, solver='gradient_descent', min_iter=60, max_iter=60)
    m.fit(ti, labels[ti]); m.fit(ti, labels[ti])
else:
    lab3, X3 = bench.config3_data()
    W3 = gl.weightmatrix.knn(X3, 20)
    ti3 = gl.trainsets.generate(lab3, rate=10, seed=0)
    m = gl.ssl.laplace(W3, reduce='tree', tol=1e-300)          # never converges: the last launch is a full one
    m.fit(ti3, lab3[ti3])
N = 1 << 20
buf = np.zeros(N, dtype=np.uint64)
lib.glx_debug_wave_probe.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.glx_debug_wave_probe(buf.ctypes.data, N) == 0
q = buf.reshape(-1, 4)
q = q[q[:, 0] > 0]
t0 = q[:, 0].min()
start = (q[:, 0] - t0) * 10e-3; loop_end = (q[:, 1] - t0) * 10e-3; end = (q[:, 3] - t0) * 10e-3     # 100 MHz counter -> us
S = (q[:, 2] & 0xffffffff).astype(int); nch = (q[:, 2] >> 32).astype(int)
print('%d wavefronts with work; launch span %.2f us (first start -> last store)' % (len(q), end.max()))
print('starts: median %.2f us, max %.2f us' % (np.median(start), start.max()))
for s in sorted(set(S)):
    mk = S == s
    print('S=%2d: %5d slices, chunks %d..%d (mean %.1f); chunk loop takes median %.2f us / max %.2f us; loop ends at median %.2f / max %.2f us; stores done at max %.2f us'
          % (s, mk.sum(), nch[mk].min(), nch[mk].max(), nch[mk].mean(), np.median((loop_end - start)[mk]), (loop_end - start)[mk].max(),
             np.median(loop_end[mk]), loop_end[mk].max(), end[mk].max()))
order = np.argsort(-loop_end)[:8]
for i in order:
    print('   late wavefront: S=%d chunks=%d start %.2f loop end %.2f end %.2f us' % (S[i], nch[i], start[i], loop_end[i], end[i]))
