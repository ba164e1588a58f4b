"""The stop margins of the tolerance-mode CG on the random systems of tests/test_gpu_auto.py: for every system the margin
(glx_cg_last_stop_margin: relative distance from tol of the residual norms that decided the stop) and whether the mode's iteration
count equals the reference-order mode's.  Shows how wide ssl.AUTO_STOP_BAND must be.  Usage: python scripts/auto_margin_probe.py [chunks]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import graphlearning_amd as gl
from oracle import gl_oracle as orc
import test_gpu_auto as ta
nchunk = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rows = []
for gi in range(nchunk * 9):
    g = ta._graph(orc, gi)
    W, lab, rng = g['W'], g['lab'], g['rng']
    with np.errstate(all='ignore'):
        for t in range(8):
            ti = ta._trainset(g, rng)
            norm = str(rng.choice(['combinatorial', 'normalized']))
            tau = ta._tau_for(g, ti, float(rng.choice([0.0, 0.0, 0.0, 0.01])))
            shift = bool(rng.random() < 0.25)
            mt = gl.ssl.laplace(W, normalization=norm, tau=tau, mean_shift=shift, reduce='tree')
            me = gl.ssl.laplace(W, normalization=norm, tau=tau, mean_shift=shift, reduce='exact')
            ut = mt.fit(ti, lab[ti]); margin = mt._cache[3].last_stop_margin()
            ue = me.fit(ti, lab[ti])
            if np.all(np.isfinite(ue)):
                rows.append((margin, mt.num_iter, me.num_iter, float(np.max(np.abs(ut - ue))), gi, t))
        for t in range(2):
            ta._trainset(g, rng)
rows.sort()
diff = [r for r in rows if r[1] != r[2]]
print('%d systems; iteration counts differ in %d' % (len(rows), len(diff)))
for r in diff:
    print('  differ: margin %.3e  tree %d exact %d  max|du| %.2e  (graph %d set %d)' % r)
print('smallest margins:')
for r in rows[:12]:
    print('  margin %.3e  tree %d exact %d  max|du| %.2e  (graph %d set %d)' % r)
for band in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
    print('band %.0e: %d of %d systems handed back (%.2f %%), differing systems outside the band: %d' % (
        band, sum(r[0] < band for r in rows), len(rows), 100.0 * sum(r[0] < band for r in rows) / len(rows), sum(r[0] >= band for r in diff)))
