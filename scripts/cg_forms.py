"""A/B of the two forms of the reference-order reduction chains (GLX_CG_CHAIN / GLX_CG_BLOCKS, include/glx.h) on the bench's
configurations: ssl.poisson's default CG on config 2 (singular system, cancelling products) and ssl.laplace(reduce='exact') on
config 3.  Prints fit times, per-iteration times, block statistics and whether the iterates are the same bits.
    python scripts/cg_forms.py [reps] [forms, e.g. blocks or chain,blocks]"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import graphlearning_amd as gl  # noqa: E402
from graphlearning_amd import _hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
forms = tuple(sys.argv[2].split(',')) if len(sys.argv) > 2 else ('chain', 'auto', 'blocks', 'chain', 'auto')
_hip.CG_EXACT_EAGER = len(sys.argv) > 3 and sys.argv[3] == 'eager'      # third argument `eager`: launch by launch instead of captured chunks


def timed(model, ti, lab, dev_of):
    out = {}
    for form in forms:
        _hip.CG_EXACT_FORM = None if form == 'auto' else form      # 'auto': the library's choice, re-decided per kind of reduction during the solve
        u = model.fit(ti, lab).copy()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            model.fit(ti, lab)
            ts.append(time.perf_counter() - t0)
        out.setdefault(form, []).append((float(np.median(ts)) * 1e3, int(model.num_iter), dev_of(model).last_block_stats() + (dev_of(model).last_block_forms(),), u))
    _hip.CG_EXACT_FORM = None
    return out


def report(name, res):
    uc = res[forms[0]][0][3]
    for form in dict.fromkeys(forms):
        for ms, its, st, u in res[form]:
            print('%s | %-6s: fit %.2f ms, %d iterations, %.1f us per iteration, blocks (plain, by record, row by row, kinds still in block form at the end) %s, same bits as chain: %s'
                  % (name, form, ms, its, ms * 1e3 / its, st, np.array_equal(u, uc, equal_nan=True)), flush=True)


labels2 = bench.load_labels(70000)
X2 = bench.make_features(labels2)
W2 = gl.weightmatrix.knn(X2, 10)
ti2 = gl.trainsets.generate(labels2, rate=1, seed=0)
report('config 2 ssl.poisson CG', timed(gl.ssl.poisson(W2), ti2, labels2[ti2], lambda m: m._cache[1]))
lab3, X3 = bench.config3_data()
W3 = gl.weightmatrix.knn(X3, 20)
ti3 = gl.trainsets.generate(lab3, rate=10, seed=0)
report('config 3 ssl.laplace exact', timed(gl.ssl.laplace(W3, reduce='exact'), ti3, lab3[ti3], lambda m: m._cache[3]))
