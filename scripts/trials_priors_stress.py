"""Developer probe (round 6): the volume projection behind ssl_trials with class priors, batched and one by one, over and over --
does any repetition differ from the first?  (One run of the suite under GLX_TEST_ABLATE=poison127,nopool with six workers saw two rows of
tests/test_gpu_groups.py::test_ssl_trials_gd_file_equals_sequential differ.)  Usage: trials_priors_stress.py REPS [ablations]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import contextlib, io
import graphlearning_amd as gl
from graphlearning_amd import _hip, ssl as glssl
from conftest import blobs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for a in (sys.argv[2].split(',') if len(sys.argv) > 2 else []):
    if a == 'nopool': _hip.pool_set_enabled(False)
    if a.startswith('poison'): _hip.pool_set_poison(int(a[6:]))
X, labels = blobs(2500, 12, 5, 9, 1.6)
W = gl.weightmatrix.knn(X, 8)
trainsets = gl.trainsets.generate(labels, rate=np.array([[1], [2], [4]]), num_trials=5, seed=3)
tmp = tempfile.mkdtemp()
glssl.results_dir = tmp
first = {}
bad = 0
for rep in range(reps):
    for tag, batched in (('b_', True), ('s_', False)):
        model = gl.ssl.poisson(W, class_priors=gl.utils.class_priors(labels), solver='gradient_descent')
        if not batched:
            model._trial_batch_size = lambda labels: 1
        with contextlib.redirect_stdout(io.StringIO()):
            model.ssl_trials(trainsets, labels, tag=tag, overwrite=True)
        txt = open(os.path.join(tmp, tag + model.get_accuracy_filename())).read()
        if not first:
            first['x'] = txt
            print(txt, flush=True)
        elif txt != first['x']:
            bad += 1
            a, b = first['x'].splitlines(), txt.splitlines()
            print('rep %d %s differs:' % (rep, tag), [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y], flush=True)
print('pid %d: %d repetitions x (batched, one by one), %d differ from the first' % (os.getpid(), reps, bad))
