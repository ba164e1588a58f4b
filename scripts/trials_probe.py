"""Developer probe: where does a trial of ssl_trials spend its time (config 2 graph)?"""
import numpy as np, sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
trainsets = gl.trainsets.generate(labels, rate=np.array([[1], [2], [3], [4], [5]]), num_trials=4, seed=0) if hasattr(gl.trainsets, 'generate') else None
for solver in ['gradient_descent', 'conjugate_gradient']:
    model = gl.ssl.poisson(W, solver=solver)
    model.ssl_trials(trainsets[:2], labels, save_results=False)      # warm: operator upload, plans, graphs
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    model.ssl_trials(trainsets, labels, save_results=False)
    pr.disable()
    wall = time.perf_counter() - t0
    print('== %s: %d trials in %.3f s = %.1f ms per trial; iterations of the last call: %s' % (solver, len(trainsets), wall, wall / len(trainsets) * 1e3, model.num_iter))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
