"""Stacked gradient-descent trials at config 2 (bench.trials_gd_block): per-launch time, step wall time and the fraction of the HBM
roofline on SURVEY 8d's bytes for B = 2 .. 16 training sets per sweep.  Usage: python scripts/groups_probe.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
labels = bench.load_labels(bench.N_PER_RANK)
X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, bench.K_NN)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
sync = lambda: _hip.check(_hip.load().glx_device_synchronize(), 'sync')
scan = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (2, 3, 4, 5, 6, 8)
blk = bench.trials_gd_block(W, labels, ti, sync, scan_B=scan)
for e in blk['scan']:
    print('B=%2d  launch %.2f us  step %.3f ms  per trial-sweep %.2f us  frac %.3f  edges*classes/s %.3e  T=%s' % (
        e['B'], e['avg_launch_us'], e['step_ms_wall'], e['step_ms_wall'] * 1e3 / (e['B'] * e['sweeps_per_step']), e['roofline']['frac'],
        e['edges_classes_per_s'], sorted(set(e['T']))))
print('parity', blk.get('parity'))
