import numpy as np, sys, os, hashlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphlearning_amd import _hip
labels = bench.load_labels(70000); X = bench.make_features(labels)
for i in range(3):
    ind, d = _hip.knn_bruteforce(X, 11)
st = _hip.knn_stats()
m = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g4_large_meta.json")))["config2"]
ok = hashlib.sha256(np.ascontiguousarray(ind).tobytes()).hexdigest()[:16] == m["J_sha"]
fl = 2.0 * 70000 * 70000 * st['dpa']
print('nsplit=%d tile %.2f ms (%.1f TFLOP/s, %.1f%% of 157.3) rerank %.2f ms fallback rows %d  matches cKDTree: %s' % (st['nsplit'], st['tile_ms'], fl / st['tile_ms'] / 1e9, fl / st['tile_ms'] / 1e9 / 157.3 * 100, st['rerank_ms'], st['fallback_rows'], ok))
