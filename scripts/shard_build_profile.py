"""cProfile of ShardedGraph (one rank) at n rows: where the host seconds of 'symmetrise + plan' go."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from graphlearning_amd import _hip, dist_build
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3000000
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
g = np.random.default_rng(2)
lab = g.integers(0, 10, size=n); X = (g.normal(size=(10, 64)) * 4)[lab] + g.normal(size=(n, 64))
perm, starts = dist_build.coarse_locality_order(X, ncells=64, seed=0, return_cells=True)
X = np.ascontiguousarray(X[perm])
J, D = _hip.knn_bruteforce(X, 11, cell_starts=starts, query_range=(0, n))
J, D = np.ascontiguousarray(J, dtype=np.int64), np.ascontiguousarray(D)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
sg = dist_build.ShardedGraph(dist, n, J, D, 10, device=torch.device('cuda', 0))
pr.disable()
print('ShardedGraph at n = %d: %.2f s' % (n, time.perf_counter() - t0))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
dist.destroy_process_group()
