"""Poisson learning across the GPUs of one node (no counterpart in the reference, which is single-process):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 examples/ssl_multi_gpu.py

One process per GPU.  Every rank searches the neighbours of its own block of points, the ranks symmetrise the graph by
owner (one all-to-all of edges), each plans its halo from the rows it owns, and the library runs all sweeps of
ssl.poisson(solver='gradient_descent') with the RCCL halo exchange inside (glx_dist_sweep).  The result is bit-identical to
the single-GPU fit for any number of ranks.  Runs with one rank too (plain `python examples/ssl_multi_gpu.py`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

if 'RANK' not in os.environ:
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29511')
local = int(os.environ['LOCAL_RANK'])
torch.cuda.set_device(local)
dist.init_process_group('nccl', device_id=torch.device('cuda', local))
import graphlearning_amd as gl
from graphlearning_amd import _hip, dist as gdist, dist_build
_hip.set_default_device(local)
rank, world = dist.get_rank(), dist.get_world_size()

n, d, k = 200000, 32, 10
rng = np.random.default_rng(0)                       # every rank draws the same points (the search needs them all)
labels = rng.integers(0, 10, size=n)
X = rng.normal(size=(10, d))[labels] * 3.0 + rng.normal(size=(n, d))
train_ind = gl.trainsets.generate(labels, rate=5, seed=0)

lo, hi = (int(b) for b in gdist.block_bounds(n, world)[rank:rank + 2])
t0 = time.perf_counter()
J, D = _hip.knn_bruteforce(X, k + 1, query_range=(lo, hi))            # this rank's queries only
u, T, graph = dist_build.poisson_fit_sharded(dist, n, J, D, k, train_ind, labels[train_ind], device=torch.device('cuda', local))
t1 = time.perf_counter()
if rank == 0:
    acc = gl.ssl.ssl_accuracy(np.argmax(u, axis=1), labels, train_ind)
    print('%d ranks: n=%d, rank 0 owns %d rows with %d halo rows; T=%d sweeps; accuracy %.2f%%; %.2f s end to end'
          % (world, n, graph.plan.n_own, graph.plan.n_halo, T, acc, t1 - t0))
dist.barrier()
dist.destroy_process_group()
