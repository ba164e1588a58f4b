"""bench.py --gpus N (N > 1): weak-scaling run of the vertex-partitioned Poisson sweep.

Launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: one
rank per GPU, torch.distributed backend "nccl" (= RCCL over xGMI).  The global graph has
N * 70000 vertices (MNIST label vector tiled, same blob generator, k = 10): every rank builds
it identically (exact kNN with the queries sharded over the ranks and all_gathered, deterministic assembly), owns one contiguous
block of the RCM-ordered vertices (boundaries placed in the gaps between the graph's pieces, dist.cut_bounds)
and exchanges boundary vertex records once per sweep -- or nothing at all when no block has a halo.
"""
import os
import sys
import json
import time
import numpy as np


def main(args):
    import torch                       # first: libglx must bind to torch's HIP runtime (see dist.py)
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    torch.cuda.set_device(local_rank)
    # a stuck collective ends the job after 5 minutes (watchdog abort) instead of holding the GPUs
    import datetime
    dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=datetime.timedelta(minutes=5))
    import bench
    import graphlearning_amd as gl
    from graphlearning_amd import _hip, dist as gdist
    _hip.set_default_device(local_rank)        # every libglx call of this process uses this rank's GPU
    dev = torch.device('cuda', local_rank)

    n = bench.N_PER_RANK * world
    labels = bench.load_labels(n)
    X = bench.make_features(labels)
    t0 = time.perf_counter()
    ind, dst = gdist.knnsearch_distributed(X, bench.K_NN + 1, dist, local_rank)   # queries sharded by rank
    W = gl.weightmatrix.knn(None, bench.K_NN, knn_data=(ind, dst))
    t_graph = time.perf_counter() - t0
    train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
    prob = gdist.poisson_problem(W, train_ind, labels[train_ind])
    P = prob['P']
    order = gdist.locality_order(P)
    plan = gdist.RankPlan(P, order, gdist.cut_bounds(P, order, world), rank)
    ops = gdist.HipOps(plan, prob['k'], local_rank)
    sweep = gdist.DistSweep(plan, ops, dist)
    own = plan.own
    sweep.setup(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
    min_iter, max_iter = 50, 1000

    T = 0
    for _ in range(args.warmup):
        T = sweep.run(min_iter, max_iter)
    dist.barrier()
    torch.cuda.set_device(local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T = sweep.run(min_iter, max_iter)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    wall = float(dt.item())

    halo = torch.tensor([plan.n_halo, plan.n_own, int(plan.P_local.nnz)], dtype=torch.int64, device=dev)
    halos = [torch.zeros_like(halo) for _ in range(world)]
    dist.all_gather(halos, halo)
    if rank == 0:
        C = prob['k']
        nnz = int(P.nnz)
        iters = args.steps * T / wall
        abytes = bench.algorithmic_bytes(n, nnz, C, 8, 8)
        line = {
            'metric': 'Poisson iters/sec', 'value': iters * world, 'unit': 'iters/s (70000-vertex-graph equivalents)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'configs[1] scaled weakly: %d x 70000 = %d vertices, k=10 kNN graph, nnz=%d, C=%d, '
                                   'vertex-partitioned over %d GPUs (RCM order, block boundaries in the gaps between the '
                                   'graph\'s pieces), %s; value = sweeps/s of the whole graph x %d'
                                   % (world, n, nnz, C, world,
                                      'one RCCL all_to_all halo exchange per sweep' if plan.global_halo > 0 else
                                      'no halo (every rank owns whole pieces): no per-sweep exchange, RCCL all_reduce for the stop test only',
                                      world),
                       'n': n, 'nnz': nnz, 'classes': C, 'sweeps_per_step': T, 'parallelism': 'vertex-partition x%d' % world},
            'global_sweeps_per_sec': iters,
            'edges_classes_per_sec': iters * nnz * C,
            'roofline': {'bound': 'hbm', 'achieved': abytes * iters / 1e9, 'peak': bench.HBM_PEAK_GBS * world, 'unit': 'GB/s',
                         'frac': abytes * iters / 1e9 / (bench.HBM_PEAK_GBS * world), 'traffic': None,
                         'note': 'whole-job algorithmic bytes per sweep / wall time incl. halo exchange'},
            'cpu_baseline': None,
            'halo': {'rows_per_rank': [int(h[0]) for h in halos], 'owned_per_rank': [int(h[1]) for h in halos],
                     'exchanges_per_sweep': 1 if plan.global_halo > 0 else 0, 'global_halo_rows': int(plan.global_halo)},
            'graph_build_s': t_graph,
        }
        print(json.dumps(line))
    # orderly teardown: captured graph first, then the operators, then the group
    torch.cuda.synchronize()
    sweep.close()
    ops.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
