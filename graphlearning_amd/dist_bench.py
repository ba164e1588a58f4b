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
    min_iter, max_iter = 50, 1000
    engine = os.environ.get('GLX_DIST_ENGINE', 'glx')     # 'glx': library-owned RCCL communicator + captured sweeps; 'torch': round-1 path
    comm = None
    if engine == 'glx':
        try:
            comm = gdist.init_comm(dist, local_rank)
        except Exception as exc:                            # noqa: BLE001 -- fall back rather than lose the measurement
            if rank == 0:
                print('glx communicator unavailable (%s); using the torch.distributed engine' % (exc,), file=sys.stderr)
            engine = 'torch'
    flag = torch.tensor([1 if engine == 'glx' else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        engine = 'torch'

    def measure(partition):
        bounds = gdist.cut_bounds(P, order, world) if partition == 'cut' else gdist.block_bounds(P.shape[0], world)
        plan = gdist.RankPlan(P, order, bounds, rank)
        own = plan.own
        if engine == 'glx':
            ds = gdist.glx_dist_sweep(comm, plan, prob['k'], force_exchange=gdist._force_collectives())
            ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
            run = lambda: ds.run(min_iter, max_iter, 8, 0.0)[0]
            close = ds.close
        else:
            ops = gdist.HipOps(plan, prob['k'], local_rank)
            sweep = gdist.DistSweep(plan, ops, dist)
            sweep.setup(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
            run = lambda: sweep.run(min_iter, max_iter)

            def close():
                torch.cuda.synchronize()
                sweep.close()
                ops.close()
        T = 0
        for _ in range(args.warmup):
            T = run()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            T = run()
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        halo = torch.tensor([plan.n_halo, plan.n_own, int(plan.P_local.nnz)], dtype=torch.int64, device=dev)
        halos = [torch.zeros_like(halo) for _ in range(world)]
        dist.all_gather(halos, halo)
        close()
        return dict(T=T, wall=float(dt.item()), halo_rows=[int(h[0]) for h in halos], owned=[int(h[1]) for h in halos],
                    global_halo=int(plan.global_halo))

    res = measure('cut')
    even = measure('even') if world > 1 else None      # equal blocks: a non-zero halo, i.e. the RCCL exchange in every sweep
    if rank == 0:
        C = prob['k']
        nnz = int(P.nnz)
        T = res['T']
        iters = args.steps * T / res['wall']
        abytes = bench.algorithmic_bytes(n, nnz, C, 8, 8)
        line = {
            'metric': 'Poisson iters/sec', 'value': iters * world, 'unit': 'iters/s (70000-vertex-graph equivalents)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['wall'] / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'configs[1] scaled weakly: %d x 70000 = %d vertices, k=10 kNN graph, nnz=%d, C=%d, '
                                   'vertex-partitioned over %d GPUs (RCM order, block boundaries in the gaps between the '
                                   'graph\'s pieces), %s; value = sweeps/s of the whole graph x %d'
                                   % (world, n, nnz, C, world,
                                      'one RCCL all-to-all-v halo exchange per sweep' if res['global_halo'] > 0 else
                                      'no halo (every rank owns whole pieces): no per-sweep exchange, one RCCL all-reduce for the stop test',
                                      world),
                       'n': n, 'nnz': nnz, 'classes': C, 'sweeps_per_step': T, 'parallelism': 'vertex-partition x%d' % world,
                       'engine': ('libglx communicator (grouped ncclSend/ncclRecv + captured device graphs)' if engine == 'glx'
                                  else 'torch.distributed all_to_all_single (eager)')},
            'global_sweeps_per_sec': iters,
            'edges_classes_per_sec': iters * nnz * C,
            'roofline': {'bound': 'hbm', 'achieved': abytes * iters / 1e9, 'peak': bench.HBM_PEAK_GBS * world, 'unit': 'GB/s',
                         'frac': abytes * iters / 1e9 / (bench.HBM_PEAK_GBS * world), 'traffic': None,
                         'note': 'whole-job algorithmic bytes per sweep / wall time incl. halo exchange'},
            'cpu_baseline': None,
            'halo': {'rows_per_rank': res['halo_rows'], 'owned_per_rank': res['owned'],
                     'exchanges_per_sweep': 1 if res['global_halo'] > 0 else 0, 'global_halo_rows': res['global_halo']},
            'graph_build_s': t_graph,
        }
        if even is not None:
            it_e = args.steps * even['T'] / even['wall']
            line['partition_even'] = {'value': it_e * world, 'global_sweeps_per_sec': it_e, 'ms_per_step': even['wall'] / args.steps * 1e3,
                                      'halo_rows_per_rank': even['halo_rows'], 'owned_per_rank': even['owned'],
                                      'note': 'equal blocks of the same order: every rank imports a halo, so every sweep carries the '
                                              'RCCL exchange (the headline partition above places the cuts between the graph\'s pieces)'}
        print(json.dumps(line))
    if comm is not None:
        comm.close()
    # orderly teardown: sweeps and communicator are closed above, then the group
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
