"""bench.py --gpus N (N > 1): the vertex-partitioned Poisson sweep on N GPUs -- by default STRONG scaling of the
70 000-vertex configs[1] graph (BASELINE.json's metric: "MNIST k=10 graph @1/2/4/8 GPU").

Launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: one
rank per GPU, torch.distributed backend "nccl" (= RCCL over xGMI).  Every rank builds the graph
identically (exact kNN with the queries sharded over the ranks and all_gathered, deterministic assembly), owns one contiguous
block of the RCM-ordered vertices -- equal blocks (`even`, the headline: every block has a halo) or blocks placed in the gaps
between the graph's pieces (`cut`, measured beside it) -- and exchanges boundary vertex records once per sweep.
--scaling weak: N * 70000 vertices (MNIST label vector tiled, same blob generator), 70 000 per rank.
"""
import os
import sys
import json
import time
import numpy as np


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner with C stdio when its first communicator comes up
    (torch's and the library's alike), flushed whenever: file descriptor 1 is pointed at stderr for the life of the process and
    the returned function writes the line to the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real, (line + '\n').encode())
    return emit


def check_world(args, n_devices=None):
    """The ranks refuse to run when the job they find is not the job the command line asks for: `--gpus N` must meet
    WORLD_SIZE = N (a plain `python bench.py --gpus 8` is turned into 8 ranks by bench.spawn_ranks before it gets here),
    and, on a GPU node, N visible devices.  Returns (rank, world, local_rank) or exits with status 2."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world != int(args.gpus):
        print('bench.py: --gpus %d but WORLD_SIZE = %d: launch with `python bench.py --gpus %d` (it starts the ranks itself) or '
              '`python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d`'
              % (args.gpus, world, args.gpus, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if n_devices is not None and n_devices < world:
        print('bench.py: %d ranks but only %d GPUs are visible (one rank per GPU)' % (world, n_devices), file=sys.stderr)
        sys.exit(2)
    return rank, world, local_rank


def main_dry_run(args):
    """bench.py --gpus N --dist-dry-run: the launch path without a GPU -- the ranks rendezvous over gloo, count
    themselves with an all_reduce and rank 0 prints one line.  tests/test_dist_gloo.py uses it to prove that a plain
    `python bench.py --gpus 2` really runs two ranks."""
    emit = _claim_stdout()
    import datetime
    import torch
    import torch.distributed as dist
    rank, world, local_rank = check_world(args)
    if 'MASTER_ADDR' not in os.environ:         # --gpus 1 without a launcher
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    # failure injection for the supervisor's tests (bench.supervise_rank): GLX_BENCH_TEST_FAIL = "<attempt>:<exit|hang>" makes the LAST
    # rank of that attempt die or hang before the rendezvous
    inject = os.environ.get('GLX_BENCH_TEST_FAIL', '')
    if inject and rank == world - 1 and inject.split(':')[0] == os.environ.get('GLX_BENCH_ATTEMPT', '0'):
        if inject.endswith('exit'):
            sys.exit(7)
        time.sleep(3600)
    dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=2))
    seen = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(seen)
    ids = [None] * world
    dist.all_gather_object(ids, (rank, local_rank, os.getpid()))
    if rank == 0:
        emit(json.dumps({'dry_run': True, 'n_gpus': world, 'ranks_seen': int(seen.item()), 'gpus_asked': int(args.gpus),
                         'ranks': [list(t) for t in ids], 'engine': getattr(args, 'engine', None),
                         'spawned_by_bench': bool(getattr(args, 'spawned', False) or os.environ.get('GLX_BENCH_SPAWNED') == '1')}))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def _load_test_ops(spec):
    """--test-ops path/to/module.py:Class (hidden; tests/test_dist_gloo.py): the rank-local sweep stand-in of the CPU tests, so that the
    launch path, the planner, the exchange and the line's schema run on a machine without a GPU.  Never a product path."""
    import importlib.util
    path, cls = spec.rsplit(':', 1)
    sp = importlib.util.spec_from_file_location('glx_bench_test_ops', path)
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    return getattr(mod, cls)


def main(args):
    """bench.py --gpus N: the STATED multi-GPU metric (BASELINE.json: "Poisson iters/sec ..., MNIST k=10 graph @1/2/4/8 GPU") --
    STRONG scaling of the 70 000-vertex configs[1] graph: the same graph as the N = 1 line, vertex-partitioned into N equal blocks of a
    locality order (`even`: every block cuts through clusters, so every sweep carries the halo exchange), value = sweeps per second of THAT
    graph.  Beside it: the `cut` partition (blocks between the graph's pieces; may need no exchange at all) and the `connected` workload
    (one component).  --scaling weak: every rank owns 70 000 vertices of an N x 70 000 graph (the line of rounds 1-5)."""
    emit = _claim_stdout()
    test_ops = getattr(args, 'test_ops', None)
    import torch                       # first: libglx must bind to torch's HIP runtime (see dist.py)
    import torch.distributed as dist
    rank, world, local_rank = check_world(args, None if test_ops else torch.cuda.device_count())
    if 'MASTER_ADDR' not in os.environ:         # python bench.py --gpus 1 --force-dist: a one-rank job without a launcher
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import datetime
    import bench
    from graphlearning_amd import dist as gdist
    if test_ops:
        dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=5))
        dev = torch.device('cpu')
        sync = lambda: None
        gl = _hip = None
    else:
        torch.cuda.set_device(local_rank)
        # a stuck collective ends the job after 5 minutes (watchdog abort) instead of holding the GPUs
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=datetime.timedelta(minutes=5))
        import graphlearning_amd as gl
        from graphlearning_amd import _hip
        _hip.set_default_device(local_rank)        # every libglx call of this process uses this rank's GPU
        dev = torch.device('cuda', local_rank)
        sync = torch.cuda.synchronize
    strong = getattr(args, 'scaling', 'strong') == 'strong'
    min_iter, max_iter = 50, 1000

    def build(workload):
        """The graph of one workload, built identically on every rank (exact kNN with the queries sharded over the ranks and
        all-gathered, deterministic device assembly), its Poisson problem and a locality order of the vertices."""
        t0 = time.perf_counter()
        if test_ops:                   # (CPU tests hand the graph in: building one needs the GPU)
            g = np.load(args.test_graph)
            from scipy import sparse
            W = sparse.csr_matrix((g['data'], g['indices'], g['indptr']), shape=(len(g['indptr']) - 1,) * 2)
            labels, train_ind = g['labels'], g['train_ind']
        else:
            n = bench.N_PER_RANK * (1 if strong else world)
            labels = bench.load_labels(n)
            X = bench.make_features(labels, scale=0.8 if workload == 'connected' else 2.0)
            ind, dst = gdist.knnsearch_distributed(X, bench.K_NN + 1, dist, local_rank)   # queries sharded by rank
            W = gl.weightmatrix.knn(None, bench.K_NN, knn_data=(ind, dst))
            train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
        prob = gdist.poisson_problem(W, train_ind, labels[train_ind])
        order = gdist.locality_order(prob['P'])
        return dict(W=W, labels=labels, train_ind=train_ind, prob=prob, P=prob['P'], order=order, workload=workload,
                    n=W.shape[0], nnz=int(prob['P'].nnz), graph_build_s=time.perf_counter() - t0)

    engine = 'test' if test_ops else getattr(args, 'engine', 'glx')   # 'glx': library-owned RCCL communicator + captured sweeps; 'torch': torch.distributed collectives
    gdist.FORCE_COLLECTIVES = bool(getattr(args, 'force_collectives', False))
    comm = None
    if engine == 'glx':
        # the library's own RCCL communicator; an error OR a rendezvous that does not come back within two minutes must not
        # cost the measurement: the ranks then agree (below) on the torch.distributed engine
        import threading
        box = {}

        def _init():
            try:
                box['comm'] = gdist.init_comm(dist, local_rank)
            except Exception as exc:                        # noqa: BLE001
                box['err'] = exc
        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(120.0)
        comm = box.get('comm')
        if comm is None:
            print('rank %d: glx communicator unavailable (%s); using the torch.distributed engine'
                  % (rank, box.get('err', 'no answer from ncclCommInitRank after 120 s')), file=sys.stderr)
            engine = 'torch'
    if not test_ops:
        flag = torch.tensor([1 if engine == 'glx' else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            engine = 'torch'
            comm = None                                         # (a communicator some ranks did get is left alone)
    N_BATCHES = 5

    def measure(G, partition, want_u=False):
        P, prob = G['P'], G['prob']
        order_p, bounds, pinfo = gdist.plan_partition(P, G['order'], world, partition, dist)      # (every rank plans for itself; the ranks compare digests)
        plan = gdist.RankPlan(P, order_p, bounds, rank)
        own = plan.own
        parts = None
        if engine == 'glx':
            ds = gdist.glx_dist_sweep(comm, plan, prob['k'], force_exchange=gdist._force_collectives())
            ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
            run = lambda: ds.run(min_iter, max_iter, 8, 0.0)[0]
            close = ds.close
            info = ds.info
            fetch = ds.fetch
        else:
            info = lambda: dict(exchange='eager' if ((world > 1 and plan.global_halo > 0) or gdist._force_collectives()) else 'none',
                                selftest='not run', overlap=True, fused=False, scatter=False)
            ops = _load_test_ops(test_ops)(plan, prob['k']) if test_ops else gdist.HipOps(plan, prob['k'], local_rank)
            sweep = gdist.DistSweep(plan, ops, dist)
            sweep.setup(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
            run = lambda: sweep.run(min_iter, max_iter)
            fetch = sweep.result_own

            def close():
                sync()
                sweep.close()
                if hasattr(ops, 'close'):
                    ops.close()
        T = 0
        for _ in range(args.warmup):
            T = run()
        # batches of EXACTLY --steps steps, each bracketed by a barrier + device synchronisation on both sides, the time of a batch = the
        # MAX over the ranks; the reported batch is the median one (the rule of the N = 1 line)
        walls = []
        for _ in range(N_BATCHES):
            dist.barrier()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                T = run()
            sync()
            dist.barrier()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            walls.append(float(dt.item()))
        walls.sort()
        halo = torch.tensor([plan.n_halo, plan.n_own, int(plan.P_local.nnz), plan.n_boundary], dtype=torch.int64, device=dev)
        halos = [torch.zeros_like(halo) for _ in range(world)]
        dist.all_gather(halos, halo)
        how = info()
        u = None
        if want_u:                                 # the iterate of the last step, every rank's rows, for the parity check on rank 0
            got = [None] * world
            dist.all_gather_object(got, (own, fetch()))
            if rank == 0:
                u = np.zeros((G['n'], prob['k']), dtype=got[0][1].dtype)
                for ids, block in got:
                    u[ids] = block
        if engine == 'glx':                        # (behind the fetch: the timed pieces run on the sweep's own state buffers)
            try:
                parts = ds.time_parts(20)          # this rank's pieces of one sweep, each timed alone (microseconds)
            except Exception:                      # noqa: BLE001
                parts = None
        close()
        return dict(T=T, wall=walls[len(walls) // 2], wall_min=walls[0], wall_max=walls[-1], halo_rows=[int(h[0]) for h in halos],
                    owned=[int(h[1]) for h in halos], boundary=[int(h[3]) for h in halos], global_halo=int(plan.global_halo), how=how,
                    parts_rank0=parts, u=u,
                    planner={k: (v if not hasattr(v, 'item') else v.item()) for k, v in pinfo.items()})

    def measure_or_fall_back(G, partition, want_u=False):
        # an error every rank sees alike (an RCCL call refused, a capture the runtime rejects) must not cost the measurement:
        # the ranks agree on it and repeat the run with the torch.distributed engine
        nonlocal engine
        if engine != 'glx':
            return measure(G, partition, want_u)
        out, ok = None, 1
        try:
            out = measure(G, partition, want_u)
        except Exception as exc:                                # noqa: BLE001
            ok = 0
            print('rank %d: libglx distributed sweep failed (%s)' % (rank, exc), file=sys.stderr)
        agreed = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 1:
            return out
        engine = 'torch'
        if rank == 0:
            print('falling back to the torch.distributed engine', file=sys.stderr)
        return measure(G, partition, want_u)

    def side(G, partition):
        """A measurement BESIDE the headline: whatever goes wrong in it (on every rank alike) is recorded in its entry, the line is
        printed all the same."""
        ok, out = 1, None
        try:
            out = measure_or_fall_back(G, partition)
        except Exception as exc:                                # noqa: BLE001
            ok, out = 0, dict(error=repr(exc))
        agreed = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        return out if int(agreed.item()) == 1 else dict(error=(out or {}).get('error', 'failed on another rank'))

    workload = getattr(args, 'workload', 'blobs')
    G = build(workload)
    # the headline partition.  strong scaling (default): `even` -- equal blocks of the locality order; they cut through clusters, so every rank
    # imports a halo and EVERY sweep carries the exchange (asserted below).  --partition cut / cells / auto are measured beside it.
    headline = getattr(args, 'partition', None) or ('even' if strong else 'cut')
    res = measure_or_fall_back(G, headline, want_u=True)
    exchanges_per_sweep = 1 if (res['global_halo'] > 0 and world > 1) or (gdist._force_collectives() and res['how']['exchange'] != 'none') else 0
    if strong and world > 1 and exchanges_per_sweep < 1:
        print('bench.py: the headline partition `%s` has no halo on %d ranks: the stated metric is a sweep WITH its halo exchange' % (headline, world),
              file=sys.stderr)
        sys.exit(4)
    sides = {}
    if world > 1 and not getattr(args, 'no_sides', False):
        for alt in ('cut', 'even'):
            if alt != headline and res['planner'].get('partition') != alt:
                sides['partition_' + alt] = (G, side(G, alt))
        if workload != 'connected' and not test_ops:
            Gc = build('connected')
            sides['workload_connected'] = (Gc, side(Gc, 'even'))
    # The verdict before the line is built: the gathered iterate against the one-process oracle (rank 0).  A library exchange that gives ANOTHER
    # iterate -- no RCCL peer has ever existed for it -- must not become the line: the ranks agree on the verdict and the measurement is
    # repeated with the torch.distributed engine, the line says so (`parity_refit`).  GLX_BENCH_TEST_FAIL=parity (a harness variable of
    # tests/test_gpu_dist.py) takes the verdict of the first measurement as "differs".
    cpu = parity = T_ref = None
    parity_refit = None
    if not test_ops:
        if rank == 0:
            cpu, parity, T_ref = bench.cpu_baseline(G['W'], G['train_ind'], G['labels'][G['train_ind']], res['u'], res['T'])
        verdict = torch.tensor([0 if (rank == 0 and (not parity or os.environ.get('GLX_BENCH_TEST_FAIL') == 'parity')) else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        if int(verdict.item()) == 0 and engine == 'glx':
            engine = 'torch'
            if rank == 0:
                print('the library\'s own exchange gave an iterate that differs from the oracle: measuring again with the torch.distributed engine',
                      file=sys.stderr)
            first_value = args.steps * res['T'] / res['wall']
            res = measure(G, headline, want_u=True)
            if rank == 0:
                cpu, parity, T_ref = bench.cpu_baseline(G['W'], G['train_ind'], G['labels'][G['train_ind']], res['u'], res['T'])
                parity_refit = {'reason': 'the iterate of the libglx exchange engine differed from the oracle', 'discarded_sweeps_per_sec': first_value,
                                'measured_again_with': 'torch.distributed all_to_all_single (eager)'}
    if rank == 0:
        prob, n, nnz, W = G['prob'], G['n'], G['nnz'], G['W']
        C = prob['k']
        T = res['T']
        iters = args.steps * T / res['wall']
        abytes = bench.algorithmic_bytes(n, nnz, C, 8, 8)
        have_rccl = comm is not None and comm.has_transport()
        what = ('configs[1]: MNIST-shaped k=10 kNN graph, n=%d, nnz=%d, C=%d, ssl.poisson gradient_descent (T=%d sweeps per step, stop test '
                'included)' % (n, nnz, C, T)) if strong else (
                'configs[1] scaled weakly: %d x 70000 = %d vertices, k=10 kNN graph, nnz=%d, C=%d' % (world, n, nnz, C))
        how = ('STRONG scaling: the one graph vertex-partitioned over %d GPUs (%s blocks of a locality order)' % (world, res['planner'].get('partition', headline))
               if strong else 'vertex-partitioned over %d GPUs (partition `%s`); value = sweeps/s of the whole graph x %d' % (world, res['planner'].get('partition', headline), world))
        xch = ('one halo exchange of boundary vertex records per sweep (RCCL all-to-all-v: grouped ncclSend/ncclRecv)' if res['global_halo'] > 0 and world > 1
               else ('one rank: no peers' if world == 1 else 'no halo (every rank owns whole pieces): no per-sweep exchange'))
        value = iters if strong else iters * world
        line = {
            'metric': 'Poisson iters/sec', 'value': value, 'unit': 'iters/s' if strong else 'iters/s (70000-vertex-graph equivalents)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['wall'] / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s; %s; %s%s' % (what, how, xch, '; workload `connected` (centre scale 0.8: one component)' if workload == 'connected' else ''),
                       'n': n, 'nnz': nnz, 'classes': C, 'sweeps_per_step': T, 'parallelism': 'vertex-partition x%d' % world,
                       'engine': {'glx': 'libglx communicator (grouped ncclSend/ncclRecv + captured device graphs)',
                                  'torch': 'torch.distributed all_to_all_single (eager)', 'test': 'CPU stand-in of the tests (gloo)'}[engine]},
            'timing': {'batches': N_BATCHES, 'steps_per_batch': args.steps, 'reported': 'median batch (each batch: barrier + synchronize on both sides, max over ranks)',
                       'ms_per_step_min': res['wall_min'] / args.steps * 1e3, 'ms_per_step_max': res['wall_max'] / args.steps * 1e3},
            'global_sweeps_per_sec': iters,
            'edges_classes_per_sec': iters * nnz * C,
            'roofline': {'bound': 'hbm', 'achieved': abytes * iters / 1e9, 'peak': bench.HBM_PEAK_GBS * world, 'unit': 'GB/s',
                         'frac': abytes * iters / 1e9 / (bench.HBM_PEAK_GBS * world), 'traffic': None,
                         'algorithmic_bytes_per_sweep': abytes, 'us_per_sweep': res['wall'] / (args.steps * T) * 1e6,
                         'note': 'whole-job algorithmic bytes per sweep / wall time per sweep incl. the halo exchange, against N x 8 TB/s; the '
                                 'one-GPU sweep of this graph takes 12 us, so at N > 1 the per-sweep exchange latency is the bound, not HBM'},
            'cpu_baseline': cpu,
            'speedup_vs_cpu_baseline': (value / cpu['value']) if cpu else None,
            'parity': None if test_ops else {'bit_identical_to_oracle': parity, 'T': T, 'T_oracle': T_ref,
                                             'note': 'the iterate of the last timed step, every rank\'s rows gathered, against the one-process scipy oracle'},
            'parity_refit': parity_refit,
            'halo': {'rows_per_rank': res['halo_rows'], 'owned_per_rank': res['owned'], 'boundary_rows_per_rank': res['boundary'],
                     'exchanges_per_sweep': exchanges_per_sweep, 'global_halo_rows': res['global_halo']},
            'sweep_parts_rank0_us': res['parts_rank0'],
            'graph_build_s': G['graph_build_s'],
            # what really ran: ranks of the RCCL communicator the sweeps used (the library's own, or torch's), the engine, and
            # whether sweeps that carry the halo exchange were replayed from captured device graphs or enqueued eagerly
            'rccl_ranks': (comm.info()['nranks'] if have_rccl else int(dist.get_world_size())),
            'rccl_owner': 'libglx' if have_rccl else ('torch.distributed (%s backend)' % dist.get_backend()),
            'engine': engine,
            'exchange': res['how']['exchange'],
            'exchange_selftest': res['how']['selftest'],
        }
        if int(line['rccl_ranks']) != int(args.gpus):
            print('bench.py: the communicator has %d ranks, --gpus is %d' % (line['rccl_ranks'], args.gpus), file=sys.stderr)
            sys.exit(3)
        line['partition'] = res['planner']
        for key, (Gs, r_alt) in sides.items():
            if 'error' in r_alt:
                line[key] = r_alt
                continue
            it_a = args.steps * r_alt['T'] / r_alt['wall']
            line[key] = {'value': it_a if strong else it_a * world, 'global_sweeps_per_sec': it_a, 'ms_per_step': r_alt['wall'] / args.steps * 1e3,
                         'sweeps_per_step': r_alt['T'], 'n': Gs['n'], 'nnz': Gs['nnz'],
                         'halo_rows_per_rank': r_alt['halo_rows'], 'owned_per_rank': r_alt['owned'], 'planner': r_alt['planner'],
                         'exchanges_per_sweep': 1 if r_alt['global_halo'] > 0 else 0,
                         'exchange': r_alt['how']['exchange'], 'exchange_selftest': r_alt['how']['selftest']}
        emit(json.dumps(line))

    def teardown():     # orderly: sweeps are closed above, then the communicator, then the group
        if comm is not None:
            comm.close()
        sync()
        dist.barrier()
        dist.destroy_process_group()
    _finish(teardown)


def _finish(teardown, seconds=30.0):
    """The line is out; what is left is tearing communicators down -- the one thing that can still hang when an exchange went wrong on
    some rank.  It gets `seconds`, then the process ends regardless (the supervising rank is waiting for this child's exit status)."""
    import threading
    th = threading.Thread(target=teardown, daemon=True)
    th.start()
    th.join(seconds)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


def config4_features(n, world_shards=64, seed=2, d=64, C=10):
    """Config-4-shaped data (isotropic Gaussian blobs, d = 64, C = 10, centers * 4) from SHARD-LOCAL random
    streams: shard q of `world_shards` equal blocks uses default_rng([seed, q]), so any rank can produce any block and
    the data do not depend on the number of GPUs.  Returns (X (n, d) float64, labels (n,) int64)."""
    centers = np.random.default_rng(seed).normal(size=(C, d)) * 4
    bounds = [(n * q) // world_shards for q in range(world_shards + 1)]
    X = np.empty((n, d))
    labels = np.empty(n, dtype=np.int64)
    for q in range(world_shards):
        lo, hi = bounds[q], bounds[q + 1]
        g = np.random.default_rng([seed, q])
        labels[lo:hi] = g.integers(0, C, size=hi - lo)
        X[lo:hi] = centers[labels[lo:hi]] + g.normal(size=(hi - lo, d))
    return X, labels


def config4_features_sharded(n, dist, dev, rank, world, world_shards=64, seed=2, d=64, C=10):
    """The same data with every rank generating only ITS share of the 64 shard-local streams (shards rank*64/N ..
    (rank+1)*64/N) -- host work proportional to n/N -- and the ranks all-gathering the blocks over RCCL.  Returns
    (X as a float64 tensor on `dev` -- the coarse order is computed there --, labels as a host array)."""
    import torch
    centers = np.random.default_rng(seed).normal(size=(C, d)) * 4
    sb = [(n * q) // world_shards for q in range(world_shards + 1)]
    q0, q1 = (world_shards * rank) // world, (world_shards * (rank + 1)) // world
    rows = [sb[(world_shards * r) // world] for r in range(world + 1)]          # row ranges of the ranks' shares
    m_own = rows[rank + 1] - rows[rank]
    # the rank's share lands in page-locked memory (the upload then runs at PCIe speed) and its shard-local streams are drawn by a
    # few host threads at once -- they are independent generators, numpy fills outside the GIL: 6.6 s -> 0.6 s for 10^7 x 64 on one rank
    pinned = dev.type == 'cuda'
    Xp_t = torch.empty((m_own, d), dtype=torch.float64, pin_memory=pinned)
    Xp = Xp_t.numpy()
    lp = np.empty(m_own, dtype=np.int64)

    def draw(q):
        lo, hi = sb[q] - rows[rank], sb[q + 1] - rows[rank]
        g = np.random.default_rng([seed, q])
        lp[lo:hi] = g.integers(0, C, size=hi - lo)
        # (the same numbers as `centers[labels] + g.normal(size=(rows, d))`: drawn in place, the centres added in row blocks)
        g.standard_normal(out=Xp[lo:hi])
        for a in range(lo, hi, 65536):
            b = min(hi, a + 65536)
            np.add(centers[lp[a:b]], Xp[a:b], out=Xp[a:b])
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(q1 - q0, (os.cpu_count() or 8) // max(1, world), 32))
    with ThreadPoolExecutor(max_workers=nthreads) as pool:
        list(pool.map(draw, range(q0, q1)))
    if world == 1:
        return Xp_t.to(dev, non_blocking=False), lp
    width = max(rows[r + 1] - rows[r] for r in range(world))
    mine = torch.zeros((width, d + 1), dtype=torch.float64, device=dev)           # last column: the label
    mine[:len(Xp), :d] = torch.from_numpy(Xp).to(dev)
    mine[:len(Xp), d] = torch.from_numpy(lp.astype(np.float64)).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    X = torch.cat([parts[r][:rows[r + 1] - rows[r], :d] for r in range(world)])
    labels = torch.cat([parts[r][:rows[r + 1] - rows[r], d] for r in range(world)]).to(torch.int64).cpu().numpy()
    return X, labels


def _knn_counting_check(torch, X, J, D, lo, K, nsample=512, seed=0):
    """The counting argument of an exact search, on `nsample` of this rank's query rows: with r the reported distance of the last list
    entry (the K-th neighbour; the lists hold the row itself first), no more than K points lie strictly within r of the row, and at
    least K + 1 within r -- computed from the features with torch (differences squared and added in fp64), no part of the search."""
    rng = np.random.default_rng(seed)
    rows = np.sort(rng.choice(len(J), size=min(nsample, len(J)), replace=False))
    n = X.shape[0]
    rk2 = torch.tensor(D[rows, -1] ** 2, dtype=torch.float64, device=X.device)
    inside = torch.zeros(len(rows), dtype=torch.int64, device=X.device)
    within = torch.zeros(len(rows), dtype=torch.int64, device=X.device)
    idx = torch.tensor(lo + rows, dtype=torch.int64, device=X.device)
    for a in range(0, len(rows), 32):
        q = X[idx[a:a + 32]].to(torch.float64)                       # (b, d)
        for c0 in range(0, n, 131072):
            blk = X[c0:c0 + 131072].to(torch.float64)
            d2 = ((blk[:, None, :] - q[None, :, :]) ** 2).sum(dim=2)  # (rows of the block, b)
            inside[a:a + 32] += (d2 < rk2[None, a:a + 32] * (1 - 1e-9)).sum(dim=0)
            within[a:a + 32] += (d2 <= rk2[None, a:a + 32] * (1 + 1e-9)).sum(dim=0)
    inside, within = inside.cpu().numpy(), within.cpu().numpy()
    self_first = bool(np.array_equal(J[rows, 0], lo + rows))
    return {'rows': int(len(rows)), 'max_points_strictly_inside': int(inside.max()), 'min_points_within': int(within.min()), 'k_plus_self': K + 1,
            'row_itself_first': self_first, 'ok': bool(inside.max() <= K and within.min() >= K + 1 and self_first)}


def _operator_checks(sg, world, nsample=1000000, seed=1):
    """This rank's rows of W: no stored diagonal entry, no stored zero, sorted columns; and (one rank: the whole matrix is here) symmetry on
    `nsample` sampled entries, each looked up transposed by a binary search over the row-major keys."""
    W = sg.W_own
    rows_global = sg.plan.own if hasattr(sg.plan, 'own') else np.arange(W.shape[0])
    nrow = W.shape[0]
    rid = np.repeat(np.arange(nrow, dtype=np.int64), np.diff(W.indptr))
    out = {'rows': int(nrow), 'nnz': int(W.nnz),
           'zero_diagonal': bool(not np.any(W.indices == np.asarray(sg.lo + rid, dtype=W.indices.dtype))) if hasattr(sg, 'lo') else None,
           'no_stored_zeros': bool(np.all(W.data != 0)), 'positive_weights': bool(np.all(W.data > 0))}
    if world == 1:
        n = W.shape[1]
        keys = rid * n + W.indices                                        # ascending: rows ascending, columns sorted inside a row
        out['sorted_columns'] = bool(np.all(np.diff(keys) > 0))
        rng = np.random.default_rng(seed)
        pick = rng.integers(0, W.nnz, size=min(nsample, W.nnz))
        tkeys = W.indices[pick].astype(np.int64) * n + rid[pick]
        pos = np.searchsorted(keys, tkeys)
        pos[pos >= len(keys)] = len(keys) - 1
        out['symmetric_on_sample'] = bool(np.all(keys[pos] == tkeys) and np.array_equal(W.data[pos], W.data[pick]))
        out['symmetry_sample'] = int(len(pick))
    out['ok'] = all(v for k, v in out.items() if isinstance(v, bool))
    return out


def main_config4(args):
    """bench.py --gpus N --config 4: STRONG scaling of config 4 (BASELINE.json configs[3]): n vertices in total (default
    10^7), d = 64, k = 10, C = 10, Poisson gradient descent with a fixed T = 200 sweeps per step, the graph built sharded
    (dist_build: every rank searches, symmetrises and plans only its own block of rows) and swept by the library-owned
    distributed sweep (glx_dist_sweep: RCCL halo exchange captured with the SpMMs)."""
    emit = _claim_stdout()
    import torch
    import torch.distributed as dist
    import datetime
    rank, world, local_rank = check_world(args, torch.cuda.device_count())
    if 'MASTER_ADDR' not in os.environ:     # plain `python bench.py --config 4`: a one-rank job without a launcher
        os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(local_rank)
    dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=datetime.timedelta(minutes=20))
    import bench
    import graphlearning_amd as gl
    from graphlearning_amd import _hip, dist as gdist, dist_build
    _hip.set_default_device(local_rank)
    dev = torch.device('cuda', local_rank)
    n, K, T = int(args.n), 10, 200
    t_start = time.perf_counter()

    def progress(what):
        if rank == 0:
            print('[config 4] %7.1f s  %s' % (time.perf_counter() - t_start, what), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    Xt, labels = config4_features_sharded(n, dist, dev, rank, world)       # every rank generates 1/N of the rows
    t_feat = time.perf_counter() - t0
    progress('features generated (n = %d; this rank: %d rows) and all-gathered' % (n, n // world))
    # points arrive in arbitrary order: a contiguous block of them would reference nearly every other vertex.  A coarse
    # geometric order first (64 cells, chained), identical on every rank and computed on the GPU; the block a rank owns
    # is then compact, and the cell starts are where block boundaries may fall between clusters
    t0 = time.perf_counter()
    perm_t, cell_starts = dist_build.coarse_locality_order_torch(Xt, ncells=64, seed=0)
    X = Xt[perm_t].contiguous()          # stays on the device: the search reads it there (no 5 GB down and up again at n = 10^7)
    labels = labels[perm_t.cpu().numpy()]
    del Xt, perm_t
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t_order = time.perf_counter() - t0
    progress('coarse locality order applied (on the device)')
    even = gdist.block_bounds(n, world)                                     # the search is balanced by rows: equal blocks
    lo, hi = int(even[rank]), int(even[rank + 1])
    t0 = time.perf_counter()
    # the cells of the coarse order double as the search's pruning structure (glx_knn_cells_range: the same lists as the all-pairs
    # search, only the cells that can hold a neighbour are visited; --knn allpairs for the search over every tile)
    knn_cells = None if getattr(args, 'knn', 'cells') == 'allpairs' else cell_starts
    J, D = _hip.knn_bruteforce(X, K + 1, device=local_rank, query_range=(lo, hi), cell_starts=knn_cells)
    st = _hip.knn_stats()
    t_knn = time.perf_counter() - t0
    progress('kNN lists of %d query rows (tile kernel %.1f s, %d fallback rows, %s)' % (
        hi - lo, st['tile_ms'] / 1e3, st['fallback_rows'],
        'all pairs' if knn_cells is None else '%d cells, sample stride %d' % (st['cells'], st['seed_sample'])))
    checks = None
    if getattr(args, 'check', False):
        checks = {'knn_counting_argument': _knn_counting_check(torch, X, np.asarray(J), np.asarray(D), lo, K)}
        progress('check: counting argument on %d sampled query rows' % checks['knn_counting_argument']['rows'])
    del X
    # the sweep's blocks follow the graph: boundaries at the cell starts that cross the fewest list entries (between clusters: none),
    # the lists move to their new owners (--partition even keeps the equal blocks)
    t0 = time.perf_counter()
    partition = getattr(args, 'partition', None) or 'cut'
    bounds = even
    if partition == 'cut' and world > 1:
        bounds = dist_build.graph_cut_bounds(dist, n, np.asarray(J), lo, cell_starts, device=dev)
        J, D = dist_build.redistribute_rows(dist, [np.ascontiguousarray(J, dtype=np.int64), np.ascontiguousarray(D, dtype=np.float64)], even, bounds,
                                            device=dev)
    t_cut = time.perf_counter() - t0
    progress('block boundaries %s' % [int(b) for b in bounds])
    t0 = time.perf_counter()
    # (a rank's rows keep the chained cells of the coarse order: the library's breadth-first order of their links among themselves
    # gained nothing -- 253 us per sweep at 10^6 rows either way -- for 3 s more set-up at 10^7)
    local_order = 'block'
    sg = dist_build.ShardedGraph(dist, n, J, D, K, device=dev, bounds=bounds, local_order=local_order)
    del J, D
    progress('own rows of W / P symmetrised, halo plan built (sharded graph)')
    train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
    progress('training set drawn')
    prob = sg.poisson_problem_rows(train_ind, labels[train_ind])
    t_build = time.perf_counter() - t0
    progress('right-hand side rows, degrees of all vertices')
    comm = gdist.init_comm(dist, local_rank)
    progress('communicator')
    ds = gdist.glx_dist_sweep(comm, sg.plan, prob['k'], force_exchange=gdist._force_collectives())
    ds.set_problem(prob['Db'], prob['w0'], prob['deg'], prob['vinf'])
    progress('sweep object on the device')
    for _ in range(max(args.warmup, 1)):
        ds.run(T, T, 8, 0.0)
    progress('warm-up done')
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_dev = 0.0
    for _ in range(args.steps):
        _, ms = ds.run(T, T, 8, 0.0)
        ms_dev += ms
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    stats = torch.tensor([sg.plan.n_halo, sg.plan.n_own, int(sg.W_own.nnz), sg.plan.n_boundary], dtype=torch.int64, device=dev)
    allst = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(allst, stats)
    u_own = ds.fetch()
    how = ds.info()
    if checks is not None:
        checks.update(_operator_checks(sg, world))
        # the sweep conserves sum_i deg_i u_i per class: u <- D^-1 (b + W^T u) with symmetric W gives sum deg u' = sum b + sum deg u, and the
        # columns of b = onehot - mean(onehot) add up to zero; u starts at 0
        du = torch.tensor(np.concatenate([prob['deg'] @ u_own, prob['deg'] @ np.abs(u_own)]), dtype=torch.float64, device=dev)   # (both in the rank's local row order)
        dist.all_reduce(du)
        du = du.cpu().numpy()
        Cc = u_own.shape[1]
        checks['degree_weighted_sum_conserved'] = {'max_abs_sum_deg_u': float(np.max(np.abs(du[:Cc]))), 'sum_deg_abs_u': float(np.min(du[Cc:])),
                                                   'ok': bool(np.all(np.abs(du[:Cc]) <= 1e-8 * du[Cc:]))}
        progress('check: operator symmetric / zero diagonal, degree-weighted sums conserved')
    pred = np.argmax(u_own, axis=1)
    hit = torch.tensor([int(np.sum(pred == labels[sg.plan.own])), len(pred)], dtype=torch.int64, device=dev)
    dist.all_reduce(hit)
    if rank == 0:
        nnz = int(sum(int(a[2]) for a in allst))
        wall = float(dt.item())
        iters = args.steps * T / wall
        abytes = bench.algorithmic_bytes(n, nnz, 10, 8, 8)
        line = {
            'metric': 'Poisson iters/sec', 'value': iters, 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': 'configs[3]: Gaussian blobs n=%d d=64 k=10 C=10, ssl.poisson gradient_descent with T=%d fixed sweeps per '
                                   'step, vertex-sharded over %d GPUs (coarse geometric order, sharded kNN search, block boundaries placed where the fewest '
                                   'list entries cross, symmetrisation by owner rank, request-based halo plan), RCCL all-to-all-v halo exchange per sweep' % (n, T, world),
                       'n': n, 'nnz': nnz, 'classes': 10, 'sweeps_per_step': T, 'parallelism': 'vertex-partition x%d' % world},
            'edges_classes_per_sec': iters * nnz * 10,
            'roofline': {'bound': 'hbm', 'achieved': abytes * iters / 1e9, 'peak': bench.HBM_PEAK_GBS * world, 'unit': 'GB/s',
                         'frac': abytes * iters / 1e9 / (bench.HBM_PEAK_GBS * world), 'traffic': None,
                         'note': 'whole-job algorithmic bytes per sweep / wall time incl. halo exchange; the kernel is bound by random '
                                 '128-byte line gathers (one per stored edge), see DESIGN.md 4.1'},
            'cpu_baseline': None,
            'halo': {'rows_per_rank': [int(a[0]) for a in allst], 'owned_per_rank': [int(a[1]) for a in allst],
                     'boundary_rows_per_rank': [int(a[3]) for a in allst]},
            'partition': {'kind': partition if world > 1 else 'one block', 'bounds': [int(b) for b in bounds]},
            'build': {'features_s': t_feat, 'locality_order_s': t_order, 'knn_own_rows_s': t_knn, 'cut_and_redistribute_s': t_cut,
                      'host_work_note': 'features: every rank generates n/N rows; order: on the device; search: n/N query rows; symmetrisation and plan: the rank\'s own rows', 'knn_tile_tflops_rank0': 2.0 * (hi - lo) * n * (st['visited_share'] if st['cells'] else 1.0) * st['dpa'] / st['tile_ms'] / 1e9,
                      'knn_tile_tflops_note': 'the (query, ref) pairs the cell-pruned search VISITS (knn_visited_share of all pairs), 2 d_padded flops each',
                      'knn_visited_share': (st['visited_share'] if st['cells'] else 1.0),
                      'local_order': local_order, 'knn_search': 'all pairs' if knn_cells is None else 'cell-pruned (%d cells)' % st['cells'], 'knn_tile_s': st['tile_ms'] / 1e3,
                      'symmetrise_plan_s': t_build},
            'accuracy_percent': 100.0 * int(hit[0]) / max(int(hit[1]), 1),
            'checks': checks,
            'rccl_ranks': comm.info()['nranks'], 'rccl_owner': 'libglx' if comm.has_transport() else 'none (one rank)', 'engine': 'glx',
            'exchange': how['exchange'], 'exchange_selftest': how['selftest'],
        }
        emit(json.dumps(line))

    def teardown():
        ds.close()
        comm.close()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    _finish(teardown)
