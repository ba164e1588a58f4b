"""Developer probe: the library-owned distributed sweep (glx_dist_sweep) on ONE rank of the config-2 graph -- plain,
with a forced self-halo through a device copy, and through a 1-rank RCCL communicator (grouped ncclSend/ncclRecv
captured inside the device graphs): microseconds per sweep."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip, dist as gdist
from test_gpu_dist import _self_halo_plan, _Solo

labels = bench.load_labels(70000)
W = gl.weightmatrix.knn(bench.make_features(labels), 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
prob = gdist.poisson_problem(W, ti, labels[ti])
P = prob['P']
order = gdist.locality_order(P)
plain = gdist.RankPlan(P, order, gdist.block_bounds(P.shape[0], 1), 0)
Pr = P[order, :][:, order]           # self-halo plan in the locality order, like a real rank
halo = _self_halo_plan(Pr, 7)
halo.own = order[halo.own]
cases = [('no exchange', plain, None, False, {}), ('self-halo, device copy', halo, None, True, {}),
         ('self-halo, device copy, pack kernel', halo, None, True, {'GLX_DIST_PACK': '1'}),
         ('self-halo, 1-rank RCCL', halo, 'rccl', True, {}),
         ('self-halo, 1-rank RCCL, pack kernel', halo, 'rccl', True, {'GLX_DIST_PACK': '1'}),
         ('self-halo, 1-rank RCCL, in line', halo, 'rccl', True, {'GLX_DIST_OVERLAP': '0'}),
         ('self-halo, 1-rank RCCL, one launch', halo, 'rccl', True, {'GLX_DIST_FUSE': '100'}),
         ('self-halo, 1-rank RCCL, self-test', halo, 'rccl', True, {'GLX_DIST_CAPTURE_EXCHANGE': '-1'}),
         ('no halo, forced RCCL group', plain, 'rccl', True, {})]
for name, plan, uid, force, env in cases:
    for k in ('GLX_DIST_PACK', 'GLX_DIST_OVERLAP', 'GLX_DIST_FUSE', 'GLX_DIST_CAPTURE_EXCHANGE'):
        os.environ.pop(k, None)
    os.environ.update(env)
    comm = _hip.Comm(1, 0, _hip.Comm.unique_id() if uid else None, 0)
    for graph in (True, False):
        ds = _hip.DistSweep(comm, plan.P_local, plan.n_boundary, plan.send_counts, plan.send_idx, plan.recv_counts, plan.n_global,
                            prob['k'], force_exchange=force, use_hipgraph=graph)
        own = plan.own
        ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
        T, _ = ds.run(50, 1000, 8, 0.0)
        tot = 0.0
        t0 = time.perf_counter()
        for _ in range(20):
            T, ms = ds.run(50, 1000, 8, 0.0)
            tot += ms
        wall = time.perf_counter() - t0
        print('%-40s graph=%d: T=%d, %.2f us/sweep (events), %.2f us/sweep (wall), halo rows %d, %s' % (
            name, graph, T, tot * 1e3 / (20 * T), wall * 1e6 / (20 * T), plan.n_halo, ds.info()['exchange']), flush=True)
        if graph:
            print('    parts: %s' % ds.time_parts(50), flush=True)
        ds.close()
    comm.close()
