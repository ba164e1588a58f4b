"""Predicted multi-GPU scaling of the vertex-partitioned Poisson sweep from measurements on ONE MI355X.

No 8-GPU node has been available to this project, so the curve is predicted instead of left blank: for world in {2, 4, 8}
every rank's share of the graph is built exactly as the multi-GPU run builds it (dist.RankPlan: locality order, blocks,
[owned | halo] columns, boundary rows first, send lists) and its glx_dist_sweep object is created on the one GPU; the
rank-local pieces of a sweep -- the one launch of the fused form, boundary rows (incl. the scatter into the send buffer) and
interior rows of the split form -- are timed with HIP events (glx_dist_sweep_time_parts), the halo records per peer are counted,
and a sweep of the N-GPU job is predicted as

    sweep(N) = max over ranks  min( fused_r + X_r ,  boundary_r + max(interior_r, X_r) + fork/join )  +  all-reduces / T
    X_r      = L + max over peers ( records(r <- p) * record bytes / link rate )        (direct all-to-all-v: one xGMI link per pair)

(every rank takes the cheaper form, as glx_dist_sweep_create does) with three transports: `bw_peak` / `bw_rccl` (L = 0: the
bandwidth bound at 76 / 50 GB/s per direction) and `rccl` (L = the fixed cost of one grouped ncclSend/ncclRecv exchange, measured
here on a 1-rank communicator: sweep with a 10 000-record self-exchange minus the same launch alone, minus the data's travel time
inside one GPU).  Link rate: 7 xGMI links x ~153 GB/s bidirectional per GPU (task statement) -> 76 GB/s per direction peak; 50 GB/s
per direction is assumed for RCCL point-to-point (ASSUMPTION, not measured: no second GPU).

  weak   config 2 (BASELINE configs[1]): N x 70000 vertices, k = 10, d = 20 -- partitions `cut`, `even` and `cells` (dist.plan_partition);
         --scale 0.8: the connected workload (bench.py --workload connected)
  strong config 4 shape (configs[3]): n vertices (default 2e6; 1e7 needs ~15 min of host planning), d = 64, k = 10, blocks of the coarse
         geometric order -- `cut` (boundaries where the fewest entries cross) and `even` --, T = 200 fixed

Writes profiles/r04_scale_model.json (or --out).  Usage: python scripts/scale_model.py [--n4 2e6] [--worlds 2,4,8] [--skip4]
"""
import os
import sys
import json
import time
import argparse
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
import graphlearning_amd as gl                          # noqa: E402
from graphlearning_amd import _hip, dist as gdist, dist_build      # noqa: E402
from graphlearning_amd import dist_bench                # noqa: E402

LINK_GBS_PEAK = 76.0      # per direction, per link (153 GB/s bidirectional)
LINK_GBS_RCCL = 50.0      # assumed achievable by RCCL send/recv per direction (ASSUMPTION)
ALLREDUCE_US = 20.0       # one small ncclAllReduce(MAX) + host read per stop-test chunk (ASSUMPTION for real peers; the 1-rank call: ~16 us)
CHECK_EVERY = 8
FORK_JOIN_US = 14.0       # cross-stream edges of the captured split form (measured on one rank: 35.0 vs 29.4 us per sweep, profiles/r03_dist_probe.txt)


def rank_measurements(P, order, bounds, prob_rows, C, reps):
    """Per-rank kernel times and halo counts for one partition (all ranks, one after another on the one GPU): the split
    form's pieces (boundary rows incl. the scatter, interior rows) and the fused form's one launch."""
    world = len(bounds) - 1
    comm = _hip.Comm(1, 0, None, 0)            # identity only: the pieces are timed without a transport
    out = []
    for r in range(world):
        t0 = time.perf_counter()
        plan = gdist.RankPlan(P, order, bounds, r)
        t_plan = time.perf_counter() - t0
        own = plan.own
        tp = {}
        for form in ('split', 'fused'):
            # a comm of one rank with the plan of rank r of `world`: counts per peer are folded into one pseudo-peer for the object
            ds = _hip.DistSweep(comm, plan.P_local, plan.n_boundary, [plan.send_idx.size], plan.send_idx, [plan.n_halo], plan.n_global, C,
                                force_exchange=False, use_hipgraph=False, form=form)
            ds.set_problem(prob_rows['Db'][own], prob_rows['w0'][own], prob_rows['deg'][own], prob_rows['vinf'][own])
            t = ds.time_parts(reps)
            if form == 'split':
                tp.update(t)
            else:
                tp['fused_us'] = t['boundary_us']
            rec = ds.lay['rec_bytes']
            ds.close()
        out.append(dict(rank=r, n_own=int(plan.n_own), n_boundary=int(plan.n_boundary), n_halo=int(plan.n_halo), nnz=int(plan.P_local.nnz),
                        recv_per_peer=[int(c) for c in plan.recv_counts], send_per_peer=[int(c) for c in plan.send_counts],
                        rec_bytes=int(rec), plan_s=t_plan, **tp))
    comm.close()
    return out


def gather_measurements(P, order, bounds, prob_rows, C, reps):
    """The all-gather form (dist.GatherPlan, GLX_DIST_FORM_GATHER): every rank's one launch over a state of `world` blocks of `cap`
    records, timed per virtual rank (a rank identity without a transport)."""
    world = len(bounds) - 1
    out = []
    for r in range(world):
        plan = gdist.GatherPlan(P, order, bounds, r)
        comm = _hip.Comm(world, r, None, 0)
        ds = gdist.glx_dist_sweep(comm, plan, C, use_hipgraph=False)
        own = plan.own
        ds.set_problem(prob_rows['Db'][own], prob_rows['w0'][own], prob_rows['deg'][own], prob_rows['vinf'][own])
        t = ds.time_parts(reps)
        out.append(dict(rank=r, n_own=int(plan.n_own), cap=int(plan.cap), fused_us=t['boundary_us'], rec_bytes=int(ds.lay['rec_bytes'])))
        ds.close()
        comm.close()
    return out


def predict_gather(granks, latency_us, link_gbs, T=50, min_iter=50):
    """sweep(N) of the all-gather form: one launch, then every rank's block to every peer -- on a fully connected xGMI node each of the
    N - 1 links of a GPU carries ONE block per direction: X = L + cap * record bytes / link rate (the same L as the grouped send/recv:
    an ASSUMPTION; RCCL's all-gather may well differ)."""
    x = [latency_us + m['cap'] * m['rec_bytes'] / (link_gbs * 1e3) for m in granks]
    per_rank = [m['fused_us'] + xi for m, xi in zip(granks, x)]
    n_allreduce = 1 + max(0, -(-(T - min_iter) // CHECK_EVERY))
    return dict(sweep_us=max(per_rank) + ALLREDUCE_US * n_allreduce / max(T, 1), per_rank_us=per_rank, exchange_us=max(x))


def predict(ranks, latency_us, link_gbs, T=50, min_iter=50):
    """sweep(N) and its parts from the per-rank measurements.  Every rank runs the cheaper of its two forms:
    fused  = one launch for all rows, exchange in line:           fused_r + X_r
    split  = boundary rows, exchange beside the interior rows:    boundary_r + max(interior_r, X_r) + fork/join
    (the library picks by the same estimate, glx_dist_sweep_create); stop test: one all-reduce after min_iter sweeps and one per
    check_every sweeps beyond -- T = min_iter = 50 on the bench graph, i.e. one per 50 sweeps."""
    per_rank, forms = [], []
    for m in ranks:
        peak_peer = max(max(m['recv_per_peer'], default=0), max(m['send_per_peer'], default=0))
        x = 0.0 if (m['n_halo'] == 0 and sum(m['send_per_peer']) == 0) else latency_us + peak_peer * m['rec_bytes'] / (link_gbs * 1e3)
        fused = m['fused_us'] + x
        split = m['boundary_us'] + max(m['interior_us'], x) + (FORK_JOIN_US if x > 0 else 0.0)
        per_rank.append(min(fused, split))
        forms.append('fused' if fused <= split else 'split')
    any_halo = any(m['n_halo'] > 0 for m in ranks)
    n_allreduce = 1 + max(0, -(-(T - min_iter) // CHECK_EVERY))
    return dict(sweep_us=max(per_rank) + ALLREDUCE_US * n_allreduce / max(T, 1), slowest_rank=int(np.argmax(per_rank)), per_rank_us=per_rank,
                forms=forms, exchanges_per_sweep=1 if any_halo else 0)


def single_rank_sweep_us(P, prob, C, reps=3, T=50):
    """The 1-GPU sweep of the same graph (whole graph on one rank, no halo): captured head of T sweeps, events."""
    n = P.shape[0]
    plan = gdist.RankPlan(P, gdist.locality_order(P) if n <= 600000 else np.arange(n), gdist.block_bounds(n, 1), 0)
    comm = _hip.Comm(1, 0, None, 0)
    ds = _hip.DistSweep(comm, plan.P_local, plan.n_boundary, plan.send_counts, plan.send_idx, plan.recv_counts, n, C)
    own = plan.own
    ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
    ds.run(T, T, CHECK_EVERY, 0.0)
    ms = 0.0
    for _ in range(reps):
        ms += ds.run(T, T, CHECK_EVERY, 0.0)[1]
    ds.close()
    comm.close()
    return ms * 1e3 / (reps * T)


def rccl_latency_us(P70k, order, prob, C):
    """Latency of one grouped ncclSend/ncclRecv exchange as this library issues it: per-sweep time of a 1-rank sweep whose
    ~10 000 boundary records travel through RCCL to itself, minus the same sweep with a plain device copy."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_gpu_dist import _self_halo_plan
    Pr = P70k[order, :][:, order]
    plan = _self_halo_plan(Pr, 7)
    plan.own = order[plan.own]
    res = {}
    for name, uid in (('copy', None), ('rccl', 'rccl')):
        try:
            comm = _hip.Comm(1, 0, _hip.Comm.unique_id() if uid else None, 0)
        except Exception as exc:                               # noqa: BLE001
            res[name] = None
            res['error'] = str(exc)
            continue
        ds = _hip.DistSweep(comm, plan.P_local, plan.n_boundary, plan.send_counts, plan.send_idx, plan.recv_counts, plan.n_global, C,
                            force_exchange=True, form='fused')      # one launch + the exchange in line: sweep - launch = the exchange itself
        own = plan.own
        ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
        ds.run(50, 50, CHECK_EVERY, 0.0)
        ms = sum(ds.run(50, 50, CHECK_EVERY, 0.0)[1] for _ in range(5))
        res[name] = ms * 1e3 / 250
        res[name + '_info'] = ds.info()
        res[name + '_launch_alone_us'] = ds.time_parts(50)['boundary_us']
        res['halo_records'] = int(plan.n_halo)
        res['halo_bytes'] = int(plan.n_halo) * ds.lay['rec_bytes']
        ds.close()
        comm.close()
    if res.get('rccl'):
        # what one grouped ncclSend/ncclRecv exchange of `halo_bytes` costs in line; the part that is not bandwidth is the latency
        res['exchange_us'] = max(0.0, res['rccl'] - res['rccl_launch_alone_us'])
        # the data of a self-exchange move inside one GPU (~1 TB/s): the rest of the exchange's time is its fixed cost
        res['latency_us'] = max(0.0, res['exchange_us'] - res['halo_bytes'] / 1e6)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--worlds', default='2,4,8')
    ap.add_argument('--n4', type=float, default=2e6)
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--skip4', action='store_true')
    ap.add_argument('--skip2', action='store_true')
    ap.add_argument('--scale', type=float, default=2.0, help='centre scale of the config-2 features: 2.0 = the headline blobs (10 separate clusters), '
                                                              '0.8 = bench.py --workload connected (one component)')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r04_scale_model.json'))
    args = ap.parse_args()
    worlds = [int(w) for w in args.worlds.split(',')]
    _hip.require_device()
    t_start = time.perf_counter()

    def log(msg):
        print('[scale_model %6.1f s] %s' % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)
    result = dict(assumptions=dict(link_GBs_peak_per_direction=LINK_GBS_PEAK, link_GBs_rccl_assumed=LINK_GBS_RCCL, allreduce_us=ALLREDUCE_US,
                                   check_every=CHECK_EVERY, fork_join_us=FORK_JOIN_US,
                                   model='sweep(N) = max_r min(fused_r + X_r, boundary_r + max(interior_r, X_r) + fork_join) + allreduces/T; '
                                         'X_r = L + max_peer bytes / link rate; kernel times measured per virtual rank on ONE MI355X '
                                         '(glx_dist_sweep_time_parts)'),
                  libglx_source_hash=__import__('graphlearning_amd._build', fromlist=['x']).source_hash())

    # ---- transport latency, measured on the 70k graph
    labels = bench.load_labels(bench.N_PER_RANK)
    W = gl.weightmatrix.knn(bench.make_features(labels), bench.K_NN)
    ti = gl.trainsets.generate(labels, rate=1, seed=0)
    prob1 = gdist.poisson_problem(W, ti, labels[ti])
    order1 = gdist.locality_order(prob1['P'])
    lat = rccl_latency_us(prob1['P'], order1, prob1, prob1['k'])
    result['rccl_self_exchange'] = lat
    L = lat.get('latency_us') if lat.get('latency_us') is not None else 20.0
    log('1-rank exchange: %s' % lat)

    # ---- weak scaling, config 2
    if not args.skip2:
        weak = {}
        t1 = single_rank_sweep_us(prob1['P'], prob1, prob1['k'])
        weak['single_gpu_sweep_us'] = t1
        log('config 2, 1 GPU: %.2f us per sweep' % t1)
        for world in worlds:
            n = bench.N_PER_RANK * world
            labels = bench.load_labels(n)
            W = gl.weightmatrix.knn(bench.make_features(labels, scale=args.scale), bench.K_NN)
            ti = gl.trainsets.generate(labels, rate=1, seed=0)
            prob = gdist.poisson_problem(W, ti, labels[ti])
            P = prob['P']
            order = gdist.locality_order(P)
            entry = dict(n=n, nnz=int(P.nnz))
            for part in ('even', 'cut', 'cells'):
                t_part = time.perf_counter()
                order_p, bounds, pinfo = gdist.plan_partition(P, order, world, part)
                t_part = time.perf_counter() - t_part
                ranks = rank_measurements(P, order_p, bounds, prob, prob['k'], args.reps)
                pr = {name: predict(ranks, lat_us, gbs) for name, lat_us, gbs in
                      (('bw_peak', 0.0, LINK_GBS_PEAK), ('bw_rccl', 0.0, LINK_GBS_RCCL), ('rccl', L, LINK_GBS_RCCL))}
                for v in pr.values():
                    v['weak_efficiency'] = t1 / v['sweep_us']
                    v['iters_per_s_70k_equivalents'] = world * 1e6 / v['sweep_us']
                share = gdist.halo_share(P, order_p, bounds)
                granks = gather_measurements(P, order_p, bounds, prob, prob['k'], args.reps)
                pg = {name: predict_gather(granks, lat_us, gbs) for name, lat_us, gbs in
                      (('bw_peak', 0.0, LINK_GBS_PEAK), ('bw_rccl', 0.0, LINK_GBS_RCCL), ('rccl', L, LINK_GBS_RCCL))}
                for v in pg.values():
                    v['weak_efficiency'] = t1 / v['sweep_us']
                entry[part] = dict(ranks=ranks, predicted=pr, imbalance=max(m['n_own'] for m in ranks) * world / n,
                                   work_imbalance=max(m['nnz'] for m in ranks) * world / float(P.nnz), planner=pinfo, planner_s=t_part,
                                   halo_share=share, gather_form=dict(ranks=granks, predicted=pg),
                                   auto_takes='gather' if (share >= gdist.GATHER_SHARE and world > 1) else 'halo')
                log('    halo share %.2f; all-gather form: predicted sweep bw %.1f us, rccl %.1f us (eff %.2f / %.2f)'
                    % (share, pg['bw_rccl']['sweep_us'], pg['rccl']['sweep_us'], pg['bw_rccl']['weak_efficiency'], pg['rccl']['weak_efficiency']))
                log('config 2 weak, N=%d, %s: halo/rank %s, predicted sweep bw %.1f us, rccl %.1f us (eff %.2f / %.2f)'
                    % (world, part, [m['n_halo'] for m in ranks], pr['bw_rccl']['sweep_us'], pr['rccl']['sweep_us'],
                       pr['bw_rccl']['weak_efficiency'], pr['rccl']['weak_efficiency']))
            weak[str(world)] = entry
        result['config2_weak'] = weak

    # ---- strong scaling, config 4 shape
    if not args.skip4:
        n = int(args.n4)
        X, labels = dist_bench.config4_features(n)
        perm = dist_build.coarse_locality_order(X, ncells=64, seed=0)
        X, labels = np.ascontiguousarray(X[perm]), labels[perm]
        t0 = time.perf_counter()
        W = gl.weightmatrix.knn(X, 10)
        log('config 4 shape, n = %d: graph built in %.1f s (nnz %d)' % (n, time.perf_counter() - t0, W.nnz))
        del X
        ti = gl.trainsets.generate(labels, rate=5, seed=0)
        prob = gdist.poisson_problem(W, ti, labels[ti])
        P = prob['P']
        del W
        order = np.arange(n)                     # the coarse geometric order IS the partition order of bench.py --config 4
        strong = dict(n=n, nnz=int(P.nnz))
        t1 = single_rank_sweep_us(P, prob, prob['k'], reps=2, T=20)
        strong['single_gpu_sweep_us'] = t1
        log('config 4 shape, 1 GPU: %.1f us per sweep' % t1)
        for world in worlds:
            entry = {}
            for part in ('cut', 'even'):       # cut: boundaries where the fewest entries cross (bench.py --config 4's default), even: equal blocks
                bounds = gdist.cut_bounds(P, order, world) if part == 'cut' else gdist.block_bounds(n, world)
                ranks = rank_measurements(P, order, bounds, prob, prob['k'], max(4, args.reps // 4))
                pr = {name: predict(ranks, lat_us, gbs, T=200, min_iter=200) for name, lat_us, gbs in
                      (('bw_peak', 0.0, LINK_GBS_PEAK), ('bw_rccl', 0.0, LINK_GBS_RCCL), ('rccl', L, LINK_GBS_RCCL))}
                for v in pr.values():
                    v['speedup'] = t1 / v['sweep_us']
                    v['strong_efficiency'] = t1 / v['sweep_us'] / world
                entry[part] = dict(ranks=ranks, predicted=pr, bounds=[int(b) for b in bounds], imbalance=max(m['n_own'] for m in ranks) * world / n)
                log('config 4 strong, N=%d, %s: own/rank %s, halo/rank %s, predicted sweep bw %.1f us rccl %.1f us (speed-up %.2f / %.2f) forms %s'
                    % (world, part, [m['n_own'] for m in ranks], [m['n_halo'] for m in ranks], pr['bw_rccl']['sweep_us'], pr['rccl']['sweep_us'],
                       pr['bw_rccl']['speedup'], pr['rccl']['speedup'], pr['rccl']['forms']))
            strong[str(world)] = entry
        result['config4_strong'] = strong

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(result, f, indent=1)
    log('written ' + args.out)


if __name__ == '__main__':
    main()
