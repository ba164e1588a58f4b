"""Worker of tests/test_dist_gloo.py::test_sharded_build_*: one rank of a gloo job building ITS rows of the kNN weight
matrix / Poisson operator / exchange plan with graphlearning_amd.dist_build (no rank holds the whole matrix) and
running the distributed sweep with a scipy stand-in for the rank-local kernel; everything is compared with the
single-process oracle."""
import os
import sys
import json
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch.distributed as dist
from scipy import sparse
from graphlearning_amd import dist as gdist, dist_build
from dist_worker import ScipyOps


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    engine = sys.argv[3] if len(sys.argv) > 3 else 'ops'      # 'glxstep': rank-local pieces by libglx on cuda:0, gloo as the transport
    partition = sys.argv[4] if len(sys.argv) > 4 else 'even'   # 'cut': blocks that follow the graph (dist_build.graph_cut_bounds) + redistributed lists
    local_order = sys.argv[5] if len(sys.argv) > 5 else 'block'  # 'rcm': a rank's rows in the library's breadth-first order (ShardPlan)
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import blobs
    from oracle import gl_oracle as orc
    kernel = 'gaussian'
    min_iter, max_iter = 50, 400
    if case == 'blobs':
        X, lab = blobs(1500, 8, 4, 21, 2.5)
        k = 8
    elif case == 'uniform':
        X, lab = blobs(900, 5, 3, 4, 2.0)
        k = 6
        kernel = 'uniform'
    elif case == 'miniter0':
        X, lab = blobs(700, 6, 3, 9, 3.0)
        k = 7
        min_iter, max_iter = 0, 40
    elif case.startswith('random'):
        # random size / dimension / neighbours / kernel / iteration bounds; clouds with a dense core so that some vertices are
        # many rows' neighbour and the blocks the ranks own differ widely in their halos
        rng = np.random.default_rng(500 + int(case[6:]))
        n_pts = int(rng.integers(150, 1400))
        dd = int(rng.choice([2, 3, 6, 12]))
        CC = int(rng.integers(2, 5))
        lab = rng.integers(0, CC, size=n_pts).astype(np.int64)
        lab[:CC] = np.arange(CC)
        X = rng.normal(size=(CC, dd))[lab] * 2.0 + rng.normal(size=(n_pts, dd)) * rng.choice([0.2, 1.0], size=(n_pts, 1))
        k = int(rng.integers(3, 12))
        kernel = str(rng.choice(['gaussian', 'gaussian', 'uniform', 'distance', 'singular']))
        min_iter, max_iter = [(50, 300), (0, 60), (20, 20), (5, 500)][int(rng.integers(0, 4))]
    else:
        raise SystemExit('unknown case')
    n = X.shape[0]
    cut_info = {}
    if partition == 'cut':      # the config-4 pipeline: coarse geometric order first, then the search, then blocks that follow the graph
        perm, cell_starts = dist_build.coarse_locality_order(X, ncells=16, seed=0, return_cells=True)
        X, lab = np.ascontiguousarray(X[perm]), lab[perm]
    J, D = orc.knnsearch(X, k + 1)                       # every rank could search its own rows; the lists are the input here
    ti = orc.trainsets_generate(lab, rate=3, seed=2)
    tl = lab[ti]
    bounds = gdist.block_bounds(n, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    J_own, D_own = J[lo:hi], D[lo:hi]
    if partition == 'cut':
        even = bounds
        bounds = dist_build.graph_cut_bounds(dist, n, J_own, lo, cell_starts)
        J_own, D_own = dist_build.redistribute_rows(dist, [np.ascontiguousarray(J_own, dtype=np.int64), np.ascontiguousarray(D_own)], even, bounds)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        # the same cuts from the global matrix's point of view: crossings of the chosen boundaries, counted directly
        cross = [int(np.sum((np.minimum(np.arange(n)[:, None], J) < b) & (np.maximum(np.arange(n)[:, None], J) >= b))) for b in bounds[1:-1]]
        cross_even = [int(np.sum((np.minimum(np.arange(n)[:, None], J) < b) & (np.maximum(np.arange(n)[:, None], J) >= b))) for b in even[1:-1]]
        cut_info = dict(moved_ok=bool(np.array_equal(J_own, J[lo:hi]) and np.array_equal(D_own, D[lo:hi])), bounds=[int(b) for b in bounds],
                        crossing=cross, crossing_even=cross_even)
    u, T, sg = dist_build.poisson_fit_sharded(dist, n, J_own, D_own, k, ti, tl, engine=engine, ops_factory=lambda plan, C: ScipyOps(plan, C),
                                             min_iter=min_iter, max_iter=max_iter, kernel=kernel, device=0, bounds=bounds, local_order=local_order)
    # oracle: the whole pipeline in one process
    W = orc.knn_weights(J, D, k, kernel=kernel)
    W.sort_indices()
    s = orc.poisson_gd_setup(W, ti, tl)
    u_ref, T_ref = orc.poisson_gd(W, ti, tl, min_iter=min_iter, max_iter=max_iter, return_T=True)
    Wb = sparse.csr_matrix(W[lo:hi, :])
    Pb = sparse.csr_matrix(s['P'])[lo:hi, :]
    w_ok = (np.array_equal(sg.W_own.indptr, Wb.indptr) and np.array_equal(sg.W_own.indices, Wb.indices)
            and np.array_equal(sg.W_own.data, Wb.data))
    p_ok = (np.array_equal(sg.P_own.indptr, Pb.indptr) and np.array_equal(sg.P_own.indices, Pb.indices)      # entry ORDER included
            and np.array_equal(sg.P_own.data, Pb.data))
    deg_ok = np.array_equal(sg.deg_own, s['deg'][lo:hi])
    # the plan equals the one the global planner derives for the same blocks
    ref_plan = gdist.RankPlan(s['P'], np.arange(n), bounds, rank)
    pl = sg.plan
    if local_order != 'block':      # another order of the rank's rows: the same SETS of boundary / interior rows, every row intact
        nb = pl.n_boundary
        inv = np.empty(hi - lo, dtype=np.int64)
        inv[pl.own - lo] = np.arange(hi - lo)
        rows_ok = True
        Pl = sparse.csr_matrix(pl.P_local)
        glob = np.concatenate([pl.own, pl.halo])
        for g_row in np.random.default_rng(1).integers(lo, hi, size=min(200, hi - lo)):
            r = inv[g_row - lo]
            a, b = Pl.indptr[r], Pl.indptr[r + 1]
            a0, b0 = Pb.indptr[g_row - lo], Pb.indptr[g_row - lo + 1]
            rows_ok = rows_ok and np.array_equal(glob[Pl.indices[a:b]], Pb.indices[a0:b0]) and np.array_equal(Pl.data[a:b], Pb.data[a0:b0])
        plan_ok = (set(pl.own[:nb]) == set(ref_plan.own[:nb]) and set(pl.own) == set(ref_plan.own) and np.array_equal(pl.halo, ref_plan.halo)
                   and pl.send_counts == ref_plan.send_counts and pl.recv_counts == ref_plan.recv_counts and pl.n_boundary == ref_plan.n_boundary
                   and np.array_equal(np.sort(pl.own[pl.send_idx]), np.sort(ref_plan.own[ref_plan.send_idx])) and rows_ok)
    else:
      plan_ok = (np.array_equal(pl.own, ref_plan.own) and np.array_equal(pl.halo, ref_plan.halo) and pl.send_counts == ref_plan.send_counts
               and pl.recv_counts == ref_plan.recv_counts and np.array_equal(pl.send_idx, ref_plan.send_idx)
               and pl.n_boundary == ref_plan.n_boundary and pl.global_halo == ref_plan.global_halo
               and np.array_equal(pl.P_local.indices, ref_plan.P_local.indices) and np.array_equal(pl.P_local.data, ref_plan.P_local.data)
               and np.array_equal(pl.P_local.indptr, ref_plan.P_local.indptr))
    res = dict(rank=rank, world=world, T=int(T), T_ref=int(T_ref), equal=bool(np.array_equal(u, u_ref)), w_ok=bool(w_ok), p_ok=bool(p_ok),
               deg_ok=bool(deg_ok), plan_ok=bool(plan_ok), n_halo=int(pl.n_halo), nnz_own=int(sg.W_own.nnz), n_own=int(hi - lo), **cut_info)
    with open(out_path + '.%d' % rank, 'w') as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
