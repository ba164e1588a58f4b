"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job running the
vertex-partitioned Poisson sweep with a scipy-backed rank-local sweep (CPU stand-in for the
HIP kernel; same record layout, same sequential row sums)."""
import os
import sys
import json
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.distributed as dist
from scipy import sparse
from graphlearning_amd import dist as gdist
from graphlearning_amd import _hip


class ScipyOps:
    """CPU stand-in for HipOps: fp64 vertex records as torch CPU tensors, products by scipy's
    csr_matvecs (the reference's arithmetic)."""

    def __init__(self, plan, C):
        self.plan, self.C = plan, C
        lay = _hip.record_layout(C, np.float64, True)       # pure host call into libglx
        self.ld, self.wcol = lay['ld'], lay['woff'] // 8
        self.P = plan.P_local
        self.off = int(getattr(plan, 'own_off', 0))          # GatherPlan: the rank's rows are block `rank` of the state

    def new_state(self, rows):
        return torch.zeros((rows, self.ld), dtype=torch.float64)

    def to_device(self, a, dtype=None):
        return torch.from_numpy(np.ascontiguousarray(a))

    def pack(self, dense, w, rows):
        rec = self.new_state(rows)
        if dense is not None:
            rec[:, :self.C] = torch.from_numpy(np.ascontiguousarray(dense, dtype=np.float64))
        if w is not None:
            rec[:, self.wcol] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float64))
        return rec

    def unpack(self, rec, rows):
        return rec[:rows, :self.C].numpy().copy()

    def set_bias(self, bias_rec):
        self.bias = bias_rec.numpy()

    def set_stop_vectors(self, deg, vinf):
        self.deg, self.vinf = np.asarray(deg), np.asarray(vinf)

    def sweep_part(self, which, xin, xout, want_err):
        nb = self.plan.n_boundary
        lo, hi = (0, nb) if which == 0 else (nb, self.plan.n_own)
        if hi <= lo:
            return None
        C = self.C
        x = xin.numpy()
        u = np.ascontiguousarray(x[:, :C])
        w = np.ascontiguousarray(x[:, self.wcol])
        Ps = self.P[lo:hi, :]
        out = xout.numpy()[self.off:]
        out[lo:hi, :C] = self.bias[lo:hi, :C] + Ps * u
        wn = Ps * w
        out[lo:hi, self.wcol] = wn
        if want_err:
            return torch.tensor([np.max(np.abs(self.deg[lo:hi] * wn - self.vinf[lo:hi]))], dtype=torch.float64)
        return None

    def sweep(self, xin, xout, want_err):
        n_own, C = self.plan.n_own, self.C
        x = xin.numpy()
        u = np.ascontiguousarray(x[:, :C])
        w = np.ascontiguousarray(x[:, self.wcol])
        out = xout.numpy()[self.off:]
        out[:n_own, :C] = self.bias[:, :C] + self.P * u
        wn = self.P * w
        out[:n_own, self.wcol] = wn
        if want_err:
            e = np.max(np.abs(self.deg * wn - self.vinf)) if n_own else 0.0
            return torch.tensor([e], dtype=torch.float64)
        return None

    def index_rows(self, rec, idx):
        return rec.index_select(0, idx)


def main():
    case = sys.argv[1]
    out_path = sys.argv[2]
    use_hip = len(sys.argv) > 3 and sys.argv[3] == 'hip'     # rank-local sweeps on the GPU (all ranks share cuda:0)
    glxstep = len(sys.argv) > 3 and sys.argv[3] == 'glxstep'  # the C-ABI sweep object (glx_dist_sweep), gloo as the transport
    partition = sys.argv[4] if len(sys.argv) > 4 else 'even'  # 'even': equal blocks (halo exchange every sweep); 'cut': graph-following blocks
    exchange = sys.argv[5] if len(sys.argv) > 5 else 'halo'   # 'halo': selected rows, all-to-all-v; 'gather': whole blocks, all-gather; 'auto'
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import csr_from, blobs
    from oracle import gl_oracle as orc
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_twomoons.npz')))
    min_iter, max_iter = 50, 1000
    if case == 'twomoons':
        W = csr_from(g, 'W_gaussian'); ti = g['train_ind']; tl = g['labels'][ti]
    elif case == 'directed':
        W = csr_from(g, 'W_gaussian_nosym'); ti = g['train_ind']; tl = g['labels'][ti]
    elif case == 'miniter0':
        W = csr_from(g, 'W_gaussian'); ti = g['train_ind']; tl = g['labels'][ti]; min_iter, max_iter = 0, 30
    elif case == 'blobs':
        X, lab = blobs(1500, 8, 4, 21, 2.5)
        W = orc.knn(X, 8)
        ti = orc.trainsets_generate(lab, rate=3, seed=2); tl = lab[ti]
    elif case == 'connected':
        X, lab = blobs(1800, 8, 4, 23, 0.9)          # overlapping blobs: one component, no cut without a halo
        W = orc.knn(X, 8)
        ti = orc.trainsets_generate(lab, rate=3, seed=2); tl = lab[ti]
    else:
        raise SystemExit('unknown case')
    factory = (lambda plan, k: gdist.HipOps(plan, k, 0)) if use_hip else (lambda plan, k: ScipyOps(plan, k))
    if glxstep:
        u, T = gdist.poisson_fit_glx(W, ti, tl, dist, device=0, min_iter=min_iter, max_iter=max_iter, partition=partition, stepwise=True,
                                     exchange=exchange)
    else:
        u, T = gdist.poisson_fit_distributed(W, ti, tl, dist, factory, min_iter=min_iter, max_iter=max_iter, partition=partition,
                                             exchange=exchange)
    u_ref, T_ref = orc.poisson_gd(W, ti, tl, min_iter=min_iter, max_iter=max_iter, return_T=True)
    # partition bookkeeping invariants
    P = gdist.poisson_problem(W, ti, tl)['P']
    order = gdist.locality_order(P)
    order, bounds, _ = gdist.plan_partition(P, order, world, partition)
    plan = gdist.RankPlan(P, order, bounds, rank)
    chosen = type(gdist.make_plan(P, order, bounds, rank, exchange)).__name__
    share = gdist.halo_share(P, order, bounds)
    counts = [None] * world
    dist.all_gather_object(counts, (plan.send_counts, plan.recv_counts, plan.n_own, plan.n_halo))
    ok_counts = all(counts[a][0][b] == counts[b][1][a] for a in range(world) for b in range(world))
    # the ranks' digest comparison of a plan (collective): silent on agreement, an error on EVERY rank when one rank differs
    gdist._agree(dist, None, [np.arange(7), bounds], 'test')
    try:
        gdist._agree(dist, None, [np.arange(7) + (1 if rank == world - 1 else 0)], 'test')
        caught = False
    except RuntimeError:
        caught = True
    res = dict(rank=rank, world=world, T=int(T), T_ref=int(T_ref), equal=bool(np.array_equal(u, u_ref)), disagree_caught=caught, plan=chosen, halo_share=float(share),
               ok_counts=bool(ok_counts), n_own=int(plan.n_own), n_halo=int(plan.n_halo), global_halo=int(plan.global_halo),
               sorted_perm=bool(np.array_equal(np.sort(order), np.arange(P.shape[0]))))
    with open(out_path + '.%d' % rank, 'w') as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
