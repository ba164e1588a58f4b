"""Worker of tests/test_dist_gloo.py::test_distributed_cg_* (and, with 'hip', of tests/test_gpu_dist.py): one rank of a gloo
job running the vertex-partitioned conjugate gradient of graphlearning_amd.dist (cg_distributed / laplace_fit_distributed /
randomwalk_fit_distributed) against the single-process oracle: tolerance mode -- identical labels, iterates within 1e-5,
iteration counts within one of the reference's."""
import os
import sys
import json
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
if len(sys.argv) > 3 and sys.argv[3] == 'hip':
    import torch                                    # before libglx: one shared HIP runtime
import torch.distributed as dist
from graphlearning_amd import dist as gdist


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    use_hip = len(sys.argv) > 3 and sys.argv[3] == 'hip'
    partition = sys.argv[4] if len(sys.argv) > 4 else 'even'
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import csr_from, blobs
    from oracle import gl_oracle as orc
    factory = (lambda plan, C: gdist.CgHipOps(plan, C, 0)) if use_hip else (lambda plan, C: gdist.CgScipyOps(plan, C))
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_twomoons.npz')))
    kw = {}
    if case.startswith('laplace'):
        if case == 'laplace_twomoons':
            W = csr_from(g, 'W_gaussian'); ti = g['train_ind']; tl = g['labels'][ti]
        elif case == 'laplace_normalized_tau':
            W = csr_from(g, 'W_gaussian'); ti = g['train_ind']; tl = g['labels'][ti]
            kw = dict(normalization='normalized', tau=0.01)
        else:                                                     # laplace_blobs
            X, lab = blobs(1500, 8, 4, 21, 2.5)
            W = orc.knn(X, 8)
            ti = orc.trainsets_generate(lab, rate=3, seed=2); tl = lab[ti]
        u, it = gdist.laplace_fit_distributed(W, ti, tl, dist, factory, partition=partition, **kw)
        u_ref, it_ref = orc.laplace_fit(W, ti, tl, return_iters=True, **kw)
    elif case == 'randomwalk':
        X, lab = blobs(1200, 6, 3, 5, 2.0)
        W = orc.knn(X, 7)
        ti = orc.trainsets_generate(lab, rate=4, seed=1); tl = lab[ti]
        u, it = gdist.randomwalk_fit_distributed(W, ti, tl, dist, factory, partition=partition)
        u_ref, it_ref = orc.randomwalk_fit(W, ti, tl, return_iters=True)
    elif case.startswith('poisson_cg'):
        # the default solver of ssl.poisson: residual contract (see dist.poisson_cg_fit_distributed)
        if case == 'poisson_cg_twomoons':
            W = csr_from(g, 'W_gaussian'); ti = g['train_ind']; tl = g['labels'][ti]
        else:                            # (one connected graph: on a disconnected one the reference's own system is inconsistent)
            X, lab = blobs(1500, 8, 4, 21, 1.5)
            W = orc.knn(X, 8)
            ti = orc.trainsets_generate(lab, rate=3, seed=2); tl = lab[ti]
        u, it = gdist.poisson_cg_fit_distributed(W, ti, tl, dist, factory, partition=partition)
        u_ref, it_ref = orc.poisson_cg(W, ti, tl, return_iters=True)
        # the residual of the distributed solution in the reference's own system
        from scipy import sparse
        n = W.shape[0]
        W0 = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))
        src, _ = orc.poisson_source(n, ti, tl)
        Dm = orc.degree_matrix(W0, p=-0.5)
        L = orc.laplacian(W0, 'normalized')
        deg = np.asarray(W0.sum(axis=1)).ravel()
        x = np.sqrt(deg)[:, None] * u
        kw = dict(residual=float(np.sqrt(np.sum((Dm * src - L * x) ** 2))),
                  label_agreement=float(np.mean(orc.predict(u) == orc.predict(u_ref))))
    else:
        raise SystemExit('unknown case')
    extra = kw if case.startswith('poisson_cg') else {}
    res = dict(rank=rank, world=world, it=int(it), it_ref=int(it_ref), max_abs_diff=float(np.max(np.abs(u - u_ref))),
               labels_equal=bool(np.array_equal(orc.predict(u), orc.predict(u_ref))), scale=float(np.max(np.abs(u_ref))), **extra)
    with open(out_path + '.%d' % rank, 'w') as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
