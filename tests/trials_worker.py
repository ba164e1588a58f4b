"""Worker of tests/test_dist_gloo.py::test_ssl_trials_shared_over_ranks: a numpy-only stand-in learner
(the real ones need the GPU) run through graphlearning_amd.dist.ssl_trials_distributed on gloo."""
import json
import os
import sys
import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphlearning_amd as gl
from graphlearning_amd import dist as gdist
from graphlearning_amd import ssl as glssl


class nearest_label(gl.ssl.ssl):
    """Labels every vertex like its nearest labelled vertex along a line (host numpy, test only)."""

    def __init__(self, W):
        super().__init__(W, None)
        self.name = 'stub'
        self.accuracy_filename = '_stub'

    def predict(self, ignore_class_priors=False):
        return np.argmax(self.prob, axis=1)

    def _fit(self, train_ind, train_labels, all_labels=None):
        n = self.graph.num_nodes
        k = len(np.unique(train_labels))
        d = np.abs(np.arange(n)[:, None] - np.asarray(train_ind)[None, :])
        return np.eye(k)[np.asarray(train_labels)[np.argmin(d, axis=1)]]


def main():
    out = sys.argv[1]
    dist.init_process_group('gloo')
    rank = dist.get_rank()
    n = 400
    labels = (np.arange(n) * 4 // n).astype(np.int64)
    from scipy import sparse
    W = sparse.identity(n, format='csr')
    trainsets = gl.trainsets.generate(labels, rate=np.array([[1], [2], [3]]), num_trials=4, seed=1)   # 12 training sets
    glssl.results_dir = out + '_results'
    rows = gdist.ssl_trials_distributed(nearest_label(W), trainsets, labels, dist, tag='d_', overwrite=True)
    seq = list(nearest_label(W)._trial_rows(trainsets, labels))
    # the results file exists now: without overwrite every rank must refuse BEFORE running a single trial
    # (reference ssl.py:330-337), consistently across the ranks
    class counting(nearest_label):
        fits = 0

        def _fit(self, *a, **k):
            counting.fits += 1
            return super()._fit(*a, **k)
    dist.barrier()
    again = gdist.ssl_trials_distributed(counting(W), trainsets, labels, dist, tag='d_', overwrite=False)
    with open(out + '.%d' % rank, 'w') as f:
        json.dump(dict(rows=rows, seq=seq, refused=again is None, fits_after_refusal=counting.fits), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
