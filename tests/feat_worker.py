"""Worker of tests/test_dist_gloo.py::test_config4_features_are_sharded: every rank generates its share of the shard-local
feature streams (dist_bench.config4_features_sharded) and the gathered result must equal the one-process generator."""
import os
import sys
import json
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from graphlearning_amd import dist_bench


def main():
    out_path, n = sys.argv[1], int(sys.argv[2])
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    X, labels = dist_bench.config4_features_sharded(n, dist, torch.device('cpu'), rank, world)
    Xr, lr = dist_bench.config4_features(n)
    with open(out_path + '.%d' % rank, 'w') as f:
        json.dump(dict(rank=rank, world=world, x_ok=bool(np.array_equal(X.numpy(), Xr)), l_ok=bool(np.array_equal(labels, lr))), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
