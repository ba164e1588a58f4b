"""CPU, world_size 2 (and 3) over gloo: the vertex partition, halo lists, per-sweep exchange,
distributed stop test and result gather of graphlearning_amd.dist reproduce the single-rank
oracle bit for bit (the rank-local sweep is a scipy stand-in for the HIP kernel)."""
import json
import os
import subprocess
import sys
import socket
import numpy as np
import pytest
from conftest import ROOT, csr_from


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(case, world, tmp_path, partition='even'):
    out = str(tmp_path / ('res_' + case + '_' + partition))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker.py'), case, out, 'scipy', partition]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [json.load(open(out + '.%d' % k)) for k in range(world)]


@pytest.mark.parametrize('case,world', [('twomoons', 2), ('directed', 2), ('miniter0', 2), ('blobs', 3)])
def test_distributed_sweep_matches_oracle(case, world, tmp_path):
    res = _run(case, world, tmp_path)
    for r in res:
        assert r['world'] == world
        assert r['T'] == r['T_ref'], r
        assert r['equal'], r                 # bit-identical to the single-rank reference iterates
        assert r['ok_counts'] and r['sorted_perm']
    assert sum(r['n_own'] for r in res) in (500, 1500)
    if case == 'twomoons':
        assert res[0]['T'] == 409            # the stop test fired at the reference's iteration
        assert all(r['n_halo'] > 0 for r in res)


def test_rank_plan_properties(golden):
    from graphlearning_amd import dist as gdist
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    P = gdist.poisson_problem(W, g['train_ind'], g['labels'][g['train_ind']])['P']
    order = gdist.locality_order(P)
    n = P.shape[0]
    bounds = gdist.block_bounds(n, 4)
    plans = [gdist.RankPlan(P, order, bounds, r) for r in range(4)]
    assert np.array_equal(np.sort(np.concatenate([p.own for p in plans])), np.arange(n))
    for p in plans:
        # local operator = the rank's rows with renumbered columns, entry order preserved
        sub = P[p.own, :]
        glob = np.concatenate([p.own, p.halo])
        assert np.array_equal(glob[p.P_local.indices], sub.indices)
        assert np.array_equal(p.P_local.data, sub.data)
        assert sum(p.recv_counts) == p.n_halo and p.recv_counts[p.rank] == 0
        for q in plans:   # what p sends to q is what q expects from p
            if q.rank != p.rank:
                off = sum(p.send_counts[:q.rank])
                sent = p.own[p.send_idx[off:off + p.send_counts[q.rank]]]
                roff = sum(q.recv_counts[:p.rank])
                assert np.array_equal(sent, q.halo[roff:roff + q.recv_counts[p.rank]])
    # locality ordering beats the natural order on halo volume
    nat = [gdist.RankPlan(P, np.arange(n), bounds, r).n_halo for r in range(4)]
    assert sum(p.n_halo for p in plans) < sum(nat)


def test_record_layout_host_call():
    from graphlearning_amd import _hip
    assert _hip.record_layout(10) == dict(ld=16, woff=96, rec_bytes=128, G=4, nvec=3, esize=8)
    assert _hip.record_layout(10, np.float32)['rec_bytes'] == 64
    assert _hip.record_layout(2, has_w=False)['woff'] == -1
    with pytest.raises(_hip.GlxError):
        _hip.record_layout(0)


@pytest.mark.parametrize('case,world', [('twomoons', 2), ('blobs', 3)])
def test_distributed_sweep_graph_following_partition(case, world, tmp_path):
    """partition='cut' (the default of poisson_fit_distributed and of bench.py --gpus N): block
    boundaries in the gaps between the graph's pieces.  Same bit-identical result; where every rank
    owns whole pieces nothing is exchanged at all."""
    res = _run(case, world, tmp_path, partition='cut')
    for r in res:
        assert r['T'] == r['T_ref'] and r['equal'] and r['ok_counts'] and r['sorted_perm'], r
    even = _run(case, world, tmp_path, partition='even')
    assert sum(r['n_halo'] for r in res) <= sum(r['n_halo'] for r in even)
    assert len({r['global_halo'] for r in res}) == 1            # every rank derives the same global halo size


def test_cut_bounds_follow_the_pieces():
    """Five disconnected random pieces of unequal size, 2/3/4 ranks: zero crossings, every block within
    the allowed imbalance; a graph without structure keeps equal blocks."""
    from scipy import sparse
    from graphlearning_amd import dist as gdist
    rng = np.random.default_rng(3)
    sizes = [700, 900, 650, 1100, 800]
    blocks = []
    for m in sizes:
        B = sparse.random(m, m, density=12.0 / m, random_state=int(rng.integers(1 << 30)), format='csr')
        blocks.append(B + B.T + sparse.identity(m))
    A = sparse.block_diag(blocks, format='csr')
    perm = rng.permutation(A.shape[0])
    A = A[perm][:, perm].tocsr()                               # hide the structure from the natural order
    order = gdist.locality_order(A)
    n = A.shape[0]
    for world in (2, 3, 4):
        b = gdist.cut_bounds(A, order, world)
        cross = gdist.crossing_counts(A, order)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) > 0)
        assert int(cross[b[1:-1]].sum()) == 0
        assert np.diff(b).max() <= 1.85 * n / world + 1
        plans = [gdist.RankPlan(A, order, b, r) for r in range(world)]
        assert all(p.n_halo == 0 and p.global_halo == 0 for p in plans)
    R = sparse.random(6000, 6000, density=8 / 6000, random_state=1, format='csr')
    R = (R + R.T).tocsr()
    o = gdist.locality_order(R)
    assert np.array_equal(gdist.cut_bounds(R, o, 4), gdist.block_bounds(6000, 4))


def test_ssl_trials_shared_over_ranks(tmp_path):
    """dist.ssl_trials_distributed (the reference's joblib axis, ssl.py:390-396): 12 training sets over
    3 ranks, rows gathered in the original order, file written once in ssl_trials' format."""
    out = str(tmp_path / 'trials')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=3', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'trials_worker.py'), out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS='1'))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open(out + '.%d' % k)) for k in range(3)]
    assert len(res[0]['rows']) == 12
    for k in range(3):
        assert res[k]['rows'] == res[0]['seq']            # same rows, same order as the one-process loop
    text = open(os.path.join(out + '_results', 'd__stub_accuracy.csv')).read().splitlines()
    assert text[0] == 'Number of labels,Accuracy' and text[1:] == res[0]['seq']
