"""GPU: the rank-local pieces of the multi-GPU path on ONE MI355X -- HipOps (libglx device-pointer
API on torch tensors) for every rank of a 2- and 4-way partition, checked against scipy; and
the distributed bench entry under torchrun with one rank (NCCL/RCCL init, all_reduce, driver)."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
from conftest import ROOT, csr_from

pytestmark = pytest.mark.gpu


def test_hipops_rank_local_sweeps_match_scipy(golden):
    import torch                      # before libglx: one shared HIP runtime
    from graphlearning_amd import dist as gdist, _hip
    _hip.require_device()
    g = golden('g3_blobs5000.npz')
    W = csr_from(g, 'W')
    ti = g['train_ind']
    prob = gdist.poisson_problem(W, ti, g['labels'][ti])
    P, C = prob['P'], prob['k']
    n = P.shape[0]
    order = gdist.locality_order(P)
    rng = np.random.default_rng(0)
    u = rng.normal(size=(n, C))
    w = rng.random(n)
    ref_u = prob['Db'] + P * u
    ref_w = P * w
    for world in (2, 4):
        for rank in range(world):
            plan = gdist.RankPlan(P, order, gdist.block_bounds(n, world), rank)
            ops = gdist.HipOps(plan, C, 0)
            glob = np.concatenate([plan.own, plan.halo])
            xin = ops.pack(u[glob], w[glob], len(glob))
            xout = ops.new_state(len(glob))
            ops.set_bias(ops.pack(prob['Db'][plan.own], None, plan.n_own))
            ops.set_stop_vectors(prob['deg'][plan.own], prob['vinf'][plan.own])
            e = ops.sweep(xin, xout, True)
            torch.cuda.synchronize()
            got = ops.unpack(xout, plan.n_own)
            assert np.array_equal(got, ref_u[plan.own]), (world, rank)
            wcol = ops.lay['woff'] // 8
            got_w = xout[:plan.n_own, wcol].cpu().numpy()
            assert np.array_equal(got_w, ref_w[plan.own])
            err_ref = np.max(np.abs(prob['deg'][plan.own] * ref_w[plan.own] - prob['vinf'][plan.own]))
            assert float(e.item()) == err_ref
            ops.close()


@pytest.mark.parametrize('force_coll', ['0', '1'])
def test_distributed_bench_entry_one_rank(tmp_path, force_coll):
    # force_coll=1: the all_to_all / all_reduce calls are issued even with one rank (eagerly: collectives are never
    # captured into a device graph, see DistSweep.run)
    env = dict(os.environ, GLX_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', GLX_DIST_FORCE_COLLECTIVES=force_coll)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'capture unavailable' not in r.stderr, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    j = json.loads(line)
    assert j['n_gpus'] == 1 and j['value'] > 0 and j['config']['sweeps_per_step'] == 50
    print(line[:400])


@pytest.mark.parametrize('case,world,partition', [('twomoons', 2, 'even'), ('blobs', 3, 'even'), ('miniter0', 2, 'even'),
                                                  ('twomoons', 2, 'cut'), ('blobs', 3, 'cut')])
def test_multi_rank_hip_sweeps_over_gloo(case, world, partition, tmp_path):
    """Several ranks, every one running its rank-local sweeps with the HIP kernel (all on cuda:0),
    exchanging halo records through gloo (host-staged): the full multi-rank GPU path minus RCCL,
    bit-identical to the single-rank oracle."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / ('res_' + case + '_' + partition))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world, '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'dist_worker.py'), case, out, 'hip', partition]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS='1'), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for k in range(world):
        res = json.load(open(out + '.%d' % k))
        assert res['T'] == res['T_ref'] and res['equal'] and res['ok_counts'], res
