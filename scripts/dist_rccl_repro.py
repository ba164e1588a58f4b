"""Developer repro: 1-rank self-halo exchange through RCCL with torch's bundled librccl mapped first."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
if os.environ.get('REPRO_TORCH', '1') == '1':
    import torch
    torch.zeros(4, device='cuda').sum().item()
from graphlearning_amd import _hip, dist as gdist
from test_gpu_dist import _self_halo_plan
from conftest import csr_from
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g3_blobs5000.npz')))
W = csr_from(g, 'W')
ti, lab = g['train_ind'], g['labels']
prob = gdist.poisson_problem(W, ti, lab[ti])
plan = _self_halo_plan(prob['P'])
print('maps:', sorted({l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l or 'amdhip' in l}), flush=True)
comm = _hip.Comm(1, 0, _hip.Comm.unique_id(), 0)
print('maps after init:', sorted({l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l or 'amdhip' in l}), flush=True)
ds = gdist.glx_dist_sweep(comm, plan, prob['k'], force_exchange=True, use_hipgraph=os.environ.get('REPRO_GRAPH', '1') == '1')
own = plan.own
ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
print('running', flush=True)
T, ms = ds.run(50, 1000, 8, 0.0)
u = ds.fetch()
full = np.zeros_like(g['poisson_gd_prob']); full[own] = u
print('T', T, 'equal', np.array_equal(full, g['poisson_gd_prob']), flush=True)
