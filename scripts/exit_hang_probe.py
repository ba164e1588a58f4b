"""Developer probe: does a process that leaves an RCCL-backed glx_dist_sweep un-closed (a failed test) exit?  Modes:
'leak' (objects alive at interpreter exit), 'gc_comm_first' (communicator collected before the sweep), 'exc' (uncaught exception)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
if len(sys.argv) > 2 and sys.argv[2] == 'torch':
    import torch
from graphlearning_amd import dist as gdist, _hip
from test_gpu_dist import _self_halo_plan
from conftest import csr_from
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g3_blobs5000.npz')))
W = csr_from(g, 'W')
ti, lab = g['train_ind'], g['labels']
prob = gdist.poisson_problem(W, ti, lab[ti])
plan = _self_halo_plan(prob['P'])
comm = _hip.Comm(1, 0, _hip.Comm.unique_id(), 0)
ds = gdist.glx_dist_sweep(comm, plan, prob['k'], force_exchange=True)
own = plan.own
ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
print('T', ds.run(50, 1000, 8, 0.0)[0], flush=True)
mode = sys.argv[1]
if mode == 'gc_comm_first':
    c2 = comm
    del comm
    c2.close()
    print('comm closed first', flush=True)
    ds.close()
    print('sweep closed', flush=True)
elif mode == 'exc':
    raise RuntimeError('simulated test failure with live objects')
print('leaving with mode', mode, flush=True)
