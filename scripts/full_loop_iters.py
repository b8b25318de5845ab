import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from tum_control_amd.closed_loop import ClosedLoopBatch
track=sys.argv[1] if len(sys.argv)>1 else 'monteblanco'
g=np.load(f'/root/repo/tests/golden/closed_loop_{track}_full_sub25.npz')
cl=ClosedLoopBatch(track, batch=26, params=g['params'], on_device=True, log_capacity=5499)
lg=cl.run(5499); it=lg['simSolverDebug'][:,:,3]
print(track,'per-set max qp_iter', it.max(axis=0).astype(int))
for s_,k in zip(*np.where(it>=25)): print('step',s_,'set',k,'it',int(it[s_,k]))
np.save('/root/repo/gpurun_out/iters_%s.npy'%track, it)
