import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
import torch
from tum_control_amd.closed_loop import ClosedLoopBatch
track=sys.argv[1] if len(sys.argv)>1 else 'monteblanco'
g=np.load(f'/root/repo/tests/golden/closed_loop_{track}_full_sub25.npz')
P=g['params']; sub=int(g['sub']); n=5499
cl=ClosedLoopBatch(track, batch=26, params=P, on_device=True, log_capacity=n)
t0=time.perf_counter(); lg=cl.run(n); print('wall',time.perf_counter()-t0)
C=lg['CiLX'].transpose(1,0,2)[:, ::sub]; U=lg['simU'].transpose(1,0,2)[:, ::sub]
dbg=lg['simSolverDebug']
print('status max',dbg[:,:,4].max(),'qp_iter mean',dbg[:,:,3].mean(),'max',dbg[:,:,3].max(), 'ref qp_iter mean',g['stats'][:,0].mean())
ep=np.hypot(C[:,:,0]-g['CiLX'][:,:,0], C[:,:,1]-g['CiLX'][:,:,1])
ev=np.abs(C[:,:,3]-g['CiLX'][:,:,3])
print('pos err per loop max', np.round(ep.max(axis=1),4))
print('pos err median', np.median(ep), 'p99', np.quantile(ep,0.99), 'max', ep.max())
print('vel err max', ev.max(), 'median', np.median(ev))
eu=np.abs(U-g['simU'][:, :U.shape[1]])
print('u err median', np.median(eu), 'max', eu.max())
print('cost mean ours', dbg[:,:,0].mean(axis=0)[:5], 'ref', g['stats'][:5,3])
