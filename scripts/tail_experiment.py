import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
B=4096
x0,yref=nominal_batch(B,N=40)
s=BatchedOcpSolver(N=40,batch=B); s.install_reference_ocp()
def run(x0,yref,label):
    s.set_x0(x0); s.set_yref_all(yref)
    ms=[]
    for _ in range(6):
        s.cold_start(); s.solve(); ms.append(s.last_kernel_ms())
    it=s.get_stats('qp_iter')
    print(label, 'ms',np.median(ms),'solves/s',B/np.median(ms)*1e3,'iters mean',it.mean(),'min',it.min(),'max',it.max())
    return it
it=run(x0,yref,'mixed')
# identical instances with the mean-ish iteration count
k=int(np.argmin(np.abs(it-9)))
run(np.tile(x0[k],(B,1)), np.tile(yref[k],(B,1,1)), 'all identical (it=%d)'%it[k])
# sorted by iteration count descending (LPT order)
o=np.argsort(-it, kind='stable')
run(x0[o],yref[o],'sorted longest first')
o2=np.argsort(it, kind='stable')
run(x0[o2],yref[o2],'sorted shortest first')
