import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
import sys as _s
B=int(_s.argv[1]) if len(_s.argv)>1 else 4096
x0,yref=nominal_batch(B,N=40)
s=BatchedOcpSolver(N=40,batch=B); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref)
s.cold_start(); s.solve(); s.cold_start()
p=s.profile_phases().astype(float)
it=s.get_stats('qp_iter').mean()
m=p.mean(axis=0)
print('iters',it,'total',m.sum())
for i,v in enumerate(m): print(i, int(v), int(v/it))
