import sys, numpy as np
sys.path.insert(0, ".")
import torch
from tum_control_amd.closed_loop import ClosedLoopBatch
from tum_control_amd import config
from tum_control_amd.r2nmpc import r2_setup
# device closed loop at N = 45 (Tp = 3.6 s), host loop vs device loop
logs = {}
for dev in (False, True):
    cl = ClosedLoopBatch("lvms", batch=3, N=45, Tp=3.6, on_device=dev, log_capacity=60)
    logs[dev] = cl.run(60)
for f in ("simU", "CiLX"):
    print(f, np.abs(logs[True][f] - logs[False][f]).max())
print("status", logs[True]["simSolverDebug"][:, :, 4].max(), "qp_iter", logs[True]["simSolverDebug"][:, :, 3].mean())
# r2 at N = 47
cl = ClosedLoopBatch("modena", batch=5, N=47, Tp=3.76, on_device=True, log_capacity=40, controller="r2")
lg = cl.run(40)
print("r2 N=47 status", lg["simSolverDebug"][:, :, 4].max(), "uh stage 3", cl.solver.constraints_get(3, "uh"))
A = cl.solver.get_from_qp_in(46, "A"); print("A46 shape", A.shape, np.isfinite(A).all())
