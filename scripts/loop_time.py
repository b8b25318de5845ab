import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from tum_control_amd.closed_loop import ClosedLoopBatch
for B, steps in ((1, 5000), (26, 5000), (4096, 500)):
    cl = ClosedLoopBatch("monteblanco", batch=B, N=38, Tp=3.04, on_device=True, log_capacity=steps)
    t0 = time.perf_counter(); lg = cl.run(steps); wall = time.perf_counter() - t0
    dbg = lg["simSolverDebug"]
    print(f"batch {B}: {1e3*wall/steps:.3f} ms/step, status0 {(dbg[:,:,4]==0).mean():.4f}, qp_iter {dbg[:,:,3].mean():.2f}, final x {lg['CiLX'][-1,0,:2]}, hipGraph chunk {cl.dev.graph_steps} steps")
