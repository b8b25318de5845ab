import sys, time
sys.path.insert(0,'/root/repo')
import torch, numpy as np
from tum_control_amd.closed_loop import ClosedLoopBatch
for B in (1, 26):
    for mode in ("loop-serial", "loop-fork", "loop-serial", "loop-fork"):
        cl = ClosedLoopBatch("monteblanco", batch=B, N=38, Tp=3.04, on_device=True, log_capacity=0)
        cl.dev.solver.set_kernel(mode)
        cl.dev.run(200); torch.cuda.synchronize()
        t = time.perf_counter(); cl.dev.run(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t
        t = time.perf_counter(); cl.dev.run(40); torch.cuda.synchronize(); dp = time.perf_counter() - t
        print(B, mode, "graph: %.4f ms/step   plain launches (40 steps): %.4f ms/step" % (1e3*dt/2000, 1e3*dp/40), flush=True)
