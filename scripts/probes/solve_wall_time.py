import os; os.environ.setdefault("TUM_NMPC_DEV", "1")      # (the fused kernel lives in the development build)
import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
import torch
for B in (1, 26, 256, 1024):
    x0, yref = nominal_batch(B, N=40)
    for mode in ("fused", "pipeline"):
        s = BatchedOcpSolver(N=40, batch=B); s.set_kernel(mode)
        s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref)
        for _ in range(5): s.cold_start(); s.solve()
        torch.cuda.synchronize(); ts = []
        for _ in range(50):
            s.cold_start(); torch.cuda.synchronize(); t = time.perf_counter(); s.solve(); st = s.get_stats("qp_iter"); ts.append(time.perf_counter() - t)
        print(f"batch {B:5d} {mode:9s} wall per solve()+status: median {np.median(ts)*1e3:.3f} ms")
