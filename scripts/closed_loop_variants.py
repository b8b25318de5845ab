import os; os.environ.setdefault("TUM_NMPC_DEV", "1")      # (the fused kernel lives in the development build)
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from tum_control_amd.closed_loop import ClosedLoopBatch
# kernel variant x interior point start: the development build's fused kernel always cold-starts the interior point method; the pipeline
# (the shipped library's only kernel) runs with its warm start from the previous QP (the shipped default) and without
for k, warm in (("fused", False), ("pipeline", False), ("pipeline", True)):
    for B, steps in ((4096, 300), (26, 1000)):
        cl = ClosedLoopBatch("monteblanco", batch=B, N=38, Tp=3.04, on_device=True, log_capacity=steps, qp_warm_start=warm)
        cl.solver.set_kernel(k)
        cl.run(50)
        t0 = time.perf_counter(); lg = cl.run(steps - 50); wall = time.perf_counter() - t0
        dbg = lg["simSolverDebug"]
        print(f"{k:9s} {'IPM warm start' if warm else 'IPM cold start'} batch {B}: {1e3 * wall / (steps - 50):.3f} ms/step, {B * (steps - 50) / wall:,.0f} closed-loop solves/s, "
              f"status0 {(dbg[:, :, 4] == 0).mean():.4f}, qp_iter {dbg[:, :, 3].mean():.2f}")
        del cl
