#!/usr/bin/env python3
"""A/B of the kernel variants on BASELINE configs[1] (4096 x N=40): results against the fused kernel, kernel times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch

N, B = 40, int(os.environ.get("B", "4096"))
x0, yref = nominal_batch(B, N=N)
ref = None
for name in sys.argv[1:] or ["fused", "pipeline"]:
    import contextlib
    from tum_control_amd import solver as _sv
    with (_sv.dev_library() if name in ("fused", "pipeline4") and "TUM_NMPC_LIB" not in os.environ else contextlib.nullcontext()):
        s = BatchedOcpSolver(N=N, batch=B, store_qp_in=False, qp_warm_start=(False if name in ("fused", "pipeline4") else None))      # (fused / pipeline4: development build, cold-started interior point method)
    s.install_reference_ocp(); s.set_kernel(name); s.set_kernel("time-ipm")
    s.set_x0(x0); s.set_yref_all(yref)
    ms, ipm = [], []
    for r in range(6):
        s.cold_start(); st = s.solve(); ms.append(s.last_kernel_ms())
        if name == "pipeline":
            ipm.append(1e3 * s.get_stats("time_ipm"))
    X, U = s.get_iterate(); it = s.get_stats("qp_iter"); cost = s.get_cost(); stat = s.get_stats("status")
    line = f"{name:9s} status {st} ok {np.mean(stat == 0):.4f} qp_iter {it.mean():.3f} kernel ms {np.median(ms[1:]):.3f} (first {ms[0]:.3f})"
    if ipm:
        line += f" ipm ms {np.median(ipm[1:]):.3f}"
    if ref is None:
        ref = (X, U, it, cost)
    else:
        line += f" | vs {sys.argv[1] if len(sys.argv) > 1 else 'fused'}: max|dU| {np.abs(U - ref[1]).max():.2e} max|dX| {np.abs(X - ref[0]).max():.2e} " \
                f"iter equal {np.mean(it == ref[2]):.4f} max|dcost|/cost {np.max(np.abs(cost - ref[3]) / np.abs(ref[3])):.2e}"
    print(line, flush=True)
    # warm second RTI step
    s.set_x0(X[:, 1]); s.solve(); print(f"          warm step: kernel ms {s.last_kernel_ms():.3f} qp_iter {s.get_stats('qp_iter').mean():.3f}", flush=True)
    del s
