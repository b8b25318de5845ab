#!/usr/bin/env python3
"""In-kernel phase breakdown of the config-2 batch (shader cycles per OCP, mean over the batch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
N, B = 40, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
KERNEL = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("TUM_NMPC_KERNEL", "auto")
import contextlib
from tum_control_amd import solver as _sv
x0, yref = nominal_batch(B, N=N)
with (_sv.dev_library() if KERNEL in ("fused", "pipeline4") else contextlib.nullcontext()):      # (development build)
    s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B, qp_warm_start=(False if KERNEL in ("fused", "pipeline4") else None))
s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.set_kernel(KERNEL)
s.cold_start(); s.solve(); ms0 = s.last_kernel_ms()
s.cold_start(); p = s.profile_phases(); ms1 = s.last_kernel_ms()
it = s.get_stats("qp_iter")
names = ["load+linearise", "condense", "ipm residuals", "M assembly (SYRK)", "Cholesky", "rhs (C'w)", "tri-solves", "C*dv + steps", "ipm exit", "expand+cost+store"]
tot = p[:, :12].sum(axis=1)
if KERNEL in ("pipeline", "auto"):
    names = ["(unused)", "load + initial point", "ipm residuals", "M assembly (SYRK)", "L D L' (rest)", "rhs (C'w)", "tri-solves", "C*dv + steps", "ipm exit", "outputs"]
print(f"kernel variant {KERNEL}; batch {B}: kernel {ms0:.3f} ms plain, {ms1:.3f} ms with timers; mean qp_iter {it.mean():.2f}; mean cycles/OCP {tot.mean():.0f}")
if KERNEL in ("pipeline", "auto"):
    names = names + ["L D L': tile loads + left-looking update", "L D L': micro-panels + block inverses"]
for i, n in enumerate(names):
    print(f"  {n:22s} {p[:, i].mean():12.0f} cyc  {100*p[:, i].mean()/tot.mean():5.1f} %   per-iter {p[:, i].mean()/it.mean():9.0f}")
