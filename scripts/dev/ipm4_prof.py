#!/usr/bin/env python3
"""phase cycles of ipm4_kernel per wavefront (config 2)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
N, B = 40, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, yref = nominal_batch(B, N=N)
from tum_control_amd import solver as _sv
with _sv.dev_library():
    s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B, qp_warm_start=False)
s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.set_kernel("pipeline4")
s.cold_start(); s.solve(); ms0 = s.last_kernel_ms()
s.cold_start(); p = s.profile_phases(); it = s.get_stats("qp_iter")
print(f"pipeline4 batch {B}: {ms0:.3f} ms, mean qp_iter {it.mean():.2f}")
for w in range(4):
    a = p[:, 3 * w:3 * w + 3].mean(axis=0) / it.mean()
    print(f"  wavefront {w}: per iteration: row phases {a[0]:8.0f}  assembly+factorisation {a[1]:8.0f}  solves {a[2]:8.0f}  total {a.sum():8.0f}")
