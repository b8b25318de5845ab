#!/usr/bin/env python3
"""Throughput of configs[1] when the boundary hands over HOST buffers every step (PCIe-inclusive): upload x0 + yref,
cold start, solve, download u0 / cost / status (what a host caller needs) -- and the same with the full iterate."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
N, B = 40, 4096
x0, yref = nominal_batch(B, N=N)
s = BatchedOcpSolver(N=N, batch=B); s.install_reference_ocp()
def step(full):
    s.set_x0(x0); s.set_yref_all(yref); s.cold_start(); s.solve()
    u0 = np.zeros((B, 2)); s_ = s.get(0, "u"); c = s.get_cost(); st = s.get_stats("status")
    if full:
        s.get_iterate()
for full in (False, True):
    for _ in range(3): step(full)
    t0 = time.perf_counter()
    for _ in range(10): step(full)
    dt = (time.perf_counter() - t0) / 10
    print(f"PCIe-inclusive step ({'u0,cost,status + full X,U' if full else 'u0,cost,status'} back): {1e3*dt:.3f} ms -> {B/dt:,.0f} solves/s (kernel alone {s.last_kernel_ms():.3f} ms)")
