#!/usr/bin/env python3
"""where does the time of results_async(with_iterate) go on one capsule? host-clock phases"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
N, B = 40, 4096
x0, yref = nominal_batch(B, N=N)
dx, dy = torch.from_numpy(x0).cuda(), torch.from_numpy(yref).cuda()
s = BatchedOcpSolver(N=N, batch=B); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref)
def enq():
    s.put_device("x0", dx.data_ptr()); s.put_device("yref", dy.data_ptr()); s.cold_start(); s.solve_async()
for mode in ("sync-then-copy", "copy-behind-solve", "copy-behind-solve, no put_device", "solve only"):
    ts = []
    for k in range(8):
        t0 = time.perf_counter()
        if mode == "copy-behind-solve, no put_device":
            s.cold_start(); s.solve_async()
        else:
            enq()
        t1 = time.perf_counter()
        if mode == "sync-then-copy":
            s.synchronize()
        t2 = time.perf_counter()
        if mode != "solve only":
            s.results_async(True)
        t3 = time.perf_counter()
        if mode != "solve only":
            s.results_wait()
        else:
            s.synchronize()
        t4 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
    m = 1e3 * np.median(np.array(ts[2:]), axis=0)
    print(f"{mode:36s} enqueue {m[0]:.3f}  sync {m[1]:.3f}  results_async call {m[2]:.3f}  wait {m[3]:.3f}  total {m[4]:.3f} ms")
