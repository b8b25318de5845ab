#!/usr/bin/env python3
"""stress_iters.py's sequence (cold start + 3 further SQP-RTI steps on the same x0, far-off initial states) with the interior point warm start on (the
Python binding's default) and off (acados' setting for the nominal controller): iteration-count tails and status, and how far the answers are apart."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for track, seed in (("lvms", 4321), ("monteblanco", 99), ("modena", 7)):
    x0, yref = nominal_batch(B, N=40, track_name=track, stride=7, seed=seed)
    sol = {}
    for warm in (True, False):
        s = BatchedOcpSolver(N=40, batch=B, qp_warm_start=warm); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        for step in range(4):
            st = s.solve(); it = s.get_stats("qp_iter"); qs = s.get_stats("qp_status")
            u0 = np.asarray(s.get(0, "u")).reshape(B, 2).copy()
            sol[(warm, step)] = (u0, it.copy())
            print(f"{track} warm={warm} step {step}: status max {st}, qp_status max {qs.max()}, qp_iter mean {it.mean():.2f} p99 {np.quantile(it, 0.99):.0f} max {it.max()}, >=25: {(it >= 25).sum()}, at cap: {(it >= 50).sum()}")
        del s
    for step in range(4):
        d = np.abs(sol[(True, step)][0] - sol[(False, step)][0]).max(axis=1)
        print(f"   step {step}: |u0 warm - u0 cold| max {d.max():.2e} median {np.median(d):.1e}; instances beyond 1e-4: {(d > 1e-4).sum()}")
