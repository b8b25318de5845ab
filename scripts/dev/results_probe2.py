#!/usr/bin/env python3
"""per-step host times of the ring loop with results on the host (one and three capsules, fresh batches)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.streaming import SolverRing
from tum_control_amd.workloads import nominal_batch
N, B, NBATCH = 40, 4096, 4
batches = [nominal_batch(B, N=N, seed=1234 + k) for k in range(NBATCH)]
dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in batches]
def mk(_=0):
    s = BatchedOcpSolver(N=N, batch=B); s.install_reference_ocp(); s.set_x0(batches[0][0]); s.set_yref_all(batches[0][1])
    return s
for S in (1, 3):
    ring = SolverRing(S, mk, [torch.cuda.Stream().cuda_stream for _ in range(S)] if S > 1 else None)
    for full in (False, True):
        rows = []
        for k in range(16):
            t0 = time.perf_counter()
            slot, c = ring.acquire()
            r = ring.take_results(slot)
            t1 = time.perf_counter()
            c.put_device("x0", dev[k % NBATCH][0].data_ptr()); c.put_device("yref", dev[k % NBATCH][1].data_ptr())
            c.cold_start(); c.solve_async()
            t2 = time.perf_counter()
            ring.request_results(slot, with_iterate=full)
            t3 = time.perf_counter()
            rows.append((t1 - t0, t2 - t1, t3 - t2))
        for _ in ring.drain(): pass
        a = 1e3 * np.array(rows)
        print(f"S={S} full={full}: take/enqueue/request per step (ms):")
        print("   take   ", " ".join(f"{v:.2f}" for v in a[:, 0]))
        print("   enqueue", " ".join(f"{v:.2f}" for v in a[:, 1]))
        print("   request", " ".join(f"{v:.2f}" for v in a[:, 2]))
