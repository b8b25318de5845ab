#!/usr/bin/env python3
"""What a HOST caller of configs[1] sees (4096 x N = 40, a fresh batch every step, inputs resident in HBM):
  (a) round 3's way: one capsule, synchronous getters after every solve (u0 / cost / status; + the whole iterate);
  (b) results through the capsules' pinned host slabs behind an event (tum_ocp_results_async / _wait) with S capsules in a
      ring (streaming.SolverRing.request_results / take_results): the copy of batch k crosses PCIe while batches k+1.. run;
  (c) the raw device-to-host rate of the iterate copy on an idle GPU, for scale.
usage: pcie_inclusive.py [S ...]   (default ring sizes 1 3)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.streaming import SolverRing
from tum_control_amd.workloads import nominal_batch
N, B, NBATCH = 40, 4096, 4
batches = [nominal_batch(B, N=N, seed=1234 + k) for k in range(NBATCH)]
dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in batches]


def mk(_=0):
    s = BatchedOcpSolver(N=N, batch=B); s.install_reference_ocp(); s.set_x0(batches[0][0]); s.set_yref_all(batches[0][1])
    return s


def enqueue(s, k):
    s.put_device("x0", dev[k % NBATCH][0].data_ptr()); s.put_device("yref", dev[k % NBATCH][1].data_ptr())
    s.cold_start(); s.solve_async()


s = mk()
for full in (False, True):
    def step(k):
        enqueue(s, k); s.synchronize()
        s.get(0, "u"); s.get_cost(); s.get_stats("status")
        if full:
            s.get_iterate()
    for k in range(3): step(k)
    t0 = time.perf_counter()
    for k in range(20): step(k)
    dt = (time.perf_counter() - t0) / 20
    print(f"PCIe-inclusive, synchronous getters, one capsule ({'u0,cost,status + full X,U' if full else 'u0,cost,status'} back): "
          f"{1e3*dt:.3f} ms -> {B/dt:,.0f} solves/s")

for S in [int(a) for a in sys.argv[1:]] or [1, 3]:
    ring = SolverRing(S, mk, [torch.cuda.Stream().cuda_stream for _ in range(S)] if S > 1 else None)
    for full in (False, True):
        seen = 0
        def step(k):
            global seen
            slot, c = ring.acquire()
            enqueue(c, k); ring.request_results(slot, with_iterate=full)
            if ring.outstanding(slot) == 2:          # (read the batch this capsule solved S steps ago AFTER the new one is on the stream)
                seen += int((ring.take_results(slot)[0][:, 3] == 0).sum())
        for k in range(4 * S): step(k)
        for _ in ring.drain(): pass
        seen = 0
        t0 = time.perf_counter(); per = []
        for k in range(30):
            ta = time.perf_counter(); step(k); per.append(time.perf_counter() - ta)
        for _, r in ring.drain(): seen += int((r[0][:, 3] == 0).sum())
        dt = (time.perf_counter() - t0) / 30
        assert seen == 30 * B, seen
        print(f"PCIe-inclusive, pinned slabs + events, {S} capsule(s) ({'u0,cost,status + full X,U' if full else 'u0,cost,status'} on the host every step): "
              f"{1e3*dt:.3f} ms -> {B/dt:,.0f} solves/s   (host time per step: median {1e3*np.median(per):.2f}, worst {1e3*max(per):.2f} ms)")

s.synchronize(); torch.cuda.synchronize()
for _ in range(2):
    s.results_async(True); s.results_wait()
t0 = time.perf_counter()
for _ in range(10):
    s.results_async(True); s.results_wait()
dt = (time.perf_counter() - t0) / 10
nbytes = 8 * B * ((N + 1) * 8 + N * 2 + 5)
print(f"raw result copy on an idle GPU: {nbytes/1e6:.1f} MB in {1e3*dt:.3f} ms = {nbytes/dt/1e9:.1f} GB/s")
