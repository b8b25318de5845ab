#!/usr/bin/env python3
"""Inter-batch overlap probe: S capsules on S streams take the fresh batches of config 2 in turn (step k -> capsule k % S), so the
tail of one batch's interior point kernel runs beside the head of the next batch. Prints solves/s for S = 1..6, for the
capsules' own streams (hipStreamNonBlocking) and for torch pool streams."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import config_groups
N, B, NBF = 40, int(os.environ.get("B", 4096)), 4
dev = torch.device("cuda", 0)
host = [config_groups(2, 0, B, B, N=N, variant=k)[:2] for k in range(NBF + 1)]
dx0 = [torch.from_numpy(np.ascontiguousarray(h[0])).to(dev) for h in host[1:]]
dyr = [torch.from_numpy(np.ascontiguousarray(h[1])).to(dev) for h in host[1:]]
torch.cuda.synchronize()
for kind in ("own", "torch"):
    for S in (1, 2, 3, 4, 6):
        streams = [torch.cuda.Stream() for _ in range(S)]
        sol = []
        for i in range(S):
            s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B, device=0)
            s.install_reference_ocp(); s.set_x0(host[0][0]); s.set_yref_all(host[0][1])
            if kind == "torch":
                s.set_stream(streams[i].cuda_stream)
            sol.append(s)
        for fresh in (True, False):
            def step(k):
                s = sol[k % S]
                if fresh:
                    s.put_device("x0", dx0[(k // 1) % NBF].data_ptr()); s.put_device("yref", dyr[(k // 1) % NBF].data_ptr())
                s.cold_start(); s.solve_async()
            for k in range(12): step(k)
            for s in sol: s.synchronize()
            torch.cuda.synchronize()
            steps = 24
            t0 = time.perf_counter()
            for k in range(steps): step(k)
            for s in sol: s.synchronize()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ok = all(float((s.get_stats("status") == 0).mean()) == 1.0 for s in sol)
            print(f"streams {kind} {S} {'fresh' if fresh else 'repeated'}: {B * steps / el / 1e6:.3f} M solves/s, {1e3 * el / steps:.3f} ms per step, status ok {ok}", flush=True)
        del sol
