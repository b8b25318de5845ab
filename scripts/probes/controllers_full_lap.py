#!/usr/bin/env python3
"""All three controllers over the full logged distance (5499 control steps = 110 s of driving) in the all-device closed
loop, both tracks: solver status, IPM iterations, tracking and gg usage."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.closed_loop import ClosedLoopBatch

steps = 5499
for track in ("monteblanco", "lvms"):
    for kind in ("nominal", "snmpc", "r2"):
        cl = ClosedLoopBatch(track, batch=1, N=38, Tp=3.04, controller=kind, on_device=True, log_capacity=steps)
        t0 = time.perf_counter(); lg = cl.run(steps); wall = time.perf_counter() - t0
        dbg = lg["simSolverDebug"][:, 0]; x = lg["CiLX"][:, 0]; ref = lg["simREF"][:, 0]
        dev = np.hypot(x[:-1, 0] - ref[:, 0], x[:-1, 1] - ref[:, 1])
        alat = x[:, 3] * x[:, 5]
        print(f"{track:12s} {kind:8s}: {1e3 * wall / steps:.3f} ms/step, status 0 {(dbg[:, 4] == 0).mean():.5f}, qp_iter mean {dbg[:, 3].mean():.2f} max {dbg[:, 3].max():.0f}, "
              f"distance to the reference point mean {dev.mean():.3f} m max {dev.max():.3f} m, |a_lat| max {np.abs(alat).max():.2f} m/s2, v mean {x[:, 3].mean():.2f} m/s")
        del cl
