#!/usr/bin/env python3
"""Where the wall time of a host-driven control step goes (one vehicle, warm): the enqueue (tum_ocp_step_async: staging copy + 5 kernel
launches), the wait (tum_ocp_results_wait: until the packing kernel's event), and the device time between the step's first and last kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller as Nominal
d = dict(np.load(os.path.join(ROOT, "tests", "golden", "replay_monteblanco_0_0_400.npz")))
n = len(d["x0"])
mpc = Nominal(None, None, dict(Tp=3.04, Ts=0.02, Ts_MPC=0.08), d["x0"][0])
s = mpc._solver
te, tw, td = [], [], []
for rep in range(3):
    for i in range(n):
        y = np.zeros((mpc.N + 1, 6)); y[:, :4] = d["yref"][i]
        t0 = time.perf_counter()
        s.step_async(d["x0"][i], y, True)
        t1 = time.perf_counter()
        summ, X, U = s.results_wait()
        t2 = time.perf_counter()
        if rep:
            te.append(t1 - t0); tw.append(t2 - t1); td.append(s.get_stats("time_tot"))
print(f"one-call step, one instance, {len(te)} warm steps: enqueue {1e3*np.median(te):.3f} ms, wait {1e3*np.median(tw):.3f} ms, sum {1e3*np.median(np.array(te)+np.array(tw)):.3f} ms; "
      f"device (first kernel start -> last kernel start) {1e3*np.median(td):.3f} ms")
