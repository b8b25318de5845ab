#!/usr/bin/env python3
"""Wall time of ONE control step through the mirrored controller class, the reference's call pattern (main.py:44-58:
set_initial_state, solve -> u0, predictions, stats), one instance, warm-started real-time iterations over a logged loop;
split into set_initial_state / solve() of the class / of which step_async + results_wait."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller

d = dict(np.load(os.path.join(ROOT, "tests", "golden", "replay_monteblanco_0_0_400.npz")))      # (a dict: NpzFile decompresses an array on EVERY access)
mpc = Nonlinear_Model_Predictive_Controller(sim_main_params=dict(Tp=3.04, Ts=0.02, Ts_MPC=0.08), X0_MPC=d["x0"][0])
mpc.update_cost_function_weights(d["params"])
n = len(d["x0"])
t_set, t_solve, t_c = [], [], []
s = mpc.acados_solver
c_step = s.step
def timed_step(*a, **k):
    t = time.perf_counter(); r = c_step(*a, **k); t_c.append(time.perf_counter() - t); return r
s.step = timed_step
for rep in range(3):
    for i in range(n):
        t0 = time.perf_counter()
        mpc.set_initial_state(d["x0"][i])
        t1 = time.perf_counter()
        y = d["yref"][i]
        u0, pred_X, stats = mpc.solve(dict(pos_x=y[:, 0], pos_y=y[:, 1], ref_yaw=y[:, 2], ref_v=y[:, 3]))
        t2 = time.perf_counter()
        if rep:
            t_set.append(t1 - t0); t_solve.append(t2 - t1)
k = len(t_solve)
print(f"controller class, one instance, {k} warm control steps: set_initial_state {1e3 * np.median(t_set):.3f} ms, solve() of the class "
      f"{1e3 * np.median(t_solve):.3f} ms (of which step_async + results_wait {1e3 * np.median(t_c[-k:]):.3f} ms, device {1e3 * stats[1]:.3f} ms); "
      f"whole step median {1e3 * np.median(np.array(t_set) + np.array(t_solve)):.3f} ms, worst {1e3 * np.max(np.array(t_set) + np.array(t_solve)):.3f} ms")
