#!/usr/bin/env python3
"""Wall time of ONE control step through the mirrored controller classes, the reference's call pattern (main.py:44-58:
set_initial_state, solve -> u0, predictions, stats), one instance, warm-started real-time iterations over a logged loop --
nominal, stochastic and robustified controller; split into set_initial_state / solve() of the class / of which
step_async + results_wait / device time as the class reports it (stats[1])."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller as Nominal
from tum_control_amd.snmpc import Stochastic_Nonlinear_Model_Predictive_Controller as Stochastic
from tum_control_amd.r2nmpc import Reduced_Robustified_Nonlinear_Model_Predictive_Controller as Robust

d = dict(np.load(os.path.join(ROOT, "tests", "golden", "replay_monteblanco_0_0_400.npz")))      # (a dict: NpzFile decompresses an array on EVERY access)
n = len(d["x0"])
# the reference's literal AcadosOcpSolver call sequence (call_pattern="acados": N+1 x set yref [+ set p], solve, get u, N x get x, get_cost,
# 3 x get_stats -- 2 N + 7 synchronous round trips, 3 N + 8 for the SNMPC; NMPC_class.py:169-206, SNMPC_class.py:181-214) beside the one-call step
for pattern in ("acados", "step"):
  for kind, cls in (("nominal", Nominal), ("stochastic (SNMPC)", Stochastic), ("robustified (R2NMPC)", Robust)):
    mpc = cls(None, None, dict(Tp=3.04, Ts=0.02, Ts_MPC=0.08), d["x0"][0], call_pattern=pattern)
    s = mpc.acados_solver
    t_set, t_solve, t_c, t_dev = [], [], [], []
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
                t_set.append(t1 - t0); t_solve.append(t2 - t1); t_dev.append(stats[1])
    k = len(t_solve)
    tot = np.array(t_set) + np.array(t_solve)
    if pattern == "acados":
        ncalls = (3 * mpc.N + 8) if "SNMPC" in kind else (2 * mpc.N + 7)
        print(f"{kind:22s} controller class, LITERAL acados call sequence ({ncalls} solver calls + 2 for set_initial_state per control step), one instance, {k} warm control steps: "
              f"set_initial_state {1e3 * np.median(t_set):.3f} ms, solve() of the class {1e3 * np.median(t_solve):.3f} ms (solver time it reports {1e3 * np.median(t_dev):.3f} ms); "
              f"whole step median {1e3 * np.median(tot):.3f} ms, 99th percentile {1e3 * np.percentile(tot, 99):.3f} ms", flush=True)
        continue
    print(f"{kind:22s} controller class, one instance, {k} warm control steps: set_initial_state {1e3 * np.median(t_set):.3f} ms, solve() of the class "
          f"{1e3 * np.median(t_solve):.3f} ms (of which step_async + results_wait {1e3 * np.median((t_c[-k:] or [0.0])):.3f} ms, solver time it reports {1e3 * np.median(t_dev):.3f} ms); "
          f"whole step median {1e3 * np.median(tot):.3f} ms, 99th percentile {1e3 * np.percentile(tot, 99):.3f} ms", flush=True)
