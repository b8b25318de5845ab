#!/usr/bin/env python3
"""Ad-hoc parity sweep HIP pipeline vs the oracle over horizons and perturbation sizes (development aid; the gated comparisons are in tests/)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
from tum_control_amd.config import default_config, SIM
from oracle.oracle import OracleOcp

mpc = default_config()["mpc"]
B = 192
worst = 0.0
for N in (6, 13, 25, 38, 40, 44, 48):
    for scale in (1.0, 3.0):
        w = np.asarray(SIM["w_state_estimation"], dtype=float) * scale
        x0, yref = nominal_batch(B, N=N, noise=w, seed=100 + N)
        s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B); s.install_reference_ocp()
        s.set_x0(x0); s.set_yref_all(yref); s.cold_start(); s.solve()
        X, U = s.get_iterate(); it = s.get_stats("qp_iter"); st = s.get_stats("status")
        o = OracleOcp(N, 0.08, 3)
        o.set_weights(mpc["q_lon"], mpc["q_yaw"], mpc["q_vel"], mpc["r_jerk"], mpc["r_steering_rate"], mpc["L1_pen"], mpc["L2_pen"], scale=0.01)
        o.zl[:] = mpc["L1_pen"]; o.zu[:] = mpc["L1_pen"]; o.Zl[:] = mpc["L2_pen"]; o.Zu[:] = mpc["L2_pen"]
        u0, X1, stats = o.solve_batch_cold(x0, yref, os.cpu_count())
        same = stats[:, 1] == it
        # forced-iteration comparison where the counts differ (termination test on the tolerance edge)
        if not same.all():
            u0f, X1f, _ = o.solve_batch_cold(x0, yref, os.cpu_count(), force_iter=it.astype(np.int32))
            u0[~same] = u0f[~same]; X1[~same] = X1f[~same]
        e = max(np.abs(U[:, 0] - u0).max(), np.abs(X[:, 1] - X1).max())
        worst = max(worst, e)
        print(f"N {N:2d} noise x{scale:.0f}: status0 {np.mean(st == 0):.3f} (oracle {np.mean(stats[:, 2] == 0):.3f}) qp_iter {it.mean():.2f} same count {same.mean():.3f} max|du0, dx1| {e:.2e}", flush=True)
        del s
print("worst", worst)
