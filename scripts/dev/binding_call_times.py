"""dev: wall time of every binding call the mirrored controller class issues per control step (one instance)"""
import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import torch  # noqa: F401
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
N = 38
x0, yref = nominal_batch(1, N=N)
s = BatchedOcpSolver(N=N, batch=1); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start(); s.solve()
calls = {
    "constraints_set(0,'lbx')": lambda: s.constraints_set(0, "lbx", x0[0]),
    "constraints_set(0,'ubx')": lambda: s.constraints_set(0, "ubx", x0[0]),
    "set_yref_all": lambda: s.set_yref_all(yref[0]),
    "solve": lambda: s.solve(),
    "get_iterate": lambda: s.get_iterate(),
    "get_cost": lambda: s.get_cost(),
    "get_stats('time_tot')": lambda: s.get_stats('time_tot'),
    "get_stats('sqp_iter')": lambda: s.get_stats('sqp_iter'),
    "get_stats('qp_iter')": lambda: s.get_stats('qp_iter'),
    "get(0,'u')": lambda: s.get(0, 'u'),
    "set(0,'x')": lambda: s.set(0, 'x', x0[0]),
}
for name, f in calls.items():
    for _ in range(20): f()
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    print(f"{name:28s} median {1e6 * np.median(t):8.1f} us   min {1e6 * np.min(t):8.1f} us")
