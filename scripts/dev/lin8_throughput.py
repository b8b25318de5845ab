import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
x0, yref = nominal_batch(4096, N=40)
for k in ("lin-lane-per-stage", "lin-eight-lanes"):
    s = BatchedOcpSolver(N=40, batch=4096); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.set_kernel(k)
    for _ in range(6):
        s.cold_start(); s.solve()
    print(k, s.last_kernel_ms())
