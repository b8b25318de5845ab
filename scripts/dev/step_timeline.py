"""dev: a few one-call control steps (one instance) for a kernel + memory-copy trace"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch  # noqa: F401
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
N = 38
x0, yref = nominal_batch(1, N=N)
s = BatchedOcpSolver(N=N, batch=1); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
for _ in range(30):
    s.step(x0=x0, yref=yref, with_iterate=True)
