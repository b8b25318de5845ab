#!/usr/bin/env python3
"""One coupled-SNMPC configuration (N, uph from argv) solved a few times: for `rocprofv3 --kernel-trace --stats`."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd import config, snmpc as snm
from tum_control_amd.solver import CoupledSnmpcSolver
from tum_control_amd.workloads import nominal_batch
N, uph, B = int(sys.argv[1]), int(sys.argv[2]), 4096
stds = np.asarray(config.MPC["stds"], dtype=float)
w = snm.hammersley_normal(10, 3); A = snm.pce_matrix(w, snm.alpha_generation(3, 2)); offs = snm.x0_offsets(w, stds)
x0, yref = nominal_batch(B, N=N)
X0 = np.concatenate([x0[:, None, :], x0[:, None, :] + offs[None]], axis=1)
s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=config.MPC["gamma"])
s.install_reference_ocp()
s.constraints_set(0, "lbx", X0.reshape(B, -1)); s.constraints_set(0, "ubx", X0.reshape(B, -1)); s.set_yref_all(yref)
for _ in range(6):
    s.cold_start(); s.solve()
print("kernel ms", s.last_kernel_ms())
