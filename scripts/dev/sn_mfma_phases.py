#!/usr/bin/env python3
"""cycle counters of the phases of snmpc_prologue_mfma_kernel (workgroup 256 of a 4096-instance launch, both wavefronts);
the counters live in the DEVELOPMENT build of the library only (libtumnmpc_dev.so)"""
import os, sys
os.environ.setdefault("TUM_NMPC_DEV", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tum_control_amd import config, snmpc as snm
from tum_control_amd.solver import CoupledSnmpcSolver
from tum_control_amd.workloads import nominal_batch
stds = np.asarray(config.MPC["stds"], dtype=float); w = snm.hammersley_normal(10, 3)
A = snm.pce_matrix(w, snm.alpha_generation(3, 2)); offs = snm.x0_offsets(w, stds)
N, uph, B = 38, int(sys.argv[1]) if len(sys.argv) > 1 else 38, 4096
x0, yref = nominal_batch(B, N=N)
X0 = np.concatenate([x0[:, None, :], x0[:, None, :] + offs[None]], axis=1)
s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=0.8)
s.set_kernel("prologue-mfma")
s.install_reference_ocp()
s.constraints_set(0, "lbx", X0.reshape(B, -1)); s.constraints_set(0, "ubx", X0.reshape(B, -1))
s.set_yref_all(yref); s.cold_start(); s.solve(); s.cold_start()
d = s.debug_dump(0)
names = ["barrier (top)", "samples (operand reads, MFMAs, FMAs)", "quad_sum + exchange + barrier", "stash (wait for the records)", "stores", "loop edge"]
for g in range(2):
    t = d[20000 + 10 * g: 20006 + 10 * g]
    print(f"wavefront {g}: total {t.sum():.0f} cycles; " + "; ".join(f"{n} {v:.0f}" for n, v in zip(names, t)))
